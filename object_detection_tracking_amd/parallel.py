"""Multi-GPU layout of the hot path: one process per GPU, one video stream per GPU.

The reference has no collectives; its only multi-GPU inference mode that scales is N independent
single-GPU processes (reference SPEED.md:61, "4 / 1*").  Tracker state is per stream and
sequential in time, so the unit of sharding is the stream: stream s runs on rank s % world with
replicated weights and no activation exchange ("weak" scaling, no data-path collective).

The one optional exchange is the per-frame appearance features for cross-camera association
(done offline on the CPU by the reference, multi_video_reid.py:308-324): an all-gather of at most
result_per_im x 256 floats per rank -- latency-bound (~100 KB), far below the per-link xGMI
bandwidth, so a single fixed-size RCCL all_gather (backend "nccl" == RCCL on ROCm; "gloo" in the
CPU tests) is the right shape: pad to the static maximum, gather once, trim by the gathered counts.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_streams(streams, rank, world):
  """Round-robin assignment of video streams to ranks (stream i -> rank i % world)."""
  return [s for i, s in enumerate(streams) if i % world == rank]


def all_gather_reid_features(feats, boxes, max_rows=100, group=None):
  """All-gather this rank's pooled features [n,D] and boxes [n,4] (n <= max_rows).

  Returns (list of [n_r,D] tensors, list of [n_r,4] tensors), index = source rank.  One
  fixed-shape collective for the payload plus one for the row counts.
  """
  if not (dist.is_available() and dist.is_initialized()):
    return [feats], [boxes]
  world = dist.get_world_size(group)
  n, D = feats.shape
  assert n <= max_rows and boxes.shape == (n, 4)
  dev = feats.device
  pack = torch.zeros((max_rows, D + 4), dtype=torch.float32, device=dev)
  pack[:n, :D] = feats
  pack[:n, D:] = boxes
  count = torch.tensor([n], dtype=torch.int64, device=dev)
  packs = [torch.empty_like(pack) for _ in range(world)]
  counts = [torch.empty_like(count) for _ in range(world)]
  dist.all_gather(packs, pack, group=group)
  dist.all_gather(counts, count, group=group)
  out_f, out_b = [], []
  for p, c in zip(packs, counts):
    k = int(c.item())
    out_f.append(p[:k, :D].clone())
    out_b.append(p[:k, D:].clone())
  return out_f, out_b
