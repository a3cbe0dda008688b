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


# ---- host placement: each rank on the cores next to its GPU -----------------------------------------------------------
# With 8 ranks on one host the per-stream path that degrades first is the host side (decode -> pinned staging -> H2D,
# tracker glue; reference SPEED.md:61, enqueuer_thread.py:236-303): a rank whose threads run on the other socket pays
# the inter-socket hop on every staging copy.  bind_rank_to_gpu_numa() pins the calling process to the CPUs of the NUMA
# node its GPU hangs off (sysfs: /sys/bus/pci/devices/<bdf>/numa_node + local_cpulist), and splits that node's CPUs
# evenly among the ranks that share it, so co-hosted ranks do not fight over the same cores either.
def _parse_cpulist(text):
  cpus = []
  for part in text.strip().split(","):
    if not part:
      continue
    a, _, b = part.partition("-")
    cpus.extend(range(int(a), int(b or a) + 1))
  return cpus


def gpu_pci_bdf(device):
  """'dddd:bb:dd.f' of a visible HIP device (hipDeviceGetPCIBusId through torch's runtime), or None."""
  try:
    p = torch.cuda.get_device_properties(device)
    if hasattr(p, "pci_bus_id") and hasattr(p, "pci_device_id"):
      return "%04x:%02x:%02x.0" % (int(getattr(p, "pci_domain_id", 0)), int(p.pci_bus_id), int(p.pci_device_id))
  except Exception:
    pass
  return None


def gpu_numa_topology(bdfs, sysfs="/sys/bus/pci/devices"):
  """[(numa_node, [cpus])] per PCI address; node -1 / empty list when the platform does not say."""
  import os
  out = []
  for bdf in bdfs:
    node, cpus = -1, []
    try:
      with open(os.path.join(sysfs, bdf, "numa_node")) as fh:
        node = int(fh.read().strip())
      with open(os.path.join(sysfs, bdf, "local_cpulist")) as fh:
        cpus = _parse_cpulist(fh.read())
    except (OSError, ValueError, TypeError):
      pass
    out.append((node, cpus))
  return out


def plan_rank_cpus(topology, allowed):
  """CPU set per rank: the rank's GPU-local CPUs (restricted to `allowed`), split evenly among the ranks that share
  those CPUs; ranks without topology information keep `allowed`."""
  allowed = sorted(allowed)
  plans = [None] * len(topology)
  groups = {}
  for r, (node, cpus) in enumerate(topology):
    local = tuple(c for c in cpus if c in set(allowed))
    if local:
      groups.setdefault(local, []).append(r)
  for local, ranks in groups.items():
    k = len(ranks)
    per = max(1, len(local) // k)
    for i, r in enumerate(ranks):
      mine = list(local[i * per:(i + 1) * per]) if i * per < len(local) else list(local)
      plans[r] = mine or list(local)
  return [p if p is not None else list(allowed) for p in plans]


def bind_rank_to_gpu_numa(local_rank, world, sysfs="/sys/bus/pci/devices", bdfs=None, apply=True):
  """Pin this process to its share of the CPUs local to GPU `local_rank`.  Returns what was done:
  {"numa_node", "cpus" (count), "cpu_range", "bound"} -- `bound` False when the platform gives no topology (single
  node, container without sysfs) and the affinity was left alone."""
  import os
  if bdfs is None:
    bdfs = [gpu_pci_bdf(i) for i in range(world)]
  topo = gpu_numa_topology([b or "?" for b in bdfs], sysfs)
  try:
    allowed = sorted(os.sched_getaffinity(0))
  except AttributeError:
    return {"numa_node": -1, "cpus": 0, "cpu_range": "", "bound": False}
  mine = plan_rank_cpus(topo, allowed)[local_rank]
  node = topo[local_rank][0]
  bound = False
  if apply and topo[local_rank][1] and set(mine) != set(allowed):
    try:
      os.sched_setaffinity(0, mine)
      bound = True
    except OSError:
      bound = False
  return {"numa_node": node, "cpus": len(mine), "cpu_range": "%d-%d" % (mine[0], mine[-1]) if mine else "",
          "bound": bound, "pci": bdfs[local_rank]}
