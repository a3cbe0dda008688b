"""N > 1 path on CPU: world_size-2 gloo processes exercising the stream sharding and the
optional appearance-feature all-gather (the only exchange step)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from object_detection_tracking_amd.parallel import all_gather_reid_features, shard_streams


def _free_port():
  s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
  return p


def _worker(rank, world, port, out):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    streams = ["cam%d.mp4" % i for i in range(5)]
    mine = shard_streams(streams, rank, world)
    rng = np.random.default_rng(100 + rank)
    n = 3 + 4 * rank
    feats = torch.from_numpy(rng.standard_normal((n, 256)).astype(np.float32))
    boxes = torch.from_numpy(rng.uniform(0, 1000, (n, 4)).astype(np.float32))
    gf, gb = all_gather_reid_features(feats, boxes, max_rows=100)
    ok = len(gf) == world
    for r in range(world):
      rr = np.random.default_rng(100 + r)
      nr = 3 + 4 * r
      ef = rr.standard_normal((nr, 256)).astype(np.float32)
      eb = rr.uniform(0, 1000, (nr, 4)).astype(np.float32)
      ok = ok and np.array_equal(gf[r].numpy(), ef) and np.array_equal(gb[r].numpy(), eb)
    # timing reduction used by bench.py: max over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t.item()) == float(world)
    out[rank] = (ok, mine)
    dist.barrier()
  finally:
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_feature_allgather():
  world = 2
  mgr = mp.Manager()
  out = mgr.dict()
  mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
  assert out[0][0] and out[1][0]
  assert out[0][1] == ["cam0.mp4", "cam2.mp4", "cam4.mp4"] and out[1][1] == ["cam1.mp4", "cam3.mp4"]


def test_allgather_is_identity_without_process_group():
  f = torch.ones(2, 256); b = torch.zeros(2, 4)
  gf, gb = all_gather_reid_features(f, b)
  assert len(gf) == 1 and gf[0] is f and gb[0] is b


def test_bench_gpus_flag_starts_that_many_ranks():
  """`python bench.py --gpus 2` (no launcher in front, as the driver calls it) must start two ranks itself,
  rendezvous on 127.0.0.1 and report n_gpus == world == 2 with both ranks seen (--gpus was parsed
  and never used).  --launcher-selftest swaps the GPU work for a sleep and RCCL for gloo; the launch path is the
  real one."""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ)
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                      "--launcher-selftest"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
  assert r.returncode == 0, r.stderr[-2000:]
  line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
  d = json.loads(line)
  assert d["n_gpus"] == 2 and d["ranks_seen"] == [0, 1] and d["max_over_ranks_ok"] and d["steps"] == 3
  # a launcher that started a different number of ranks than --gpus asks for is an error, not a silent n_gpus
  env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
  r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=env2,
                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
  assert r.returncode != 0 and "--gpus 2" in r.stderr


def test_bench_self_launch_world_8():
  """The launch / rendezvous / reduction path at the size the driver's scaling run uses: 8 ranks on 127.0.0.1 (gloo,
  no GPU work), all seen, max-over-ranks timing."""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ)
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  env["OMP_NUM_THREADS"] = "1"
  r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                      "--launcher-selftest"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
  assert d["n_gpus"] == 8 and d["ranks_seen"] == list(range(8)) and d["max_over_ranks_ok"]
  assert len(d["per_rank"]) == 8 and [p["rank"] for p in d["per_rank"]] == list(range(8))
  assert all("affinity" in p for p in d["per_rank"])


def test_rank_cpu_affinity_plan(tmp_path):
  """Each rank gets its share of the CPUs local to its GPU's NUMA node (sysfs numa_node / local_cpulist); ranks on one
  node do not overlap; without topology information the affinity is left alone."""
  from object_detection_tracking_amd import parallel as P
  bdfs = ["0000:%02x:00.0" % (0x10 + i) for i in range(8)]
  for i, b in enumerate(bdfs):
    d = tmp_path / b
    d.mkdir()
    (d / "numa_node").write_text("%d\n" % (i // 4))
    (d / "local_cpulist").write_text("0-31,64-95\n" if i < 4 else "32-63,96-127\n")
  topo = P.gpu_numa_topology(bdfs, str(tmp_path))
  assert [t[0] for t in topo] == [0, 0, 0, 0, 1, 1, 1, 1] and len(topo[0][1]) == 64 and topo[5][1][0] == 32
  plans = P.plan_rank_cpus(topo, range(128))
  assert all(len(p) == 16 for p in plans)
  for a in range(8):
    assert set(plans[a]) <= set(topo[a][1])
    for b in range(a + 1, 8):
      assert not set(plans[a]) & set(plans[b])
  # a restricted cpuset (container): only allowed CPUs are used
  plans = P.plan_rank_cpus(topo, range(0, 40))
  assert set(plans[0]) <= set(range(32)) and set(plans[4]) <= set(range(32, 40))
  # no topology -> everything allowed, and bind_rank_to_gpu_numa reports bound = False without touching the affinity
  before = os.sched_getaffinity(0)
  info = P.bind_rank_to_gpu_numa(0, 2, sysfs=str(tmp_path / "nope"), bdfs=["0000:aa:00.0", "0000:ab:00.0"])
  assert info["bound"] is False and os.sched_getaffinity(0) == before
  info = P.bind_rank_to_gpu_numa(5, 8, sysfs=str(tmp_path), bdfs=bdfs, apply=False)
  assert info["numa_node"] == 1 and info["bound"] is False and info["pci"] == bdfs[5]


def test_bench_counter_csv_reader(tmp_path):
  """bench.py's roofline.traffic comes from its own rocprofv3 --pmc child passes (round 6): the CSV reader sums one counter over
  the fp16x2 conv dispatches only (the guard's bring-up forward on the bf16x3 twin must not count) and counts dispatches once."""
  import bench
  p = tmp_path / "pmc_counter_collection.csv"
  rows = [("1", "void odt::(anonymous namespace)::conv_h2k_kernel<4, false, true, 2>(odt::ConvParams const*)", "FETCH_SIZE", "1000"),
          ("1", "void odt::(anonymous namespace)::conv_h2k_kernel<4, false, true, 2>(odt::ConvParams const*)", "FETCH_SIZE", "24"),
          ("2", "void odt::(anonymous namespace)::conv_split3_kernel<4, 2, 4, false>(odt::ConvParams const*)", "FETCH_SIZE", "7777"),
          ("3", "void odt::(anonymous namespace)::conv_stem_kernel(odt::ConvParams const*)", "FETCH_SIZE", "500"),
          ("4", "odt::(anonymous namespace)::split_weights_h2_kernel(float const*, int)", "FETCH_SIZE", "9"),
          ("5", "void odt::(anonymous namespace)::conv_h2_kernel<4, 4, false, 2>(odt::ConvParams const*)", "WRITE_SIZE", "3")]
  p.write_text("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n" + "\n".join('%s,"%s",%s,%s' % r for r in rows) + "\n")
  is_h2 = lambda k: ("conv_h2" in k or "conv_stem" in k) and "kernel" in k and "split_weights" not in k
  assert bench.sum_counter_csv(str(p), "FETCH_SIZE", is_h2) == (1524.0, 2)
  assert bench.sum_counter_csv(str(p), "WRITE_SIZE", is_h2) == (3.0, 1)
