"""Build libodt_hip.so (gfx950) in-tree with hipcc; no cmake, no JIT cache.

``python -m object_detection_tracking_amd.build`` or :func:`build_hip`.
The simulator build (:func:`build_emu`) compiles the *same* kernel sources with
g++ against tests/emu/include/hip/hip_runtime.h -- test infrastructure only,
it lands under tests/emu/ and the package never loads it.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["runtime.hip", "plan_common.hip", "plan_fpn.hip", "plan_effdet.hip", "op_shims.hip", "conv_igemm.hip", "conv_split.hip", "conv_split1.hip", "conv_split3.hip", "conv_h2.hip", "conv_h2d.hip", "conv_h2k.hip", "conv_stem.hip", "elementwise.hip", "proposals.hip",
           "roi_align.hip", "detections.hip", "tracker.hip", "tracker_core.cpp", "knobs.cpp", "effnet.hip", "effnet_mbconv.hip", "effdet_post.hip", "probe.hip"]
HEADERS = ["odt_common.hpp", "knobs.hpp", "odt_model.hpp", "conv_split_common.hpp", "conv_split_epilogue.hpp", "select_device.hpp", os.path.join(ROOT, "include", "odt.h")]
LIB_HIP = os.path.join(HERE, "libodt_hip.so")
LIB_EMU = os.path.join(ROOT, "tests", "emu", "libodt_emu.so")


def _newer(target, deps):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
  r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
  if r.returncode != 0:
    raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
  return r.stdout


def _deps():
  return [os.path.join(CSRC, h) if not os.path.isabs(h) else h for h in HEADERS]


def build_hip(force=False, verbose=False):
  """hipcc --offload-arch=gfx950 each TU -> .o (parallel), link the .so."""
  hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
  objdir = os.path.join(HERE, "build", "hip")
  os.makedirs(objdir, exist_ok=True)
  flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-Wno-unused-result", "-I", CSRC]
  jobs = []
  for s in SOURCES:
    src = os.path.join(CSRC, s)
    obj = os.path.join(objdir, s + ".o")
    if force or _newer(obj, [src] + _deps()):
      jobs.append([hipcc] + flags + ["-c", src, "-o", obj])
  with ThreadPoolExecutor(max_workers=8) as ex:
    for out in ex.map(_run, jobs):
      if verbose and out:
        print(out)
  objs = [os.path.join(objdir, s + ".o") for s in SOURCES]
  if force or jobs or _newer(LIB_HIP, objs):
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_HIP] + objs)
  return LIB_HIP


def build_emu(force=False):
  """g++ build of the same sources against the HIP-on-CPU simulator header
  (tests only)."""
  emu_dir = os.path.join(ROOT, "tests", "emu")
  objdir = os.path.join(emu_dir, "build")
  os.makedirs(objdir, exist_ok=True)
  inc = os.path.join(emu_dir, "include")
  flags = ["-O2", "-g0", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-x", "c++",
           "-I", inc, "-I", CSRC, "-Wno-attributes"]
  deps = _deps() + [os.path.join(inc, "hip", "hip_runtime.h")]
  jobs = []
  srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(emu_dir, "hipemu.cpp")]
  objs = []
  for src in srcs:
    obj = os.path.join(objdir, os.path.basename(src) + ".o")
    objs.append(obj)
    if force or _newer(obj, [src] + deps):
      jobs.append(["g++"] + flags + ["-c", src, "-o", obj])
  with ThreadPoolExecutor(max_workers=8) as ex:
    list(ex.map(_run, jobs))
  if force or jobs or _newer(LIB_EMU, objs):
    _run(["g++", "-shared", "-pthread", "-o", LIB_EMU] + objs)
  return LIB_EMU


if __name__ == "__main__":
  which = sys.argv[1] if len(sys.argv) > 1 else "hip"
  print(build_emu(force="--force" in sys.argv) if which == "emu"
        else build_hip(force="--force" in sys.argv, verbose=True))
