"""The C-ABI shared library loads and exports every symbol include/odt.h declares (no compute
calls, no GPU needed), and the product loader refuses to run without it."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
  txt = open(os.path.join(ROOT, "include", "odt.h")).read()
  txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
  return sorted(set(re.findall(r"\b(odt_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_by_hip_library():
  from object_detection_tracking_amd.build import build_hip
  lib = ctypes.CDLL(build_hip())
  names = _declared()
  assert len(names) >= 20
  for n in names:
    assert hasattr(lib, n), "libodt_hip.so does not export %s" % n


def test_binding_covers_header():
  from object_detection_tracking_amd._lib import OdtLib
  assert sorted(OdtLib.SYMBOLS) == _declared()


def test_no_cpu_fallback_when_library_missing(tmp_path, monkeypatch):
  from object_detection_tracking_amd import _lib
  monkeypatch.setattr(_lib, "LIB_HIP_PATH", str(tmp_path / "missing.so"))
  monkeypatch.setattr(_lib, "_LIB", None)
  with pytest.raises(_lib.OdtError):
    _lib.get_lib()


def test_product_package_never_imports_oracle_or_emulator():
  pkg = os.path.join(ROOT, "object_detection_tracking_amd")
  for dp, _, files in os.walk(pkg):
    for f in files:
      if f.endswith((".py", ".hip", ".hpp", ".h")):
        src = open(os.path.join(dp, f)).read()
        assert "import oracle" not in src and "from oracle" not in src, f
        assert "libodt_emu" not in src or f == "build.py", f


def test_error_reporting_through_abi(emu_lib):
  """Errors come back as status + message, never abort."""
  import numpy as np
  from object_detection_tracking_amd import ops
  from object_detection_tracking_amd._lib import OdtError
  with pytest.raises(OdtError, match="Cin must be a multiple of 32"):
    ops.conv2d(np.zeros((1, 4, 4, 3), np.float32), np.zeros((1, 1, 3, 8), np.float32), lib=emu_lib)
  with pytest.raises(OdtError, match="4096"):
    ops.nms(np.zeros((5000, 4), np.float32), np.zeros(5000, np.float32), 10, 0.5, lib=emu_lib)


def test_one_reader_of_the_environment():
  """Knob hygiene (round 6): csrc/knobs.cpp is the only file of the library that calls getenv -- plan builders and launchers
  read the parsed table (knobs.hpp) -- and the wrong-result ablation knob ODT_FUSE_DEBUG no longer exists."""
  csrc = os.path.join(ROOT, "object_detection_tracking_amd", "csrc")
  for f in sorted(os.listdir(csrc)):
    if not f.endswith((".hip", ".hpp", ".cpp", ".h")):
      continue
    src = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, f)).read())
    if f != "knobs.cpp":
      assert "getenv" not in src, f
    assert "ODT_FUSE_DEBUG" not in src and "FUSE_DEBUG" not in src, f
  # every ODT_* variable the Python side, the tests and the tools set is in the table (a typo cannot become a silent no-op)
  table = set(re.findall(r"X\(([A-Z0-9_]+)\)", open(os.path.join(csrc, "knobs.hpp")).read()))
  used = set()
  import glob
  files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
  for pat in ("tests/*.py", "tools/*.py", "object_detection_tracking_amd/*.py", "object_detection_tracking_amd/*/*.py"):
    files += glob.glob(os.path.join(ROOT, pat))
  for f in files:
    used |= set(re.findall(r"\bODT_((?:CONV|FUSE|EFFDET|DW|STEM|TAIL|SIDE|COSINE|TRACKER|SPLIT|AMAX)_[A-Z0-9_]+)", open(f).read()))
  assert used - table - {"FUSE_DEBUG"} == set(), sorted(used - table)


def test_describe_lists_active_overrides_by_name(emu_lib, monkeypatch):
  """odt_describe names every ODT_* override that was set when the handle's plan was built, launcher-level ones included
  (ODT_CONV_H2_BK64, ODT_DW_PX ... were invisible before), and a default handle reports none."""
  from common import make_config
  from object_detection_tracking_amd import models
  from object_detection_tracking_amd.weights import synthetic_weights
  for k in list(os.environ):
    if k.startswith("ODT_"):
      monkeypatch.delenv(k)
  cfg = make_config(rpn_test_post_nms_topk=50, max_size=96, short_edge_size=64)
  w = synthetic_weights(cfg, 0)
  m = models.get_model(cfg, 0, weights=w, lib=emu_lib)
  d = m.engine(1, 64, 96).describe()
  m.close()
  assert d["env_overrides_applied"] == 0 and d["env_overrides"] == [], d
  monkeypatch.setenv("ODT_CONV_H2_BK64", "0")
  monkeypatch.setenv("ODT_SPLIT_REDUCE_BLOCKS", "128")
  monkeypatch.setenv("ODT_CONV_SPLIT_MINTILES", "1")
  m = models.get_model(cfg, 0, weights=w, lib=emu_lib)
  d = m.engine(1, 64, 96).describe()
  m.close()
  assert d["env_overrides_applied"] == 3, d
  assert sorted(d["env_overrides"]) == ["ODT_CONV_H2_BK64=0", "ODT_CONV_SPLIT_MINTILES=1", "ODT_SPLIT_REDUCE_BLOCKS=128"], d
