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
