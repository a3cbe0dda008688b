"""Test configuration.

Markers
  gpu : needs a real MI355X; runs the product library libodt_hip.so.

Backends
  "hip" : object_detection_tracking_amd/libodt_hip.so (product, `-m gpu`)
  "emu" : tests/emu/libodt_emu.so -- the SAME kernel sources compiled with g++ against the
          HIP-on-CPU simulator header (tests/emu/include/hip/hip_runtime.h).  Test
          infrastructure only: it lets the CPU suite execute the kernels' indexing / LDS /
          MFMA-fragment / shuffle logic without a GPU.  Never a product fallback.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (libodt_hip.so)")
  config.addinivalue_line("markers", "slow: long-running CPU test")


_cache = {}


def _get_backend(name):
  if name in _cache:
    return _cache[name]
  from object_detection_tracking_amd import _lib
  if name == "hip":
    lib = _lib.get_lib()          # raises if the .so or the GPU is missing
  else:
    from object_detection_tracking_amd.build import build_emu
    lib = _lib.OdtLib(build_emu())
  _cache[name] = lib
  return lib


BACKENDS = [pytest.param("emu", id="emu"),
            pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def backend(request):
  """(name, lib) for each backend; the hip one carries the gpu marker."""
  return request.param, _get_backend(request.param)


@pytest.fixture
def hip_lib():
  return _get_backend("hip")


@pytest.fixture
def emu_lib():
  return _get_backend("emu")
