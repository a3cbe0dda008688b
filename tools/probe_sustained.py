#!/usr/bin/env python
"""The sustained-rate probe behind bench.py's roofline.sustained_peak, stand-alone (odt_probe_mfma_bf16, csrc/probe.hip):
the bf16x3 split kernels' MFMA mix on random bf16 operands, 100 ms warm-up + >= 300 ms timed, registers-only and with the
kernels' LDS fragment reads, three rounds each.  python tools/probe_sustained.py [device]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_tracking_amd import _lib

def main():
  dev = int(sys.argv[1]) if len(sys.argv) > 1 else 0
  lib = _lib.get_lib()
  print("# odt_probe_mfma_bf16(device %d, warm 100 ms, timed >= 300 ms): v_mfma_f32_32x32x16_bf16, 2 x 4 tiles per wave, six piece products per k16 step" % dev)
  print("%-28s %12s %12s %10s %9s %9s" % ("variant", "16-bit TF", "f32-work TF", "clock GHz", "ms", "launches"))
  for rnd in range(3):
    for name, lds, prod in (("operands in registers", 0, 6), ("LDS fragment reads per step", 1, 6),
                            ("fp16x2 mix (f16, 3 products)", 2, 3)):
      tf = C.c_double(); ghz = C.c_double(); ms = C.c_double(); n = C.c_int()
      lib.check(lib.dll.odt_probe_mfma_bf16(dev, 100.0, 300.0, lds, C.byref(tf), C.byref(ghz), C.byref(ms), C.byref(n)))
      print("%-28s %12.1f %12.1f %10.3f %9.1f %9d" % (name, tf.value, tf.value / prod, ghz.value, ms.value, n.value))
  print("# datasheet: 2500 TF dense 16-bit MFMA at 2.4 GHz (MI355X_MICROARCH.md); f32 work = 16-bit rate / products per MAC (6 bf16x3, 3 fp16x2)")

if __name__ == "__main__":
  main()
