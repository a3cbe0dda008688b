#!/usr/bin/env python
"""Per-layer roofline table from a tools/profile_layers.py table (profiles/r01_conv_layers_*.txt).

For every conv shape: the measured time next to the two ceilings -- matrix pipe (dense 16-bit MFMA peak / 6
products for the bf16x3 kernels' layers, / 3 for the fp16x2 kernels', f32 MFMA peak for the exact-f32 kernel's) and HBM (algorithmic
bytes: input + output activations once, weights once, residual once; 8 TB/s) -- and which one binds.
  python tools/roofline_table.py profiles/r01_conv_layers_b8_v11.txt
"""
import sys

F32_PEAK, SPLIT_PEAK, H2_PEAK, HBM = 157.3e12, 2500e12 / 6, 2500e12 / 3, 8.0e12


def main(path):
  rows = []
  for line in open(path):
    f = line.split()
    if len(f) == 8 and f[1].isdigit():
      name, n, M, N, K, ms, tf = f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4]), float(f[5]), float(f[6])
      rows.append((name, n, M, N, K, ms, tf))
  print("%-44s %4s %9s %6s %6s %8s %8s %8s %8s  %-5s %6s" %
        ("layer (first of its shape)", "n", "M", "N", "K", "ms", "TF", "mfma ms", "hbm ms", "bound", "frac"))
  tot = [0.0, 0.0]
  for name, n, M, N, K, ms, tf in sorted(rows, key=lambda r: -r[5]):
    split, h2 = name.endswith("[bf16x3]"), name.endswith("[fp16x2]")
    base = name.replace("[bf16x3]", "").replace("[fp16x2]", "")
    fused3 = base.endswith("+conv3")            # conv2 with the block's conv3 (1x1 to 4 N channels + shortcut) evaluated in its kernel
    pooled = base.endswith("+pool0")              # conv0 with pool0 evaluated in its kernel: a quarter of the map is written, the map itself never
    base = base.replace("+conv3", "").replace("+head", "").replace("+pool0", "")
    k3 = base.endswith("conv2") or "posthoc_3x3" in base or base.startswith("rpn/conv0")
    cin_bytes = M * (K // 9 if k3 else K) * 4.0            # every input pixel once (stride-1 3x3: K/9 channels)
    if base == "conv0":
      cin_bytes = M * 4 * 4 * 4.0                             # 7x7 s2 over the 4-channel padded frame
    byt = cin_bytes + M * N * 4.0 * (0.25 if pooled else 1.0) + N * K * 4.0
    if "conv3" in base or "lateral" in base:
      byt += M * N * 4.0 * (0.25 if "lateral" in base else 1.0)   # residual (2x-upsampled: a quarter)
    flops = 2.0 * M * N * K
    if fused3:
      # the [M, N] tensor between the two convs is neither written nor read; the 1x1 conv's output and residual (4 N wide) are
      byt = cin_bytes + N * K * 4.0 + 4 * N * N * 4.0 + 2 * M * 4 * N * 4.0
      flops += 2.0 * M * N * 4 * N
    t_m = n * flops / (H2_PEAK if h2 else (SPLIT_PEAK if split else F32_PEAK)) * 1e3
    t_h = n * byt / HBM * 1e3
    bound = max(t_m, t_h)
    tot[0] += ms; tot[1] += bound
    print("%-44s %4d %9d %6d %6d %8.3f %8.1f %8.3f %8.3f  %-5s %6.2f" %
          (name[:44], n, M, N, K, ms, tf, t_m, t_h, "mfma" if t_m >= t_h else "hbm", bound / ms))
  print("total measured %.2f ms, sum of per-layer ceilings %.2f ms (%.0f %%)" % (tot[0], tot[1], 100 * tot[1] / tot[0]))


if __name__ == "__main__":
  main(sys.argv[1])
