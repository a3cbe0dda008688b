for v in $VARIANTS; do
  cp ab/$v.so object_detection_tracking_amd/libodt_hip.so
  timeout 300 python tools/profile_layers.py --batch ${BATCH:-8} --steps 3 2>&1 | tail -40 > gpurun_out/layers_$v.txt
done
python - <<'PY'
import os
vs = os.environ["VARIANTS"].split()
rows = {}
for v in vs:
  for line in open("gpurun_out/layers_%s.txt" % v):
    f = line.split()
    if len(f) == 8 and f[1].isdigit():
      rows.setdefault((f[0], f[2], f[3], f[4]), {})[v] = (float(f[5]), float(f[6]))
    elif line.startswith("conv total"):
      print(v, line.strip())
print("%-36s %9s %6s %6s " % ("layer", "M", "N", "K") + " ".join("%9s" % (v + " ms") for v in vs) + " " + " ".join("%7s" % (v + " TF") for v in vs))
for k, d in sorted(rows.items(), key=lambda kv: -kv[1][vs[0]][0]):
  print("%-36s %9s %6s %6s " % k + " ".join("%9.3f" % d[v][0] for v in vs) + " " + " ".join("%7.1f" % d[v][1] for v in vs))
PY
