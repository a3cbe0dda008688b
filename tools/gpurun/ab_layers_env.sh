# per-layer in-network comparison of environment knobs: BATCH=8 bash tools/gpurun/ab_layers_env.sh "ENV_A" "ENV_B" ...
i=0
for v in "$@"; do
  env $v timeout 300 python tools/profile_layers.py --batch ${BATCH:-8} --steps 3 2>&1 | tail -40 > gpurun_out/layers_env_$i.txt
  i=$((i+1))
done
python - "$@" <<'PY'
import sys
vs = sys.argv[1:]
rows = {}
for i, v in enumerate(vs):
  for line in open("gpurun_out/layers_env_%d.txt" % i):
    f = line.split()
    if len(f) == 8 and f[1].isdigit():
      rows.setdefault((f[0], f[2], f[3], f[4]), {})[i] = (float(f[5]), float(f[6]))
    elif line.startswith("conv total"):
      print("[%s]" % v, line.strip())
print("%-34s %8s %5s %6s " % ("layer", "M", "N", "K") + " ".join("%10s" % ("ms[%d]" % i) for i in range(len(vs))) + " " + " ".join("%8s" % ("TF[%d]" % i) for i in range(len(vs))))
# (a knob may rename / regroup layers -- fused launches, other shape classes: a row missing under one setting prints blanks)
for k, d in sorted(rows.items(), key=lambda kv: -max(x[0] for x in kv[1].values())):
  print("%-34s %8s %5s %6s " % k + " ".join("%10.3f" % d[i][0] if i in d else "%10s" % "-" for i in range(len(vs))) + " " +
        " ".join("%8.1f" % d[i][1] if i in d else "%8s" % "-" for i in range(len(vs))))
PY
