run() { env $1 timeout 400 python bench.py --no-extras --no-cpu-baseline --no-d7 --batch $2 --streams $3 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%.2f FPS  %.3f ms/step verified %s fused %s' % (d['value'], d['ms_per_step'], d.get('verified'), d.get('handle',{}).get('bottleneck_tails_fused')))"; }
echo "b8 s1: $(run X=1 8 1)"
echo "b8 s2: $(run X=1 8 2)"
echo "b4 s1 default policy: $(run X=1 4 1)"
echo "b4 s2 default policy: $(run X=1 4 2)"
echo "b4 s1 mintiles3=100: $(run ODT_CONV_SPLIT3_MINTILES=100 4 1)"
echo "b4 s2 mintiles3=100: $(run ODT_CONV_SPLIT3_MINTILES=100 4 2)"
echo "b2 s4 mintiles3=50: $(run ODT_CONV_SPLIT3_MINTILES=50 2 4)"
