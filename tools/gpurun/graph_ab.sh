timeout 900 python -m pytest tests/test_efficientnet.py tests/test_ingest.py tests/test_e2e.py -m gpu -q -p no:cacheprovider -k "not b8_1080p" 2>&1 | tail -3
for g in 0 1; do
  echo "--- ODT_GRAPH=$g"
  ODT_GRAPH=$g python tools/bench_efficientdet.py --frame 1080x1920 --no-cpu-baseline | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('D7 %.2f FPS %.2f ms host %.2f ms' % (d['value'], d['ms_per_step'], d['extra']['host_to_host_ms']))"
  ODT_GRAPH=$g python tools/bench_efficientdet.py --model efficientdet-d0 --no-cpu-baseline | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('D0 %.2f FPS %.2f ms host %.2f ms' % (d['value'], d['ms_per_step'], d['extra']['host_to_host_ms']))"
  ODT_GRAPH=$g python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FPN b1 %.2f FPS' % d['value'])"
  ODT_GRAPH=$g python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FPN b8 %.2f FPS' % d['value'])"
done
