#!/bin/bash
for d in 0 $((256+512*2)) $((256+512*4)) $((256+512*6)); do
  echo "--- ODT_CONV_DEBUG=$d"; ODT_CONV_DEBUG=$d timeout 100 python tools/conv_trace.py conv3 conv3nores 2>&1 | grep -E "^==|per block avg"  | cut -c1-230
done
