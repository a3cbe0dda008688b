#!/bin/bash
# last sanity of the linked library on the GPU: the fp16x2 e2e parity tests + one 3x3 / 1x1 op case per tile shape
export TMPDIR=/tmp
timeout 50 python -m pytest tests/test_e2e.py tests/test_ops.py -q -m gpu -x -k "fp16x2 or (conv2d_split and 2/256 and (case0 or case1 or case7 or case12))" 2>&1 | tail -3
