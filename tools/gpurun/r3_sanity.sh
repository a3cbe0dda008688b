#!/bin/bash
# last sanity of the linked library on the GPU: the fp16x2 property / e2e parity tests
export TMPDIR=/tmp
timeout 40 python -m pytest tests/test_ops.py tests/test_e2e.py -q -m gpu -x -k "equivariance or fp16x2 or cache_hints" 2>&1 | tail -3
