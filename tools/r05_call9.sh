#!/bin/bash
# Round 5, GPU call 9: are the determinism tests stable on the in-tree library (streaming hints, no paired stores)?
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
for rep in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_determinism.py -m gpu -q -p no:cacheprovider > $O/tests_9_det_$rep.log 2>&1
  echo "rep $rep rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_9_det_$rep.log | tail -6
done
