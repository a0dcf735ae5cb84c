#!/bin/bash
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_training_parity.py tests/test_gpu_determinism.py -q -x -p no:cacheprovider > gpurun_out/r03/tests_j.log 2>&1
echo "tests rc $?"; tail -3 gpurun_out/r03/tests_j.log
bash tools/ab.sh 3 2>&1 | tee gpurun_out/r03/ab_j.log
