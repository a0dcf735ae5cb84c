#!/bin/bash
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_determinism.py "tests/test_gpu_training_parity.py" tests/test_gpu_reference_pins.py -q -x -rA -p no:cacheprovider > gpurun_out/r03/tests_c.log 2>&1
echo "tests rc $?"
grep -E "passed|failed" gpurun_out/r03/tests_c.log | tail -3
grep -E "input grad in mlp|determinism" gpurun_out/r03/tests_c.log | grep -v print | head -20
bash tools/ab.sh 3 2>&1 | tee gpurun_out/r03/ab_c.log
