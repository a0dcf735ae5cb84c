#!/bin/bash
# round 3, first GPU call: the new parity tests + the reworked bench line
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_trained_state.py "tests/test_gpu_training_parity.py::test_losses_and_all_gradients_at_the_real_configuration" tests/test_gpu_distributed.py -q -rA -p no:cacheprovider > gpurun_out/r03/tests_a.log 2>&1
echo "tests rc $?"
grep -E "passed|failed" gpurun_out/r03/tests_a.log | tail -3
timeout 600 python bench.py > gpurun_out/r03/bench_a.log 2> gpurun_out/r03/bench_a.err
echo "bench rc $?"
tail -c 600 gpurun_out/r03/bench_a.err
