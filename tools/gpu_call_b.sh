#!/bin/bash
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_determinism.py "tests/test_gpu_trained_state.py::test_eval_outputs_at_a_trained_state_meet_the_output_bar" -q -rA -p no:cacheprovider > gpurun_out/r03/tests_b.log 2>&1
echo "tests rc $?"
grep -E "passed|failed" gpurun_out/r03/tests_b.log | tail -3
timeout 600 python bench.py --no-quality --no-cpu-baseline > gpurun_out/r03/bench_b.log 2> gpurun_out/r03/bench_b.err
echo "bench rc $?"
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_determinism.py --deselect tests/test_gpu_trained_state.py > gpurun_out/r03/tests_b_all.log 2>&1
echo "all tests rc $?"
tail -3 gpurun_out/r03/tests_b_all.log
