#!/bin/bash
# one gpurun call of a kernel-tuning iteration: the GPU suite (stop at the first failure), then tools/ab.sh — bench.py of
# ab_prev/ (a built copy of the previous commit) against the working tree on the SAME box.
#   gpurun --timeout 1800 -- 'bash tools/gpu_tests_and_ab.sh [pytest args]'
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 1200 python -m pytest ${@:-tests -m gpu} -q -x -p no:cacheprovider > gpurun_out/r03/tests_iter.log 2>&1
echo "tests rc $?"; tail -3 gpurun_out/r03/tests_iter.log
[ -d ab_prev ] && bash tools/ab.sh 3 2>&1 | tee gpurun_out/r03/ab_iter.log
