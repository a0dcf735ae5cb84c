#!/bin/bash
# Round 5, GPU call 5: the round's profile (tools/prof_round.sh: bench lines, kernel trace, the three PMC passes, the
# fruit_nerf_big trace) on the tree that ships, then the GPU suite.
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
bash tools/prof_round.sh r05 > gpurun_out/r05/prof_round.out 2>&1
tail -25 gpurun_out/r05/prof_round.out
cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05/tests_5.log 2>&1
echo "gpu tests rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r05/tests_5.log | tail -12
