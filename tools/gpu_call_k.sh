#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
bash tools/prof_round.sh r03 > gpurun_out/r03/prof_round.out 2>&1
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03/gputest_final.log 2>&1
tail -3 gpurun_out/r03/gputest_final.log
tail -4 gpurun_out/r03/prof_round.out
