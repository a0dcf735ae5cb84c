#!/bin/bash
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r03/tests_g.log 2>&1
echo "tests rc $?"; tail -5 gpurun_out/r03/tests_g.log
bash tools/ab.sh 2 2>&1 | tee gpurun_out/r03/ab_g.log
