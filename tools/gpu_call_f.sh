#!/bin/bash
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
bash tools/ab.sh 3 2>&1 | tee gpurun_out/r03/ab_f.log
timeout 600 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_distributed.py -q -x -p no:cacheprovider 2>&1 | tail -2
