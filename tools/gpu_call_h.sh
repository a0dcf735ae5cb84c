#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
bash tools/prof_round.sh r03 > gpurun_out/r03/prof_round.out 2>&1
cd /root/repo
python -m tests.quality_matched --side hip --method fruit_nerf_big --rays 1024 --steps 300 --eval-at 100,200,300 --eval-pixels 8192 --out gpurun_out/r03/quality_big_hip.json > gpurun_out/r03/quality_big_hip.log 2>&1
FNR_MLP_PRECISION=fp32 python -m tests.quality_matched --side hip --method fruit_nerf_big --rays 1024 --steps 300 --eval-at 100,200,300 --eval-pixels 8192 --out gpurun_out/r03/quality_big_hip_fp32.json > gpurun_out/r03/quality_big_hip_fp32.log 2>&1
tail -3 gpurun_out/r03/quality_big_hip.log gpurun_out/r03/quality_big_hip_fp32.log
tail -5 gpurun_out/r03/prof_round.out
