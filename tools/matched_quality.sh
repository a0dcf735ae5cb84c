#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -m tests.quality_matched --side hip --method fruit_nerf --load-state tests/golden/_oracle_trained_1736.pt --count --out gpurun_out/r03/quality_hip_on_oracle_state.json > gpurun_out/r03/quality_hip_on_oracle_state.log 2>&1
tail -3 gpurun_out/r03/quality_hip_on_oracle_state.log
python -m tests.quality_matched --side hip --method fruit_nerf --steps 1736 --count --out gpurun_out/r03/quality_hip.json > gpurun_out/r03/quality_hip.log 2>&1
grep "^{" gpurun_out/r03/quality_hip.log
FNR_MLP_PRECISION=fp32 python -m tests.quality_matched --side hip --method fruit_nerf --steps 1736 --count --out gpurun_out/r03/quality_hip_fp32.json > gpurun_out/r03/quality_hip_fp32.log 2>&1
grep "^{" gpurun_out/r03/quality_hip_fp32.log
