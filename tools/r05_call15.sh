#!/bin/bash
# Round 5, GPU call 15: the two-stream soak once more on the shipped library, at the HIP runtime's default flush setting
# (calls 2 and 11 ran it under AMD_OPT_FLUSH=0): fruit_nerf_big, 3000 steps per run.
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
( time timeout 900 python tests/diagnostics/digest_perstep.py fruit_nerf_big ${1:-42} 3000 ) > gpurun_out/r05/soak_default_flush.log 2>&1
grep -E "DIFFERS|   step" gpurun_out/r05/soak_default_flush.log | cut -c1-300 | head -20
echo "identical runs: $(grep -c identical gpurun_out/r05/soak_default_flush.log)"; tail -4 gpurun_out/r05/soak_default_flush.log | cut -c1-200
