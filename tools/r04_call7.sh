#!/bin/bash
# round 4, GPU call 7: localise the rare fruit_nerf_big divergence (which tensors move first) + the GPU suite with sparse-touch
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r04/tests_7.log 2>&1
echo "gpu tests rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r04/tests_7.log | tail -5
( time timeout 840 python tests/diagnostics/digest_localize.py fruit_nerf_big 48 3000 50 ) > gpurun_out/r04/digest_localize.log 2>&1
grep -E "DIFFERS|tensors|and at|reference" gpurun_out/r04/digest_localize.log | cut -c1-600 | head -20; tail -4 gpurun_out/r04/digest_localize.log | cut -c1-200
