#!/bin/bash
# round 4, GPU call 1: the GPU suite with the tightened trained-state bars + same-samples gradient leg + stream-safe test,
# the emit-scan A/B (select form vs v_fmac_f32_dpp) on one box, and the bench-flow digest hunt for the stray fruit_nerf_big run
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -rA > gpurun_out/r04/tests_1.log 2>&1
echo "gpu tests rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r04/tests_1.log | tail -15
bash tools/ab_lib.sh scan_select default 3 2>&1 | tee gpurun_out/r04/ab_emit.log
( time timeout 400 python tests/diagnostics/bench_flow_digest.py fruit_nerf_big 10 ) > gpurun_out/r04/digest_big.log 2>&1
tail -12 gpurun_out/r04/digest_big.log
( time FNR_STREAM_SAFE=1 timeout 120 python tests/diagnostics/bench_flow_digest.py fruit_nerf_big 2 ) > gpurun_out/r04/digest_big_stream_safe.log 2>&1
tail -4 gpurun_out/r04/digest_big_stream_safe.log
