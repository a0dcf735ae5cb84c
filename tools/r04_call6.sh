#!/bin/bash
# round 4, GPU call 6: the long leg of the digest hunt: bench.py's fruit_nerf_big flow repeated until
# 200 k steps are clean or a run leaves the others
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
( time timeout 1100 python tests/diagnostics/bench_flow_digest.py fruit_nerf_big 55 ) > gpurun_out/r04/digest_big_long.log 2>&1
grep -c "run " gpurun_out/r04/digest_big_long.log; grep "differs" gpurun_out/r04/digest_big_long.log | head -3; tail -4 gpurun_out/r04/digest_big_long.log
