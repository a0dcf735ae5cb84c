#!/bin/bash
# round 4, GPU call 8: per-step records of the fruit_nerf_big bench flow: at which step, and in which parameters first, does a
# run leave the others?
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
( time timeout 800 python tests/diagnostics/digest_perstep.py fruit_nerf_big 46 3000 ) > gpurun_out/r04/digest_perstep.log 2>&1
grep -E "DIFFERS|   step|      |reference" gpurun_out/r04/digest_perstep.log | cut -c1-300 | head -40; tail -4 gpurun_out/r04/digest_perstep.log | cut -c1-200
