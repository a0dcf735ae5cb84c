#!/bin/bash
# round 4, GPU call 10: the per-step divergence hunt with EVERYTHING on one stream (FNR_OVERLAP_PROPOSAL_BACKWARD=0): do the
# events need the second stream?
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
( time FNR_OVERLAP_PROPOSAL_BACKWARD=0 timeout 760 python tests/diagnostics/digest_perstep.py fruit_nerf_big 44 3000 ) > gpurun_out/r04/digest_perstep_one_stream.log 2>&1
grep -E "DIFFERS|   step|      |reference|overlap" gpurun_out/r04/digest_perstep_one_stream.log | cut -c1-300 | head -30; grep -c identical gpurun_out/r04/digest_perstep_one_stream.log; tail -3 gpurun_out/r04/digest_perstep_one_stream.log
