#!/bin/bash
# round 4, GPU call 14 (the round's last minutes): the per-step hunt with the scatter's INPUT in the record (FNR_DIGEST_WS=1:
# checksum of each proposal level's d_feats region) under AMD_OPT_FLUSH=0, where the events were twice as frequent
cd /root/repo; mkdir -p gpurun_out/r04
export TMPDIR=/tmp
( time FNR_DIGEST_WS=1 AMD_OPT_FLUSH=0 timeout 335 python tests/diagnostics/digest_perstep.py fruit_nerf_big 40 3000 ) > gpurun_out/r04/digest_perstep_ws.log 2>&1
grep -E "DIFFERS|   step|      " gpurun_out/r04/digest_perstep_ws.log | cut -c1-300 | head -40; grep -c identical gpurun_out/r04/digest_perstep_ws.log; tail -4 gpurun_out/r04/digest_perstep_ws.log | cut -c1-200
