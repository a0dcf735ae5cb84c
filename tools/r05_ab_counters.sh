#!/bin/bash
# Round 5: what the counter-path fix candidates cost (same box, back to back; build them first on the CPU side with
# tools/build_hunt_variants.sh).  bench.py --no-cpu-baseline --no-quality, default 200-step windows, both methods.
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
V=$PWD/fruitnerf_amd/lib/variants
run() { label=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-quality --no-big ${METHOD:+--method $METHOD} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$label ${METHOD:-fruit_nerf}', d['value'], d['ms_per_step'], r['kernel'], r.get('avg_launch_ms'))"; }
for METHOD in "" fruit_nerf_big; do
  for rep in 1 2; do
    run default A=1
    run atomic FNR_LIB_PATH=$V/atomic_counters/libfruitnerf_hip.so
    run rmw FNR_LIB_PATH=$V/rmw_counters/libfruitnerf_hip.so
    run memset FNR_SCATTER_MEMSET=1
    run zero_early FNR_LIB_PATH=$V/zero_early/libfruitnerf_hip.so
  done
done | tee gpurun_out/r05/ab_counters.log
