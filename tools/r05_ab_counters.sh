#!/bin/bash
# Round 5: what the accumulate kernel's variants cost (same box, back to back; build them first on the CPU side with
# tools/build_hunt_variants.sh).  bench.py --no-cpu-baseline --no-quality --no-big, default 200-step windows.
#   default     every wave waits for its counter loads ahead of the reset barrier (the fix of round 4's divergence)
#   zero_early  + the accumulator zeroed while those loads are in flight, one barrier fewer
#   rmw         counters only ever touched by device-scope atomic read-modify-writes (thread 0 + LDS broadcast)
# usage: bash tools/r05_ab_counters.sh [reps = 1] [method = fruit_nerf]
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
V=$PWD/fruitnerf_amd/lib/variants
REPS=${1:-1}; METHOD=${2:-fruit_nerf}
run() { label=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-quality --no-big --method $METHOD 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$label $METHOD', d['value'], d['ms_per_step'], r['kernel'], r.get('avg_launch_ms'))"; }
for rep in $(seq $REPS); do
  run default A=1
  run zero_early FNR_LIB_PATH=$V/zero_early/libfruitnerf_hip.so
  run rmw FNR_LIB_PATH=$V/rmw_counters/libfruitnerf_hip.so
done | tee gpurun_out/r05/ab_counters.log
