#!/bin/bash
# Round 5 A/B (one box, back to back): scheduling knobs for the two-stream step written at the end of round 4 — same
# launches, nothing to validate but the clock.  bench.py --no-cpu-baseline --no-quality, 200-step windows.
#   FNR_PROP_BWD_WGS_PER_CU = 1 | 2   fewer persistent workgroups of the proposal backward (second stream)
#   FNR_BENCH_MAIN_PRIORITY = high    the launch stream on a high-priority hardware queue
# usage: bash tools/r05_ab_schedule.sh [reps = 1]   (5 bench runs of ~1.2 min per rep)
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
run() { env "$@" python bench.py --no-cpu-baseline --no-quality --no-big 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$*', d['value'], d['ms_per_step'])"; }
for rep in $(seq ${1:-1}); do
  run A=default
  run FNR_PROP_BWD_WGS_PER_CU=2
  run FNR_PROP_BWD_WGS_PER_CU=1
  run FNR_BENCH_MAIN_PRIORITY=high
  run FNR_BENCH_MAIN_PRIORITY=high FNR_PROP_BWD_WGS_PER_CU=2
done | tee gpurun_out/r05/ab_schedule.log
