#!/bin/bash
# Round 5, GPU call 12: the round's profile on the tree that ships (tools/prof_round.sh r05: bench lines, kernel trace, PMC
# passes, fruit_nerf_big trace) and the driver's own command.
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
bash tools/prof_round.sh r05 > gpurun_out/r05/prof_round.out 2>&1
cd /root/repo
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_driver_window.log 2>/dev/null
python - <<'P'
import json
for f in ('bench_fruit_nerf.log', 'bench_driver_window.log'):
    d = json.loads([l for l in open('gpurun_out/r05/' + f) if l.startswith('{')][-1])
    r = d['roofline']
    print(f, d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], 'allocs', d.get('device_allocs_in_window'),
          r['kernel'], r['frac'], r['avg_launch_ms'], d['cpu_baseline'] and d['cpu_baseline']['value'])
P
