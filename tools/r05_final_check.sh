#!/bin/bash
# What the driver runs at the end of a round, on the final commit: pytest -m gpu -x, smoke(), bench.py --gpus 1 --steps 20 --warmup 5.
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05
timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/final_tests.log 2>&1; echo "pytest -m gpu -x rc $?"; tail -1 $O/final_tests.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/final_smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/final_smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/final_bench.log 2>/dev/null; echo "bench rc $?"
python - <<'P'
import json
d = json.loads([l for l in open('gpurun_out/r05/final_bench.log') if l.startswith('{')][-1])
r = d['roofline']
print(json.dumps({k: d[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'dtype', 'vs_baseline', 'device_allocs_in_window')}))
print('roofline', r['kernel'], r['bound'], r['achieved'], r['frac'], r['avg_launch_ms'], r['traffic'], r['traffic_source'][-40:])
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['kind'])
P
