#!/bin/bash
# round 4, GPU call 5: accumulate sweep-preload A/B, emit levels-per-workgroup, the exchange path after the host-order fix
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
R=/root/repo
cd $R
bash tools/ab_lib.sh default sweep_preload 3 2>&1 | tee gpurun_out/r04/ab_sweep.log
for l in 2 8; do
  FNR_EMIT_LPB=$l python bench.py --no-cpu-baseline --no-quality 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']
print('lpb $l', d['value'], d['ms_per_step'], ' '.join(f'{k}={v*1e3:.1f}' for k,v in list(b.items())[:6]))" | tee -a gpurun_out/r04/ab_sweep.log
done
show() { python - "$1" "$2" <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])
P
}
python bench.py --no-cpu-baseline --no-quality --no-big > gpurun_out/r04/z_single.log 2>/dev/null; show gpurun_out/r04/z_single.log single
FNR_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-quality --no-big > gpurun_out/r04/z_dist.log 2>/dev/null; show gpurun_out/r04/z_dist.log rccl1-8queues-new-order
FNR_BENCH_FORCE_DIST=1 FNR_EXCHANGE_LEVEL_GROUPS=4 python bench.py --no-cpu-baseline --no-quality --no-big > gpurun_out/r04/z_dist4.log 2>/dev/null; show gpurun_out/r04/z_dist4.log rccl1-8queues-4groups
timeout 600 python -m pytest tests/test_gpu_distributed.py -m gpu -q -p no:cacheprovider > gpurun_out/r04/tests_5.log 2>&1; tail -2 gpurun_out/r04/tests_5.log
