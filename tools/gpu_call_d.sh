#!/bin/bash
# exchange path's fixed cost on one GPU: single process vs a one-rank RCCL group with the exchange forced on
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
show() { python - "$1" "$2" <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); b=d['breakdown_ms']
print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], ' '.join(f"{k}={v*1e3:.1f}" for k,v in list(b.items())[:14]))
P
}
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-quality > gpurun_out/r03/d_single_$i.log 2>/dev/null; show gpurun_out/r03/d_single_$i.log single
  FNR_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-quality > gpurun_out/r03/d_dist_$i.log 2>/dev/null; show gpurun_out/r03/d_dist_$i.log rccl1
done
