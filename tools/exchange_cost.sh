#!/bin/bash
# the exchange path's fixed cost on ONE GPU: single process vs a one-rank RCCL group with the exchange forced
# on, with / without the deferred field update, and the round-2 schedule of four level groups.
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
show() { python - "$1" "$2" <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); b=d['breakdown_ms']
print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], ' '.join(f"{k}={v*1e3:.1f}" for k,v in list(b.items())[:10]))
P
}
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-quality > gpurun_out/r03/x_single_$i.log 2>/dev/null; show gpurun_out/r03/x_single_$i.log single
  FNR_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-quality > gpurun_out/r03/x_dist_$i.log 2>/dev/null; show gpurun_out/r03/x_dist_$i.log rccl1-deferred
  FNR_BENCH_FORCE_DIST=1 FNR_DEFER_FIELD_UPDATE=0 python bench.py --no-cpu-baseline --no-quality > gpurun_out/r03/x_dist0_$i.log 2>/dev/null; show gpurun_out/r03/x_dist0_$i.log rccl1-nodefer
done
FNR_BENCH_FORCE_DIST=1 FNR_EXCHANGE_LEVEL_GROUPS=4 FNR_DEFER_FIELD_UPDATE=0 python bench.py --no-cpu-baseline --no-quality > gpurun_out/r03/x_dist4.log 2>/dev/null; show gpurun_out/r03/x_dist4.log rccl1-4groups
