#!/bin/bash
# round 4, GPU call 4: does the second stream get its own hardware queue on the exchange path?  one-rank RCCL bench with the
# side stream at normal / high priority and with GPU_MAX_HW_QUEUES=8, + a kernel-trace timeline of the default
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
R=/root/repo
show() { python - "$1" "$2" <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])
P
}
cd $R
python bench.py --no-cpu-baseline --no-quality --no-big > gpurun_out/r04/y_single.log 2>/dev/null; show gpurun_out/r04/y_single.log single
for v in normal high; do
  FNR_SIDE_STREAM_PRIORITY=$v FNR_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-quality --no-big > gpurun_out/r04/y_dist_$v.log 2>/dev/null; show gpurun_out/r04/y_dist_$v.log rccl1-side-$v
done
GPU_MAX_HW_QUEUES=8 FNR_SIDE_STREAM_PRIORITY=normal FNR_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-quality --no-big > gpurun_out/r04/y_dist_q8.log 2>/dev/null; show gpurun_out/r04/y_dist_q8.log rccl1-normal-8queues
FNR_SIDE_STREAM_PRIORITY=high python bench.py --no-cpu-baseline --no-quality --no-big > gpurun_out/r04/y_single_high.log 2>/dev/null; show gpurun_out/r04/y_single_high.log single-side-high
cd /tmp
FNR_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_x -o p -- python $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-quality --no-big > /dev/null 2>&1
python $R/tools/kt_step.py $(find /tmp/kt_x -name "*kernel_trace.csv" | head -1) 43 > $R/gpurun_out/r04/kt_exchange_high_step.txt 2>&1; tail -1 $R/gpurun_out/r04/kt_exchange_high_step.txt
