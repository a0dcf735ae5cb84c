#!/bin/bash
# on the GPU box: rocprofv3 kernel trace of the bench command in each MLP mode -> gpurun_out/kt_<mode>.txt
mkdir -p /root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in ${MODES:-fp32 bf16x3 bf16}; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$m -o p -- python /root/repo/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-quality --mlp-precision $m ${BENCH_ARGS} > /root/repo/gpurun_out/kt_bench_$m.json 2>/tmp/kt_$m.err
  python /root/repo/tools/kt_agg.py /tmp/kt_$m/p_kernel_trace.csv fnr:: > /root/repo/gpurun_out/kt_$m.txt
done
