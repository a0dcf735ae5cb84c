#!/bin/bash
# on the GPU box: kernel traces of the local and the (one-rank RCCL) exchange step, aggregated to text
mkdir -p /root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in local exchange; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/px_$m -o p -- python /root/repo/tools/microbench/rccl_single_rank.py $m 2>/dev/null | grep "ms/step" > /root/repo/gpurun_out/px_$m.txt
  python /root/repo/tools/kt_agg.py /tmp/px_$m/p_kernel_trace.csv >> /root/repo/gpurun_out/px_$m.txt
done
