#!/bin/bash
# on the GPU box: rocprofv3 kernel trace of the counting-stage front-end microbench, aggregated to text
mkdir -p /root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o p -- python /root/repo/tools/microbench/cloud_time.py 80 > /root/repo/gpurun_out/cloud_time.txt 2>/dev/null
python /root/repo/tools/kt_agg.py /tmp/pc/p_kernel_trace.csv > /root/repo/gpurun_out/cloud_kernel_trace.txt
