#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pf_i -o p -- python /root/repo/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-quality > /dev/null 2>&1
python /root/repo/tools/kt_agg.py /tmp/pf_i/p_kernel_trace.csv fnr > /root/repo/gpurun_out/r03/kt_i.txt
python /root/repo/tools/kt_step.py /tmp/pf_i/p_kernel_trace.csv > /root/repo/gpurun_out/r03/kt_i_step.txt
cd /root/repo
bash tools/ab.sh 2 2>&1 | tee gpurun_out/r03/ab_i.log
