#!/bin/bash
# on the GPU box: kernel trace + PMC passes of the bench command, aggregated to text under gpurun_out/
mkdir -p /root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-quality"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_kt -o p -- $CMD > /root/repo/gpurun_out/prof_bench.json 2>/tmp/pf_kt.err
python /root/repo/tools/kt_agg.py /tmp/pf_kt/p_kernel_trace.csv > /root/repo/gpurun_out/prof_kernel_trace.txt
python /root/repo/tools/kt_step.py /tmp/pf_kt/p_kernel_trace.csv > /root/repo/gpurun_out/prof_step_timeline.txt
head -40 /tmp/pf_kt/p_kernel_stats.csv > /root/repo/gpurun_out/prof_kernel_stats_head.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_f -o p -- $CMD > /dev/null 2>&1
python /root/repo/tools/pmc_agg.py /tmp/pf_f/p_counter_collection.csv fnr:: > /root/repo/gpurun_out/prof_fetch.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pf_w -o p -- $CMD > /dev/null 2>&1
python /root/repo/tools/pmc_agg.py /tmp/pf_w/p_counter_collection.csv fnr:: > /root/repo/gpurun_out/prof_write.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/pf_s -o p -- $CMD > /dev/null 2>&1
python /root/repo/tools/pmc_agg.py /tmp/pf_s/p_counter_collection.csv fnr:: > /root/repo/gpurun_out/prof_sq.txt
ls -la /root/repo/gpurun_out/prof_*
