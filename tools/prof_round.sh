#!/bin/bash
# on the GPU box: everything profiles/r03_* is made from.  usage: bash tools/prof_round.sh [round tag, default r03]
# bench lines (default arithmetic, fp32, fruit_nerf_big), rocprofv3 kernel trace + the three PMC passes (separate runs:
# FETCH_SIZE | WRITE_SIZE | SQ counters, each with --kernel-trace only) of ONE bench command, kernel trace of fruit_nerf_big.
TAG=${1:-r04}
mkdir -p /root/repo/gpurun_out/$TAG
OUT=/root/repo/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
python /root/repo/bench.py > $OUT/bench_fruit_nerf.log 2>$OUT/bench_fruit_nerf.err
python /root/repo/bench.py --mlp-precision fp32 --no-cpu-baseline --no-quality > $OUT/bench_fruit_nerf_fp32.log 2>&1
python /root/repo/bench.py --method fruit_nerf_big --no-cpu-baseline > $OUT/bench_fruit_nerf_big.log 2>&1
# the profiled command runs every step with the two HIP streams serialised (FNR_SERIALIZE_STREAMS=1: what bench.py does on the
# steps it brackets with events), so that a kernel's duration is that kernel's; the default two-stream run is traced once
# more for the step timeline (prof_kernel_trace_two_streams.txt)
export FNR_SERIALIZE_STREAMS=1
CMD="python /root/repo/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-quality"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_kt -o p -- $CMD > $OUT/prof_bench.json 2>/tmp/pf_kt.err
python /root/repo/tools/kt_agg.py /tmp/pf_kt/p_kernel_trace.csv > $OUT/prof_kernel_trace_top40.txt
python /root/repo/tools/kt_agg.py /tmp/pf_kt/p_kernel_trace.csv fnr > $OUT/prof_kernel_trace.txt
python /root/repo/tools/kt_step.py /tmp/pf_kt/p_kernel_trace.csv > $OUT/prof_step_timeline.txt
head -60 /tmp/pf_kt/p_kernel_stats.csv > $OUT/prof_kernel_stats_head.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_f -o p -- $CMD > /dev/null 2>&1
python /root/repo/tools/pmc_agg.py /tmp/pf_f/p_counter_collection.csv fnr > $OUT/prof_fetch.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pf_w -o p -- $CMD > /dev/null 2>&1
python /root/repo/tools/pmc_agg.py /tmp/pf_w/p_counter_collection.csv fnr > $OUT/prof_write.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/pf_s -o p -- $CMD > /dev/null 2>&1
python /root/repo/tools/pmc_agg.py /tmp/pf_s/p_counter_collection.csv fnr > $OUT/prof_sq.txt
unset FNR_SERIALIZE_STREAMS
rocprofv3 --kernel-trace --output-format csv -d /tmp/pf_kt2 -o p -- $CMD > $OUT/prof_bench_two_streams.json 2>/dev/null
python /root/repo/tools/kt_agg.py /tmp/pf_kt2/p_kernel_trace.csv fnr > $OUT/prof_kernel_trace_two_streams.txt
python /root/repo/tools/kt_step.py /tmp/pf_kt2/p_kernel_trace.csv > $OUT/prof_step_timeline_two_streams.txt
export FNR_SERIALIZE_STREAMS=1
rocprofv3 --kernel-trace --output-format csv -d /tmp/pf_big -o p -- python /root/repo/bench.py --method fruit_nerf_big --steps 40 --warmup 10 --no-cpu-baseline --no-quality > $OUT/prof_bench_big.json 2>/dev/null
python /root/repo/tools/kt_agg.py /tmp/pf_big/p_kernel_trace.csv fnr > $OUT/prof_kernel_trace_big.txt
unset FNR_SERIALIZE_STREAMS
ls -la $OUT
