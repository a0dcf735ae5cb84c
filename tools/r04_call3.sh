#!/bin/bash
# round 4, GPU call 3: the two fixed tests, a kernel-trace timeline of the exchange path (one-rank RCCL) next to the single
# process, host enqueue after the host-side fixes
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
R=/root/repo
cd $R && timeout 600 python -m pytest tests/test_gpu_cloud.py tests/test_gpu_trained_state.py tests/test_gpu_distributed.py -m gpu -q -p no:cacheprovider -rA -k "end_to_end or identical_samples or two_stream" > gpurun_out/r04/tests_3.log 2>&1
echo "tests rc $?"; grep -E "^(FAILED|ERROR)|passed|failed|\[count\]|worst gradient" gpurun_out/r04/tests_3.log | tail -12
cd /tmp
for m in single exchange; do
  if [ $m = exchange ]; then export FNR_BENCH_FORCE_DIST=1; else unset FNR_BENCH_FORCE_DIST; fi
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$m -o p -- python $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-quality --no-big > $R/gpurun_out/r04/kt_${m}_bench.log 2>/dev/null
  f=$(find /tmp/kt_$m -name "*kernel_trace.csv" | head -1)
  python $R/tools/kt_step.py $f 40 > $R/gpurun_out/r04/kt_${m}_step.txt 2>&1
  tail -1 $R/gpurun_out/r04/kt_${m}_step.txt
done
unset FNR_BENCH_FORCE_DIST
cd $R && timeout 200 python tools/host_profile.py 300 2>/dev/null | grep "host enqueue"
