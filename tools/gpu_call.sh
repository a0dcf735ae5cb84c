#!/bin/bash
# ONE runner for everything that goes to the GPU box through gpurun (replaces the per-call scripts of rounds 4 - 5).
#   gpurun --timeout 1500 -- 'bash tools/gpu_call.sh <tag> <leg> [<leg> ...]'
# Every leg writes gpurun_out/<tag>/<leg name>.log (merged back by gpurun) and prints one summary line.  Legs:
#   tests[=<pytest args>]     pytest -m gpu (default: the whole suite, -x)
#   smoke                     __graft_entry__.build() + smoke()
#   bench[=<bench.py args>]   python bench.py <args>            (log name: bench, bench2, ... in order of appearance)
#   driver                    the driver's command: bench.py --gpus 1 --steps 20 --warmup 5
#   ab=<ENV=a|b>[,<bench args>]   tools/ab_quick.py-style same-process A/B is python-side; this leg runs bench.py twice with ENV
#                             set to a, then b, on the quick settings (--no-cpu-baseline --no-quality), alternating 2x
#   kt[=<bench.py args>]      rocprofv3 --kernel-trace --stats of a short serialised-streams bench run -> per-kernel table
#   kt2[=<bench.py args>]     the same with the default two streams (timeline)
#   ktbig                     kernel trace of the fruit_nerf_big method (serialised streams) -> prof_kernel_trace_big.txt
#   pmc[=<bench.py args>]     the three counter passes (FETCH_SIZE | WRITE_SIZE | SQ_*), each its own run with --kernel-trace only
#   ktpy=<script and args>    rocprofv3 --kernel-trace of python <script and args> -> per-kernel table
#   pmcpy=<script and args>   SQ counter passes over python <script and args> -> per-kernel counter means
#   py=<script and args>      python <script and args>
#   sh=<command>              bash -c <command>
TAG=$1; shift
cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out/$TAG; mkdir -p $O
QUICK="--no-cpu-baseline --no-quality"
nb=0
for leg in "$@"; do
  name=${leg%%=*}; arg=""; [[ "$leg" == *=* ]] && arg=${leg#*=}
  t0=$(date +%s)
  case $name in
    tests)
      [ -z "$arg" ] && arg="tests/ -x"
      timeout 1700 python -m pytest $arg -q -m gpu -p no:cacheprovider > $O/tests$nb.log 2>&1; rc=$?
      echo "[tests $arg] rc $rc: $(tail -1 $O/tests$nb.log)"; nb=$((nb+1));;
    smoke)
      python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "[smoke] rc $?: $(tail -1 $O/smoke.log)";;
    driver)
      python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver.log 2>$O/driver.err; echo "[driver] rc $?"
      python tools/bench_line.py $O/driver.log;;
    bench)
      n=bench$nb; nb=$((nb+1))
      python bench.py $arg > $O/$n.log 2>$O/$n.err; echo "[bench $arg] rc $? -> $n.log"
      python tools/bench_line.py $O/$n.log;;
    ab)
      envs=${arg%%,*}; extra=""; [[ "$arg" == *,* ]] && extra=${arg#*,}
      var=${envs%%=*}; vals=${envs#*=}; a=${vals%%|*}; b=${vals#*|}
      for rep in 1 2; do for v in "$a" "$b"; do
        env $var=$v python bench.py $QUICK $extra > $O/ab_${var}_${v}_$rep.log 2>/dev/null
        echo "[ab $var=$v #$rep] $(python tools/bench_line.py $O/ab_${var}_${v}_$rep.log --short)"
      done; done;;
    kt|kt2)
      [ -z "$arg" ] && arg="--steps 60 --warmup 10 $QUICK"
      [ $name = kt ] && export FNR_SERIALIZE_STREAMS=1
      (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$name -o p -- python /root/repo/bench.py $arg > $O/${name}_bench.json 2>/tmp/pf_$name.err)
      unset FNR_SERIALIZE_STREAMS
      # (file names as tools/make_profile_summary.py reads them from profiles/<tag>_raw/)
      sfx=""; [ $name = kt2 ] && sfx="_two_streams"
      cp $O/${name}_bench.json $O/prof_bench$sfx.json
      python tools/kt_agg.py /tmp/pf_$name/p_kernel_trace.csv fnr > $O/prof_kernel_trace$sfx.txt 2>&1
      python tools/kt_agg.py /tmp/pf_$name/p_kernel_trace.csv > $O/prof_kernel_trace_top40$sfx.txt 2>&1
      python tools/kt_step.py /tmp/pf_$name/p_kernel_trace.csv > $O/prof_step_timeline$sfx.txt 2>&1
      head -60 /tmp/pf_$name/p_kernel_stats.csv > $O/prof_kernel_stats_head$sfx.csv 2>/dev/null
      echo "[$name] $(wc -l < $O/prof_kernel_trace$sfx.txt) kernel rows";;
    ktbig)
      export FNR_SERIALIZE_STREAMS=1
      (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pf_big -o p -- python /root/repo/bench.py --method fruit_nerf_big --steps 40 --warmup 10 $QUICK > $O/prof_bench_big.json 2>/dev/null)
      unset FNR_SERIALIZE_STREAMS
      python tools/kt_agg.py /tmp/pf_big/p_kernel_trace.csv fnr > $O/prof_kernel_trace_big.txt 2>&1
      echo "[ktbig] $(wc -l < $O/prof_kernel_trace_big.txt) kernel rows";;
    pmc)
      [ -z "$arg" ] && arg="--steps 60 --warmup 10 $QUICK"
      export FNR_SERIALIZE_STREAMS=1
      (cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_f -o p -- python /root/repo/bench.py $arg > /dev/null 2>&1)
      python tools/pmc_agg.py /tmp/pf_f/p_counter_collection.csv fnr > $O/prof_fetch.txt 2>&1
      (cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pf_w -o p -- python /root/repo/bench.py $arg > /dev/null 2>&1)
      python tools/pmc_agg.py /tmp/pf_w/p_counter_collection.csv fnr > $O/prof_write.txt 2>&1
      (cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/pf_s -o p -- python /root/repo/bench.py $arg > /dev/null 2>&1)
      python tools/pmc_agg.py /tmp/pf_s/p_counter_collection.csv fnr > $O/prof_sq.txt 2>&1
      unset FNR_SERIALIZE_STREAMS
      echo "[pmc] fetch $(wc -l < $O/prof_fetch.txt) write $(wc -l < $O/prof_write.txt) sq $(wc -l < $O/prof_sq.txt) rows";;
    ktpy)   # rocprofv3 kernel trace of a python script -> per-kernel table ktpy<n>.txt
      n=ktpy$nb; nb=$((nb+1))
      (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pf_$n -o p -- python /root/repo/$arg > $O/$n.out 2>/tmp/pf_$n.err)
      python tools/kt_agg.py /tmp/pf_$n/p_kernel_trace.csv fnr > $O/$n.txt 2>&1
      echo "[ktpy $arg] $(wc -l < $O/$n.txt) kernel rows -> $n.txt";;
    pmcpy)  # two SQ counter passes over a python script (each its own run, --kernel-trace only) -> pmcpy<n>.txt
      n=pmcpy$nb; nb=$((nb+1))
      (cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/pf_${n}a -o p -- python /root/repo/$arg > /dev/null 2>&1)
      python tools/pmc_agg.py /tmp/pf_${n}a/p_counter_collection.csv fnr > $O/$n.txt 2>&1
      (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM --kernel-trace --output-format csv -d /tmp/pf_${n}b -o p -- python /root/repo/$arg > /dev/null 2>&1)
      python tools/pmc_agg.py /tmp/pf_${n}b/p_counter_collection.csv fnr >> $O/$n.txt 2>&1
      echo "[pmcpy $arg] $(wc -l < $O/$n.txt) rows -> $n.txt";;
    py)
      n=$(basename ${arg%% *} .py)
      python $arg > $O/$n.log 2>&1; echo "[py $arg] rc $?: $(tail -2 $O/$n.log | tr '\n' ' ' | cut -c1-300)";;
    sh)
      bash -c "$arg" > $O/sh$nb.log 2>&1; echo "[sh] rc $?: $(tail -2 $O/sh$nb.log | tr '\n' ' ' | cut -c1-300)"; nb=$((nb+1));;
    *) echo "unknown leg $leg";;
  esac
  echo "    ($name took $(( $(date +%s) - t0 )) s)"
done
