#!/bin/bash
# Round 5, GPU call 4 (~7 box-minutes): the emit kernel with 4 instead of 6 barriers per level (kernel traces of the
# serialised loop + a same-box step A/B against the `emit6` build), fruit_nerf_big trained to 20 000 steps once, and the
# exchange path's fixed cost on a one-rank RCCL group.
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
V=$PWD/fruitnerf_amd/lib/variants
trace() {  # label, env...
  label=$1; shift
  rm -rf /tmp/kt_$label
  ( cd /tmp && env FNR_SERIALIZE_STREAMS=1 "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$label -o p -- \
      python /root/repo/tools/ab_quick.py --pairs 1 --steps 120 > /tmp/kt_$label.out 2>&1 )
  f=$(find /tmp/kt_$label -name "*kernel_trace.csv" | head -1)
  echo "== $label: $(grep 'arm A' /tmp/kt_$label.out | cut -c1-120)"
  python tools/kt_agg.py $f fnr | grep -E "k_scatter_emit|k_scatter_accumulate" | cut -c1-175
}
{
  trace emit4 A=1
  trace emit6 FNR_LIB_PATH=$V/emit6/libfruitnerf_hip.so
  trace emit4_again A=1
  trace emit6_again FNR_LIB_PATH=$V/emit6/libfruitnerf_hip.so
} 2>&1 | tee $O/kt_emit.log
for rep in 1 2; do
  timeout 200 python tools/ab_quick.py --pairs 3 2>/dev/null | grep -E "arm"
  FNR_LIB_PATH=$V/emit6/libfruitnerf_hip.so timeout 200 python tools/ab_quick.py --pairs 3 2>/dev/null | grep -E "arm"
done | tee $O/ab_quick_4.log
timeout 600 python bench.py --method fruit_nerf_big --quality-steps 20000 --no-cpu-baseline > $O/bench_big_20k.log 2>$O/bench_big_20k.err
python - <<'P'
import json
d = json.loads([l for l in open('gpurun_out/r05/bench_big_20k.log') if l.startswith('{')][-1])
print('fruit_nerf_big', d['value'], d['ms_per_step'], json.dumps(d['quality'])[:600])
P
show() { python - "$1" "$2" <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'), d['config'].get('rccl_ranks'))
P
}
python bench.py --no-cpu-baseline --no-quality --no-big > $O/x_single.log 2>/dev/null; show $O/x_single.log single-two-streams | tee $O/exchange_cost.log
FNR_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-quality --no-big > $O/x_dist.log 2>/dev/null; show $O/x_dist.log rccl1 | tee -a $O/exchange_cost.log
FNR_BENCH_FORCE_DIST=1 FNR_DEFER_FIELD_UPDATE=1 python bench.py --no-cpu-baseline --no-quality --no-big > $O/x_dist_def.log 2>/dev/null; show $O/x_dist_def.log rccl1-deferred | tee -a $O/exchange_cost.log
FNR_BENCH_FORCE_DIST=1 FNR_SHARDED_FIELD_OPTIMIZER=1 python bench.py --no-cpu-baseline --no-quality --no-big > $O/x_dist_sh.log 2>/dev/null; show $O/x_dist_sh.log rccl1-sharded | tee -a $O/exchange_cost.log
