#!/bin/bash
# Round 5, GPU call 6: streaming (`nt`) hints on write-once / read-once data — record queues, optimiser moments, the encode's
# Jacobian — as kernel traces of the serialised loop and step A/Bs against the in-tree library.
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
  python tools/kt_agg.py $f fnr | grep -E "k_scatter_emit<fnr::RaySource, true> +grid +196608|k_scatter_accumulate<|k_hash_encode|base_coop|k_prop_density" | cut -c1-175
  python tools/kt_agg.py $f fnr | awk '{t+=$(NF-6)} END {print "   sum of fnr kernels:", t, "ms over the run"}'
}
{
  trace default A=1
  for v in ntq ntm ntqm ntall; do trace $v FNR_LIB_PATH=$V/$v/libfruitnerf_hip.so; done
  trace default_again A=1
} 2>&1 | tee $O/kt_nt.log
for rep in 1 2; do
  timeout 200 python tools/ab_quick.py --pairs 3 2>/dev/null | grep -E "arm"
  for v in ntqm ntall; do FNR_LIB_PATH=$V/$v/libfruitnerf_hip.so timeout 200 python tools/ab_quick.py --pairs 3 2>/dev/null | grep -E "arm"; done
done | tee $O/ab_quick_6.log
