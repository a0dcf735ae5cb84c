#!/bin/bash
# Round 5, GPU call 7: the streaming hints that paid in call 6, without the narrow `nt` stores that cost the emit kernel 45 us:
# j = Jacobian (encode's store, base branch's load); m = optimiser moments; q = queue LOADS; d = d_feats loads of the emit;
# p = the proposal networks' saved features and d_feats.  Kernel traces of the serialised loop + step A/Bs.
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
  python tools/kt_agg.py $f fnr | grep -E "k_scatter_emit|k_scatter_accumulate|k_hash_encode|base_coop|k_prop_density|k_prop_bwd|k_field_mlp_fwd_bf16|color_coop" | cut -c1-175
  python tools/kt_agg.py $f fnr | awk '{t+=$(NF-6)} END {print "   sum of fnr kernels:", t, "ms over the run"}'
}
{
  trace default A=1
  for v in j mj qmj qmjd qmjdp; do trace $v FNR_LIB_PATH=$V/$v/libfruitnerf_hip.so; done
  trace default_again A=1
} 2>&1 | tee $O/kt_nt2.log
for rep in 1 2 3; do
  timeout 200 python tools/ab_quick.py --pairs 3 2>/dev/null | grep -E "arm"
  for v in mj qmjd qmjdp; do FNR_LIB_PATH=$V/$v/libfruitnerf_hip.so timeout 200 python tools/ab_quick.py --pairs 3 2>/dev/null | grep -E "arm"; done
done | tee $O/ab_quick_7.log
