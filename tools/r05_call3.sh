#!/bin/bash
# Round 5, GPU call 3 (~7 box-minutes): per-KERNEL durations of the round's kernel changes (call 2 showed that the step time
# does not resolve them: +-0.5 % window noise) from rocprofv3 kernel traces of the serialised loop, the losses-on-the-second-
# stream A/B, and the suite with the two repaired tests.
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
  python tools/kt_agg.py $f fnr | grep -E "k_prop_bwd|k_scatter|k_prop_reduce|k_train_losses|k_weights_bwd|k_field_mlp_bwd|k_color_ray|k_reduce_dw|k_embedding" | cut -c1-175
}
{
  trace default A=1                                                       # k_prop_bwd: LDS weights, 3 waves per SIMD (14 spilled registers)
  trace oldprop FNR_LIB_PATH=$V/oldprop/libfruitnerf_hip.so               # k_prop_bwd as it was (302 scalar registers spilled to lanes)
  trace oldprop_acc2w4 FNR_LIB_PATH=$V/oldprop_acc2w4/libfruitnerf_hip.so # ... and the paired accumulate at one workgroup per CU
  trace propbwd_w2 FNR_LIB_PATH=$V/propbwd_w2/libfruitnerf_hip.so         # k_prop_bwd: LDS weights, 188 registers, 2 waves per SIMD, no spills
  trace rows12 FNR_SCATTER_LOG2_ROWS=12                                   # main table in 4096-row bins (two accumulate workgroups per CU)
} 2>&1 | tee $O/kt_variants.log
{
  timeout 200 python tools/ab_quick.py T.LOSSES_ON_SIDE=0,1
  timeout 200 python tools/ab_quick.py T.LOSSES_ON_SIDE=0,1 T.MLP_TAILS_ON_SIDE=0,1
} 2>/dev/null | grep -E "arm|vs" | tee $O/ab_quick_3.log
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_3.log 2>&1
echo "gpu tests rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_3.log | tail -12
