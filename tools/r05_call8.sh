#!/bin/bash
# Round 5, GPU call 8: the emit kernel's queue stores two records at a time (16 + 4 bytes; even-sized reservations, zero pads),
# with and without the streaming hint on the 16-byte value stores: correctness of the variant on the scatter / training
# suites, kernel traces, step A/Bs.
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
V=$PWD/fruitnerf_amd/lib/variants
FNR_LIB_PATH=$V/pairnt/libfruitnerf_hip.so timeout 900 python -m pytest tests/test_gpu_properties.py tests/test_gpu_training_parity.py tests/test_gpu_forward_parity.py tests/test_golden.py tests/test_gpu_determinism.py -m gpu -q -p no:cacheprovider > $O/tests_8_pairnt.log 2>&1
echo "pairnt tests rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_8_pairnt.log | tail -8
trace() {  # label, env...
  label=$1; shift
  rm -rf /tmp/kt_$label
  ( cd /tmp && env FNR_SERIALIZE_STREAMS=1 "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$label -o p -- \
      python /root/repo/tools/ab_quick.py --pairs 1 --steps 120 > /tmp/kt_$label.out 2>&1 )
  f=$(find /tmp/kt_$label -name "*kernel_trace.csv" | head -1)
  echo "== $label: $(grep 'arm A' /tmp/kt_$label.out | cut -c1-120)"
  python tools/kt_agg.py $f fnr | grep -E "k_scatter_emit|k_scatter_accumulate|k_hash_encode|k_prop_density" | cut -c1-175
  python tools/kt_agg.py $f fnr | awk '{t+=$(NF-6)} END {print "   sum of fnr kernels:", t, "ms over the run"}'
}
{
  trace default A=1
  trace pair FNR_LIB_PATH=$V/pair/libfruitnerf_hip.so
  trace pairnt FNR_LIB_PATH=$V/pairnt/libfruitnerf_hip.so
  trace default_again A=1
} 2>&1 | tee $O/kt_pair.log
for rep in 1 2 3; do
  timeout 200 python tools/ab_quick.py --pairs 3 2>/dev/null | grep -E "arm"
  for v in pair pairnt; do FNR_LIB_PATH=$V/$v/libfruitnerf_hip.so timeout 200 python tools/ab_quick.py --pairs 3 2>/dev/null | grep -E "arm"; done
done | tee $O/ab_quick_8.log
