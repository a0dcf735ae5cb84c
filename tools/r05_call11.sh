#!/bin/bash
# Round 5, GPU call 11: the shipped set of streaming hints (everything but the Jacobian's loads): determinism suite x 4, the
# whole GPU suite, same-box step A/B against a build without hints, then the two-stream soak on this library.
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
V=$PWD/fruitnerf_amd/lib/variants
fails=0
for rep in 1 2 3 4; do
  timeout 300 python -m pytest tests/test_gpu_determinism.py -m gpu -q -p no:cacheprovider > $O/det_final_$rep.log 2>&1 || fails=$((fails+1))
done
echo "determinism suite on the in-tree library: $fails of 4 repetitions with a failing test"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_11.log 2>&1
echo "gpu tests rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_11.log | tail -8
for rep in 1 2 3; do
  timeout 200 python tools/ab_quick.py --pairs 3 2>/dev/null | grep -E "arm"
  FNR_LIB_PATH=$V/nt_none/libfruitnerf_hip.so timeout 200 python tools/ab_quick.py --pairs 3 2>/dev/null | grep -E "arm"
done | tee $O/ab_quick_11.log
export AMD_OPT_FLUSH=0
( time timeout 900 python tests/diagnostics/digest_perstep.py fruit_nerf_big ${1:-40} 3000 ) > $O/soak_final.log 2>&1
grep -E "DIFFERS|   step" $O/soak_final.log | cut -c1-300 | head -20
echo "identical runs: $(grep -c identical $O/soak_final.log)"; tail -4 $O/soak_final.log | cut -c1-200
