#!/bin/bash
# Round 5, GPU call 2 (~18 box-minutes): the suite on the tree with round 5's tests, same-process A/Bs of the round's
# scheduling / occupancy knobs (tools/ab_quick.py), then the soak of the two-stream divergence fix on the library that will
# ship: fruit_nerf_big, 3000 steps per run, AMD_OPT_FLUSH=0 (the setting with twice the event rate in round 4).
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05
V=$PWD/fruitnerf_amd/lib/variants
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_2.log 2>&1
echo "gpu tests rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_2.log | tail -12
{
  timeout 200 python tools/ab_quick.py T.MLP_TAILS_ON_SIDE=0,1
  timeout 200 python tools/ab_quick.py env.FNR_SCATTER_LOG2_ROWS=-,12
  for rep in 1 2; do
    timeout 200 python tools/ab_quick.py --pairs 3
    FNR_LIB_PATH=$V/acc2w4/libfruitnerf_hip.so timeout 200 python tools/ab_quick.py --pairs 3
    FNR_LIB_PATH=$V/zero_early/libfruitnerf_hip.so timeout 200 python tools/ab_quick.py --pairs 3
  done
  FNR_PROP_BWD_WGS_PER_CU=2 timeout 200 python tools/ab_quick.py --pairs 3
} 2>/dev/null | grep -E "arm|vs" | tee $O/ab_quick.log
# the soak: on zero_early if it won by more than 0.7 % (it would then become the default), else on the in-tree library
LIB=$(python - <<'P'
import re, statistics
d, z = [], []
for l in open('gpurun_out/r05/ab_quick.log'):
    m = re.search(r"median ([0-9.]+) ms/step", l)
    if not m or not l.startswith("default (lib"):
        continue
    (z if "zero_early" in l else d if "in-tree" in l else []).append(float(m.group(1)))
if d and z and statistics.median(z) < statistics.median(d) * 0.993:
    print("zero_early")
else:
    print("default")
P
)
echo "soak on: $LIB"
if [ "$LIB" = zero_early ]; then export FNR_LIB_PATH=$V/zero_early/libfruitnerf_hip.so; fi
export AMD_OPT_FLUSH=0
( time timeout 800 python tests/diagnostics/digest_perstep.py fruit_nerf_big ${1:-40} 3000 ) > $O/soak_default.log 2>&1
grep -E "DIFFERS|SELF-CHECK|   step|overlap" $O/soak_default.log | cut -c1-300 | head -30
echo "identical runs: $(grep -c identical $O/soak_default.log)"; tail -4 $O/soak_default.log | cut -c1-200
