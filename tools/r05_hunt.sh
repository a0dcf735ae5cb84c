#!/bin/bash
# Round 5, first GPU call: the two-stream divergence hunt with the instruments written (on CPU) at the end of round 4.
# (build the variants on the CPU side first: bash tools/build_hunt_variants.sh)
# usage: bash tools/r05_hunt.sh <leg> [runs]   legs: seen | atomic | rmw | memset | unpaired | base
#   seen     the `seen` library variant (-DFNR_SCATTER_DEBUG_SEEN): every step the harness copies out what the proposal
#            scatters' accumulate workgroups READ (queue count, level maximum) and what their emit kernels PLACED, checks
#            them against each other inside the run (self_check: needs no reference run) and names the (level, bin) of
#            the table that differs at an event step
#   atomic   the `atomic_counters` variant: counters read / reset with agent-scope atomics (vector path, past L1 and the
#            scalar cache) — events gone => the scalar-load path of the counters is the mechanism
#   rmw      the `rmw_counters` variant: every counter access in every kernel is a device-scope atomic read-modify-write
#            (exchange / max-with-0 by thread 0, broadcast through LDS): no cached copy of a counter line is ever read —
#            past the scalar cache, L1 AND the XCD's L2
#   memset   default library, FNR_SCATTER_MEMSET=1: counters zeroed by a memset node per call instead of self-cleaning
#   unpaired default library, FNR_PAIR_PROPOSAL_LEVELS=0: one accumulate launch per proposal level
#   base     the same hunt on `fruit_nerf` (never hunted at this length: 3000 steps x runs)
# AMD_OPT_FLUSH=0 everywhere: the setting with twice the event rate (round 4: 5 of 49 runs).  ~15 s per run.
cd /root/repo; mkdir -p gpurun_out/r05
export TMPDIR=/tmp AMD_OPT_FLUSH=0
LEG=${1:-seen}; RUNS=${2:-36}
V=fruitnerf_amd/lib/variants
case $LEG in
  seen)     ENV="FNR_LIB_PATH=$PWD/$V/seen/libfruitnerf_hip.so FNR_DIGEST_SEEN=1"; M=fruit_nerf_big ;;
  atomic)   ENV="FNR_LIB_PATH=$PWD/$V/atomic_counters/libfruitnerf_hip.so"; M=fruit_nerf_big ;;
  rmw)      ENV="FNR_LIB_PATH=$PWD/$V/rmw_counters/libfruitnerf_hip.so"; M=fruit_nerf_big ;;
  memset)   ENV="FNR_SCATTER_MEMSET=1"; M=fruit_nerf_big ;;
  unpaired) ENV="FNR_PAIR_PROPOSAL_LEVELS=0"; M=fruit_nerf_big ;;
  base)     ENV=""; M=fruit_nerf ;;
  *) echo "unknown leg $LEG"; exit 2 ;;
esac
LOG=gpurun_out/r05/hunt_$LEG.log
( time env $ENV timeout $((RUNS * 17 + 120)) python tests/diagnostics/digest_perstep.py $M $RUNS 3000 ) > $LOG 2>&1
grep -E "DIFFERS|SELF-CHECK|   step|      |overlap" $LOG | cut -c1-400 | head -60
echo "identical runs: $(grep -c identical $LOG)"; tail -3 $LOG | cut -c1-200
