#!/bin/bash
# Round 5, first GPU call: the two-stream divergence hunt with the instruments written (on CPU) at the end of round 4.
# (build the variants on the CPU side first: bash tools/build_hunt_variants.sh)
# usage: bash tools/r05_hunt.sh <leg> [runs]
# Every leg runs tests/diagnostics/digest_perstep.py on a `seen` build of the library (-DFNR_SCATTER_DEBUG_SEEN): each step
# the harness copies out, for both proposal scatters, (a) the queue count and level maximum every accumulate workgroup READ,
# (b) the records every emit level PLACED, (c) an order-independent checksum of every bin's records as WRITTEN by the emit
# kernel and as READ BACK by the accumulate kernel.  Inside each run (no reference needed) it checks (a) against (b), the
# bins of a level against each other, and (c) written against read back; against the reference run it compares per-bin
# checksums of both proposal tables and (c).  Reading an event:
#   counts read != records placed, or uneven maxima   -> the counter path (scalar loads of words that atomics wrote)
#   counters fine, records read back != written       -> the queue came back different (cache maintenance / lost stores)
#   both fine, WRITTEN differs from the reference run -> the emit kernel's input (d_feats / positions): FNR_DIGEST_WS=1 next
#   records equal, gradient sums in LDS differ        -> the LDS accumulation (ds_add_u64) lost or doubled an add
#   parameters / moments READ by the sweep differ     -> the sweep read something else than the arena held at the end of
#                                                        the previous step (its checksums were equal): stale parameter lines
#   all equal to the reference, table differs         -> the update arithmetic or its stores
# THE MECHANISM WAS FOUND BY READING THE ISA AFTER THIS KIT WAS WRITTEN (hash_scatter.hip, accumulate_bin; DESIGN 2 round 4
# (c)): in k_scatter_accumulate2<true>'s copy for proposal network 0 the counter loads are vector loads that the workgroup
# barrier does not wait for, so a reset store can overtake them.  The first two legs are the proof to collect:
#   nowait    the code AS IT WAS, no wait ahead of the barrier, with the counter loads forced onto the vector path in both
#             copies (the instrumented builds would otherwise get harmless scalar loads): expect events (round 4: 5 of 49
#             runs with ONE hazardous copy), each with "the WAVES of a workgroup read different counters" at its step
#   vec       the FIX under the same forced vector loads: expect NO event, no self-check line
#   seen      the fixed default as the compiler builds it: expect NO event either
# further legs (kept from before the finding):
#   atomic    counters read / reset with agent-scope atomic loads / stores (vector path, past L1 and the scalar cache)
#   rmw       every counter access a device-scope atomic read-modify-write (past the XCD's L2 too)
#   memset    default counter code, FNR_SCATTER_MEMSET=1: counters zeroed by a memset node per call
#   unpaired  FNR_PAIR_PROPOSAL_LEVELS=0: one accumulate launch per proposal level
#   serial    FNR_SERIALIZE_STREAMS=1: both streams, both allocator pools, both hardware queues — nothing concurrent
#   onestream FNR_OVERLAP_PROPOSAL_BACKWARD=0: one stream (round 4: 0 of 44 runs; is that still so under AMD_OPT_FLUSH=0?)
#   base      the same hunt on `fruit_nerf` (no kernel of its step uses scratch memory; fruit_nerf_big's semantic backward
#             spills 34 registers) — never hunted at this length
# AMD_OPT_FLUSH=0 everywhere: the setting with twice the event rate (round 4: 5 of 49 runs).  ~15 s per run.
cd /root/repo; mkdir -p gpurun_out/r05
export TMPDIR=/tmp AMD_OPT_FLUSH=0 FNR_DIGEST_SEEN=1
LEG=${1:-seen}; RUNS=${2:-36}
V=$PWD/fruitnerf_amd/lib/variants
M=fruit_nerf_big; LIB=seen; ENV=""
case $LEG in
  seen)      ;;
  nowait)    LIB=seen_nowait ;;
  vec)       LIB=seen_vec ;;
  atomic)    LIB=seen_atomic ;;
  rmw)       LIB=seen_rmw ;;
  memset)    ENV="FNR_SCATTER_MEMSET=1" ;;
  unpaired)  ENV="FNR_PAIR_PROPOSAL_LEVELS=0" ;;
  serial)    ENV="FNR_SERIALIZE_STREAMS=1" ;;
  onestream) ENV="FNR_OVERLAP_PROPOSAL_BACKWARD=0" ;;
  base)      M=fruit_nerf ;;
  *) echo "unknown leg $LEG"; exit 2 ;;
esac
test -f $V/$LIB/libfruitnerf_hip.so || { echo "missing $V/$LIB: run tools/build_hunt_variants.sh on the CPU side"; exit 2; }
LOG=gpurun_out/r05/hunt_$LEG.log
( time env FNR_LIB_PATH=$V/$LIB/libfruitnerf_hip.so $ENV timeout $((RUNS * 22 + 120)) python tests/diagnostics/digest_perstep.py $M $RUNS 3000 ) > $LOG 2>&1
grep -E "DIFFERS|SELF-CHECK|   step|      |overlap" $LOG | cut -c1-400 | head -60
echo "identical runs: $(grep -c identical $LOG)"; tail -3 $LOG | cut -c1-200
