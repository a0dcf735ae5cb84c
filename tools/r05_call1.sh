#!/bin/bash
# Round 5, GPU call 1 (~20 box-minutes; build the hunt variants on the CPU side first: bash tools/build_hunt_variants.sh):
#   1. the GPU suite on the tree that ends round 4 (the accumulate kernels wait for their counter loads now);
#   2. the unvalidated legs written without a GPU (sharded optimiser step over a one-rank RCCL group), non-fatal;
#   3. the default bench line (the fix costs a wait per accumulate workgroup: compare with round 4's 5.43 M rays/s);
#   4. the hazard in isolation (tools/microbench/barrier_load_race.hip: stale counter reads with / without the wait);
#   5. the proof of the divergence's mechanism: leg nowait (old code, hazardous loads: expect events + "WAVES ... read
#      different counters"), leg vec (the fix under the same loads: expect none).
# gpurun --timeout 1800 -- bash tools/r05_call1.sh   (suite 2 min, bench 2, microbenchmark 1, two hunt legs of 24 runs ~8 min each)
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05/tests_1.log 2>&1
echo "gpu tests rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r05/tests_1.log | tail -5
FNR_RUN_UNVALIDATED=1 timeout 300 python -m pytest tests/test_gpu_distributed.py -q -p no:cacheprovider -k single_process_step > gpurun_out/r05/tests_unvalidated.log 2>&1
echo "unvalidated legs rc $?"; tail -3 gpurun_out/r05/tests_unvalidated.log
timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/r05/bench_1.err | tail -1 > gpurun_out/r05/bench_1.log
python - <<'P'
import json
d = json.loads(open('gpurun_out/r05/bench_1.log').read())
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms'))
P
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/microbench/barrier_load_race.hip -o /tmp/barrier_load_race 2>/dev/null && \
  timeout 120 /tmp/barrier_load_race 200000 | tee gpurun_out/r05/barrier_load_race.log   # the hazard in isolation (6 x ~10 s; round 4 saw ~1.6e-8 stale reads per wave and launch)
bash tools/r05_hunt.sh nowait 13
bash tools/r05_hunt.sh vec 13
