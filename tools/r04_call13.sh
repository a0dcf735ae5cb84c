#!/bin/bash
# round 4, GPU call 13: (1) the default bench line with the end-to-end count in scene coordinates; (2) the two-stream per-step
# divergence hunt under AMD_OPT_FLUSH=0 (the runtime's system-scope fences on every packet): do the events need the
# device-scope / elided cache maintenance between kernels?
cd /root/repo; mkdir -p gpurun_out/r04
export TMPDIR=/tmp
python bench.py --no-cpu-baseline 2>gpurun_out/r04/bench_final.err | tail -1 > gpurun_out/r04/bench_final.log
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04/bench_final.log').read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['quality'].get('fruit_count'), d['quality'].get('fruit_count_first_stage'), d['secondary']['fruit_count_end_to_end'])
P
AMD_OPT_FLUSH=0 python bench.py --no-cpu-baseline --no-quality 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('AMD_OPT_FLUSH=0', d['value'], d['ms_per_step'])" | tee gpurun_out/r04/opt_flush_speed.log
( time AMD_OPT_FLUSH=0 timeout 800 python tests/diagnostics/digest_perstep.py fruit_nerf_big 50 3000 ) > gpurun_out/r04/digest_perstep_opt_flush0.log 2>&1
grep -E "DIFFERS|   step|      |overlap" gpurun_out/r04/digest_perstep_opt_flush0.log | cut -c1-300 | head -30; grep -c identical gpurun_out/r04/digest_perstep_opt_flush0.log; tail -3 gpurun_out/r04/digest_perstep_opt_flush0.log
