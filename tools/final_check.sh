#!/bin/bash
# what the driver runs at round end, in one call: the GPU suite, smoke(), the default bench command
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03/gputest_final.log 2>&1
echo "gpu tests rc $?"; grep -E "passed|failed" gpurun_out/r03/gputest_final.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r03/smoke.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_driver_style.log 2> gpurun_out/r03/bench_driver_style.err ) 2>&1 | grep real
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_driver_style.log') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_over_algorithmic'), d['quality']['psnr_heldout'], d['quality']['semantic_iou_heldout'], d['quality']['fruit_count_first_stage'], d['secondary']['fruit_nerf_big']['value'])
P
