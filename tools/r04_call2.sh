#!/bin/bash
# round 4, GPU call 2: the GPU suite (new: image metrics, end-to-end count, exchange schedule, same-samples gradients),
# host-side profiles of the step loop, the exchange path's fixed cost, the IoU anchor (HIP 10k state continued in the oracle)
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 1000 python -m pytest tests -m gpu -q -p no:cacheprovider -rA > gpurun_out/r04/tests_2.log 2>&1
echo "gpu tests rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r04/tests_2.log | tail -15
timeout 300 python tools/host_profile.py 300 > gpurun_out/r04/host_profile_single.log 2>&1; head -1 gpurun_out/r04/host_profile_single.log | cut -c1-200; grep "host enqueue" gpurun_out/r04/host_profile_single.log
timeout 300 python tools/host_profile.py 300 exchange > gpurun_out/r04/host_profile_exchange.log 2>&1; grep "host enqueue" gpurun_out/r04/host_profile_exchange.log
timeout 400 bash tools/exchange_cost_quick.sh > gpurun_out/r04/exchange_cost.log 2>&1; tail -6 gpurun_out/r04/exchange_cost.log
( time timeout 600 python -m tests.iou_anchor --pre-steps 10000 --steps 150 --eval-every 50 --out gpurun_out/r04/iou_anchor.json ) > gpurun_out/r04/iou_anchor.log 2>&1
tail -5 gpurun_out/r04/iou_anchor.log | cut -c1-1500
