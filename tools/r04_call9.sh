#!/bin/bash
# round 4, GPU call 9: the proposal backward in isolation, repeated from one restored state: is it reproducible?
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
for args in "pair 30000" "pair 30000 --load" "scatter 30000 --load"; do
  timeout 200 python tests/diagnostics/scatter_repeat.py $args 2>&1 | grep -v amdgpu.ids | tail -4
done | tee gpurun_out/r04/scatter_repeat.log
