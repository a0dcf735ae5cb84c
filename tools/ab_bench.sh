#!/bin/bash
# A/B on ONE box: bench.py with the previous build of the library (FNR_LIB_PATH=fruitnerf_amd/lib/libfruitnerf_hip_prev.so,
# built from the parent commit with `make OUT=../lib/libfruitnerf_hip_prev.so OBJDIR=../../build/obj_prev`) and with
# the current one, alternating, 3 rounds.  Boxes differ by a few per cent; runs on one box by < 1 %.
ARGS="$@"
one() {
  FNR_LIB_PATH=$1 python bench.py --no-cpu-baseline --no-quality $ARGS 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"
}
for i in 1 2 3; do
  one /root/repo/fruitnerf_amd/lib/libfruitnerf_hip_prev.so prev
  one /root/repo/fruitnerf_amd/lib/libfruitnerf_hip.so      curr
done
