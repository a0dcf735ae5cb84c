#!/bin/bash
# A/B on ONE box of two builds of the library (tools/build_variant.sh), alternating bench.py runs:
#   bash tools/ab_lib.sh <variant A | default> <variant B | default> [rounds] [bench args...]
A=$1; B=$2; ROUNDS=${3:-3}; shift 3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
libof() { [ "$1" = default ] && echo "" || echo "$ROOT/fruitnerf_amd/lib/variants/$1/libfruitnerf_hip.so"; }
one() {
  FNR_LIB_PATH=$(libof $1) python $ROOT/bench.py --no-cpu-baseline --no-quality "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']
print('$1'.ljust(14), d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), ' '.join(f'{k}={v*1e3:.1f}' for k,v in list(b.items())[:10]))"
}
for i in $(seq $ROUNDS); do one $A "$@"; one $B "$@"; done
