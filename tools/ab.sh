#!/bin/bash
# A/B on ONE box (boxes differ by up to 20 % on latency-bound kernels): bench.py of ab_prev/ (a built copy of an earlier
# commit: rm -rf ab_prev && mkdir ab_prev && git archive <commit> | tar -x -C ab_prev && make -C ab_prev/fruitnerf_amd/csrc)
# against the working tree, alternating.   usage: bash tools/ab.sh [rounds] [bench args...]
ROUNDS=${1:-3}; shift
ROOT=$(pwd)
mkdir -p gpurun_out/ab
one() {
  (cd $1 && python bench.py --no-cpu-baseline --no-quality "${@:3}" 2>/dev/null | grep "^{" > $ROOT/gpurun_out/ab/$2.json; python - $ROOT/gpurun_out/ab/$2.json $2 <<'P'
import json,sys
d=json.load(open(sys.argv[1])); b=d['breakdown_ms']
keys=['hash_encode_bwd[196608]','field_mlp_bwd[196608]','hash_encode_fwd[196608]','position_grad[196608]','prop_density_bwd[1048576]','prop_density_bwd[393216]','field_mlp_fwd[196608]','losses_fwd[4096]']
print(sys.argv[2], d['value'], d['ms_per_step'], ' '.join(f"{k.split('[')[0]}={b.get(k,0)*1e3:.1f}" for k in keys))
P
)
}
for i in $(seq $ROUNDS); do
  one $ROOT/ab_prev prev "$@"
  one $ROOT curr "$@"
done
