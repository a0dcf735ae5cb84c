#!/bin/bash
# A/B on ONE box: bench.py of the committed HEAD (a copy under build/prev_tree, made with
#   rm -rf build/prev_tree && mkdir -p build/prev_tree && git archive HEAD | tar -x -C build/prev_tree && make -C build/prev_tree/fruitnerf_amd/csrc
# ) against the working tree, alternating, 3 rounds.  Boxes differ by a few per cent; runs on one box by < 1 %.
ARGS="$@"
ROOT=$(pwd)
one() {
  (cd $1 && python bench.py --no-cpu-baseline --no-quality $ARGS 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])")
}
for i in 1 2 3; do
  one $ROOT/build/prev_tree prev
  one $ROOT curr
done
