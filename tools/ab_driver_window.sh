#!/bin/bash
# 20-step windows at increasing warm-up of ab_prev/ against the working tree on one box, alternating
# (the driver's command is --steps 20 --warmup 5)
export TMPDIR=/tmp
ROOT=$(pwd)
for w in ${@:-5 5 25 45 100 200}; do
  for d in ab_prev .; do
    (cd $ROOT/$d && python bench.py --gpus 1 --steps 20 --warmup $w --no-quality --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']
print('warmup $w', '$d'.ljust(8), d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], ' '.join(f'{k.split(chr(91))[0]}={v*1e3:.0f}' for k,v in list(b.items())[:9]))")
  done
done
