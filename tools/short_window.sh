#!/bin/bash
# does the driver's short window (20 steps after 5) read the same as a long one on the same box?
export TMPDIR=/tmp
for cfg in "20 5" "20 5" "200 20" "20 5" "20 200" "200 200"; do
  set -- $cfg
  python bench.py --gpus 1 --steps $1 --warmup $2 --no-quality --no-cpu-baseline --no-big 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('steps $1 warmup $2', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"
done
