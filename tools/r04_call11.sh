#!/bin/bash
# round 4, GPU call 11: the profile round (bench lines, kernel trace, PMC passes) + sparse-touch A/B
export TMPDIR=/tmp
cd /root/repo
bash tools/prof_round.sh r04 > gpurun_out/r04_prof_round.out 2>&1
tail -3 gpurun_out/r04_prof_round.out
for v in 1 0; do
  FNR_SPARSE_TOUCH=$v python bench.py --no-cpu-baseline --no-quality 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']
print('sparse_touch $v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], ' '.join(f'{k}={v*1e3:.1f}' for k,v in list(b.items())[:4]))" | tee -a gpurun_out/r04/ab_sparse_touch.log
done
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_fruit_nerf.log') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['value_fp32_arithmetic'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('record_queue_bytes'), d['quality']['psnr_heldout'], d['quality']['semantic_iou_heldout'], d['quality'].get('fruit_count'), d['quality'].get('fruit_count_first_stage'), d['secondary']['fruit_count_end_to_end'])
P
