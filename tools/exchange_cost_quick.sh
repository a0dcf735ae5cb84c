#!/bin/bash
# the exchange path's fixed cost on ONE GPU, short form: single process (two streams / one stream) vs a one-rank
# RCCL group with the exchange forced on (deferred field update; the exchange path has the second stream for the proposal
# backward only and the look-ahead in-stream)
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
show() { python - "$1" "$2" <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])
P
}
python bench.py --no-cpu-baseline --no-quality --no-big > gpurun_out/r04/x_single.log 2>/dev/null; show gpurun_out/r04/x_single.log single-two-streams
FNR_OVERLAP_PROPOSAL_BACKWARD=0 python bench.py --no-cpu-baseline --no-quality --no-big > gpurun_out/r04/x_single1.log 2>/dev/null; show gpurun_out/r04/x_single1.log single-one-stream
FNR_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-quality --no-big > gpurun_out/r04/x_dist.log 2>/dev/null; show gpurun_out/r04/x_dist.log rccl1-deferred
FNR_BENCH_FORCE_DIST=1 FNR_OVERLAP_PROPOSAL_BACKWARD=0 python bench.py --no-cpu-baseline --no-quality --no-big > gpurun_out/r04/x_dist1.log 2>/dev/null; show gpurun_out/r04/x_dist1.log rccl1-deferred-one-stream
