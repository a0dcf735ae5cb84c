#!/bin/bash
# Round 5, GPU call 13: the exchange path (one-rank RCCL group) with the losses launch on the second stream and on the launch
# stream, twice each; then smoke() and the GPU suite on the tree that ships.
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
show() { python - "$1" "$2" <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'), d['config'].get('rccl_ranks'))
P
}
for rep in 1 2; do
  FNR_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-quality --no-big > $O/x13_on_$rep.log 2>/dev/null; show $O/x13_on_$rep.log rccl1-losses-on-side
  FNR_BENCH_FORCE_DIST=1 FNR_LOSSES_ON_SIDE=0 python bench.py --no-cpu-baseline --no-quality --no-big > $O/x13_off_$rep.log 2>/dev/null; show $O/x13_off_$rep.log rccl1-losses-on-main
done | tee $O/exchange_losses_ab.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -3 $O/smoke.log
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_13.log 2>&1
echo "gpu tests rc $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_13.log | tail -8
