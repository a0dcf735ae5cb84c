#!/bin/bash
# k_prop_bwd alone (rocprofv3 kernel trace of tools/microbench/prop_bwd_time.py) for several built copies of the tree
export TMPDIR=/tmp
ROOT=$(pwd)
for d in "$@"; do
  rm -rf /tmp/pb_$$; mkdir -p /tmp/pb_$$
  (cd $ROOT/$d && rocprofv3 --kernel-trace --output-format csv -d /tmp/pb_$$ -- python tools/microbench/prop_bwd_time.py > /tmp/pb_$$/out.txt 2>&1)
  echo "== $d"; grep "proposal net" /tmp/pb_$$/out.txt
  f=$(find /tmp/pb_$$ -name "*kernel_trace.csv" | head -1)
  python $ROOT/tools/kt_agg.py $f k_prop_bwd; python $ROOT/tools/kt_agg.py $f k_scatter; python $ROOT/tools/kt_agg.py $f k_prop_reduce
done
