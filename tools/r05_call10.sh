#!/bin/bash
# Round 5, GPU call 10: which streaming hints break run-to-run determinism?  tests/test_gpu_determinism.py, repeated, per class
# mask (common.hpp NtClass): 0x00 none, 0x33 the loads, 0xcc the stores (+ the proposal features), then one class at a time.
cd /root/repo; mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=$PWD/gpurun_out/r05
V=$PWD/fruitnerf_amd/lib/variants
for m in 0x00 0x33 0xcc 0x01 0x02 0x04 0x08 0x10 0x20 0x40 0x80; do
  fails=0
  for rep in 1 2 3 4; do
    FNR_LIB_PATH=$V/nt_$m/libfruitnerf_hip.so timeout 300 python -m pytest tests/test_gpu_determinism.py -m gpu -q -p no:cacheprovider > $O/det_${m}_$rep.log 2>&1 || fails=$((fails+1))
  done
  echo "mask $m: $fails of 4 repetitions with a failing test; $(cat $O/det_${m}_*.log | grep -c '^FAILED') failed tests in all: $(cat $O/det_${m}_*.log | grep '^FAILED' | sed -E 's/.*::(test_[a-z_0-9]+).*/\1/' | sort | uniq -c | tr '\n' ';')"
done | tee $O/nt_bisect.log
