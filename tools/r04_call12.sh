#!/bin/bash
# round 4, GPU call 12: dump the 512^3 semantic export of the bench's trained model (for offline work on the count)
cd /root/repo; mkdir -p gpurun_out/r04
FNR_BENCH_DUMP_CLOUD=gpurun_out/r04/semantic_cloud_512.npz python bench.py --no-cpu-baseline --no-big 2>gpurun_out/r04/dump.err | tail -1 > gpurun_out/r04/dump_bench.log
ls -la gpurun_out/r04/semantic_cloud_512.npz
