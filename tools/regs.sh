#!/bin/bash
# usage: tools/regs.sh field_mlp_bwd  -> vgpr/spill/scratch per kernel
cd /root/repo/fruitnerf_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off --save-temps=obj -c $1.hip -o /tmp/$1.o 2>&1 | grep -E "error|warning: var" 
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size):" /tmp/$1-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - | sed -E 's/\s+/ /g' | cut -c1-200
