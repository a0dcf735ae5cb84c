#!/bin/bash
# usage: tools/regs.sh field_mlp_bwd [extra hipcc flags]  -> vgpr / spill / scratch / LDS per kernel
f=$1; shift
cd /root/repo/fruitnerf_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off "$@" --save-temps=obj -c $f.hip -o /tmp/$f.o 2>&1 | grep -E "error|warning: var"
python3 - /tmp/$f-hip-amdgcn-amd-amdhsa-gfx950.s <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.sgpr_spill_count:\s*(\d+).*?\.vgpr_count:\s*(\d+).*?\.vgpr_spill_count:\s*(\d+)", t, re.S):
    lds, name, scratch, sspill, vg, vspill = m.groups()
    name = re.sub(r"^_ZN3fnr(2pw)?\d+", "", name)
    print(f"{name[:70]:70s} vgpr {vg:>3s} spill {vspill:>3s} scratch {scratch:>4s} sgpr_spill {sspill:>3s} lds {lds}")
PY
