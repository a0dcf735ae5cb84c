"""Which kernels carry VECTOR loads across a workgroup barrier?  (No GPU needed: hipcc cross-compiles.)

On gfx950 neither `s_barrier` nor the workgroup-scope fence of `__syncthreads()` waits for outstanding vector loads
(`vmcnt`) — only LDS / scalar traffic (`lgkmcnt`).  A load that is still in flight when its wave leaves the barrier is
harmless while nobody rewrites the address inside the kernel (input prefetches: the MLP kernels' h / d_logit tiles, the emit
kernel's next-level gradient), and a defect when another thread does — round 4's divergence was `accumulate_bin` resetting
queue counters that other waves were still reading.  This lists every (kernel, barrier) with a vector
load between it and the previous full `s_waitcnt vmcnt(0)`; read each hit for a store to the same address after the barrier.
usage: python tools/isa_barrier_scan.py [file.hip ...]      (default: every kernel source of the library)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fruitnerf_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffp-contract=off", "-S",
         "--cuda-device-only"]
srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
with tempfile.TemporaryDirectory() as tmp:
    for src in srcs:
        out = os.path.join(tmp, os.path.basename(src) + ".s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + ["-o", out, src], check=True,
                       capture_output=True)
        lines = open(out).read().split("\n")
        kern, hits = None, {}
        for i, l in enumerate(lines):
            m = re.match(r"^(_Z\w+):", l)
            if m:
                kern = m.group(1)
            if kern and l.strip().startswith("s_barrier"):
                j = i - 1
                while j >= 0 and not re.match(r"^(_Z\w+):", lines[j]):
                    t = lines[j].strip()
                    if (t.startswith("s_waitcnt") and "vmcnt(0)" in t) or t.startswith("s_barrier"):
                        break
                    if re.match(r"(global|buffer|flat)_load", t):
                        hits.setdefault(kern, []).append(t)
                        break
                    j -= 1
        for k, v in hits.items():
            if "rocprim" in k:
                continue
            print(f"{os.path.basename(src)}: {k[:100]}: {len(v)} barrier(s) with a load in flight, e.g. `{v[0]}`")
