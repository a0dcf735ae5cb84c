"""Static instruction mix of one kernel from hipcc's gfx950 assembly (no GPU needed): VALU / SALU / LDS / VMEM / MFMA /
waitcnt / barrier counts for the whole kernel and for every loop (back edge), plus the most frequent VALU mnemonics of the
largest loop.  With the counters' SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (profiles/*_kernel_trace_pmc.md) x the resident waves
per SIMD this tells whether a "latency-bound" kernel is in fact out of VALU issue slots (k_scatter_emit: 609 VALU
instructions per (sample, level) thread, 0.14 x 6 waves = 0.84 of the slots).
usage: python tools/isa_mix.py <source base name, e.g. hash_scatter> <substring of the mangled kernel name>"""
import collections
import re
import subprocess
import sys

base, want = sys.argv[1], sys.argv[2]
asm = f"/tmp/{base}-hip-amdgcn-amd-amdhsa-gfx950.s"
subprocess.run(["bash", "/root/repo/tools/regs.sh", base], check=False, capture_output=True)
src = open(asm).read().split("\n")
start = next(i for i, l in enumerate(src) if re.match(r"^_Z\w*:", l) and want in l)
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
lines = [l.strip() for l in src[start:end + 1]]
print(src[start].split(":")[0], f"({len(lines)} lines)")


def cat(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.split("_")[0] in ("global", "buffer", "flat", "scratch"):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    return "salu" if op.startswith("s_") else "other"


def instrs(a, b):
    for l in lines[a:b + 1]:
        if l and not l.startswith((";", ".")) and not l.endswith(":"):
            yield l


def mix(a, b):
    c = collections.Counter(cat(l.split()[0]) for l in instrs(a, b))
    return ", ".join(f"{k} {v}" for k, v in c.most_common())


labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = []
for i, l in enumerate(lines):
    m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and labels.get(m.group(1), len(lines)) < i:
        loops.append((labels[m.group(1)], i))
print("whole kernel:", mix(0, len(lines) - 1))
for a, b in sorted(set(loops)):
    print(f"  loop lines {a}..{b} ({b - a} lines): {mix(a, b)}")
if loops:
    a, b = max(loops, key=lambda ab: ab[1] - ab[0])
    ops = collections.Counter(re.sub(r"_e32|_e64|_dpp|_sdwa", "", l.split()[0]) for l in instrs(a, b) if l.startswith("v_"))
    dpp = sum("dpp" in l or "row_sh" in l or "row_bcast" in l for l in instrs(a, b))
    print(f"largest loop: {dpp} DPP-modified instructions; VALU mnemonics:",
          ", ".join(f"{k} {v}" for k, v in ops.most_common(14)))
