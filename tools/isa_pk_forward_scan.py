"""Packed-FP32 results read one NON-VALU instruction later (gfx950).  No GPU needed: hipcc cross-compiles.

Round 6 finding (profiles/r06_raw/nt_hunt.md): in k_field_mlp_bwd_base_coop a `v_pk_add_f32` wrote v[22:23], ONE `ds_bpermute_b32`
followed, and the next `v_pk_add_f32` read v[22:23] — hipcc counts the DS instruction as the wait state the part needs after a
packed / op_sel VALU result ("dst forwarding"; it puts `s_nop 0` between two such VALU instructions that are adjacent), the
hardware evidently does not always: ~10 of 12 288 waves per launch computed a wrong sum (timing-dependent: only the waves that
reach the sequence while the LDS pipe is idle), which is what broke run-to-run reproducibility when the Jacobian's loads became
`nt` (another schedule).  This lists every place where a packed-FP32 VALU result (v_pk_*_f32) is read by a VALU instruction with
ONLY non-VALU instructions (DS / VMEM / SALU other than s_nop) in between, at a distance the compiler treats as covered by them.
usage: python tools/isa_pk_forward_scan.py [-D...] [file.hip ...]"""
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
MAX_GAP = 2          # non-VALU instructions between producer and consumer that the scan still reports


def vregs(op):
    m = re.match(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", op)
    return {int(m.group(1))} if m else set()


def operands(t):
    parts = t.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    ops = [o.strip() for o in re.split(r",(?![^\[]*\])", parts[1].split(" op_sel")[0].split(" neg_")[0].split(" clamp")[0])]
    return parts[0], ops


def scan_asm(lines):
    out, kern, body = {}, None, []

    def finish():
        if kern is None:
            return
        hits = []
        for i, (no, t) in enumerate(body):
            mn, ops = operands(t)
            if not (mn.startswith("v_pk_") and mn.endswith("_f32")) or not ops:
                continue
            dst = vregs(ops[0])
            gap = []
            for no2, t2 in body[i + 1:i + 2 + MAX_GAP]:
                mn2, ops2 = operands(t2)
                if mn2.startswith("s_nop") or mn2.startswith("s_waitcnt") or mn2.startswith("s_barrier"):
                    break                                     # explicit wait states / a stall
                if mn2.startswith("v_") and not mn2.startswith("v_mfma"):
                    srcs = set().union(*[vregs(o) for o in ops2[1:]]) if len(ops2) > 1 else set()
                    if gap and dst & srcs:
                        hits.append((no, t, [g for _, g in gap], t2))
                    break                                     # the next VALU instruction ends the window either way
                gap.append((no2, t2))
        out[kern] = hits

    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            finish()
            kern, body = m.group(1), []
            continue
        if kern is not None:
            t = l.strip()
            if t.startswith("s_endpgm"):
                finish()
                kern, body = None, []
            elif t and not t.startswith((";", ".")):
                body.append((i + 1, t))
    finish()
    return out


def scan_source(src, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, os.path.basename(src) + ".s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + list(extra) + ["-o", out, src], check=True,
                       capture_output=True)
        return scan_asm(open(out).read().split("\n"))


if __name__ == "__main__":
    srcs = [a for a in sys.argv[1:] if not a.startswith("-D")] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    total = 0
    for src in srcs:
        for k, hits in scan_source(src, extra).items():
            if "rocprim" in k:
                continue
            for no, prod, gap, cons in hits:
                print(f"{os.path.basename(src)}: {k[:70]}: `{prod}` -> {gap} -> `{cons}`")
            total += len(hits)
    print(f"total: {total}")
