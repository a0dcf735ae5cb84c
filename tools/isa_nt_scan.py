"""Where does a kernel wait PARTIALLY (`s_waitcnt vmcnt(n)`, n > 0) while vector-memory operations of BOTH cache policies —
streaming (`nt`) and plain — can be outstanding?  (No GPU needed: hipcc cross-compiles.)

vmcnt is one counter on gfx950: a partial wait is only correct if the wave's vector-memory operations complete in issue
order.  tools/microbench/nt_load_order.hip measures that they do, across policies (profiles/r06_raw/nt_load_order.log); this
scan keeps the inventory of the places that RELY on it, so that the rule of csrc/common.hpp ("no streaming access where the
compiler may count accesses of both kinds at a partial wait", or a documented exception) is checked by
tests/test_isa_invariants.py instead of by a comment.

Model: instructions in program order per kernel (labels and branches ignored: a loop's back edge can only ADD operations
to what is outstanding, which the first partial wait of the next iteration would see — the scan walks every kernel body
twice for that); a queue of outstanding operations, cut to its youngest n at `vmcnt(n)` and emptied at `vmcnt(0)`; a hit =
a partial wait with an `nt` and a plain operation both in the queue BEFORE the cut.
usage: python tools/isa_nt_scan.py [file.hip ...]   -> one line per (kernel, kinds in flight), and a total"""
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
VMEM = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)\w*")


def scan_asm(lines):
    """-> {kernel: {"partial_waits": n, "mixed": [(line no, n, kinds)...], "nt_ops": n}}"""
    out, kern, body = {}, None, []

    def finish():
        if kern is None:
            return
        queue, mixed, partial, nt_ops = [], [], 0, 0
        for rep in range(2):                       # second pass: what the first iteration left outstanding is still there
            for no, t in body:
                m = VMEM.match(t)
                if m:
                    kind = ("nt" if re.search(r"\bnt\b", t) else "plain") + "-" + ("load" if m.group(2) == "load" else "store")
                    queue.append(kind)
                    nt_ops += rep == 0 and kind.startswith("nt")
                    continue
                w = re.match(r"s_waitcnt\b(.*)", t)
                if w:
                    v = re.search(r"vmcnt\((\d+)\)", w.group(1))
                    if v is None:
                        continue
                    n = int(v.group(1))
                    if n == 0:
                        queue = []
                        continue
                    if rep == 0:
                        partial += 1
                    if len(queue) > n:
                        kinds = set(queue)
                        if any(k.startswith("nt") for k in kinds) and any(k.startswith("plain") for k in kinds):
                            if rep == 0 or not any(x[0] == no for x in mixed):
                                mixed.append((no, n, tuple(sorted(kinds))))
                        queue = queue[len(queue) - n:]
        out[kern] = {"partial_waits": partial, "mixed": mixed, "nt_ops": nt_ops}

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


def scan_source(src, hipcc=None, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, os.path.basename(src) + ".s")
        subprocess.run([hipcc or os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + list(extra) + ["-o", out, src],
                       check=True, capture_output=True)
        return scan_asm(open(out).read().split("\n"))


if __name__ == "__main__":
    srcs = [a for a in sys.argv[1:] if not a.startswith("-D")] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    total = 0
    for src in srcs:
        for k, r in scan_source(src, extra=extra).items():
            if "rocprim" in k or not r["nt_ops"]:
                continue
            kinds = sorted({kk for _, _, ks in r["mixed"] for kk in ks})
            print(f"{os.path.basename(src)}: {k[:90]}: {r['nt_ops']} nt ops, {r['partial_waits']} partial waits, "
                  f"{len(r['mixed'])} with both policies in flight {kinds if kinds else ''}")
            total += len(r["mixed"])
    print(f"total partial waits with both policies in flight: {total}")
