#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) as a per-kernel table (markdown)."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:90]


def main(path: str, skip_first: int = 0) -> None:
    db = sqlite3.connect(path)
    rows = db.execute("select name, duration, grid_x, grid_y, workgroup_x, vgpr_count, accum_vgpr_count, lds_size, "
                      "scratch_size from kernels order by start").fetchall()
    agg = {}
    for name, dur, gx, gy, wx, vg, ag, lds, scr in rows:
        key = (short(name), gx * max(gy, 1) // max(wx, 1), wx)
        a = agg.setdefault(key, [0, 0.0, 1e30, 0.0, vg, ag, lds, scr])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    print(f"total kernel time {total / 1e6:.3f} ms over {len(rows)} dispatches\n")
    print("| kernel | workgroups x threads | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B | scratch |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for (name, wgs, wx), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"| {name} | {wgs}x{wx} | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | "
              f"{a[3] / 1e3:.1f} | {100 * a[1] / total:.1f} | {a[4]} | {a[5]} | {a[6]} | {a[7]} |")


if __name__ == "__main__":
    main(sys.argv[1])
