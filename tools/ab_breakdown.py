"""Per-entry-point A/B of two bench lines (stdin: two JSON lines, prev then curr)."""
import json, sys
a, b = [json.loads(l) for l in sys.stdin if l.startswith("{")][:2]
print("step", a["ms_per_step"], b["ms_per_step"])
for k in a["breakdown_ms"]:
    x, y = a["breakdown_ms"][k], b["breakdown_ms"].get(k, 0)
    print(f"{k:32s} {x*1e3:8.1f} {y*1e3:8.1f}  {1e3*(y-x):+7.1f} us")
