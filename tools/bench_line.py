#!/usr/bin/env python3
"""Prints the fields of a bench.py JSON line that a round is steered by (tools/gpu_call.sh).  usage: bench_line.py <log> [--short]"""
import json
import sys


def main():
    path = sys.argv[1]
    short = "--short" in sys.argv
    lines = [ln for ln in open(path) if ln.startswith("{")]
    if not lines:
        print("  no JSON line in", path)
        return
    d = json.loads(lines[-1])
    head = f"{d['value'] / 1e6:.3f} M rays/s, {d['ms_per_step']} ms/step, host {d.get('host_enqueue_ms_per_step')} ms/step"
    if short:
        print(head)
        return
    print(" ", head, "| by kind", d.get("host_enqueue_ms_by_step_kind"), "| sequencer", d.get("native_sequencer"))
    for key in ("roofline", "roofline_other_bound"):
        r = d.get(key)
        if r:
            print(f"  {key}: {r['kernel']}[{r.get('units_per_launch')}] {r['bound']} achieved {r['achieved']:.1f} {r['unit']} "
                  f"frac {r['frac']} avg {r['avg_launch_ms']} ms traffic {r.get('traffic')}")
    if d.get("value_fp32_arithmetic"):
        print("  fp32 arithmetic:", d["value_fp32_arithmetic"])
    q = d.get("quality")
    if q:
        print("  quality:", {k: q[k] for k in q if k in ("train_steps", "psnr_heldout", "semantic_iou_heldout", "fruit_count",
                                                        "train_rays_per_s_over_these_steps", "scatter_queue_overflows",
                                                        "parameter_checksum", "second_seed_stream")})
    s = d.get("secondary")
    if s:
        for k in ("train_rays_per_s_plugin_api", "train_rays_per_s_camera_optimizer_off", "train_rays_per_s_by_proposal_backward_stream",
                  "train_rays_per_s_by_mlp_precision", "eval_rays_per_s", "export_samples_per_s"):
            if k in s:
                v = s[k]
                if isinstance(v, dict):
                    v = {a: b for a, b in v.items() if a != "note"}
                print(f"  {k}: {v}")
        if s.get("fruit_nerf_big"):
            b = s["fruit_nerf_big"]
            print(f"  fruit_nerf_big: {b['value'] / 1e6:.3f} M rays/s, {b['ms_per_step']} ms/step, host {b.get('host_enqueue_ms_per_step')}")
    if d.get("cpu_baseline"):
        print("  cpu_baseline:", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "cores")
    bd = d.get("breakdown_ms")
    if bd:
        print("  breakdown:", {k: v for k, v in list(bd.items())[:14]})


if __name__ == "__main__":
    main()
