"""Diagnostic scripts, run by hand on a GPU box (`gpurun -- python tests/diagnostics/<script>.py ...`), not collected by
pytest.  Test infrastructure: some replay a step / an export in the CPU oracle next to the HIP path (they import oracle/),
the others watch the HIP path itself.

oracle next to HIP     spike_vs_oracle, trained_export_check, golden_big_dbg, golden_raygrad_dbg, adam_zero_grad_dbg
timing of the loop     early_steps (per-window step time, host enqueue time, allocator growth from step 0)
reproducibility        big_determinism (fruit_nerf_big, one / two streams, sampling ahead, eval passes in between),
                       long_divergence (parameter digests every N steps of long runs, across modes and processes),
                       kernel_stress (one forward + backward on fixed inputs, thousands of times, next to a busy stream),
                       bench_flow_digest (bench.py's own flow — timed window, breakdown pass, evals — with digests at marks)
scatter health         scatter_overflows (queue overflows during training: none), overflow_probe (inputs that do overflow)
"""
