"""Properties of the COMPILED scatter kernels that no functional test can see (hipcc cross-compiles without a GPU).

Round 4's rare divergence of long two-stream runs was a missing wait: `accumulate_bin` reads its queue counters, passes a
workgroup barrier and lets thread 0 reset them — but on gfx950 neither the barrier nor the workgroup-scope fence of
`__syncthreads()` waits for outstanding VECTOR loads, and hipcc had issued the counter loads as vector loads in one inlined
copy of `k_scatter_accumulate2<true>` (`s_barrier` ahead of `s_waitcnt vmcnt`).  A wave could read a counter after its reset.
The fix is an explicit `s_waitcnt vmcnt(0) lgkmcnt(0)` (inline assembly, counters as inputs) ahead of that barrier; this
test keeps it there, in every copy of the function, whatever loads the compiler picks."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fruitnerf_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def scatter_asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "hash_scatter.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffp-contract=off",
           "-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "hash_scatter.hip")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    return out.read_text().split("\n")


def _kernel(lines, mangled_prefix):
    """The instructions of the kernel whose label starts with `mangled_prefix` (label line .. s_endpgm)."""
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled_prefix) and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return [l.strip() for l in lines[start:end + 1]]


@pytest.mark.parametrize("name,copies", [("_ZN3fnr20k_scatter_accumulateILb1EEE", 1), ("_ZN3fnr20k_scatter_accumulateILb0EEE", 1),
                                         ("_ZN3fnr21k_scatter_accumulate2ILb1EEE", 2), ("_ZN3fnr21k_scatter_accumulate2ILb0EEE", 2)])
def test_counter_loads_are_waited_for_ahead_of_the_reset_barrier(scatter_asm, name, copies):
    body = _kernel(scatter_asm, name)
    waits = [i for i, l in enumerate(body) if l.startswith("s_waitcnt vmcnt(0) lgkmcnt(0)") and body[i - 1].startswith(";;#ASMSTART")]
    assert len(waits) == copies, f"{name}: {len(waits)} explicit counter waits for {copies} inlined copies of accumulate_bin"
    for w in waits:
        # the next barrier follows within a few instructions, with no memory instruction in between ...
        nxt = next(i for i in range(w, len(body)) if body[i].startswith("s_barrier"))
        between = [l for l in body[w + 1:nxt] if l and not l.startswith(";")]
        assert len(between) <= 6 and not any(re.match(r"(global|buffer|flat|scratch)_", l) for l in between), between
        # ... and every counter load of this copy sits ahead of the wait: the first store / atomic to global memory (the
        # reset, the qdone increment) comes after the barrier
        first_write = next(i for i in range(w, len(body)) if re.match(r"global_(store|atomic)", body[i]))
        assert first_write > nxt


def test_scatter_kernels_use_no_scratch_and_keep_three_emit_workgroups_per_cu(scatter_asm):
    """Resource facts DESIGN 4 relies on: no scatter kernel spills (scratch would also make them the only scratch users of the
    `fruit_nerf` step), and the emit kernel's static LDS lets three workgroups share a CU's 160 KiB."""
    text = "\n".join(scatter_asm)
    meta = re.findall(r"\.group_segment_fixed_size:\s*(\d+)\s*\n(?:.*\n)*?\s*\.name:\s*(\S+)\s*\n(?:.*\n)*?\s*\.private_segment_fixed_size:\s*(\d+)"
                      r"(?:.*\n)*?\s*\.vgpr_spill_count:\s*(\d+)", text)
    kernels = {name: (int(lds), int(scratch), int(spill)) for lds, name, scratch, spill in meta if name.startswith("_ZN3fnr")}
    assert len(kernels) >= 20, sorted(kernels)
    for name, (lds, scratch, spill) in kernels.items():
        assert scratch == 0 and spill == 0, (name, scratch, spill)
        if "k_scatter_emit" in name:
            assert 3 * ((lds + 511) // 512 * 512) <= 160 * 1024, (name, lds)



# ---- round 6: the position-gradient reduction of k_field_mlp_bwd_base_coop (profiles/r06_raw/nt_hunt.md) ----------------
# With `nt` loads of the Jacobian hipcc chose packed FP32 with cross-half operand selects threaded through the six
# ds_bpermute shuffles of that reduction, and ~10 of 12 288 waves per launch computed a wrong y component.  The partial sums are
# pinned in registers since; this keeps every build's reduction free of packed math, for both load policies.

def _compile(src, out, extra=()):
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffp-contract=off",
           "-S", "--cuda-device-only", *extra, "-o", str(out), os.path.join(CSRC, src)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return out.read_text().split("\n")


@pytest.mark.parametrize("mask", ["0xef", "0xff"])
def test_position_gradient_reduction_is_free_of_packed_math(tmp_path, mask):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    lines = _compile("field_mlp_bf16.hip", tmp_path / "field_mlp_bf16.s", extra=[f"-DFNR_NT_MASK={mask}"])
    kernels = [m.group(1) for m in (re.match(r"^(_ZN3fnr25k_field_mlp_bwd_base_coop\w+):", l) for l in lines) if m]
    posgrad = [k for k in kernels if "Lb1EEE" in k]
    assert len(posgrad) >= 4, kernels                     # (FieldCfgBase | FieldCfgBig) x (bf16 | bf16x3)
    for name in posgrad:
        body = [l for l in _kernel(lines, name) if l and not l.startswith((";", "."))]
        shuffles = [i for i, l in enumerate(body) if l.startswith("ds_bpermute_b32")]
        assert len(shuffles) == 6, (name, len(shuffles))   # x, y, z by 16 lanes, then by 32
        window = body[shuffles[0] - 24:shuffles[-1] + 8]   # the sums that feed the first shuffle .. the adds behind the last
        packed = [l for l in window if l.startswith("v_pk_")]
        assert not packed, f"{name}: packed math in the position-gradient reduction (FNR_NT_MASK={mask}): {packed[:3]}"
        if mask == "0xff":
            assert any(l.startswith("global_load_dwordx2") and l.endswith(" nt") for l in body), "the variant under test streams the Jacobian"


# ---- round 6: the per-wave MLP backward (field_mlp_bwd_pw.hip) ---------------------------------------------------------------
# Its kernels sit at the register limit of two waves per SIMD by design (116 dW accumulator registers in the colour branch): a
# spill would put accumulators into scratch memory inside the loop.  And its base branch carries the same position-gradient
# reduction as k_field_mlp_bwd_base_coop, once per tile.

@pytest.fixture(scope="module")
def per_wave_asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    return _compile("field_mlp_bwd_pw.hip", tmp_path_factory.mktemp("isa_pw") / "field_mlp_bwd_pw.s")


def test_per_wave_backward_kernels_do_not_spill(per_wave_asm):
    text = "\n".join(per_wave_asm)
    meta = re.findall(r"\.name:\s*(\S+)\s*\n(?:.*\n)*?\s*\.private_segment_fixed_size:\s*(\d+)(?:.*\n)*?\s*\.vgpr_count:\s*(\d+)"
                      r"(?:.*\n)*?\s*\.vgpr_spill_count:\s*(\d+)", text)
    kernels = {name: (int(scratch), int(vgpr), int(spill)) for name, scratch, vgpr, spill in meta if name.startswith("_ZN3fnr2pw")}
    # fruit_nerf: (colour | semantic | base | base + position gradient) x (bf16 | bf16x3); fruit_nerf_big: the same without semantic
    assert len(kernels) == 14, sorted(kernels)
    for name, (scratch, vgpr, spill) in kernels.items():
        assert scratch == 0 and spill == 0 and vgpr <= 256, (name, scratch, vgpr, spill)


def test_per_wave_position_gradient_reduction_is_free_of_packed_math(per_wave_asm):
    lines = per_wave_asm
    kernels = [m.group(1) for m in (re.match(r"^(_ZN3fnr2pw23k_field_mlp_bwd_base_pw\w+):", l) for l in lines) if m]
    posgrad = [k for k in kernels if "Lb1EEE" in k]
    assert len(posgrad) == 4, kernels                     # (fruit_nerf | fruit_nerf_big) x (bf16 | bf16x3)
    for name in posgrad:
        body = [l for l in _kernel(lines, name) if l and not l.startswith((";", "."))]
        shuffles = [i for i, l in enumerate(body) if l.startswith("ds_bpermute_b32")]
        assert len(shuffles) in (6, 12, 24), (name, len(shuffles))   # x, y, z by 16 lanes, then by 32 — per tile of the wave
        for k in range(0, len(shuffles), 6):
            window = body[shuffles[k] - 24:shuffles[k + 5] + 8]
            packed = [l for l in window if l.startswith("v_pk_")]
            assert not packed, f"{name}: packed math in the position-gradient reduction: {packed[:3]}"


def test_streaming_accesses_share_partial_waits_only_in_known_kernels():
    """tools/isa_nt_scan.py: partial `s_waitcnt vmcnt(n)` with `nt` and plain accesses both in flight are fine (measured:
    tools/microbench/nt_load_order.hip) — this pins WHICH kernel families have them, so that a new one gets looked at."""
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_nt_scan
    known = ("k_scatter_accumulate", "k_prop_bwd", "k_scatter_emit", "k_hash_encode", "k_prop_density", "k_adam", "k_radam")
    seen = set()
    for src in ("hash_scatter.hip", "hashgrid.hip", "field_mlp_bf16.hip", "field_mlp_bwd_pw.hip", "position_grad.hip", "train.hip"):
        for k, r in isa_nt_scan.scan_source(os.path.join(CSRC, src)).items():
            if r["mixed"]:
                fam = [f for f in known if f in k]
                assert fam, f"{src}: {k[:80]} waits partially with nt and plain accesses in flight: a new place — read it, then list it"
                seen.add(fam[0])
    assert {"k_scatter_accumulate", "k_hash_encode"} <= seen     # the scan does find what is there


def test_transposition_chunk_order_is_bank_conflict_free():
    """field_mlp_bwd_pw.hip's chunk order (chunk (s, c) at 8 (16 c + (s ^ 8 (c >> 1))) bytes) against the LDS banking rules of
    MI355X_MICROARCH.md: ds_write_b64 — groups of 16 contiguous lanes, bank = dword address mod 32; ds_read_b64_tr_b16 — two
    groups of 32 lanes, mod 64.  Every group must touch every bank at most once; the row-major order (32 s + 8 c) must not (it
    is what the counters showed: half of the kernel's LDS cycles were conflicts)."""
    def chunk(s, c):
        return 8 * (16 * c + (s ^ ((c >> 1) << 3)))

    def row_major(s, c):
        return 32 * s + 8 * c

    def worst(order):
        ways = 1
        # writes: lane (j, g) = lane j + 16 g holds chunk (s = j, c = g)
        for g in range(4):
            banks = [((order(j, g) // 4) + d) % 32 for j in range(16) for d in (0, 1)]
            ways = max(ways, max(banks.count(b) for b in set(banks)))
        # transposing reads: lane (t, g) reads chunk (4 g + t / 4, t % 4); lanes 0..31 and 32..63 are serviced together
        for half in range(2):
            banks = [((order(4 * g + (t >> 2), t & 3) // 4) + d) % 64 for g in (2 * half, 2 * half + 1) for t in range(16) for d in (0, 1)]
            ways = max(ways, max(banks.count(b) for b in set(banks)))
        return ways

    assert sorted(chunk(s, c) for s in range(16) for c in range(4)) == list(range(0, 512, 8))   # a permutation of the tile
    assert worst(chunk) == 1
    assert worst(row_major) == 4
