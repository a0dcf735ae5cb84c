"""profiles/<tag>_kernel_trace_pmc.md and profiles/pmc_traffic.json from the aggregates tools/gpu_call.sh (legs kt, pmc, kt2) left under
profiles/<tag>_raw/.   usage: python tools/make_profile_summary.py [tag, default r03] [git revision the run was made from]"""
import json
import re
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
rev = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True,
                                                           text=True, cwd='/root/repo').stdout.strip()
base = f'/root/repo/profiles/{tag}_raw/'
dst = f'/root/repo/profiles/{tag}_kernel_trace_pmc.md'


def kt(path):
    rows = []
    for l in open(path):
        m = re.match(r'(\S.*?)\s+grid\s+(\d*)\s+n\s+(\d+)\s+total\s+([\d.]+) ms median\s+([\d.]+) us min\s+([\d.]+)', l)
        if m:
            rows.append((m.group(1).strip(), m.group(2), int(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6))))
    return rows


def pmc(path):
    d, cur = {}, None
    for l in open(path):
        if 'dispatches' in l:
            cur = l.split(' dispatches')[0].strip()
            d[cur] = {'_n': int(l.split('dispatches')[1])}
        else:
            p = l.split()
            if len(p) == 2 and cur:
                d[cur][p[0]] = float(p[1])
    return d


def short(n):
    n = n.replace('fnr::', '')
    m = re.match(r'_ZN3fnr(?:2pw)?\d+(k_[a-z_0-9]+?)INS', n)
    if m:
        n = m.group(1) + '<...> (mangled)'
    return n[:72]


def line(fn):
    for l in open(base + fn):
        if l.startswith('{'):
            return json.loads(l)


def rf(x):
    s = "%s: %s %s = %.3f of %s" % (x['kernel'], x['achieved'], x['unit'], x['frac'], x['peak'])
    if 'frac_of_fp32_mfma_peak' in x:
        s += " (%.2f of the fp32-MFMA peak; issued %s TF = %.2f of the bf16 peak)" % (
            x['frac_of_fp32_mfma_peak'], x['issued_bf16_tflops'], x['issued_frac_of_bf16_peak'])
    if x.get('alg_bytes_per_launch_fixed'):
        s += " (incl. the table optimiser's %.0f MB per launch; %.0f GB/s without)" % (
            x['alg_bytes_per_launch_fixed'] / 1e6, x['achieved_without_fixed_bytes'])
    return s + ", %.0f us/launch" % (x['avg_launch_ms'] * 1e3)


def find(d, name_part, also=None):
    for k in d:
        if name_part in k and (also is None or also in k):
            return d[k]
    return {}


rows = kt(base + 'prof_kernel_trace.txt')
f, w, sq = pmc(base + 'prof_fetch.txt'), pmc(base + 'prof_write.txt'), pmc(base + 'prof_sq.txt')
steps = max(n for name, g, n, *_ in rows if 'k_train_losses' in name)   # one per training step
round_no = tag[1:].lstrip("0")
out = [f'# Round {round_no} - rocprofv3 of `FNR_SERIALIZE_STREAMS=1 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-quality` (default arithmetic: bf16x3), build {rev}\n',
       f'Collected by `tools/gpu_call.sh (legs kt, pmc, kt2)` on 1xMI355X (gfx950, ROCm 7.2): one `--kernel-trace --stats` pass and three separate '
       f'`--pmc` passes (FETCH_SIZE | WRITE_SIZE | SQ counters, each with `--kernel-trace` only), {steps} training steps each (10 '
       'warm-up + 60 timed + 12 of the per-entry-point breakdown).  FNR_SERIALIZE_STREAMS=1: the second HIP stream\'s launches '
       '(proposal-network backward, ray-gradient reduction, camera step, the next step\'s sampling) run before / after the launch '
       'stream\'s instead of next to them on EVERY step, as bench.py does on the steps it brackets with events — a duration below '
       'is one kernel with the GPU to itself; `prof_kernel_trace_two_streams.txt` under the raw aggregates is the default run.  '
       'Raw aggregates: `profiles/' + tag + '_raw/`; this table: '
       '`tools/make_profile_summary.py`.  Times are per STEP (kernel total / steps; the proposal-network backward runs on about '
       'half of the steps of this window).  FETCH_SIZE is doubled per MI355X_MICROARCH.md "HBM" (gfx950 reports half of a wide '
       'streaming read) and, like WRITE_SIZE, given in MB per dispatch (rocprofv3 reports KB).  SQ columns are ratios of '
       'per-dispatch counters (SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader '
       'engines: the last column is MFMA_BUSY / (32 x SQ_BUSY)).  A kernel that serves several entry points appears once per '
       'grid size (x dimension).\n',
       '| kernel | grid x | launches/step | us/step | median us | FETCHx2 MB | WRITE MB | VALU-active / wave-cycles | WAIT_ANY / wave-cycles | MFMA-busy / SIMD-cycles |',
       '|---|---|---|---|---|---|---|---|---|---|']
tot = 0.0


def pmc_for(name, grid_x, d):
    """counter record of a kernel-trace row: same kernel; the emit's main-field call is the one after the MLP backward's last
    launch (k_finish_weights; k_reduce_dw before round 6; k_position_contract in the fp32 window)"""
    cands = [k for k in d if k.split(' grid ')[0].split(' after ')[0][:50] == name[:50]]
    if 'k_scatter_emit' in name:
        main = [k for k in cands if 'after k_reduce_dw' in k or 'after k_finish_weights' in k]
        prop = [k for k in cands if 'after k_prop_reduce' in k]
        if grid_x == '196608' and main:
            return d[main[0]]
        if prop:   # both proposal levels' calls: dispatch-weighted mean is what the aggregate holds per key
            want = {'1048576': '1048576', '393216': '786432'}.get(grid_x)
            hit = [k for k in prop if want and k.endswith('grid ' + want)]
            return d[hit[0]] if hit else {}
        return {}
    if len(cands) == 1:
        return d[cands[0]]
    # several grid sizes (proposal levels): counter files carry the TOTAL grid, the trace the x dimension — same for 1-D grids
    hit = [k for k in cands if k.endswith('grid ' + grid_x)]
    return d[hit[0]] if hit else (d[cands[0]] if cands else {})


for name, grid_x, n, total, med, mn in rows:
    fs = pmc_for(name, grid_x, f).get('FETCH_SIZE')
    ws = pmc_for(name, grid_x, w).get('WRITE_SIZE')
    s = pmc_for(name, grid_x, sq)
    wc = s.get('SQ_WAVE_CYCLES')
    tot += total
    c_f = '' if fs is None else '%.1f' % (2 * fs / 1024)
    c_w = '' if ws is None else '%.1f' % (ws / 1024)
    c_v = '' if not wc else '%.2f' % (s.get('SQ_ACTIVE_INST_VALU', 0) / wc)
    c_a = '' if not wc else '%.2f' % (s.get('SQ_WAIT_ANY', 0) / wc)
    c_m = '' if not s or not s.get('SQ_BUSY_CYCLES') else '%.2f' % (s.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (32.0 * s['SQ_BUSY_CYCLES']))
    out.append('| `%s` | %s | %.2f | %.1f | %.1f | %s | %s | %s | %s | %s |' % (short(name), grid_x, n / steps, total * 1e3 / steps, med, c_f, c_w, c_v, c_a, c_m))
out.append('\nSum of kernel time: %.0f us per step (the un-profiled step time is in the bench lines below).\n' % (tot * 1e3 / steps))

# ---- HBM traffic per entry-point launch (what bench.py prints as roofline.traffic) --------------------------------------
ENTRY = {
    'hash_encode_bwd[196608]': [('k_scatter_emit', '196608'), ('k_scatter_accumulate<true>', None)],
    'hash_encode_fwd[196608]': [('k_hash_encode', None)],
    'field_mlp_bwd[196608]': [('k_field_mlp_bwd_color_pw', None), ('k_color_ray_grads', None), ('k_field_mlp_bwd_sem_pw', None),
                              ('k_field_mlp_bwd_base_pw', None), ('k_finish_weights', None)],
    'field_mlp_fwd[196608]': [('k_prepare_field', None), ('k_field_mlp_fwd_bf16', None)],
}
traffic = {}
out.append('## HBM traffic per entry-point launch (counters: FETCH_SIZE x 2 + WRITE_SIZE, summed over the entry point\'s kernels)\n')
out.append('| entry point | kernels | counter MB / launch | algorithmic MB / launch (bench.py) | counter / algorithmic |')
out.append('|---|---|---|---|---|')
bench = line('bench_fruit_nerf.log')
ALG = {'hash_encode_bwd[196608]': (2 * 1024.0 + 128.0) * 196608 + 24.0 * 16777216,   # fused sweep: p, m, v read + written
       'hash_encode_fwd[196608]': (1024.0 + 128.0) * 196608}
for ep, parts in ENTRY.items():
    total_b, names = 0.0, []
    ok = True
    for part, gx in parts:
        row = [r for r in rows if part in r[0] and (gx is None or r[1] == gx)]
        if not row:
            ok = False
            break
        fs = pmc_for(row[0][0], row[0][1], f).get('FETCH_SIZE')
        ws = pmc_for(row[0][0], row[0][1], w).get('WRITE_SIZE')
        if fs is None or ws is None:
            ok = False
            break
        total_b += (2 * fs + ws) * 1024.0
        names.append(short(row[0][0]).split('<')[0])
    if not ok:
        continue
    traffic[ep] = {'bytes_per_launch': round(total_b), 'kernels': names}
    alg = ALG.get(ep)
    out.append('| `%s` | %s | %.1f | %s | %s |' % (ep, ', '.join(names), total_b / 1e6, '%.1f' % (alg / 1e6) if alg else '(MFMA-bound: FLOP)',
                                                   '%.2f' % (total_b / alg) if alg else ''))
json.dump({'source': f'profiles/{tag}_kernel_trace_pmc.md: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only) of '
                     '`FNR_SERIALIZE_STREAMS=1 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-quality`, FETCH_SIZE x 2 per MI355X_MICROARCH.md "HBM" '
                     '(tools/gpu_call.sh (legs kt, pmc, kt2), tools/make_profile_summary.py)',
           'build': rev, 'entry_points': {'fruit_nerf': traffic}}, open('/root/repo/profiles/pmc_traffic.json', 'w'), indent=1)

out.append('\n## Bench lines of the same build (un-profiled)\n')
out.append('| command | rays/s | ms/step | `roofline` | `roofline_other_bound` |')
out.append('|---|---|---|---|---|')
for fn, cmd in (('bench_fruit_nerf.log', 'python bench.py'), ('bench_fruit_nerf_fp32.log', 'python bench.py --mlp-precision fp32'),
                ('bench_fruit_nerf_big.log', 'python bench.py --method fruit_nerf_big')):
    import os
    if not os.path.exists(base + fn):
        continue
    d = line(fn)
    out.append('| `%s` | %.0f | %s | %s | %s |' % (cmd, d['value'], d['ms_per_step'], rf(d['roofline']), rf(d['roofline_other_bound'])))
d = bench
sec = d['secondary']
out.append('\nDefault line, other fields: cpu_baseline ' + json.dumps(d['cpu_baseline'])[:460] + '; quality ' + json.dumps(d['quality'])
           + '; rays/s by MLP arithmetic ' + json.dumps(sec['train_rays_per_s_by_mlp_precision'])
           + '; eval %.3g rays/s; export 256^3 %.3g samples/s (passes %s ms); export points %s; fruit_nerf_big window %s rays/s (%s ms/step).\n'
           % (sec['eval_rays_per_s'], sec['export_samples_per_s'], sec['export_pass_ms'], json.dumps(sec['export_points']),
              sec.get('fruit_nerf_big', {}).get('value'), sec.get('fruit_nerf_big', {}).get('ms_per_step')))
import os
if os.path.exists(base + 'bench_fruit_nerf_big.log'):
    dbig = line('bench_fruit_nerf_big.log')
    out.append('`fruit_nerf_big` line: quality ' + json.dumps(dbig['quality']) + '.\n')
else:   # the default line's own 20 000-step look at the big method
    out.append('`fruit_nerf_big` (the default line\'s secondary run): quality ' + json.dumps(sec.get('fruit_nerf_big', {}).get('quality')) + '.\n')
big = kt(base + 'prof_kernel_trace_big.txt')
sb = max(n for name, g, n, *_ in big if 'k_train_losses' in name)
out.append(f'## `fruit_nerf_big` (8192 rays, samples 512/256/128, T = 2^21): kernel trace of `bench.py --method fruit_nerf_big --steps 40 --warmup 10`, {sb} steps\n')
out.append('| kernel | grid x | launches/step | us/step | median us |')
out.append('|---|---|---|---|---|')
for name, grid_x, n, total, med, mn in big[:26]:
    out.append('| `%s` | %s | %.2f | %.0f | %.0f |' % (short(name), grid_x, n / sb, total * 1e3 / sb, med))
open(dst, 'w').write('\n'.join(out) + '\n')
print(dst)
