"""profiles/r02_kernel_trace_pmc.md from the aggregates tools/prof_round.sh left under profiles/r02_raw/."""
import json
import re
import sys

base = sys.argv[1] if len(sys.argv) > 1 else '/root/repo/profiles/r02_raw/'
dst = sys.argv[2] if len(sys.argv) > 2 else '/root/repo/profiles/r02_kernel_trace_pmc.md'


def kt(path):
    rows = []
    for l in open(path):
        m = re.match(r'(\S.*?)\s+grid\s+n\s+(\d+)\s+total\s+([\d.]+) ms median\s+([\d.]+) us min\s+([\d.]+)', l)
        if m:
            rows.append((m.group(1).strip(), int(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5))))
    return rows


def pmc(path):
    d, cur = {}, None
    for l in open(path):
        if 'dispatches' in l:
            cur = l.split(' dispatches')[0].strip()
            d[cur] = {}
        else:
            p = l.split()
            if len(p) == 2 and cur:
                d[cur][p[0]] = float(p[1])
    return d


def short(n):
    n = n.replace('fnr::', '')
    m = re.match(r'_ZN3fnr\d+(k_[a-z_0-9]+?)INS', n)
    if m:
        n = m.group(1) + '<...> (mangled in the trace)'
    return n[:64]


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


rows = kt(base + 'prof_kernel_trace.txt')
f, w, sq = pmc(base + 'prof_fetch.txt'), pmc(base + 'prof_write.txt'), pmc(base + 'prof_sq.txt')
steps = 86
out = ['# Round 2 - rocprofv3 of `python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-quality` (default arithmetic: bf16x3)\n',
       'Collected by `tools/prof_round.sh` on 1xMI355X (gfx950, ROCm 7.2): one `--kernel-trace --stats` pass and three separate '
       '`--pmc` passes (FETCH_SIZE | WRITE_SIZE | SQ counters), 86 training steps each (10 warm-up + 4 instrumented + 60 timed + '
       '12 of the per-entry-point breakdown).  Raw aggregates: `profiles/r02_raw/`; this table: `tools/make_profile_summary.py`.  '
       'Times are per STEP (kernel total / 86; the proposal-network backward runs on 48 of the 86 steps).  FETCH_SIZE is doubled '
       'per MI355X_MICROARCH.md (gfx950 reports half of a wide streaming read) and, like WRITE_SIZE, given in MB per dispatch '
       '(rocprofv3 reports KB).  SQ columns are ratios of per-dispatch counters (wave-cycles count quad-cycles; '
       'SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader engines: the last column is '
       'MFMA_BUSY / (32 x SQ_BUSY)).\n',
       '| kernel | launches/step | us/step | median us | FETCHx2 MB | WRITE MB | VALU-active / wave-cycles | WAIT_ANY / wave-cycles | MFMA-busy / SIMD-cycles |',
       '|---|---|---|---|---|---|---|---|---|']
tot = 0.0
for name, n, total, med, mn in rows:
    key = [k for k in f if k[:50] == name[:50]]
    k = key[0] if key else None
    fs = f.get(k, {}).get('FETCH_SIZE')
    ws = w.get(k, {}).get('WRITE_SIZE')
    s = sq.get(k, {})
    wc = s.get('SQ_WAVE_CYCLES')
    tot += total
    c_f = '' if fs is None else '%.1f' % (2 * fs / 1024)
    c_w = '' if ws is None else '%.1f' % (ws / 1024)
    c_v = '' if not wc else '%.2f' % (s.get('SQ_ACTIVE_INST_VALU', 0) / wc)
    c_a = '' if not wc else '%.2f' % (s.get('SQ_WAIT_ANY', 0) / wc)
    c_m = '' if not s or not s.get('SQ_BUSY_CYCLES') else '%.2f' % (s.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (32.0 * s['SQ_BUSY_CYCLES']))
    out.append('| `%s` | %.2f | %.1f | %.1f | %s | %s | %s | %s | %s |' % (short(name), n / steps, total * 1e3 / steps, med, c_f, c_w, c_v, c_a, c_m))
out.append('\nSum of kernel time: %.0f us per step (the un-profiled step time is in the bench lines below).\n' % (tot * 1e3 / steps))
out.append('## Bench lines of the same build (un-profiled)\n')
out.append('| command | rays/s | ms/step | dominant entry point (`roofline`) | `roofline_other_bound` |')
out.append('|---|---|---|---|---|')
for fn, cmd in (('bench_fruit_nerf.log', 'python bench.py'), ('bench_fruit_nerf_fp32.log', 'python bench.py --mlp-precision fp32'),
                ('bench_fruit_nerf_big.log', 'python bench.py --method fruit_nerf_big'),
                ('bench_fruit_nerf_big_fp32.log', 'python bench.py --method fruit_nerf_big --mlp-precision fp32')):
    d = line(fn)
    out.append('| `%s` | %.0f | %s | %s | %s |' % (cmd, d['value'], d['ms_per_step'], rf(d['roofline']), rf(d['roofline_other_bound'])))
d = line('bench_fruit_nerf.log')
sec = d['secondary']
out.append('\nDefault line, other fields: cpu_baseline ' + json.dumps(d['cpu_baseline'])[:420] + '; quality ' + json.dumps(d['quality'])
           + '; rays/s by MLP arithmetic after 1846 steps ' + json.dumps(sec['train_rays_per_s_by_mlp_precision'])
           + '; eval %.3g rays/s; export 256^3 %.3g samples/s (passes %s ms).\n' % (sec['eval_rays_per_s'], sec['export_samples_per_s'], sec['export_pass_ms']))
big = kt(base + 'prof_kernel_trace_big.txt')
out.append('## `fruit_nerf_big` (8192 rays, samples 512/256/128, T = 2^21): kernel trace of `bench.py --method fruit_nerf_big --steps 40 --warmup 10`, 66 steps\n')
out.append('| kernel | launches/step | us/step | median us |')
out.append('|---|---|---|---|')
for name, n, total, med, mn in big[:24]:
    out.append('| `%s` | %.2f | %.0f | %.0f |' % (short(name), n / 66, total * 1e3 / 66, med))
open(dst, 'w').write('\n'.join(out) + '\n')
print(dst)
