"""Which rows of the hash tables can ever receive a gradient?  With every level hashed (nerfstudio's torch layout:
16 x 2^19 rows), level l only ever addresses hash(v) for the (s_l + 1)^3 lattice vertices v of its resolution
s_l = floor(16 * g^l): for the coarse levels that is a small, STATIC subset of the 524 288 rows — rows outside it keep
gradient, moment and update exactly zero for the whole training run.  CPU only.
  python tools/microbench/static_rows.py"""
import numpy as np

L, T, BASE, MAXRES = 16, 1 << 19, 16, 2048
g = np.exp((np.log(MAXRES) - np.log(BASE)) / (L - 1))
total = 0
for lvl in range(L):
    s = int(np.floor(BASE * g ** lvl))
    n = s + 1
    if n ** 3 > 40 * T:
        print(f"level {lvl:2d} res {s:5d}: {n ** 3:>12d} vertices -> all {T} rows (saturated)")
        total += T
        continue
    ax = np.arange(n, dtype=np.int64)
    x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")
    # int32 wrap-around arithmetic of the oracle's hash (x * 1) ^ (y * 2654435761) ^ (z * 805459861), mod T
    h = (x.astype(np.uint32) * np.uint32(1)) ^ (y.astype(np.uint32) * np.uint32(2654435761)) ^ \
        (z.astype(np.uint32) * np.uint32(805459861))
    rows = np.unique(h.ravel() % np.uint32(T)).size
    total += rows
    print(f"level {lvl:2d} res {s:5d}: {n ** 3:>12d} vertices -> {rows:7d} distinct rows ({100.0 * rows / T:5.1f} %)")
print(f"addressable rows: {total} of {L * T} ({100.0 * total / (L * T):.1f} %); never touched: {L * T - total} rows = "
      f"{(L * T - total) * 8 / 1e6:.1f} MB of fp32 gradient per exchange")
