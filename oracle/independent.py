"""Independent float64 NumPy derivations of the arithmetic that `oracle/ns_torch.py` restates from memory.

TEST INFRASTRUCTURE ONLY (never imported by fruitnerf_amd/).  Nothing here shares code with `ns_torch.py`: every
function is written from the PUBLISHED definition of the quantity, not from nerfstudio's implementation, so a test
that finds `ns_torch` == `independent` on random inputs pins the oracle's arithmetic on something external to it.

  hash_grid            Instant-NGP (Mueller et al. 2022) eq. 4 spatial hash  h(x) = (XOR_i x_i * pi_i) mod T  with
                       pi = (1, 2654435761, 805459861) in uint32 arithmetic, d-linear interpolation of the 2^3 cell
                       corners (weight of a corner = product over axes of `o` or `1 - o`)
  real_sh16            real spherical harmonics up to l = 3 from scipy's complex Y_l^m (Condon-Shortley phase):
                       m < 0: sqrt2 (-1)^m Im Y_l^|m| ; m = 0: Y_l^0 ; m > 0: sqrt2 (-1)^m Re Y_l^m
  render_weights       NeRF quadrature (Mildenhall et al. 2020, eq. 3): w_i = T_i (1 - exp(-sigma_i delta_i)),
                       T_i = prod_{j<i} exp(-sigma_j delta_j)
  contract_linf        mip-NeRF 360 (Barron et al. 2022) eq. 10 with the infinity norm
  histogram_cdf        piecewise-linear CDF of a step pdf (inverse-transform sampling, mip-NeRF 360 / NeRF appendix)
  outer_measure_loss   mip-NeRF 360 eq. 13: bound_i = sum of the proposal weights of all proposal intervals that
                       overlap interval i, loss = sum_i max(0, w_i - bound_i)^2 / w_i
  distortion           mip-NeRF 360 eq. 15: sum_ij w_i w_j |m_i - m_j| + 1/3 sum_i w_i^2 (t_{i+1} - t_i)
  weighted_median      first sample whose cumulative weight reaches 1/2

What these canNOT pin (constants and conventions that exist only in nerfstudio 0.3.2's source, absent from this
image — they stay "recalled"): the float32 `pow` that produces the integer level scalings,
the absence of a +0.5 cell offset and the ceil/floor (not floor/floor+1) corner pair, "every level hashed", the SH
basis being evaluated on the SHIFTED UN-normalised direction, histogram_padding = 0.01 / eps = 1e-5 of PDFSampler,
the `+1e-7` of the interlevel loss denominator and its mean over rays x intervals, the "last_sample" background rule.
"""
import itertools

import numpy as np

PRIMES = (np.uint32(1), np.uint32(2654435761), np.uint32(805459861))


def spatial_hash(ix, iy, iz, T):
    """Instant-NGP eq. 4 on uint32 (wrap-around) arithmetic; T must be a power of two or any modulus."""
    with np.errstate(over="ignore"):
        h = (ix.astype(np.uint32) * PRIMES[0]) ^ (iy.astype(np.uint32) * PRIMES[1]) ^ (iz.astype(np.uint32) * PRIMES[2])
    return (h % np.uint32(T)).astype(np.int64)


def hash_grid(x, table, scalings, T):
    """x [N,3] in [0,1]; table [L*T, F]; scalings [L] integers.  Returns [N, L*F] (level-major), float64 blend.

    The CELL a point falls into is a discrete decision and is taken on the float32 product x*scale exactly as any
    float32 implementation must; the interpolation itself is float64."""
    x = np.asarray(x, dtype=np.float32)
    N, L, F = x.shape[0], len(scalings), table.shape[1]
    out = np.zeros((N, L, F), dtype=np.float64)
    tab = np.asarray(table, dtype=np.float64)
    for lvl, s in enumerate(scalings):
        p32 = x * np.float32(s)                        # float32 product decides the cell
        lo = np.floor(p32).astype(np.int64)
        hi = np.ceil(p32).astype(np.int64)
        o = p32.astype(np.float64) - lo                # offset inside the cell, in [0,1)
        for bits in itertools.product((0, 1), repeat=3):
            idx = [hi[:, a] if bits[a] else lo[:, a] for a in range(3)]
            w = np.ones(N)
            for a in range(3):
                w = w * (o[:, a] if bits[a] else 1.0 - o[:, a])
            row = spatial_hash(idx[0], idx[1], idx[2], T) + lvl * T
            out[:, lvl, :] += w[:, None] * tab[row]
    return out.reshape(N, L * F)


def real_sh16(d):
    """Real spherical harmonics l = 0..3 of UNIT vectors d [N,3], ordered (l, m = -l..l)."""
    from scipy import special
    d = np.asarray(d, dtype=np.float64)
    theta = np.arccos(np.clip(d[:, 2], -1.0, 1.0))     # polar angle
    phi = np.arctan2(d[:, 1], d[:, 0])                 # azimuth
    cols = []
    for l in range(4):
        for m in range(-l, l + 1):
            if hasattr(special, "sph_harm_y"):
                Y = special.sph_harm_y(l, abs(m), theta, phi)
            else:
                Y = special.sph_harm(abs(m), l, phi, theta)
            if m < 0:
                cols.append(np.sqrt(2.0) * (-1) ** m * Y.imag)
            elif m == 0:
                cols.append(Y.real)
            else:
                cols.append(np.sqrt(2.0) * (-1) ** m * Y.real)
    return np.stack(cols, axis=-1)


def render_weights(sigma, delta):
    """sigma, delta [R,S] -> w [R,S] by an explicit running product of per-interval transmittances."""
    sigma = np.asarray(sigma, dtype=np.float64)
    delta = np.asarray(delta, dtype=np.float64)
    R, S = sigma.shape
    w = np.zeros((R, S))
    for r in range(R):
        T = 1.0
        for i in range(S):
            a = 1.0 - np.exp(-sigma[r, i] * delta[r, i])
            w[r, i] = T * a
            T *= np.exp(-sigma[r, i] * delta[r, i])
    return w


def contract_linf(p):
    p = np.asarray(p, dtype=np.float64)
    n = np.max(np.abs(p), axis=-1, keepdims=True)
    safe = np.where(n > 0, n, 1.0)
    return np.where(n <= 1.0, p, (2.0 - 1.0 / safe) * (p / safe))


def histogram_cdf(edges, pdf_mass, s):
    """CDF at positions s of the step density with mass pdf_mass[i] on [edges[i], edges[i+1]] (masses sum to 1)."""
    c = np.concatenate([[0.0], np.cumsum(np.asarray(pdf_mass, dtype=np.float64))])
    return np.interp(np.asarray(s, dtype=np.float64), np.asarray(edges, dtype=np.float64), c)


def outer_measure_loss(t, w, t_env, w_env, eps=0.0):
    """One ray.  t [n+1], w [n] fine histogram; t_env [m+1], w_env [m] proposal histogram.  Brute force over pairs:
    an envelope interval counts if it has a non-empty intersection with the fine interval."""
    n, m = len(w), len(w_env)
    loss = np.zeros(n)
    for i in range(n):
        bound = 0.0
        for j in range(m):
            if t_env[j] < t[i + 1] and t_env[j + 1] > t[i]:
                bound += w_env[j]
        loss[i] = max(0.0, w[i] - bound) ** 2 / (w[i] + eps)
    return loss


def distortion(t, w):
    n = len(w)
    mid = [(t[i] + t[i + 1]) / 2 for i in range(n)]
    inter = sum(w[i] * w[j] * abs(mid[i] - mid[j]) for i in range(n) for j in range(n))
    intra = sum(w[i] ** 2 * (t[i + 1] - t[i]) for i in range(n)) / 3.0
    return inter + intra


def weighted_median(values, w):
    acc = 0.0
    for v, wi in zip(values, w):
        acc += wi
        if acc >= 0.5:
            return v
    return values[-1]


def lin_disp_piecewise(u, near, far):
    """Distance for normalised coordinate u in [0,1]: the warp s(t) is linear in t up to t = 1 (s = t/2) and linear in
    DISPARITY beyond (s = 1 - 1/(2t)); samples are uniform in s between s(near) and s(far)."""
    def s_of(t):
        return t / 2.0 if t < 1.0 else 1.0 - 1.0 / (2.0 * t)

    def t_of(s):
        return 2.0 * s if s < 0.5 else 1.0 / (2.0 - 2.0 * s)
    s = u * s_of(far) + (1.0 - u) * s_of(near)
    return t_of(s)
