"""The host side of the compressed-catalogue option (csrc/lf_compress.h) through the C ABI's host-only
helper lf_compress_keys: no GPU needed.  The sums over the pseudo-sources are compared with the sums over the
sources themselves, both taken in long double, for walkers all over (and on the corners of) the prior box."""
import numpy as np
import pytest

from lumfuncmcmc_amd.capi import compress_keys

FCMIN = 0.1
FR = abs((2 * FCMIN - 1) ** 2 / (1 - (2 * FCMIN - 1) ** 2))          # VmaxLumFunc.py:164-165
LD = np.longdouble


def g_free(x, aC):
    """ln fc(alpha_C x) / (1 - exp(-10^(x - b))): VmaxLumFunc.py:118-127, :141, in a cancellation-free form"""
    x = x.astype(LD)
    num = LD(aC) * x
    s = np.sqrt(1 + num * num)
    lnfc = np.where(num >= 0, np.log1p(-0.5 / (s * (s + num))), -np.log(2 * s * (s - num)))
    b = -np.sqrt(LD(FR) / LD(aC) ** 2)
    return lnfc / -np.expm1(-(LD(10) ** (x - b)))


def quad(y, zp, x):
    z1, z2, z3 = zp
    return (y[0] * (x - z2) * (x - z3) / ((z1 - z2) * (z1 - z3)) + y[1] * (x - z1) * (x - z3) / ((z2 - z1) * (z2 - z3))
            + y[2] * (x - z1) * (x - z2) / ((z3 - z1) * (z3 - z2)))


@pytest.mark.parametrize("n,dist", [(200000, "uniform"), (50000, "steep"), (3000, "clumped")])
def test_free_pseudo_sources_reproduce_the_sums(n, dist):
    rng = np.random.default_rng(n)
    if dist == "uniform":
        logf = rng.uniform(-18.2, -14.7, n)
    elif dist == "steep":                                  # number counts rising to the faint end
        logf = -16.9 + rng.exponential(0.3, n)
    else:                                                  # a few tight clumps and isolated sources
        logf = np.concatenate([rng.normal(c, 1e-3, n // 3) for c in (-17.5, -16.2, -15.0)] + [rng.uniform(-18, -14, n - 3 * (n // 3))])
    node, w, bound = compress_keys(0, [FR, 1.0, 6.0, 1.0, 6.0], logf)
    assert bound <= 1e-16
    assert node.size < max(n // 20, 1200) and node.size <= n
    assert abs(w.sum() - n) <= 1e-9 * n
    worst = 0.0
    for t in range(40):
        aC = (1.0, 6.0, 1.0, 6.0)[t] if t < 4 else rng.uniform(1, 6)
        Fl = (1.0, 1.0, 6.0, 6.0)[t] if t < 4 else rng.uniform(1, 6)
        lF = np.log10(Fl) - 17
        exact = float(g_free(logf - lF, aC).sum())
        approx = float((w.astype(LD) * g_free(node - lF, aC)).sum())
        worst = max(worst, abs(approx - exact) / abs(exact))
    print("free %s n=%d -> %d pseudo-sources, bound %.1e, worst rel %.2e" % (dist, n, node.size, bound, worst))
    assert worst < 2e-15


@pytest.mark.parametrize("pivots", [(1.20, 1.53, 1.86), (1.20, 1.76, 2.32), (1.18, 1.36, 1.54)])
def test_zevol_pseudo_sources_reproduce_the_sums(pivots):
    rng = np.random.default_rng(7)
    n = 100000
    z = rng.uniform(1.16, 1.90, n)
    P = 10.0 ** (rng.uniform(41.0, 43.5, n) - 42.0)
    node, w, bound = compress_keys(1, [40.0, 45.0] + list(pivots), z, P)
    assert bound <= 1e-16 and node.size < 8000
    worst = 0.0
    for t in range(30):
        y = np.array([(45.0 if t & 1 else 40.0), (45.0 if t & 2 else 40.0), (45.0 if t & 4 else 40.0)] if t < 8
                     else rng.uniform(40, 45, 3), dtype=LD)
        exact = float((P.astype(LD) * LD(10) ** (42 - quad(y, pivots, z.astype(LD)))).sum())
        approx = float((w.astype(LD) * LD(10) ** (42 - quad(y, pivots, node.astype(LD)))).sum())
        worst = max(worst, abs(approx - exact) / abs(exact))
    print("zevol pivots %s -> %d pseudo-sources, worst rel %.2e" % (pivots, node.size, worst))
    assert worst < 5e-14        # node rounding: d ln h / dz is up to ~200 with the close pivots


def test_small_and_degenerate_inputs_are_kept_as_they_are():
    key = np.array([-16.5, -16.4, -17.0])
    node, w, _ = compress_keys(0, [FR, 1.0, 6.0, 1.0, 6.0], key)
    assert np.array_equal(node, key) and np.array_equal(w, np.ones(3))
    node, w, _ = compress_keys(0, [FR, 1.0, 6.0, 1.0, 6.0], np.full(100, -16.5))
    assert np.array_equal(node, np.full(100, -16.5))
    node, w, _ = compress_keys(0, [FR, 1.0, 6.0, 1.0, 6.0], np.zeros(0))
    assert node.size == 0
    with pytest.raises(RuntimeError):
        compress_keys(0, [FR, 1.0, 6.0, 1.0, 6.0], np.array([np.nan] * 40))


def test_compressed_grid_reproduces_the_double_sum():
    """sum_j wL_j T_j sum_k c_k F(L_j - D_k - lF) over the S^2 lattice against the shared-node form
    sum_b sum_n F(u_bn - lF) sum_r T_{row0_b + r} omega_b[r][n] (csrc/lf_compress.h: compress_grid), long double."""
    from lumfuncmcmc_amd.capi import compress_grid
    S = 101
    L = np.linspace(41.0, 43.5, S)
    wL = np.full(S, L[1] - L[0]); wL[[0, -1]] *= 0.5
    z = np.linspace(1.16, 1.90, S)
    Dk = 57.9 + 0.5 * (z - 1.16) / 0.74 - 0.1 * (z - 1.16) * (z - 1.9)       # log10(4 pi DL^2)-like, monotonic
    ck = (1 + 0.1 * z) * (z[1] - z[0]); ck[[0, -1]] *= 0.5
    g = compress_grid([FR, 1.0, 6.0, 1.0, 6.0], L, wL, ck, Dk)
    nb = g["u"].shape[0]
    assert g["bound"] <= 1e-16 and nb * 16 < S * S // 8
    rng = np.random.default_rng(3)
    worst = 0.0
    for t in range(12):
        aC = (1.0, 6.0, 6.0)[t] if t < 3 else rng.uniform(1, 6)
        Fl = (1.0, 6.0, 1.0)[t] if t < 3 else rng.uniform(1, 6)
        lF = np.log10(Fl) - 17
        T = np.exp(LD(-0.8) * (L.astype(LD) - 42) - LD(10) ** (L.astype(LD) - rng.uniform(41.8, 43.0)))   # Schechter-like row factor
        F = lambda u: np.exp(g_free(np.asarray(u, dtype=LD) - lF, aC))
        full = sum(wL[j] * T[j] * (ck.astype(LD) * F(L[j] - Dk)).sum() for j in range(S))
        comp = LD(0)
        for b in range(nb):
            om = g["omega"][g["off"][b]:g["off"][b] + g["nrows"][b] * 16].reshape(g["nrows"][b], 16).astype(LD)
            R = (T[g["row0"][b]:g["row0"][b] + g["nrows"][b], None] * om).sum(axis=0)
            comp += (R * F(g["u"][b])).sum()
        worst = max(worst, float(abs(comp - full) / full))
    print("compressed grid: %d bins -> %d nodes (of %d lattice points), worst rel %.2e" % (nb, nb * 16, S * S, worst))
    assert worst < 5e-16
