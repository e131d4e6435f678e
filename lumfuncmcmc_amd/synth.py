"""Synthetic emission-line catalogues and walker positions.

One generator shared by the golden-vector script (run under the conda
interpreter that can import the reference), the tests, and bench.py, so the
three always see the same catalogue for a given (N, seed).  numpy-only and
python-3.9 compatible on purpose.

Instrument constants are the values of the reference's configLF.py:1-42
(restated here as data, they are inputs of the path, not code).
"""
import numpy as np

# configLF.py:6, :10, :18-21, :9, :13, :24-32
FLIM = [2.72, 3.61, 2.55, 3.31, 3.30]
ALPHA_C = 4.56
OMEGA_0 = [val * 0.85 * 3600 for val in [121.9, 122.2, 116.0, 147.3, 118.7]]
FLIM_LIMS = [1.0, 6.0]
ALPHA_LIMS = [1.0, 7.0]
SCH_AL = -1.49
SCH_AL_LIMS = [-3.0, 1.0]
LSTAR = 42.5
LSTAR_LIMS = [40.0, 45.0]
PHISTAR = -2.0
PHISTAR_LIMS = [-8.0, 5.0]
LC, LH = 40.0, 46.0
MIN_COMP_FRAC = 0.0
FCMIN = 0.1

ZLO, ZHI = 1.16, 1.90          # [OIII] grism range, cf. VmaxLumFunc.py:235
LUMLO, LUMHI = 41.0, 43.5

# finite box used for timing walkers: no walker takes the -inf early-out
BOX = {
    "Lstar": (41.5, 44.5), "phistar": (-4.0, 0.0), "sch_al": (-2.5, 0.5),
    "Flim": (1.5, 5.5), "alpha": (1.5, 6.5),
}


def field_index(n, nf=5):
    """Contiguous, (almost) equal field ranges: field_ind[nf+1]."""
    return np.round(np.linspace(0, n, nf + 1)).astype(np.int64)


def catalogue(n, seed=20241016, nf=5, zslices=0):
    """Return dict(z, lum, lum_e, field_ind) for an n-source catalogue.

    zslices > 0 lays the redshifts out in that many contiguous equal-width
    slices over [ZLO, ZHI] (BASELINE config 5: "8 z-bins" is only a layout).
    """
    rng = np.random.default_rng(seed)
    if zslices and zslices > 0:
        edges = np.linspace(ZLO, ZHI, zslices + 1)
        cnt = np.diff(np.round(np.linspace(0, n, zslices + 1)).astype(np.int64))
        z = np.concatenate([rng.uniform(edges[s], edges[s + 1], int(cnt[s]))
                            for s in range(zslices)])
    else:
        z = rng.uniform(ZLO, ZHI, n)
    lum = rng.uniform(LUMLO, LUMHI, n)
    lum_e = np.full(n, 0.05)
    return {"z": z, "lum": lum, "lum_e": lum_e, "field_ind": field_index(n, nf)}


def split_fields(arr, field_ind):
    """The reference constructors take per-field lists (lumfuncmcmc.py:143)."""
    return [arr[field_ind[i]:field_ind[i + 1]] for i in range(len(field_ind) - 1)]


def ndim_of(variant, fix_sch_al=False, nf=5):
    if variant == "free":
        return 2 + (0 if fix_sch_al else 1) + nf + 1
    if variant == "fixcomp":
        return 2 + (0 if fix_sch_al else 1)
    if variant == "zevol":
        return 6 + (0 if fix_sch_al else 1)
    raise ValueError(variant)


def walkers(variant, nwalkers, seed=1, fix_sch_al=False, nf=5):
    """theta[nwalkers, ndim] uniform in the finite BOX (layouts:
    lumfuncmcmc.py:320-337, lumfuncmcmc_z.py:332-341)."""
    rng = np.random.default_rng(seed)
    cols = []
    if variant in ("free", "fixcomp"):
        cols += [BOX["Lstar"], BOX["phistar"]]
        if not fix_sch_al:
            cols += [BOX["sch_al"]]
        if variant == "free":
            cols += [BOX["Flim"]] * nf + [BOX["alpha"]]
    elif variant == "zevol":
        cols += [BOX["Lstar"]] * 3 + [BOX["phistar"]] * 3
        if not fix_sch_al:
            cols += [BOX["sch_al"]]
    else:
        raise ValueError(variant)
    lims = np.array(cols, dtype=np.float64)
    u = rng.random((nwalkers, len(cols)))
    return u * (lims[:, 1] - lims[:, 0]) + lims[:, 0]
