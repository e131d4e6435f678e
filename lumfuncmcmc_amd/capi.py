"""ctypes binding of liblfmcmc.so (include/lfmcmc.h).

There is no CPU fallback: if the shared object is missing or does not load, importing
callers get a RuntimeError that says so.  Build it with `python -m lumfuncmcmc_amd.build`
(or `__graft_entry__.build()`).
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liblfmcmc.so")

LF_ABI_VERSION = 2
LF_MAX_FIELDS = 8
LF_FREE, LF_FIXCOMP, LF_ZEVOL = 0, 1, 2
VARIANTS = {"free": LF_FREE, "fixcomp": LF_FIXCOMP, "zevol": LF_ZEVOL}
LF_OK = 0
LIM_ORDER = ("Lstar", "phistar", "sch_al", "Flim", "alpha")

MPC_CM = 3.086e24                                  # lumfuncmcmc.py:70

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int64_p = ctypes.POINTER(ctypes.c_int64)


class LfDesc(ctypes.Structure):
    _fields_ = [
        ("variant", ctypes.c_int32), ("fix_sch_al", ctypes.c_int32),
        ("nf", ctypes.c_int32), ("S", ctypes.c_int32), ("N", ctypes.c_int64),
        ("field_ind", _c_int64_p), ("lum", _c_double_p), ("logf", _c_double_p),
        ("z", _c_double_p), ("om_arr", _c_double_p), ("omega0", _c_double_p),
        ("logL", _c_double_p), ("zarr", _c_double_p), ("volume_part", _c_double_p),
        ("dl_zarr", _c_double_p), ("integ_part", _c_double_p), ("flim0", _c_double_p),
        ("alpha0", ctypes.c_double), ("sch_al0", ctypes.c_double), ("fcmin", ctypes.c_double),
        ("lims", (ctypes.c_double * 2) * 5), ("pivots", ctypes.c_double * 3),
        ("device", ctypes.c_int32), ("max_batch", ctypes.c_int32),
    ]


EXPORTS = ("lf_abi_version", "lf_create", "lf_destroy", "lf_ndim", "lf_lnprob_batch",
           "lf_lnprob_batch_device", "lf_lnprob_batch_device_n", "lf_lnprob_pieces", "lf_set_profiling", "lf_kernel_times",
           "lf_set_option", "lf_last_error", "lf_sampler_create", "lf_sampler_destroy", "lf_sampler_start",
           "lf_sampler_run", "lf_sampler_read", "lf_sampler_steps", "lf_sampler_half_eval",
           "lf_sampler_half_accept", "lf_compress_keys", "lf_compress_grid", "lf_grid_bins", "lf_deal_table", "lf_form_counts", "lf_last_launch", "lf_veff")

_lib = None


def load():
    """dlopen liblfmcmc.so and declare the prototypes.  Raises RuntimeError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "liblfmcmc.so not found at %s: the HIP library has not been built "
            "(run `python -m lumfuncmcmc_amd.build`). There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64,
    # and a second runtime initialised in the same process sees no GPU.  Importing torch first
    # makes the dynamic loader resolve this library's libamdhip64.so.7 to torch's copy, so device
    # pointers and streams can be shared with torch (and with RCCL through torch.distributed).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise RuntimeError("liblfmcmc.so failed to load (%s). There is no CPU fallback." % e)
    lib.lf_abi_version.restype = ctypes.c_int
    lib.lf_abi_version.argtypes = []
    lib.lf_create.restype = ctypes.c_void_p
    lib.lf_create.argtypes = [ctypes.POINTER(LfDesc)]
    lib.lf_destroy.restype = None
    lib.lf_destroy.argtypes = [ctypes.c_void_p]
    lib.lf_ndim.restype = ctypes.c_int
    lib.lf_ndim.argtypes = [ctypes.c_void_p]
    lib.lf_lnprob_batch.restype = ctypes.c_int
    lib.lf_lnprob_batch.argtypes = [ctypes.c_void_p, _c_double_p, ctypes.c_int, _c_double_p]
    lib.lf_lnprob_batch_device.restype = ctypes.c_int
    lib.lf_lnprob_batch_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_void_p]
    lib.lf_lnprob_batch_device_n.restype = ctypes.c_int
    lib.lf_lnprob_batch_device_n.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_void_p]
    lib.lf_lnprob_pieces.restype = ctypes.c_int
    lib.lf_lnprob_pieces.argtypes = [ctypes.c_void_p, _c_double_p, ctypes.c_int, _c_double_p, _c_double_p]
    lib.lf_set_profiling.restype = ctypes.c_int
    lib.lf_set_profiling.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.lf_kernel_times.restype = ctypes.c_int
    lib.lf_kernel_times.argtypes = [ctypes.c_void_p, _c_double_p, _c_int64_p]
    lib.lf_set_option.restype = ctypes.c_int
    lib.lf_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64]
    lib.lf_form_counts.restype = ctypes.c_int
    lib.lf_form_counts.argtypes = [ctypes.c_void_p, _c_int64_p]
    lib.lf_last_launch.restype = ctypes.c_int
    lib.lf_last_launch.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]
    lib.lf_veff.restype = ctypes.c_int
    lib.lf_veff.argtypes = [ctypes.c_int, ctypes.c_int64, _c_double_p, _c_double_p, _c_double_p, ctypes.c_double, ctypes.c_double,
                            ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32,
                            _c_int64_p, ctypes.c_uint64, _c_double_p, _c_double_p]
    lib.lf_last_error.restype = ctypes.c_char_p
    lib.lf_last_error.argtypes = [ctypes.c_void_p]
    lib.lf_sampler_create.restype = ctypes.c_void_p
    lib.lf_sampler_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_uint64, ctypes.c_int64]
    lib.lf_sampler_destroy.restype = None
    lib.lf_sampler_destroy.argtypes = [ctypes.c_void_p]
    lib.lf_sampler_start.restype = ctypes.c_int
    lib.lf_sampler_start.argtypes = [ctypes.c_void_p, _c_double_p, _c_double_p]
    lib.lf_sampler_run.restype = ctypes.c_int
    lib.lf_sampler_run.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    lib.lf_sampler_read.restype = ctypes.c_int
    lib.lf_sampler_read.argtypes = [ctypes.c_void_p, _c_double_p, _c_double_p, _c_int64_p, _c_double_p, _c_double_p]
    lib.lf_sampler_steps.restype = ctypes.c_int64
    lib.lf_sampler_steps.argtypes = [ctypes.c_void_p]
    lib.lf_sampler_half_eval.restype = ctypes.c_int
    lib.lf_sampler_half_eval.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.lf_sampler_half_accept.restype = ctypes.c_int
    lib.lf_sampler_half_accept.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.lf_compress_keys.restype = ctypes.c_int64
    lib.lf_compress_keys.argtypes = [ctypes.c_int, _c_double_p, _c_double_p, _c_double_p, ctypes.c_int64,
                                     _c_double_p, _c_double_p, ctypes.c_int64, _c_double_p]
    v = lib.lf_abi_version()
    if v != LF_ABI_VERSION:
        raise RuntimeError("liblfmcmc.so ABI %d != binding ABI %d: rebuild the library" % (v, LF_ABI_VERSION))
    _lib = lib
    return lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def compress_keys(kind, params, key, weight=None):
    """Host-only helper behind the "compress" option (csrc/lf_compress.h): replace the coordinates `key`
    (weights `weight`, default 1) by weighted pseudo-sources.  kind 0 = FREE (params = |a/(1-a)|, alpha_lo,
    alpha_hi, flim_lo, flim_hi), kind 1 = ZEVOL (params = L_lo, L_hi, z1, z2, z3).
    Returns (node, weight, bound).  Touches no GPU."""
    lib = load()
    params, key = _f64(params), _f64(key)
    wt = _f64(weight) if weight is not None else None
    cap = key.size + 64
    node, w = np.empty(cap), np.empty(cap)
    bound = np.zeros(1)
    n = lib.lf_compress_keys(int(kind), _ptr(params), _ptr(key), _ptr(wt), key.size, _ptr(node), _ptr(w), cap, _ptr(bound))
    if n < 0:
        raise RuntimeError("lf_compress_keys failed (%d): the error bound cannot be met" % n)
    return node[:n].copy(), w[:n].copy(), float(bound[0])


def compress_grid(params, L, wL, ck, Dk):
    """Host-only helper behind the compressed FREE grid (csrc/lf_compress.h: compress_grid).  Returns a dict with
    u (nb, 16), row0, nrows, off, omega (flat, per bin [row][node]) and bound.  Touches no GPU."""
    lib = load()
    params, L, wL, ck, Dk = (_f64(a) for a in (params, L, wL, ck, Dk))
    S = L.size
    capb, capo = 4096, 4096 * 16 * 64
    u = np.empty(capb * 16)
    row0, nrows, off = (np.empty(capb, dtype=np.int32) for _ in range(3))
    omega = np.empty(capo)
    bound = np.zeros(1)
    ip = ctypes.POINTER(ctypes.c_int32)
    lib.lf_compress_grid.restype = ctypes.c_int64
    lib.lf_compress_grid.argtypes = [_c_double_p, ctypes.c_int] + [_c_double_p] * 5 + [ip, ip, ip, _c_double_p,
                                                                                       ctypes.c_int64, ctypes.c_int64, _c_double_p]
    nb = lib.lf_compress_grid(_ptr(params), S, _ptr(L), _ptr(wL), _ptr(ck), _ptr(Dk), _ptr(u), row0.ctypes.data_as(ip),
                              nrows.ctypes.data_as(ip), off.ctypes.data_as(ip), _ptr(omega), capb, capo, _ptr(bound))
    if nb < 0:
        raise RuntimeError("lf_compress_grid failed (%d)" % nb)
    nom = int(off[nb - 1] + nrows[nb - 1] * 16)
    return {"u": u[:nb * 16].reshape(nb, 16).copy(), "row0": row0[:nb].copy(), "nrows": nrows[:nb].copy(),
            "off": off[:nb].copy(), "omega": omega[:nom].copy(), "bound": float(bound[0])}


def deal_table(n_cell_chunks, n_bins, grid_part=0, grid_parts=0):
    """Host-only helper behind lf_free's deal of its cell chunks and flux bins to its 32 virtual workgroups (lfmcmc.hip:
    make_deal).  Returns (cells, bins): two lists of 32 lists - the chunk / bin numbers each rank
    takes, in the order it takes them.  Touches no GPU."""
    lib = load()
    lib.lf_deal_table.restype = ctypes.c_int
    lib.lf_deal_table.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_int32), ctypes.c_int64]
    n = lib.lf_deal_table(int(n_cell_chunks), int(n_bins), int(grid_part), int(grid_parts), None, 0)
    if n < 0:
        raise ValueError("lf_deal_table: bad arguments")
    t = np.empty(n, dtype=np.int32)
    assert lib.lf_deal_table(int(n_cell_chunks), int(n_bins), int(grid_part), int(grid_parts), t.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n) == n
    VF = (n - n_cell_chunks - n_bins) // 2 - 1
    lst = t[2 * (VF + 1):]
    cells = [lst[t[v]:t[v + 1]].tolist() for v in range(VF)]
    bins = [lst[n_cell_chunks + t[VF + 1 + v]:n_cell_chunks + t[VF + 2 + v]].tolist() for v in range(VF)]
    return cells, bins


def grid_bins(params, L, wL, ck, Dk):
    """Host-only helper behind piece B over flux bins (csrc/lf_gridbound.h: build_gridq; the default of FREE contexts with
    a separable grid).  params = |a/(1-a)|, alpha_lo, alpha_hi, flim_lo, flim_hi.  Returns a dict with edges (nb, 2), rec
    (nb, 64, 4) = {x_n, 10^(x_n + 17), L of row row0 + n, 10^(that - 42)}, rows (nb, 4) = {row0, nrows, offset, 0}, omega
    (flat, per bin [row][64]) and margin = ln(proven bound / allowance) <= 0.  Touches no GPU."""
    lib = load()
    params, L, wL, ck, Dk = (_f64(a) for a in (params, L, wL, ck, Dk))
    S = L.size
    capb, capo = 1024, 1024 * 64 * 64
    edges, rec = np.empty(capb * 2), np.empty(capb * 64 * 4)
    rows = np.empty(capb * 4, dtype=np.int32)
    omega = np.empty(capo)
    margin = np.zeros(1)
    ip = ctypes.POINTER(ctypes.c_int32)
    lib.lf_grid_bins.restype = ctypes.c_int64
    lib.lf_grid_bins.argtypes = [_c_double_p, ctypes.c_int] + [_c_double_p] * 6 + [ip, _c_double_p, ctypes.c_int64, ctypes.c_int64,
                                                                                   _c_double_p]
    nb = lib.lf_grid_bins(_ptr(params), S, _ptr(L), _ptr(wL), _ptr(ck), _ptr(Dk), _ptr(edges), _ptr(rec), rows.ctypes.data_as(ip),
                          _ptr(omega), capb, capo, _ptr(margin))
    if nb < 0:
        raise RuntimeError("lf_grid_bins failed (%d): no proven set of bins for this box" % nb)
    rows = rows[:nb * 4].reshape(nb, 4).copy()
    nom = int(rows[-1, 2] + rows[-1, 1] * 64)
    return {"edges": edges[:nb * 2].reshape(nb, 2).copy(), "rec": rec[:nb * 256].reshape(nb, 64, 4).copy(), "rows": rows,
            "omega": omega[:nom].copy(), "margin": float(margin[0])}


def veff_device(flux, flim, vol, pref0, alpha, fcmin, bin_of=None, nbin=0, nboot=0, boot_idx=None, seed=0, device=0):
    """lf_veff (include/lfmcmc.h): 1/Veff weights, and optionally their binned sums over the catalogue and over nboot
    bootstrap resamples, on the GPU.  vol: scalar or per-source array.  Returns (phi[n], sums[(nboot + 1), nbin] or None)."""
    lib = load()
    flux, flim = _f64(flux), _f64(flim)
    n = flux.size
    volarr = None if np.ndim(vol) == 0 else _f64(vol)
    phi = np.empty(n)
    sums = np.zeros((nboot + 1, nbin)) if nbin > 0 else None
    b32 = None if bin_of is None else np.ascontiguousarray(bin_of, dtype=np.int32)
    bidx = None if boot_idx is None else np.ascontiguousarray(boot_idx, dtype=np.int64)
    if bidx is not None and bidx.shape != (nboot, n):
        raise ValueError("boot_idx must be (nboot, n)")
    rc = lib.lf_veff(int(device), n, _ptr(flux), _ptr(flim), _ptr(volarr), float(vol) if volarr is None else 0.0, float(pref0),
                     float(alpha), float(fcmin) if fcmin else 0.0,
                     None if b32 is None else b32.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), int(nbin), int(nboot),
                     None if bidx is None else bidx.ctypes.data_as(_c_int64_p), ctypes.c_uint64(int(seed)), _ptr(phi),
                     None if sums is None else _ptr(sums))
    if rc != LF_OK:
        raise LFError("lf_veff failed (%d)" % rc)
    return phi, sums


def _ptr(a):
    return a.ctypes.data_as(_c_double_p) if a is not None else None


def log_flux(lum, dl_mpc):
    """logf_i = lum_i - log10(4 pi (3.086e24 DLf(z_i))^2): log10 of the flux of lumfuncmcmc.py:70."""
    return np.asarray(lum, dtype=np.float64) - np.log10(4.0 * np.pi * (MPC_CM * np.asarray(dl_mpc, dtype=np.float64)) ** 2)


class LFError(RuntimeError):
    pass


class LFContext(object):
    """One catalogue + grids resident on one MI355X.  `inp` uses the reference's own names
    (the attributes LumFuncMCMC / LumFuncMCMCz build in __init__):

        variant, fix_sch_al, sch_al0, field_ind, lum, DLz | logf, z, Om_arr, Omega_0, Flim0, alpha0,
        logL (S,S), zarr, volume_part, DL_zarr, integ_part (nf,S,S) | integ_sum (S,S), fcmin,
        lims {Lstar, phistar, sch_al, Flim, alpha}, pivots
    """

    def __init__(self, inp, device=0, max_batch=0):
        lib = load()
        self._lib = lib
        self._h = None
        variant = inp["variant"]
        if variant not in VARIANTS:
            raise ValueError("variant must be one of %s" % (sorted(VARIANTS),))
        self.variant = variant
        fi = np.ascontiguousarray(inp["field_ind"], dtype=np.int64)
        nf = len(fi) - 1
        lum = _f64(inp["lum"])
        N = lum.shape[0]
        logL = _f64(inp["logL"])
        S = logL.shape[0]
        if logL.shape != (S, S):
            raise ValueError("logL must be (S, S)")
        keep = {"fi": fi, "lum": lum, "logL": logL, "zarr": _f64(inp["zarr"]),
                "omega0": _f64(inp["Omega_0"])}
        d = LfDesc()
        d.variant = VARIANTS[variant]
        d.fix_sch_al = 1 if inp.get("fix_sch_al", False) else 0
        d.nf, d.S, d.N = nf, S, N
        if variant == "free":
            if inp.get("logf") is not None:
                keep["logf"] = _f64(inp["logf"])
            else:
                keep["logf"] = _f64(log_flux(lum, inp["DLz"]))
            keep["volume_part"] = _f64(inp["volume_part"])
            keep["dl_zarr"] = _f64(inp["DL_zarr"])
        else:
            keep["om_arr"] = _f64(inp["Om_arr"])
            ip = inp.get("integ_part")
            if ip is None:        # a field-summed table: field 0 carries the sum, the rest are zero
                ip = np.zeros((nf, S, S))
                ip[0] = inp["integ_sum"]
            keep["integ_part"] = _f64(ip)
            if keep["integ_part"].shape != (nf, S, S):
                raise ValueError("integ_part must be (nf, S, S)")
            if variant == "zevol":
                keep["z"] = _f64(inp["z"])
            else:
                keep["flim0"] = _f64(inp["Flim0"])
                d.alpha0 = float(inp["alpha0"])
        for k in ("logf", "z", "om_arr", "volume_part", "dl_zarr", "integ_part", "flim0"):
            setattr(d, k, _ptr(keep.get(k)))
        d.field_ind = fi.ctypes.data_as(_c_int64_p)
        d.lum = _ptr(lum)
        d.omega0 = _ptr(keep["omega0"])
        d.logL = _ptr(logL)
        d.zarr = _ptr(keep["zarr"])
        d.sch_al0 = float(inp.get("sch_al0", 0.0))
        d.fcmin = float(inp.get("fcmin", 0.1))
        lims = inp["lims"]
        for i, name in enumerate(LIM_ORDER):
            d.lims[i][0], d.lims[i][1] = float(lims[name][0]), float(lims[name][1])
        piv = inp.get("pivots", (1.20, 1.53, 1.86))
        for i in range(3):
            d.pivots[i] = float(piv[i])
        d.device = int(device)
        d.max_batch = int(max_batch)
        h = lib.lf_create(ctypes.byref(d))
        if not h:
            raise LFError(lib.lf_last_error(None).decode())
        self._h = ctypes.c_void_p(h)
        self.ndim = lib.lf_ndim(self._h)
        self.N, self.nf, self.S = N, nf, S
        self.device = int(device)

    # ------------------------------------------------------------------ calls
    def _check(self, rc):
        if rc != LF_OK:
            raise LFError("liblfmcmc error %d: %s" % (rc, self._lib.lf_last_error(self._h).decode()))

    def _theta(self, theta):
        th = np.ascontiguousarray(np.atleast_2d(np.asarray(theta, dtype=np.float64)))
        if th.ndim != 2 or th.shape[1] != self.ndim:
            raise ValueError("theta must be (B, %d), got %s" % (self.ndim, th.shape))
        return th

    def lnprob_batch(self, theta):
        """theta (B, ndim) or (ndim,) host array -> lnprob (B,) float64."""
        th = self._theta(theta)
        out = np.empty(th.shape[0], dtype=np.float64)
        if th.shape[0] == 0:
            return out
        self._check(self._lib.lf_lnprob_batch(self._h, _ptr(th), th.shape[0], _ptr(out)))
        return out

    def lnprob_pieces(self, theta):
        th = self._theta(theta)
        a = np.empty(th.shape[0], dtype=np.float64)
        b = np.empty(th.shape[0], dtype=np.float64)
        if th.shape[0] == 0:
            return a, b
        self._check(self._lib.lf_lnprob_pieces(self._h, _ptr(th), th.shape[0], _ptr(a), _ptr(b)))
        return a, b

    def lnprob_batch_device(self, theta_ptr, B, out_ptr, stream=0):
        """Raw device pointers (ints) and a hipStream_t handle (int, 0 = default stream)."""
        self._check(self._lib.lf_lnprob_batch_device(self._h, ctypes.c_void_p(theta_ptr), int(B),
                                                     ctypes.c_void_p(out_ptr), ctypes.c_void_p(stream)))

    def lnprob_torch(self, theta, out=None):
        """theta: CUDA(=HIP) float64 tensor (B, ndim) on this context's device; enqueues on
        torch's current stream and returns a (B,) tensor.  torch is plumbing here: device memory
        and streams only."""
        import torch
        if theta.dtype != torch.float64 or not theta.is_cuda or theta.dim() != 2 or theta.shape[1] != self.ndim:
            raise ValueError("theta must be a float64 device tensor of shape (B, %d)" % self.ndim)
        if theta.device.index != self.device:
            raise ValueError("theta is on device %s, context is on %d" % (theta.device, self.device))
        theta = theta.contiguous()
        B = theta.shape[0]
        if out is None:
            out = torch.empty(B, dtype=torch.float64, device=theta.device)
        if B:
            stream = torch.cuda.current_stream(theta.device).cuda_stream
            self.lnprob_batch_device(theta.data_ptr(), B, out.data_ptr(), stream)
        return out

    def lnprob_torch_n(self, theta, out=None):
        """K independent blocks in one C call (lf_lnprob_batch_device_n): theta (K, B, ndim) device tensor -> (K, B)."""
        import torch
        if theta.dtype != torch.float64 or not theta.is_cuda or theta.dim() != 3 or theta.shape[2] != self.ndim:
            raise ValueError("theta must be a float64 device tensor of shape (K, B, %d)" % self.ndim)
        theta = theta.contiguous()
        K, B = theta.shape[0], theta.shape[1]
        if out is None:
            out = torch.empty((K, B), dtype=torch.float64, device=theta.device)
        if K and B:
            stream = torch.cuda.current_stream(theta.device).cuda_stream
            self._check(self._lib.lf_lnprob_batch_device_n(self._h, ctypes.c_void_p(theta.data_ptr()), int(B), int(K),
                                                           ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream)))
        return out

    def set_profiling(self, level):
        """0 off, 1 the main kernel only, 2 every launch (True = 2)."""
        self._check(self._lib.lf_set_profiling(self._h, 2 if level is True else int(level)))

    def kernel_times(self):
        ms = (ctypes.c_double * 4)()
        n = (ctypes.c_int64 * 4)()
        self._check(self._lib.lf_kernel_times(self._h, ms, n))
        names = ("prepare", "main", "unused", "finalize")
        return {k: {"ms": ms[i], "launches": int(n[i])} for i, k in enumerate(names)}

    FORM_NAMES = ("general", "general_noexp", "table", "table_noexp", "careful", "skipped", "node_general", "node_bright", "cell")

    def form_counts(self):
        """Census of the term forms since set_option("count_forms", 1): dict name -> count (include/lfmcmc.h)."""
        n = (ctypes.c_int64 * 9)()
        self._check(self._lib.lf_form_counts(self._h, n))
        return {k: int(n[i]) for i, k in enumerate(self.FORM_NAMES)}

    def last_launch(self):
        """Shape of the most recent lf_main launch: dict st, tw, twb, compressed, workgroups, chunks_a, chunks_b, rows."""
        n = (ctypes.c_int32 * 8)()
        self._check(self._lib.lf_last_launch(self._h, n))
        d = dict(zip(("st", "tw", "twb", "kind", "workgroups", "chunks_a", "chunks_b", "rows"), (int(v) for v in n)))
        d["compressed"] = int(d["kind"] == 1)
        d["fused"] = d["kind"] in (3, 5)            # lf_free / lf_pers doing lf_prepare's and lf_finalize's work too: one launch
        if d["fused"]:
            d["kind"] -= 1
        d["kernel"] = "lf_free<%d>" % d["st"] if d["kind"] == 2 else ("lf_pers" if d["kind"] == 4 else "lf_main")
        return d

    def set_option(self, key, value):
        self._check(self._lib.lf_set_option(self._h, key.encode(), int(value)))

    def close(self):
        if self._h is not None:
            self._lib.lf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
