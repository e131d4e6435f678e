"""ctypes loader of oracle/liblforacle.so (the C restatement).  TEST INFRASTRUCTURE."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int64)
VARIANTS = {"free": 0, "fixcomp": 1, "zevol": 2}
LIM_ORDER = ("Lstar", "phistar", "sch_al", "Flim", "alpha")


class LfoInputs(ctypes.Structure):
    _fields_ = [("variant", ctypes.c_int32), ("fix_sch_al", ctypes.c_int32), ("nf", ctypes.c_int32),
                ("S", ctypes.c_int32), ("N", ctypes.c_int64), ("field_ind", _ip),
                ("lum", _dp), ("z", _dp), ("dl_src", _dp), ("om_arr", _dp), ("omega0", _dp), ("logL", _dp),
                ("zarr", _dp), ("volume_part", _dp), ("dl_zarr", _dp), ("integ_part", _dp), ("flim0", _dp),
                ("alpha0", ctypes.c_double), ("sch_al0", ctypes.c_double), ("fcmin", ctypes.c_double),
                ("lims", (ctypes.c_double * 2) * 5), ("pivots", ctypes.c_double * 3)]


def load():
    so = os.path.join(HERE, "liblforacle.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "lf_oracle.c")):
        subprocess.run(["make", "-s", "-C", HERE], check=True)
    lib = ctypes.CDLL(so)
    lib.lfo_lnprob_batch.restype = ctypes.c_int
    lib.lfo_lnprob_batch.argtypes = [ctypes.POINTER(LfoInputs), _dp, ctypes.c_int, _dp, _dp, _dp, ctypes.c_int]
    return lib


class COracle(object):
    """Same input dict as lf_oracle.lnprob (the reference's attribute names)."""

    def __init__(self, inp):
        self.lib = load()
        k = {}
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        k["fi"] = np.ascontiguousarray(inp["field_ind"], dtype=np.int64)
        nf = len(k["fi"]) - 1
        S = int(np.asarray(inp["logL"]).shape[0])
        for name, key in (("lum", "lum"), ("z", "z"), ("dl_src", "DLz"), ("om_arr", "Om_arr"), ("omega0", "Omega_0"),
                          ("logL", "logL"), ("zarr", "zarr"), ("volume_part", "volume_part"),
                          ("dl_zarr", "DL_zarr"), ("flim0", "Flim0")):
            k[name] = f64(inp[key])
        ip = inp.get("integ_part")
        if ip is None and inp.get("integ_sum") is not None:
            ip = np.zeros((nf, S, S))
            ip[0] = inp["integ_sum"]
        k["integ_part"] = f64(ip) if ip is not None else None
        d = LfoInputs()
        d.variant, d.fix_sch_al, d.nf, d.S, d.N = VARIANTS[inp["variant"]], int(bool(inp["fix_sch_al"])), nf, S, len(k["lum"])
        d.field_ind = k["fi"].ctypes.data_as(_ip)
        for name in ("lum", "z", "dl_src", "om_arr", "omega0", "logL", "zarr", "volume_part", "dl_zarr", "flim0", "integ_part"):
            setattr(d, name, k[name].ctypes.data_as(_dp) if k[name] is not None else None)
        d.alpha0, d.sch_al0, d.fcmin = float(inp["alpha0"]), float(inp["sch_al0"]), float(inp["fcmin"])
        for i, n in enumerate(LIM_ORDER):
            d.lims[i][0], d.lims[i][1] = float(inp["lims"][n][0]), float(inp["lims"][n][1])
        for i in range(3):
            d.pivots[i] = float(inp["pivots"][i])
        self._keep, self.desc = k, d

    def lnprob_batch(self, theta, nthreads=1, pieces=False):
        th = np.ascontiguousarray(np.atleast_2d(theta), dtype=np.float64)
        B = th.shape[0]
        out, a, b = np.empty(B), np.empty(B), np.empty(B)
        self.lib.lfo_lnprob_batch(ctypes.byref(self.desc), th.ctypes.data_as(_dp), B, out.ctypes.data_as(_dp),
                                  a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), int(nthreads))
        return (out, a, b) if pieces else out
