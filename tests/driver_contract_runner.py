"""Runs the REFERENCE's own driver, in place, over this package's classes (tests/test_driver_contract.py starts it
under an interpreter that has astropy).  argv: <workdir> <reference dir> <nsources> <nwalkers> <nsteps> [--z] [--default-config]
[-fc] [-fsa].  --z: run_lumfuncmcmc_z.py over LumFuncMCMCz instead of run_lumfuncmcmc.py over LumFuncMCMC; --default-config:
configLF.output_dict as the reference ships it ('triangle plot': True), i.e. the driver with NO config edits.

sys.path order: dropin/ (lumfuncmcmc, lumfuncmcmc_z, VmaxLumFunc resolve to this package), the repo, oracle/ and
only then the reference checkout (run_lumfuncmcmc.py, configLF.py).  The GPU is not needed: `_Base._evaluate` is
pointed at the oracle (tests may use it) and the host sampler runs the chain; everything else - the constructor call
with the driver's keywords, fit_model, set_median_fit, the attributes the driver writes out, add_fitinfo_to_table - is
the product code.  Nothing of the reference is copied: it is imported where it lies."""
import json
import os
import sys

import numpy as np

work, ref, nsrc, nwalk, nsteps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
flags = sys.argv[6:]
ZDRIVER = "--z" in flags
DEFAULT_CONFIG = "--default-config" in flags
flags = [f for f in flags if f not in ("--z", "--default-config")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "dropin"), ROOT, os.path.join(ROOT, "oracle"), ref]

if not hasattr(np, "asscalar"):                       # astropy 4.3 predates its numpy
    for name, f in (("asscalar", lambda a: a.item()), ("alen", len), ("rank", np.ndim)):
        setattr(np, name, f)

import lf_oracle as O                                  # noqa: E402
from lumfuncmcmc_amd import model, synth               # noqa: E402

# ---- a synthetic catalogue in the driver's input format (flux columns, 5 field names sorting into
# Flim order, ID), and the hard-coded second table read_input_file insists on (run_lumfuncmcmc.py:241)
os.chdir(work)
cat = synth.catalogue(nsrc, seed=7)
fi = cat["field_ind"]
DL = np.asarray(model._cosmo.luminosity_distance(cat["z"]))
flux = 10.0 ** cat["lum"] / (4.0 * np.pi * (3.086e24 * DL) ** 2) / 1.0e-17
fields = np.concatenate([["F%d" % f] * int(fi[f + 1] - fi[f]) for f in range(5)])
ids = np.arange(nsrc)
with open("cat.dat", "w") as fh:
    fh.write("Field ID z OIII_flux OIII_flux_e\n")
    for i in range(nsrc):
        fh.write("%s %d %.10f %.10e %.10e\n" % (fields[i], ids[i], cat["z"][i], flux[i], 0.05 * flux[i]))
with open("combined_all_Swift_NoDust_Donley_removed.dat", "w") as fh:
    fh.write("Field ID E(B-V) E(B-V)err SFR100 SFR100err\n")
    for i in range(nsrc):
        fh.write("%s %d 0.1 0.01 1.0 0.1\n" % (fields[i], ids[i]))

# ---- the reference's driver, imported where it lies
import configLF                                        # noqa: E402
if ZDRIVER:
    import run_lumfuncmcmc_z as drv                    # noqa: E402
    assert drv.LumFuncMCMCz is model.LumFuncMCMCz, "the driver did not pick up the drop-in class"
    cls, outdir = model.LumFuncMCMCz, "LFMCMCzOut"
    # run_lumfuncmcmc_z.py:274-279 writes the (zlen, Llen) medianLF matrix as ONE table column next to two 1-d columns:
    # astropy (4.3.1 here) refuses that in the REFERENCE's own code, whatever class produced the arrays - the one output the
    # z driver cannot write in this container
    configLF.output_dict["bestfitLF"] = False
else:
    import run_lumfuncmcmc as drv                      # noqa: E402
    assert drv.LumFuncMCMC is model.LumFuncMCMC, "the driver did not pick up the drop-in class"
    cls, outdir = model.LumFuncMCMC, "LFMCMCOut"
assert os.path.dirname(os.path.abspath(drv.__file__)) == os.path.abspath(ref)
if not DEFAULT_CONFIG:
    configLF.output_dict["triangle plot"] = False      # (the other branch of the driver: set_median_fit)

cls.device_sampler = False                             # host sampler: no GPU in this test


def evaluate(self, theta):                             # the oracle in place of liblfmcmc.so (tests only)
    inp = self.kernel_inputs()
    inp["lims"] = {k: list(v) for k, v in inp["lims"].items()}
    th = np.asarray(theta, dtype=np.float64)
    out = O.lnprob_batch(inp, np.atleast_2d(th))
    return float(out[0]) if th.ndim == 1 else out


model._Base._evaluate = evaluate
np.random.seed(12345)
drv.main(["-f", "cat.dat", "-o", "contract.dat", "-nw", str(nwalk), "-ns", str(nsteps), "-nbins", "10", "-nboot", "20"] + flags)

out = sorted(os.listdir(outdir))
from astropy.table import Table                        # noqa: E402
post = [f for f in out if f.startswith("fitposterior")][0]
t = Table.read(os.path.join(outdir, post), format="ascii.fixed_width_two_line")
print(json.dumps({"files": out, "posterior_columns": len(t.colnames), "posterior_rows": len(t),
                  "finite_lnprob": int(np.isfinite(np.asarray(t[t.colnames[-1]], dtype=float)).sum())}))
