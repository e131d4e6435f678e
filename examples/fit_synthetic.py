#!/usr/bin/env python
"""End to end on the MI355X path, the way run_lumfuncmcmc.py drives the reference
(run_lumfuncmcmc.py:230-323), without astropy / emcee / corner:

    catalogue file -> per-field lists -> LumFuncMCMC -> fit_model (device-resident sampler)
    -> set_median_fit (median LF + 1/Veff estimate) -> the reference's output tables.

    python examples/fit_synthetic.py [--nsrc 20000] [--nwalkers 64] [--nsteps 300] [--fix-comp]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lumfuncmcmc_amd import synth, tableio                      # noqa: E402
from lumfuncmcmc_amd.cosmology import cosmo                     # noqa: E402
from lumfuncmcmc_amd.model import LumFuncMCMC                   # noqa: E402


def write_catalogue(path, n, seed):
    """A synthetic catalogue in the driver's input format (Field, ID, z, OIII_flux, OIII_flux_e)."""
    cat = synth.catalogue(n, seed=seed)
    dl = cosmo.luminosity_distance(cat["z"])
    flux17 = 10 ** cat["lum"] / (4.0 * np.pi * (dl * 3.086e24) ** 2) / 1.0e-17
    names = np.array(["AEGIS", "COSMOS", "GOODSN", "GOODSS", "UDS"])
    field = np.concatenate([np.full(int(cat["field_ind"][f + 1] - cat["field_ind"][f]), names[f]) for f in range(5)])
    with open(path, "w") as f:
        f.write("Field ID z OIII_flux OIII_flux_e\n")
        for i in range(n):
            f.write("%s %d %r %r %r\n" % (field[i], i, float(cat["z"][i]), float(flux17[i]), float(0.1 * flux17[i])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsrc", type=int, default=20000)
    ap.add_argument("--nwalkers", type=int, default=64)
    ap.add_argument("--nsteps", type=int, default=300)
    ap.add_argument("--fix-comp", action="store_true")
    ap.add_argument("--out", default="LFMCMCOut")
    ap.add_argument("--compress", action="store_true", help="compressed catalogue and grid (DESIGN.md section 3.5)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    cpath = os.path.join(args.out, "synthetic_catalogue.dat")
    write_catalogue(cpath, args.nsrc, seed=5)

    z, flux, flux_e, field_names, field_ind, _ = tableio.read_input_catalogue(cpath, "OIII", list(synth.FLIM), synth.ALPHA_C)
    t0 = time.time()
    LFmod = LumFuncMCMC(z, flux=flux, flux_e=flux_e, Flim=list(synth.FLIM), alpha=synth.ALPHA_C, line_name="OIII",
                        Omega_0=list(synth.OMEGA_0), nbins=50, nboot=100, sch_al=synth.SCH_AL,
                        sch_al_lims=synth.SCH_AL_LIMS, Lstar=synth.LSTAR, Lstar_lims=synth.LSTAR_LIMS,
                        phistar=synth.PHISTAR, phistar_lims=synth.PHISTAR_LIMS, Lc=synth.LC, Lh=synth.LH,
                        nwalkers=args.nwalkers, nsteps=args.nsteps, fix_sch_al=False, fix_comp=args.fix_comp,
                        min_comp_frac=0.0, Flim_lims=synth.FLIM_LIMS, alpha_lims=synth.ALPHA_LIMS,
                        field_names=field_names, field_ind=field_ind, compress=args.compress)
    print("setup %.2f s for %d sources" % (time.time() - t0, len(LFmod.lum)))
    np.random.seed(3)
    LFmod.fit_model()
    LFmod.set_median_fit()

    names = LFmod.get_param_names() + ["Ln Prob"]
    tag = "synthetic_nw%d_ns%d" % (args.nwalkers, args.nsteps)
    tableio.write_fixed_width_two_line(os.path.join(args.out, "fitposterior_%s.dat" % tag),
                                       list(LFmod.samples.T), names)
    tableio.write_fixed_width_two_line(os.path.join(args.out, "bestfitLF_%s.dat" % tag),
                                       [LFmod.lum, LFmod.lum_e, LFmod.medianLF], ["Luminosity", "Luminosity_Err", "MedianLF"])
    tableio.write_fixed_width_two_line(os.path.join(args.out, "VeffLF_%s.dat" % tag),
                                       [LFmod.Lavg, LFmod.lfbinorig, np.sqrt(LFmod.var)], ["Luminosity", "BinLF", "BinLFErr"])
    percentiles = [5, 16, 50, 84, 95]
    labels = ["Line"] + [n + "_%02d" % p for n in names[:-1] for p in percentiles]
    LFmod.table = [["OIII"] + [0.0] * (len(labels) - 1)]
    LFmod.add_fitinfo_to_table(percentiles)
    tableio.write_fixed_width_two_line(os.path.join(args.out, "%s.dat" % tag), [np.array([v]) for v in LFmod.table[-1]],
                                       labels, formats={l: ("%s" if l == "Line" else "%0.3f") for l in labels})
    med = np.median(LFmod.samples[:, :-1], axis=0)
    print("posterior medians:", dict(zip(names[:-1], np.round(med, 3))))
    print("wrote", sorted(os.listdir(args.out)))
    LFmod.close()


if __name__ == "__main__":
    main()
