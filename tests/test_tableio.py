"""Catalogue reader / result writer (lumfuncmcmc_amd/tableio.py) against what astropy and the
reference driver's read_input_file produce (fixtures written by oracle/gen_golden.py --only tableio)."""
import os

import numpy as np

from lumfuncmcmc_amd import synth, tableio

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_read_input_catalogue_matches_the_driver():
    cat = os.path.join(GOLDEN, "catalogue_n60.dat")
    for tag, mcf in (("mcf0", 0.0), ("mcf50", 0.5)):
        g = np.load(os.path.join(GOLDEN, "readinput_%s.npz" % tag))
        z, flux, flux_e, field_names, field_ind, ids = tableio.read_input_catalogue(
            cat, "OIII", list(synth.FLIM), synth.ALPHA_C, synth.FCMIN, mcf)
        assert list(field_names) == list(g["field_names"])          # np.unique order
        assert np.array_equal(field_ind, g["field_ind"])
        assert np.array_equal(np.concatenate(z), g["z"])
        assert np.array_equal(np.concatenate(flux), g["flux"])
        assert np.array_equal(np.concatenate(flux_e), g["flux_e"])
        assert sum(len(i) for i in ids) == field_ind[-1]
    assert field_ind[-1] < 60                                        # the 0.5 cut removed sources


def test_fixed_width_two_line_is_byte_identical(tmp_path):
    cols = np.load(os.path.join(GOLDEN, "fwtl_plain_cols.npy"))
    out = tmp_path / "a.dat"
    tableio.write_fixed_width_two_line(str(out), list(cols), ["Luminosity", "Luminosity_Err", "MedianLF"])
    assert out.read_text() == open(os.path.join(GOLDEN, "fwtl_plain.dat")).read()
    names = ["Line", r"$\log L_*$_05", r"$\log L_*$_50"]
    out = tmp_path / "b.dat"
    tableio.write_fixed_width_two_line(str(out), [np.array(["OIII"]), np.array([42.123456]), np.array([42.5])], names,
                                       formats={"Line": "%s", names[1]: "%0.3f", names[2]: "%0.3f"})
    assert out.read_text() == open(os.path.join(GOLDEN, "fwtl_formats.dat")).read()
    back = tableio.read_ascii_table(os.path.join(GOLDEN, "fwtl_plain.dat"))
    assert np.array_equal(back["Luminosity"], cols[0]) and np.array_equal(back["MedianLF"], cols[2])


def test_catalogue_to_constructor(tmp_path):
    """File -> per-field lists -> LumFuncMCMC: the flux-input constructor path of the drivers."""
    from lumfuncmcmc_amd.model import LumFuncMCMC
    z, flux, flux_e, field_names, field_ind, _ = tableio.read_input_catalogue(
        os.path.join(GOLDEN, "catalogue_n60.dat"), "OIII", list(synth.FLIM), synth.ALPHA_C)
    o = LumFuncMCMC(z, flux=flux, flux_e=flux_e, Flim=list(synth.FLIM), alpha=synth.ALPHA_C,
                    Omega_0=list(synth.OMEGA_0), min_comp_frac=0.0, Flim_lims=synth.FLIM_LIMS,
                    alpha_lims=synth.ALPHA_LIMS, field_names=field_names, field_ind=field_ind, nwalkers=32, nsteps=10)
    assert len(o.lum) == 60 and np.all(np.isfinite(o.lum)) and np.all(o.lum_e > 0)
    assert o.kernel_inputs()["variant"] == "free"
