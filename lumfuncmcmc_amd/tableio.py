"""Catalogue and result tables without astropy.

The reference drivers read the catalogue with `astropy.table.Table.read(fn, format='ascii')` and
split it by field (run_lumfuncmcmc.py:165-228), and write their outputs with
`format='ascii.fixed_width_two_line'` (:298-323).  These are small host-side text formats; this
module reads and writes the same files with NumPy only, so that the class can be fed and its
results saved on a machine without astropy (the GPU box).
"""
import numpy as np

from . import hostsetup as hs


def _convert(col):
    for typ in (np.int64, np.float64):
        try:
            return np.array(col, dtype=typ)
        except ValueError:
            pass
    return np.array(col)


def read_ascii_table(path):
    """Whitespace-delimited table with one header line (what astropy's basic 'ascii' reader accepts),
    or a fixed_width_two_line file (header, dashes, rows).  Returns {column name: array}."""
    with open(path) as f:
        lines = [ln.rstrip("\n") for ln in f if ln.strip() and not ln.lstrip().startswith("#")]
    if not lines:
        raise ValueError("%s: empty table" % path)
    names = lines[0].split()
    body = lines[1:]
    if body and set(body[0].strip()) <= set("- "):
        body = body[1:]
    rows = [ln.split() for ln in body]
    for r in rows:
        if len(r) != len(names):
            raise ValueError("%s: row with %d fields, header has %d" % (path, len(r), len(names)))
    return {n: _convert([r[i] for r in rows]) for i, n in enumerate(names)}


def write_fixed_width_two_line(path, columns, names, formats=None):
    """astropy's 'ascii.fixed_width_two_line': names right-aligned over columns as wide as their
    widest entry, a line of dashes, one row per line; floats print as repr unless a %-format is given."""
    formats = formats or {}
    cells = []
    for name, col in zip(names, columns):
        fmt = formats.get(name)
        out = []
        for v in np.asarray(col).tolist():
            if isinstance(v, bytes):
                v = v.decode()
            if fmt is not None:
                out.append(fmt % v)
            elif isinstance(v, float):
                out.append(repr(v))
            else:
                out.append(str(v))
        cells.append(out)
    widths = [max([len(n)] + [len(c) for c in col]) for n, col in zip(names, cells)]
    with open(path, "w") as f:
        f.write(" ".join(n.rjust(w) for n, w in zip(names, widths)) + "\n")
        f.write(" ".join("-" * w for w in widths) + "\n")
        for i in range(len(cells[0]) if cells else 0):
            f.write(" ".join(col[i].rjust(w) for col, w in zip(cells, widths)) + "\n")


def read_input_catalogue(filename, line_name, Flim, alpha, fcmin=0.1, min_comp_frac=0.0):
    """The catalogue split the way read_input_file does it (run_lumfuncmcmc.py:165-210): fields in
    sorted order (np.unique), per-field flux cut at the minimum-completeness flux (no cut when
    min_comp_frac = 0), `<line>_flux` / `<line>_flux_e` columns in 1e-17 erg/cm^2/s.
    Returns z, flux, flux_e (lists of per-field arrays), field_names, field_ind, ids."""
    t = read_ascii_table(filename)
    for need in ("Field", "z", "ID", "%s_flux" % line_name, "%s_flux_e" % line_name):
        if need not in t:
            raise KeyError("%s: column %r not found (have %s)" % (filename, need, sorted(t)))
    fields, zfull, idfull = t["Field"], t["z"], t["ID"]
    fluxfull, fluxfull_e = t["%s_flux" % line_name], t["%s_flux_e" % line_name]
    field_names = np.unique(fields)
    if abs(min_comp_frac - 0.0) < 1.0e-6:
        roots = np.zeros(len(field_names))
    else:
        from scipy.optimize import fsolve
        roots = np.array([fsolve(lambda x: hs.fleming(x, Flim[i], alpha, fcmin) - min_comp_frac, [Flim[i]])[0]
                          for i in range(len(field_names))])
    z, flux, flux_e, ids = [], [], [], []
    field_ind = np.array([0])
    for i, field in enumerate(field_names):
        cond = np.logical_and(fields == field, fluxfull > roots[i])
        z.append(zfull[cond]); flux.append(fluxfull[cond]); flux_e.append(fluxfull_e[cond]); ids.append(idfull[cond])
        field_ind = np.append(field_ind, field_ind[i] + int(cond.sum()))
    return z, flux, flux_e, field_names, field_ind, ids
