"""Readers for the reference's on-disk inputs (bayes_io.py): the MCMC config script,
the stacked measurement CSV and the initial-condition CSV.  Same names, arguments
and return values as the reference's functions, so a `main.py`-style launcher
works unchanged; the parsing itself is table-driven rather than the reference's
if-chain.  Host-side only.
"""
import csv
import os
import sys

import numpy as np


# ---------------------------------------------------------------------------
# small field parsers (bayes_io.py:17-72)
def get_split_and_clean_line(line):
    return [part.strip() for part in line.split(":")]


def extract_values(string, delimiter, dtype=float):
    return np.array(string.split(delimiter), dtype=dtype)


def _convert(tok, dtype):
    if tok == "inf":
        return np.inf
    if tok == "-inf":
        return -np.inf
    if dtype in (float, int):
        try:
            return dtype(tok)
        except ValueError:
            return tok
    return tok


def extract_tuples(string, delimiter, dtype=float):
    """"(a, b)<delim>(c, d, e)" -> [(a, b), (c, d, e)]; entries that do not parse as dtype
    stay strings; inf / -inf are understood (bayes_io.py:36-72)."""
    return [tuple(_convert(tok, dtype) for tok in chunk.strip("()").split(", "))
            for chunk in string.split(delimiter)]


# ---------------------------------------------------------------------------
# measurement / initial-condition files (bayes_io.py:75-149)
def get_data(exp_file, ic_flags, MCMC_fields):
    """Stacked `[time, value, uncertainty]` CSV -> (times, values, uncertainties), one array
    per measurement.  A new measurement starts at every row with time == 0; rows are cut to
    ic_flags["time_cutoff"]; with MCMC_fields["log_y"] values become log10|y| (floored at the
    smallest positive double) and uncertainties u / (|y| ln 10); select_obs_sets keeps a subset."""
    cutoff = ic_flags["time_cutoff"]
    select = ic_flags["select_obs_sets"]
    noise = ic_flags.get("noise_level", 0)
    data = np.loadtxt(exp_file, delimiter=",")
    times, y, unc = data[:, 0], np.array(data[:, 1]), np.array(data[:, 2])
    if noise is not None:
        # numpy's global generator, as the reference (bayes_io.py:93); with the default level 0
        # this only consumes draws
        y = y + noise * np.random.normal(0, 1, len(y))
    starts = list(np.where(times == 0)[0]) + [None]
    t_list = [times[a:b] for a, b in zip(starts[:-1], starts[1:])]
    y_list = [y[a:b] for a, b in zip(starts[:-1], starts[1:])]
    u_list = [unc[a:b] for a, b in zip(starts[:-1], starts[1:])]
    if cutoff is not None:
        for i, t in enumerate(t_list):
            lo = np.searchsorted(t, cutoff[0], side="left")
            hi = np.searchsorted(t, cutoff[1], side="right")
            t_list[i], y_list[i], u_list[i] = t[lo:hi], y_list[i][lo:hi], u_list[i][lo:hi]
    if MCMC_fields["log_y"]:
        tiny = sys.float_info.min
        for i in range(len(t_list)):
            a = np.abs(y_list[i])
            a[a < tiny] = tiny
            u_list[i] = u_list[i] / a / np.log(10)
            y_list[i] = np.log10(a)
    if select is not None:
        keep = [i for i in range(len(t_list)) if i in select]
        return ([t_list[i] for i in keep], [y_list[i] for i in keep], [u_list[i] for i in keep])
    return (t_list, y_list, u_list)


def get_initpoints(init_file, ic_flags):
    """One CSV row per measurement (a density profile or [fluence, alpha, direction])."""
    select = ic_flags["select_obs_sets"]
    with open(init_file, newline="") as f:
        rows = [row for row in csv.reader(f) if len(row) > 0]
    if select is not None:
        rows = np.array(rows)[select]
    return np.array(rows, dtype=float)


def make_dir(dirname):
    os.makedirs(dirname, exist_ok=True)


# ---------------------------------------------------------------------------
# fittable-parameter bookkeeping (bayes_io.py:152-257)
_FITTABLES = (("scale_factor", "_s"), ("fittable_fluences", "_f"), ("fittable_absps", "_a"))
_MODES = {"scale_f": 0, "fluences": 1, "absorptions": 2}


def put_into_param_info(param_info, vals, new_key):
    if "names" not in param_info:
        raise KeyError("Entry \"Param names\" not found in MCMC config file.\n"
                       "Check whether this entry is present and FIRST in\nthe Param Info subsection.")
    param_info[new_key] = {name: vals[i] for i, name in enumerate(param_info["names"])}


def insert_param(param_info, MCMC_fields, mode="fluences"):
    """Append the `_f{i}` / `_a{i}` / `_s{i}` parameters a fittable spec
    [move size, measurement indices, constraint groups, guesses] asks for: one per listed
    measurement, except the non-leading members of a constraint group."""
    if mode not in _MODES:
        raise NotImplementedError("Unsupported mode for insert_param()")
    field, prefix = _FITTABLES[_MODES[mode]]
    spec = MCMC_fields.get(field, None)
    if spec is None:
        return
    followers = [m for grp in (spec[2] or []) for m in list(grp)[1:]]
    for i in spec[1]:
        if i in followers:
            continue
        name = f"{prefix}{i}"
        param_info["names"].append(name)
        param_info["do_log"][name] = 1
        param_info["prior_dist"][name] = (0, np.inf)
        param_info["init_guess"][name] = spec[3][i]
        param_info["trial_move"][name] = spec[0]
        param_info["active"][name] = 1


def remap_fittable_inds(fittables, select_obs_sets):
    """Positions (within the selected subset) of the selected measurements that are fittable."""
    return np.array([i for i, s in enumerate(select_obs_sets) if s in fittables])


def remap_constraint_grps(c_grps, select_obs_sets):
    """Constraint groups renumbered to the selected subset; groups left with one member vanish."""
    out = []
    for grp in c_grps:
        kept = [np.where(select_obs_sets == v)[0][0] for v in grp if v in select_obs_sets]
        if len(kept) > 1:
            out.append(tuple(kept))
    return out


# ---------------------------------------------------------------------------
# config script (bayes_io.py:259-541)
def _tabs(dtype):
    return lambda s, ctx: extract_values(s, delimiter="\t", dtype=dtype)


def _per_param(key, dtype=None, tuples=False):
    def handler(s, ctx):
        vals = extract_tuples(s, delimiter="\t") if tuples else extract_values(s, delimiter="\t", dtype=dtype)
        put_into_param_info(ctx["param_info"], vals, key)
        return None
    return handler


def _per_meas_type(s, ctx):
    """one number for every measurement type, or "(TRPL, 1)|(TRTS, 10)"."""
    try:
        v = float(s)
        return {m: v for m in ctx["grid"]["meas_types"]}
    except ValueError:
        return {t[0]: float(t[1]) for t in extract_tuples(s, delimiter="|", dtype=float)}


def _fittable(label):
    def handler(s, ctx):
        if s == "None":
            return None
        parts = s.split("\t")
        if len(parts) == 3:
            move, inds, grps = parts
            guesses = [1] * len(inds)  # (sic) one per CHARACTER of the index list, bayes_io.py:405
        elif len(parts) == 4:
            move, inds, grps, guesses = parts
            guesses = extract_values(guesses.strip("([])"), delimiter=", ", dtype=float)
        else:
            raise ValueError(f"Invalid {label}")
        inds = extract_values(inds.strip("([])"), delimiter=", ", dtype=int)
        grps = None if grps == "None" else extract_tuples(grps, delimiter="|", dtype=int)
        return [float(move), inds, grps, guesses]
    return handler


def _none_or(fn):
    return lambda s, ctx: None if s == "None" else fn(s, ctx)


# section letter -> [(line prefix, destination key or None, parser)]; first matching prefix wins
_TABLE = {
    "g": [("Length(s)", "lengths", _tabs(float)),
          ("nx", "nx", _tabs(int)),
          ("Measurement type(s)", "meas_types", lambda s, c: s.split("\t")),
          ("Number of measurements", "num_meas", lambda s, c: int(s))],
    "p": [("Param Names", "names", lambda s, c: s.split("\t")),
          ("Unit conversions", None, _per_param("unit_conversions", float)),
          ("Do logscale", None, _per_param("do_log", int)),
          ("Active", None, _per_param("active", int)),
          ("Initial guess", None, _per_param("init_guess", float)),
          ("Prior", None, _per_param("prior_dist", tuples=True)),
          ("Trial move size", None, _per_param("trial_move", float)),
          ("Mu constraint", "do_mu_constraint", _tabs(float))],
    "m": [("Time cutoffs", "time_cutoff", _tabs(float)),
          ("Select measurement", "select_obs_sets", _none_or(lambda s, c: list(extract_values(s, "\t", int))))],
    "s": [("Num iters", "num_iters", lambda s, c: int(s)),
          ("Starting iter", "starting_iter", lambda s, c: int(s)),
          ("Solver name", "solver", lambda s, c: tuple(s.split("\t"))),
          ("Model name", "model", lambda s, c: s),
          ("Solver rtol", "rtol", lambda s, c: float(s)),
          ("Solver atol", "atol", lambda s, c: float(s)),
          ("Solver hmax", "hmax", lambda s, c: float(s)),
          ("Init mode", "ini_mode", lambda s, c: s),
          ("Likelihood-to-trial-move", "likel2move_ratio", _per_meas_type),
          ("Model uncertainty", "model_uncertainty", _per_meas_type),
          ("Use log of measurements", "log_y", lambda s, c: int(s)),
          ("Scale factor", "scale_factor", _fittable("scale factor")),
          ("Fittable fluences", "fittable_fluences", _fittable("fittable_fluence")),
          ("Fittable absorptions", "fittable_absps", _fittable("fittable_absp")),
          ("Use hard boundaries", "hard_bounds", lambda s, c: int(s)),
          ("Force min y", "force_min_y", lambda s, c: int(s)),
          ("IRF", "irf_convolution", _none_or(_tabs(float))),
          ("Parallel tempering", "parallel_tempering", _tabs(float)),
          ("Tempering frequency", "temper_freq", lambda s, c: int(s)),
          ("Checkpoint freq", "checkpoint_freq", lambda s, c: int(s)),
          ("Load checkpoint", "load_checkpoint", _none_or(lambda s, c: s)),
          ("Initial condition path", "init_cond_path", lambda s, c: os.path.join(s)),
          ("Measurement path", "measurement_path", lambda s, c: os.path.join(s)),
          ("Output path", "output_path", lambda s, c: os.path.join(s))],
}
_SECTIONS = (("p$ Space Grid", "g"), ("p$ Param Info", "p"), ("p$ Measurement handling flags", "m"),
             ("p$ MCMC Control flags", "s"))


def read_config_script_file(path):
    """-> (sim_info, param_info, meas_fields, MCMC_fields) exactly as the reference returns them,
    including the re-indexing of fittables / constraint groups / IRF list to select_obs_sets and
    the `_s*`, `_f*`, `_a*` parameters appended to param_info."""
    grid, param_info, meas_flags, MCMC_fields = {}, {}, {}, {}
    dest = {"g": grid, "p": param_info, "m": meas_flags, "s": MCMC_fields}
    ctx = {"grid": grid, "param_info": param_info}
    with open(path, "r") as f:
        if "$$ MCMC CONFIG CREATED" not in next(f):
            raise OSError("Error: this file is not a valid MCMC config file")
        section = None
        for line in f:
            line = line.strip("\n")
            if "#" in line or not line:
                continue
            for marker, letter in _SECTIONS:
                if marker in line:
                    section = letter
            parts = get_split_and_clean_line(line)
            if len(parts) < 2 or section is None or line.startswith("p$"):
                continue
            for prefix, key, parse in _TABLE[section]:
                if line.startswith(prefix):
                    val = parse(parts[1], ctx)
                    if key is not None:
                        dest[section][key] = val
                    break

    _require(grid, ("lengths", "nx", "meas_types", "num_meas"), "Space Grid")
    _require(param_info, ("names", "unit_conversions", "do_log", "active", "init_guess", "prior_dist", "trial_move"),
             "Param Info")
    _require(meas_flags, ("time_cutoff", "select_obs_sets"), "Measurement handling flags")
    _require(MCMC_fields, ("num_iters", "solver", "model", "ini_mode", "log_y", "checkpoint_freq", "init_cond_path",
                           "measurement_path", "output_path"), "MCMC Control flags")

    if meas_flags.get("select_obs_sets", None) is None:
        meas_flags["select_obs_sets"] = np.arange(grid["num_meas"])
    else:
        meas_flags["select_obs_sets"] = np.array(meas_flags["select_obs_sets"], dtype=int)
    sel = meas_flags["select_obs_sets"]
    for field, _ in _FITTABLES:
        spec = MCMC_fields.get(field, None)
        if spec is None:
            continue
        spec[1] = remap_fittable_inds(spec[1], sel)
        if spec[2] is not None:
            spec[2] = remap_constraint_grps(spec[2], sel)
        spec[3] = list(np.array(spec[3])[sel])
    insert_param(param_info, MCMC_fields, mode="scale_f")
    insert_param(param_info, MCMC_fields, mode="fluences")
    insert_param(param_info, MCMC_fields, mode="absorptions")
    # (the reference leaves grid["nx"] at its full length, bayes_io.py:531-536)
    grid["meas_types"] = [grid["meas_types"][i] for i in sel]
    grid["lengths"] = [grid["lengths"][i] for i in sel]
    grid["num_meas"] = len(sel)
    if MCMC_fields.get("irf_convolution", None) is not None:
        MCMC_fields["irf_convolution"] = [MCMC_fields["irf_convolution"][i] for i in sel]
    return grid, param_info, meas_flags, MCMC_fields


def _require(d, keys, section):
    missing = [k for k in keys if k not in d]
    if missing:
        raise ValueError(f"MCMC config: section '{section}' lacks {missing}")
