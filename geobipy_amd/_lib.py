"""ctypes binding of libgeobipy_amd.so (C ABI in include/geobipy_amd.h).

There is deliberately NO fallback: if the HIP library has not been built, importing a symbol from
here raises.  Build it with ``python __graft_entry__.py`` (or ``geobipy_amd.build.build_native()``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgeobipy_amd.so")

c_int = ctypes.c_int
c_void_p = ctypes.c_void_p
c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)



class RjOptions(ctypes.Structure):
    """gbp_rj_options (include/geobipy_amd.h)."""
    _fields_ = ([(n, ctypes.c_int32) for n in ("max_layers", "n_channels", "solve_gradient", "solve_value", "solve_relative_error",
                                               "solve_additive_error", "exact_jacobian", "n_depth_bins", "n_value_bins", "n_error_bins", "schedule",
                                               "burn_in_min_iterations", "n_markov_chains", "forward_waves")]
                + [(n, ctypes.c_double) for n in ("min_edge", "max_edge", "min_width", "p_birth", "p_death", "p_perturb", "p_none",
                                                  "value_precision", "value_min", "value_max", "gradient_precision", "alpha")]
                + [("n_rel_groups", ctypes.c_int32), ("n_add_groups", ctypes.c_int32)]
                + [(n, ctypes.c_double * 4) for n in ("rel_min", "rel_max", "rel_sd", "add_min", "add_max", "add_sd")]
                + [("depth_bin_width", ctypes.c_double), ("value_half_width", ctypes.c_double)]
                + [("seed", ctypes.c_uint64), ("first_chain", ctypes.c_uint64)]
                + [("solve_height", ctypes.c_int32), ("height_half_width", ctypes.c_double), ("height_scale", ctypes.c_double),
                   ("additive_independent", ctypes.c_int32), ("add_centre", ctypes.c_double * 4), ("extra_log_prior", ctypes.c_double),
                   ("trace_every", ctypes.c_int32), ("trace_length", ctypes.c_int32)])


RJ_CHAIN_FIELDS = ("rel_group", "add_group", "add_scale", "chain_id", "data", "height", "log_mean_prior", "k", "edges", "sigma", "rel", "add", "pred", "J", "prior", "like", "misfit",
                   "action", "k_r", "nl_a", "nl_c", "nl_b", "edges_r", "sigma_r", "thk_r", "rel_p", "add_p", "pred_r", "J_r", "chol",
                   "log_prop", "sigma_p", "pred_p", "misfit_p", "like_p", "J_p", "log_ratio", "n_accepted", "k_hist", "edge_hist",
                   "rel_hist", "add_hist", "hitmap", "hit_dwell", "burned_in_iteration", "status", "best_posterior", "best_k", "best_edges", "best_sigma",
                   "best_rel", "best_add", "iteration0", "height_p", "height0", "height_hist", "best_height", "step_flags", "trace_misfit", "trace_accept",
                   "best_iteration")


class RjChains(ctypes.Structure):
    """gbp_rj_chains: B and the device pointers in declaration order."""
    _fields_ = [("B", ctypes.c_int32)] + [(n, c_void_p) for n in RJ_CHAIN_FIELDS]


class TdMix(ctypes.Structure):
    """gbp_td_mix."""
    _fields_ = [("n_in", ctypes.c_int32), ("terms", ctypes.c_int32), ("n_weights", ctypes.c_int32), ("src", c_void_p),
                ("col", c_void_p), ("weights", c_void_p), ("offset", c_void_p)]


class TdMoves(ctypes.Structure):
    """gbp_td_moves."""
    _fields_ = [("n_moves", ctypes.c_int32), ("entry", ctypes.c_int32 * 6), ("sign", ctypes.c_double * 6), ("half_width", ctypes.c_double * 6),
                ("scale", ctypes.c_double * 6), ("n_bins", ctypes.c_int32 * 6), ("geom", c_void_p), ("geom_p", c_void_p), ("geom0", c_void_p),
                ("weights", c_void_p), ("weights_p", c_void_p), ("offset", c_void_p), ("offset_p", c_void_p), ("hist", c_void_p),
                ("best_geom", c_void_p), ("n_blocks", ctypes.c_int32), ("n_basis", ctypes.c_int32), ("loop", ctypes.c_int32),
                ("on_axis", ctypes.c_int32), ("basis", ctypes.c_int32 * 5), ("block_comp", c_void_p), ("block_scale", c_void_p),
                ("block_primary", c_void_p), ("block_windows", c_void_p), ("rho_scale", c_void_p), ("rho_scale_p", c_void_p),
                ("rho_set", c_void_p), ("dz_set", c_void_p)]


class TdOperator(ctypes.Structure):
    """gbp_td_operator."""
    _fields_ = [("n_nodal", ctypes.c_int32), ("W", c_void_p), ("nodal", c_void_p), ("J_nodal", c_void_p), ("mix", TdMix),
                ("table_set", c_void_p), ("moves", TdMoves)]


_rj_o, _rj_c = ctypes.POINTER(RjOptions), ctypes.POINTER(RjChains)

# name -> (restype, argtypes); must list every symbol include/geobipy_amd.h declares
SIGNATURES = {
    "gbp_rj_propose": (c_int, [_rj_o, _rj_c, ctypes.c_int64, c_void_p]),
    "gbp_rj_debug_propose_variant": (c_int, [_rj_o, _rj_c, ctypes.c_int64, c_int, c_void_p]),
    "gbp_rj_newton": (c_int, [_rj_o, _rj_c, ctypes.c_int64, c_void_p]),
    "gbp_rj_accept": (c_int, [_rj_o, _rj_c, ctypes.c_int64, c_int, c_void_p]),
    "gbp_rj_run": (c_int, [c_void_p, _rj_o, _rj_c, ctypes.c_int64, c_int, c_int, c_void_p]),
    "gbp_rj_run_mode": (c_int, [c_void_p, _rj_o, _rj_c, ctypes.c_int64, c_int, c_int, c_int, c_void_p]),
    "gbp_rj_debug_stage_ticks": (c_int, [ctypes.POINTER(ctypes.c_int64), c_int]),
    "gbp_tdem_system_create": (c_int, [ctypes.c_char_p, c_double_p, c_double_p, ctypes.POINTER(c_void_p)]),
    "gbp_tdem_system_destroy": (None, [c_void_p]),
    "gbp_tdem_system_info": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_double_p]),
    "gbp_tdem_system_tables": (c_int, [c_void_p, c_double_p, c_double_p, c_double_p]),
    "gbp_tdem_forward": (c_int, [c_void_p, c_int, c_double_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gbp_tdem_fm_dlogc": (c_int, [c_void_p, c_int, c_double_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gbp_td_apply": (c_int, [c_int, c_int, c_int, c_int] + [c_void_p] * 6 + [c_void_p]),
    "gbp_td_apply_mix": (c_int, [c_int, c_int, c_int, c_int] + [c_void_p] * 6 + [ctypes.POINTER(TdMix), c_void_p]),
    "gbp_rj_flush_posteriors": (c_int, [_rj_o, _rj_c, c_void_p]),
    "gbp_rj_run_td": (c_int, [c_void_p, ctypes.POINTER(TdOperator), _rj_o, _rj_c, ctypes.c_int64, c_int, c_int, c_void_p]),
    "gbp_rj_debug_random": (c_int, [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gbp_version": (ctypes.c_char_p, []),
    "gbp_last_error": (ctypes.c_char_p, []),
    "gbp_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "gbp_fdem_system_create": (c_int, [c_int, c_int32_p] + [c_double_p] * 11 + [ctypes.POINTER(c_void_p)]),
    "gbp_fdem_system_create_windowed": (c_int, [c_int, c_int32_p] + [c_double_p] * 11 + [ctypes.c_double, ctypes.c_double,
                                                                                         ctypes.POINTER(c_void_p)]),
    "gbp_fdem_system_create_binned": (c_int, [c_int, c_int32_p] + [c_double_p] * 11 + [ctypes.c_double, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "gbp_fdem_system_bin_points": (c_int, [c_void_p, c_int, ctypes.POINTER(c_int)]),
    "gbp_fdem_system_npoints": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "gbp_hankel_system_add_bins": (c_int, [c_void_p, ctypes.c_double, c_int, c_int, c_int]),
    "gbp_hankel_system_clear_bins": (c_int, [c_void_p]),
    "gbp_hankel_system_add_set": (c_int, [c_void_p, c_double_p, c_double_p]),
    "gbp_tdem_system_set_hankel_eps": (c_int, [c_void_p, ctypes.c_double]),
    "gbp_hankel_system_create_raw": (c_int, [c_int, c_int32_p] + [c_double_p] * 4 + [ctypes.POINTER(c_void_p)]),
    "gbp_fdem_system_destroy": (None, [c_void_p]),
    "gbp_fdem_system_nfreq": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "gbp_fdem_system_h0": (c_int, [c_void_p, c_double_p]),
    "gbp_fdem_forward": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 5 + [c_void_p]),
    "gbp_fdem_forward_ex": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 5 + [c_int, c_void_p]),
    "gbp_fdem_forward_rows_ex": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 5 + [c_void_p, c_int, c_void_p]),
    "gbp_fdem_fm_dlogc_rows_ex": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 6 + [c_int, c_int, c_void_p, c_int, c_void_p]),
    "gbp_fdem_forward_rows_scaled": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 5 + [c_void_p, c_void_p, c_int, c_void_p]),
    "gbp_fdem_fm_dlogc_rows_scaled": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 6 + [c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "gbp_fdem_validate": (c_int, [c_int, c_int, c_int] + [c_void_p] * 6 + [c_void_p]),
    "gbp_gauss_loglike": (c_int, [c_int, c_int] + [c_void_p] * 6 + [c_void_p]),
    "gbp_gauss_loglike_std": (c_int, [c_int, c_int] + [c_void_p] * 5 + [c_void_p]),
    "gbp_fdem_forward_loglike": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 10 + [c_void_p]),
    "gbp_fdem_forward_loglike_ex": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 10 + [c_int, c_void_p]),
    "gbp_fdem_sensitivity": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 5 + [c_void_p]),
    "gbp_fdem_sensitivity_ex": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "gbp_fdem_fm_dlogc": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 6 + [c_int, c_int, c_void_p]),
    "gbp_fdem_fm_dlogc_ex": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 6 + [c_int, c_int, c_int, c_void_p]),
    "gbp_hitmap_statistics": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, ctypes.c_double] + [c_void_p] * 4 + [c_void_p]),
    "gbp_hitmap_runs": (c_int, [c_int, ctypes.c_int64] + [c_void_p] * 5 + [c_void_p]),
    "gbp_runs_to_zlib": (c_int, [c_int, ctypes.c_int64] + [c_void_p] * 4 + [ctypes.c_int64, c_void_p]),
    "gbp_debug_math": (c_int, [c_int, c_int] + [c_void_p] * 4 + [c_void_p]),
    "gbp_bench_time_forward_loglike": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 10 + [c_void_p, c_int,
                                                                                        ctypes.POINTER(ctypes.c_float)]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise NativeLibraryError loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: the gfx950 HIP library has not been built. "
            "Run `python __graft_entry__.py` in the repo root (needs hipcc). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != 0:
        msg = load().gbp_last_error().decode(errors="replace")
        raise NativeLibraryError(f"geobipy_amd native call failed (status {status}): {msg}")
