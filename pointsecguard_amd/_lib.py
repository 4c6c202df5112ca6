"""ctypes binding of libpsg.so (the C ABI declared in include/psg.h).

The HIP library IS the product: there is no CPU or PyTorch fallback.  Importing this module never
touches the GPU; the first call that needs the library loads it and fails loudly if it is missing.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PSG_LIBRARY_OVERRIDE: a variant build (tools/build_variant.sh -> build/libpsg_NAME.so) for probes; bench.py refuses to
# run with it unless --allow-env-switches is given, and then lists it in config.env_switches
LIB_PATH = os.environ.get("PSG_LIBRARY_OVERRIDE") or os.path.join(_HERE, "libpsg.so")

c_f = ctypes.POINTER(ctypes.c_float)
c_i = ctypes.POINTER(ctypes.c_int32)
c_u8 = ctypes.POINTER(ctypes.c_uint8)
c_ll = ctypes.POINTER(ctypes.c_longlong)
vp = ctypes.c_void_p
ci = ctypes.c_int
cf = ctypes.c_float

# name -> (restype, argtypes); mirrors include/psg.h one to one (tests/test_abi.py checks the set)
SIGNATURES = {
    "psg_last_error": (ctypes.c_char_p, []),
    "psg_version": (ctypes.c_char_p, []),
    "psg_env_switches": (ci, [ctypes.c_char_p, ci]),
    "psg_diag_build": (ci, []),
    "psg_ctx_create": (ci, [ci, ctypes.POINTER(vp)]),
    "psg_ctx_destroy": (ci, [vp]),
    "psg_square_distance": (ci, [vp, vp, vp, ci, ci, ci, vp, vp]),
    "psg_fps": (ci, [vp, vp, ci, ci, ci, ci, vp, vp, vp]),
    "psg_gather_points": (ci, [vp, vp, ci, ci, ci, ci, vp, ci, vp, vp]),
    "psg_ball_query": (ci, [vp, vp, ci, vp, ci, ci, ci, cf, ci, vp, vp]),
    "psg_three_nn": (ci, [vp, vp, ci, vp, ci, ci, ci, vp, vp, vp]),
    "psg_group_rows": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp]),
    "psg_group_rows_bwd": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, vp, vp]),
    "psg_pw_mlp_fwd": (ci, [vp, ci, ci, ci, vp, vp, ci, ci, vp, ci, vp, vp, vp, vp]),
    "psg_pw_mlp_bwd": (ci, [vp, ci, ci, ci, vp, vp, ci, vp, ci, vp]),
    "psg_apply_relu_bits": (ci, [vp, ci, vp, vp, ci, ci, vp]),
    "psg_mrconv_gather_fwd": (ci, [vp, ci, ci, ci, ci, vp, vp, vp, vp]),
    "psg_mrconv_gather_bwd": (ci, [vp, ci, ci, ci, vp, vp, vp, ci, vp]),
    "psg_sa_mlp_max_fwd": (ci, [vp, ci, ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(vp), ctypes.POINTER(vp), vp, vp,
                                ctypes.POINTER(vp), vp, vp, vp]),
    "psg_sa_mlp_max_bwd": (ci, [vp, vp, ci, ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(vp), ctypes.POINTER(vp), vp, vp, vp, vp]),
    "psg_three_interp_fwd": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp]),
    "psg_three_interp_bwd": (ci, [vp, ci, ci, vp, vp, ci, ci, ci, ci, vp, vp]),
    "psg_gcn_pairwise_distance": (ci, [vp, ci, ci, ci, vp, vp, vp]),
    "psg_global_max": (ci, [vp, ci, ci, ci, vp, vp, vp, vp]),
    "psg_edgeconv_fwd": (ci, [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp]),
    "psg_edgeconv_bwd": (ci, [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, ci, vp]),
    "psg_pn2_model_create": (ci, [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]),
    "psg_pn2_model_create_arch": (ci, [vp, ci, ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ctypes.POINTER(vp)]),
    "psg_pn2_model_destroy": (ci, [vp]),
    "psg_pn2_ws_create": (ci, [vp, ci, ci, ci, ctypes.POINTER(vp)]),
    "psg_pn2_ws_create_arch": (ci, [vp, ci, ci, ci, ci, ctypes.POINTER(vp)]),
    "psg_pn2_ws_destroy": (ci, [vp]),
    "psg_pn2_ws_bytes": (ctypes.c_size_t, [vp]),
    "psg_pn2_debug_read": (ci, [vp, ctypes.POINTER(ctypes.c_ulonglong), ci]),
    "psg_pn2_prof_enable": (ci, [vp, ci]),
    "psg_pn2_prof_read": (ci, [vp, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    "psg_pn2_plan_build": (ci, [vp, vp, vp, ci, vp]),
    "psg_pn2_plan_ptr": (vp, [vp, ci, ci, ci, ci]),
    "psg_pn2_forward": (ci, [vp, vp, ci, vp, vp, vp, vp]),
    "psg_pn2_backward": (ci, [vp, vp, ci, vp, vp, vp]),
    "psg_pn2_forward_lean": (ci, [vp, vp, ci, vp, vp, vp]),
    "psg_pn2_backward_colour": (ci, [vp, vp, ci, vp, vp, vp]),
    "psg_pn2_backward_colour_pgd": (ci, [vp, vp, ci, vp, vp, vp, vp, cf, cf, cf, ci, vp]),
    "psg_pn2_activation_ptr": (vp, [vp, ci]),
    "psg_pn2_activation_channels": (ci, [vp, ci]),
    "psg_to_point_major": (ci, [vp, ci, ci, ci, vp, vp]),
    "psg_to_channel_major": (ci, [vp, ci, ci, ci, vp, vp]),
    "psg_ce_logp_grad": (ci, [vp, vp, ci, ci, ci, ci, cf, vp, vp, vp]),
    "psg_pgd_step": (ci, [vp, vp, vp, vp, ci, ci, cf, cf, cf, ci, vp]),
    "psg_pn2_nb_attack": (ci, [vp, vp, vp, vp, vp, vp, cf, cf, ci, ci, ci, vp, vp]),
    "psg_nu_inverse_tanh": (ci, [vp, ci, ci, vp, vp]),
    "psg_nu_tanh_color": (ci, [vp, vp, ci, ci, vp, vp]),
    "psg_nu_f_loss_grad": (ci, [vp, vp, ci, ci, ci, cf, cf, vp, vp, vp, vp]),
    "psg_gcn_f_loss_grad": (ci, [vp, vp, ci, vp, ci, ci, ci, ci, cf, cf, cf, vp, vp, vp, vp]),
    "psg_smooth_knn": (ci, [vp, ci, vp, ci, ci, ci, vp, vp, vp]),
    "psg_nu_adam_step": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, cf, cf, cf, cf, ci, ci, ci, vp, vp]),
    "psg_pn2_nu_window": (ci, [vp, vp, vp]),
    "psg_nu_graph_create": (ci, [ctypes.POINTER(vp)]),
    "psg_nu_graph_destroy": (ci, [vp]),
    "psg_nu_graph_stats": (ci, [vp, c_ll]),
    "psg_capture_stats": (ci, [c_ll]),
    "psg_nu_step_latch": (ci, [vp, vp, ci, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, ci, vp]),
    "psg_nu_tanh_color_rooms": (ci, [vp, vp, ci, ci, vp, vp]),
    "psg_nu_f_loss_grad_rooms": (ci, [vp, vp, ci, ci, ci, ci, cf, cf, vp, vp, vp, vp]),
    "psg_smooth_knn_rooms": (ci, [vp, ci, ctypes.c_size_t, vp, ci, ctypes.c_size_t, ci, ci, ci, vp, vp, vp, ci, vp]),
    "psg_nu_adam_step_rooms": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, cf, cf, cf, cf, ci, ci, ci, vp, vp, vp]),
    "psg_gcn_model_create": (ci, [vp, ctypes.POINTER(vp), ci, ci, ctypes.POINTER(vp)]),
    "psg_gcn_model_create_cfg": (ci, [vp, ctypes.POINTER(vp), ci, ci, ci, ci, ctypes.POINTER(vp)]),
    "psg_gcn_model_destroy": (ci, [vp]),
    "psg_gcn_ws_create": (ci, [vp, ci, ci, ci, ctypes.POINTER(vp)]),
    "psg_gcn_ws_create_cfg": (ci, [vp, ci, ci, ci, ci, ci, ctypes.POINTER(vp)]),
    "psg_gcn_ws_destroy": (ci, [vp]),
    "psg_gcn_ws_bytes": (ctypes.c_size_t, [vp]),
    "psg_gcn_prof_enable": (ci, [vp, ci]),
    "psg_gcn_prof_read": (ci, [vp, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]),
    "psg_gcn_knn": (ci, [vp, vp, ci, ci, vp, vp]),
    "psg_gcn_knn_stats": (ci, [vp, ctypes.POINTER(ctypes.c_ulonglong), ci]),
    "psg_gcn_forward": (ci, [vp, vp, vp, vp, vp]),
    "psg_gcn_backward": (ci, [vp, vp, vp, vp, vp]),
    "psg_gcn_nb_attack": (ci, [vp, vp, vp, vp, cf, cf, ci, vp, vp]),
    "psg_gcn_set_graphs": (ci, [vp, vp, vp]),
    "psg_gcn_edge_ptr": (vp, [vp, ci]),
    "psg_gcn_feats_ptr": (vp, [vp]),
    "psg_knn_points": (ci, [vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "psg_rla_sampler_create": (ci, [vp, vp, vp, ci, ctypes.POINTER(vp)]),
    "psg_rla_sampler_destroy": (ci, [vp]),
    "psg_rla_sampler_argmin": (ci, [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double), vp]),
    "psg_rla_sampler_query": (ci, [vp, vp, ci, vp, vp]),
    "psg_rla_sampler_update": (ci, [vp, vp, ci, vp, vp, vp]),
    "psg_rla_sampler_possibility": (ci, [vp, vp]),
    "psg_rla_model_create": (ci, [vp, ctypes.POINTER(vp), ci, ctypes.POINTER(vp)]),
    "psg_rla_model_destroy": (ci, [vp]),
    "psg_rla_ws_create": (ci, [vp, ci, ctypes.POINTER(vp)]),
    "psg_rla_ws_create_batch": (ci, [vp, ci, ci, ctypes.POINTER(vp)]),
    "psg_rla_ws_destroy": (ci, [vp]),
    "psg_rla_ws_bytes": (ctypes.c_size_t, [vp]),
    "psg_rla_prof_enable": (ci, [vp, ci]),
    "psg_rla_prof_read": (ci, [vp, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]),
    "psg_rla_prof_read_kernels": (ci, [vp, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "psg_rla_set_cloud": (ci, [vp, vp, vp]),
    "psg_rla_index_ptr": (vp, [vp, ci, ci]),
    "psg_rla_forward": (ci, [vp, vp, vp, vp, vp]),
    "psg_rla_backward": (ci, [vp, vp, vp, vp, vp]),
    "psg_rla_colper_grad": (ci, [vp, vp, ci, vp, vp, vp]),
    "psg_rla_colper_grad_masked": (ci, [vp, vp, vp, cf, ci, vp, vp, vp]),
    "psg_rla_bim_step": (ci, [vp, vp, vp, ci, cf, cf, ci, vp, vp, vp]),
    "psg_rla_nu_color": (ci, [vp, vp, vp, ci, vp, vp, vp]),
    "psg_rla_nu_adam_step": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, cf, cf, ci, vp]),
    "psg_rla_bim_attack": (ci, [vp, vp, vp, vp, cf, cf, ci, ci, vp, vp]),
    "psg_seg_stats": (ci, [vp, vp, ci, ci, vp, vp, vp]),
    "psg_vote_add": (ci, [vp, vp, vp, vp, ci, ci, ci, vp, vp, vp]),
    "psg_vote_stats": (ci, [vp, vp, ci, ci, vp, vp, vp]),
    "psg_l2_dist": (ci, [vp, vp, ctypes.c_size_t, vp, vp, vp]),
}



class NuWindowArgs(ctypes.Structure):
    """psg_nu_window_args of include/psg.h, field for field."""
    _fields_ = ([("model", vp), ("ws", vp)] +
                [(n, ci) for n in ("slot0", "step0", "n_steps", "G", "rows", "N", "mode", "use_target", "target", "neighbour",
                                   "warm_first", "adam_t0")] +
                [(n, cf) for n in ("kappa", "tsign", "c_smooth", "c_l2", "lr", "beta1", "beta2", "eps")] +
                [(n, vp) for n in ("w", "m", "v", "mask", "n_mask", "x0", "ori", "labels", "logp", "dlogp", "dx0", "sgrad", "pred",
                                   "scal", "nn_state", "hist", "out", "active", "exit_step")])


_lib = None


class PsgError(RuntimeError):
    pass


def load():
    """Load libpsg.so (once) and attach the prototypes.  Raises if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PsgError(
                "%s not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C pointsecguard_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        # torch first: PyTorch-ROCm ships its own libamdhip64, and libpsg.so must bind to THAT copy (the process then
        # has one HIP runtime, the one that owns torch's allocator and streams).  Loading libpsg.so before torch pulled in
        # /opt/rocm's copy as a second runtime, and device discovery failed ("no ROCm-capable device is detected").
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def env_switches():
    """[(name, value, kind)] of the libpsg environment switches set in this process (include/psg.h: psg_env_switches)."""
    buf = ctypes.create_string_buffer(4096)
    n = load().psg_env_switches(buf, len(buf))
    if n < 0:
        check(n, "psg_env_switches")
    out = []
    for item in buf.value.decode().split(";"):
        if item:
            nv, kind = item.rsplit(":", 1)
            name, value = nv.split("=", 1)
            out.append((name, value, kind))
    return out


CAPTURE_KEYS = ("captures_tried", "captures_failed", "replays", "eager")


def capture_stats(graph=None):
    """hipGraph bookkeeping {captures_tried, captures_failed, replays, eager} of one psg_nu_graph handle, or (graph=None) of
    every replayed loop of the process (include/psg.h: psg_nu_graph_stats / psg_capture_stats)."""
    out = (ctypes.c_longlong * 4)()
    if graph is None:
        call("psg_capture_stats", out)
    else:
        call("psg_nu_graph_stats", graph, out)
    return dict(zip(CAPTURE_KEYS, (int(v) for v in out)))


def check(rc, what=""):
    if rc != 0:
        msg = load().psg_last_error().decode("utf-8", "replace")
        raise PsgError("%s failed (rc=%d): %s" % (what or "libpsg call", rc, msg))


def call(name, *args):
    """Call an int-returning entry point and raise PsgError on a non-zero return code."""
    check(getattr(load(), name)(*args), name)
