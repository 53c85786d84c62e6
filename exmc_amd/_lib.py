"""ctypes binding of libexmc_hip.so (include/exmc_hip.h). No CPU fallback: if the library or a
HIP device is missing, calls raise ExmcHipError."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EXMC_HIP_LIB") or os.path.join(HERE, "lib", "libexmc_hip.so")

MAX_D = 256
OK, ERR_BADARG, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED = range(5)

EXPORTS = [
    "exmc_hip_last_error", "exmc_hip_device_count", "exmc_hip_model_create",
    "exmc_hip_model_destroy", "exmc_hip_model_dim", "exmc_hip_model_default_lanes", "exmc_hip_model_default_warmup_lanes",
    "exmc_hip_model_default_dense_lanes",
    "exmc_hip_model_stream", "exmc_hip_logp_grad_host", "exmc_hip_multi_step",
    "exmc_hip_multi_step_host", "exmc_hip_transitions_host", "exmc_hip_warmup",
    "exmc_hip_sample_chains", "exmc_hip_sample_chains_host", "exmc_hip_sample_host",
    "exmc_hip_chains_init", "exmc_hip_chains_advance", "exmc_hip_build_full_tree_host",
    "exmc_hip_ess", "exmc_hip_last_kernel_ms",
    "exmc_hip_traj_create", "exmc_hip_traj_destroy", "exmc_hip_traj_get_endpoint_host",
    "exmc_hip_traj_build_and_merge_host", "exmc_hip_traj_is_terminated_host",
    "exmc_hip_traj_get_result_host", "exmc_hip_build_subtree_host",
    "exmc_hip_stream_begin", "exmc_hip_stream_next_host", "exmc_hip_stream_start", "exmc_hip_stream_finish", "exmc_hip_rhat", "exmc_hip_ess_bulk",
    "exmc_hip_model_set_flat_order", "exmc_hip_warmup_from", "exmc_hip_sample_warm_host",
    "exmc_hip_warmup_dense", "exmc_hip_model_set_dense_mass", "exmc_hip_model_clear_dense_mass",
    "exmc_hip_sample_dense_host",
    "exmc_hip_sample_independent", "exmc_hip_sample_independent_host",
    "exmc_hip_leapfrog_chain_normal_host",
]


class ExmcHipError(RuntimeError):
    pass


class Opts(C.Structure):
    _fields_ = [("num_warmup", C.c_int), ("num_samples", C.c_int), ("max_tree_depth", C.c_int),
                ("target_accept", C.c_double), ("seed", C.c_uint64), ("lanes_per_chain", C.c_int)]


class Tuning(C.Structure):
    _fields_ = [("epsilon", C.c_double), ("inv_mass", C.c_double * MAX_D),
                ("warmup_divergences", C.c_int)]


class Trace(C.Structure):
    _fields_ = [("draws", C.c_void_p), ("logp", C.c_void_p), ("tree_depth", C.c_void_p),
                ("n_steps", C.c_void_p), ("divergent", C.c_void_p), ("accept_prob", C.c_void_p),
                ("energy", C.c_void_p)]


_libs = {}


def load():
    """Load libexmc_hip.so. torch (if used in the same process) must be imported first so both
    share one HIP runtime; exmc_amd/__init__ takes care of the order."""
    return bind(LIB_PATH)


def bind(path):
    """Load a library with the include/exmc_hip.h ABI: libexmc_hip.so itself or a plug-in build
    made for one generated model (exmc_amd/codegen.py)."""
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ExmcHipError(
            "%s is not built: run `python -m exmc_amd.build` (or codegen.build_plugin for a "
            "generated model); there is no CPU fallback" % path)
    L = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    vp = C.c_void_p
    L.exmc_hip_last_error.restype = C.c_char_p
    L.exmc_hip_device_count.restype = C.c_int
    L.exmc_hip_model_create.argtypes = [C.c_int, C.c_int, dp, C.c_int, C.c_int, C.POINTER(vp)]
    L.exmc_hip_model_destroy.argtypes = [vp]
    L.exmc_hip_model_destroy.restype = None
    L.exmc_hip_model_dim.argtypes = [vp]
    L.exmc_hip_model_set_flat_order.argtypes = [vp, C.POINTER(C.c_int32), C.c_int]
    L.exmc_hip_model_default_lanes.argtypes = [vp]
    L.exmc_hip_model_default_warmup_lanes.argtypes = [vp]
    L.exmc_hip_model_default_dense_lanes.argtypes = [vp]
    L.exmc_hip_model_stream.argtypes = [vp]
    L.exmc_hip_model_stream.restype = vp
    L.exmc_hip_logp_grad_host.argtypes = [vp, dp, C.c_int, C.c_int, dp, dp]
    L.exmc_hip_multi_step.argtypes = [vp, vp, vp, vp, C.c_double, dp, C.c_int, C.c_int, C.c_int,
                                      vp, vp, vp, vp]
    L.exmc_hip_multi_step_host.argtypes = [vp, dp, dp, dp, C.c_double, dp, C.c_int, C.c_int,
                                           C.c_int, dp, dp, dp, dp]
    L.exmc_hip_transitions_host.argtypes = [vp, dp, dp, dp, C.POINTER(C.c_uint64), C.c_int,
                                            C.c_int, C.c_double, dp, C.c_int, C.c_int, Trace]
    L.exmc_hip_warmup.argtypes = [vp, dp, Opts, C.POINTER(Tuning)]
    L.exmc_hip_sample_chains.argtypes = [vp, C.POINTER(Tuning), dp, C.c_int, C.c_int, C.c_int,
                                         Opts, Trace, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    L.exmc_hip_sample_chains_host.argtypes = L.exmc_hip_sample_chains.argtypes
    L.exmc_hip_chains_init.argtypes = [vp, C.POINTER(Tuning), dp, C.c_int, C.c_int, C.c_int, Opts]
    L.exmc_hip_chains_advance.argtypes = [vp, C.c_int, C.c_int, Trace, C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int32)]
    L.exmc_hip_sample_host.argtypes = [vp, dp, Opts, Trace, C.POINTER(Tuning),
                                       C.POINTER(C.c_int32)]
    L.exmc_hip_warmup_dense.argtypes = [vp, dp, Opts, C.POINTER(Tuning), dp, dp]
    L.exmc_hip_model_set_dense_mass.argtypes = [vp, dp, dp, C.c_int]
    L.exmc_hip_sample_dense_host.argtypes = [vp, dp, Opts, Trace, C.POINTER(Tuning), dp, dp, C.POINTER(C.c_int32)]
    L.exmc_hip_model_clear_dense_mass.argtypes = [vp]
    L.exmc_hip_warmup_from.argtypes = [vp, dp, Opts, C.POINTER(Tuning), C.POINTER(Tuning)]
    L.exmc_hip_sample_warm_host.argtypes = [vp, dp, Opts, C.POINTER(Tuning), Trace, C.POINTER(Tuning),
                                            C.POINTER(C.c_int32)]
    ip = C.POINTER(C.c_int32)
    L.exmc_hip_build_full_tree_host.argtypes = [
        C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, C.c_int, dp, dp, dp, dp,
        C.c_int, dp, dp, C.c_int, C.POINTER(C.c_uint64), dp, dp, dp, ip, ip, dp, ip]
    L.exmc_hip_stream_begin.argtypes = [vp, dp, Opts, C.POINTER(Tuning)]
    L.exmc_hip_stream_next_host.argtypes = [vp, C.c_int, Trace, ip]
    L.exmc_hip_stream_start.argtypes = [vp, C.c_int, C.POINTER(Trace), C.POINTER(ip)]
    L.exmc_hip_stream_finish.argtypes = [vp, ip]
    up = C.POINTER(C.c_uint64)
    L.exmc_hip_traj_create.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, C.POINTER(vp)]
    L.exmc_hip_traj_destroy.argtypes = [vp]
    L.exmc_hip_traj_destroy.restype = None
    L.exmc_hip_traj_get_endpoint_host.argtypes = [vp, ip, dp, dp, dp]
    L.exmc_hip_traj_build_and_merge_host.argtypes = [vp, dp, dp, dp, dp, C.c_int, dp, dp, ip, ip, up]
    L.exmc_hip_traj_is_terminated_host.argtypes = [vp, ip]
    L.exmc_hip_traj_get_result_host.argtypes = [vp, dp, dp, dp, ip, ip, dp, ip]
    L.exmc_hip_build_subtree_host.argtypes = [
        C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, C.c_int, dp, dp, ip, ip, up,
        dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, ip, ip, dp, ip, ip, dp]
    L.exmc_hip_leapfrog_chain_normal_host.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, dp, C.c_int, C.c_double,
                                                      C.c_double, C.c_double, dp, dp, dp, dp]
    L.exmc_hip_sample_independent.argtypes = [vp, dp, C.c_int, C.c_int, C.c_int, Opts, Trace, dp,
                                              C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    L.exmc_hip_sample_independent_host.argtypes = L.exmc_hip_sample_independent.argtypes
    L.exmc_hip_ess.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.exmc_hip_rhat.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.exmc_hip_ess_bulk.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.exmc_hip_last_kernel_ms.argtypes = [vp]
    L.exmc_hip_last_kernel_ms.restype = C.c_double
    _libs[path] = L
    return L


def check(rc, L=None):
    if rc != OK:
        msg = (L or load()).exmc_hip_last_error()
        raise ExmcHipError("libexmc_hip error %d: %s" % (rc, msg.decode() if msg else ""))
