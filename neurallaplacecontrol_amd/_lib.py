"""ctypes binding of ``libnlc_hip.so`` (C ABI declared in ``include/nlc.h``).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C neurallaplacecontrol_amd/csrc``).
There is NO fallback: if the shared object is missing or no MI355X is visible, every product entry
point raises.  HIP is initialised lazily (first ``get_ctx()``), never at import, so the package is
safe to import in ``multiprocessing`` *spawn* workers before they pick a device
(cf. reference ``run_exp_multi.py:145,207``).
"""

import ctypes as C
import os
import threading

NLC_MAX_NU = 2
NLC_MAX_NIN = 3
NLC_MAX_D = 8

ILT_ALGOS = {"fourier": 0, "dehoog": 1, "fixed_tablot": 2, "stehfest": 3}
# "oderl-cartpole-notrig": CTCartpole(obs_trans=False), the 4-dim state [x, xdot, theta, thetadot] (ctcartpole.py:60, 297-300)
ENV_IDS = {"oderl-cartpole": 0, "oderl-pendulum": 1, "oderl-acrobot": 2, "oderl-cartpole-notrig": 3}
DYN_NL, DYN_ORACLE, DYN_EXTERNAL, DYN_DTRNN, DYN_NODE = 0, 1, 2, 3, 4

ERRORS = {-1: "BAD_ARG", -2: "BAD_SHAPE", -3: "HIP_ERROR", -4: "UNSUPPORTED", -5: "STATE", -6: "COMM"}
COMM_ID_BYTES = 128
NLC_ERR_UNSUPPORTED = -4
NLC_AGAIN = 1  # include/nlc.h: nlc_mppi_finish re-ran the command on every rank; gather the partials again and call again

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnlc_hip.so")


def use_library(path):
    """tools/ only (A/B builds of one kernel on one box): bind another build of the SAME library.  Must be called before
    the first ``load_library()``; the product path never reads an environment variable for this."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_library() must be called before the library is loaded")
    LIB_PATH = os.path.abspath(path)

# every symbol include/nlc.h declares (tests check the built library exports them all)
SYMBOLS = [
    "nlc_abi_version",
    "nlc_create",
    "nlc_destroy",
    "nlc_last_error",
    "nlc_set_stream",
    "nlc_set_option",
    "nlc_get_stat",
    "nlc_synchronize",
    "nlc_device_info",
    "nlc_ilt_rep_inputs",
    "nlc_ilt_reconstruct",
    "nlc_ilt_reconstruct_backward",
    "nlc_model_blob_size",
    "nlc_set_model",
    "nlc_gru_encode",
    "nlc_model_workspace_bytes",
    "nlc_model_forward",
    "nlc_model_forward_const_t",
    "nlc_rep_func",
    "nlc_rnn_blob_size",
    "nlc_set_rnn_model",
    "nlc_rnn_forward",
    "nlc_node_blob_size",
    "nlc_set_node_model",
    "nlc_node_forward",
    "nlc_mppi_configure",
    "nlc_mppi_workspace_bytes",
    "nlc_mppi_set_U",
    "nlc_mppi_get_U",
    "nlc_mppi_rollout",
    "nlc_mppi_weights",
    "nlc_mppi_finish",
    "nlc_comm_unique_id",
    "nlc_comm_init",
    "nlc_comm_destroy",
    "nlc_comm_self_test",
    "nlc_env_step",
    "nlc_env_obs",
    "nlc_profile_enable",
    "nlc_profile_reset",
    "nlc_profile_count",
    "nlc_profile_read",
]


class IltDesc(C.Structure):
    _fields_ = [("algo", C.c_int32), ("terms", C.c_int32), ("alpha", C.c_double), ("tol", C.c_double), ("scale", C.c_double)]


class NodeDesc(C.Structure):
    _fields_ = [
        ("d", C.c_int32),
        ("nu", C.c_int32),
        ("hidden", C.c_int32),
        ("augment_dim", C.c_int32),
        ("time_div", C.c_double),
        ("step_size", C.c_double),
        ("state_mean", C.c_double * NLC_MAX_D),
        ("state_std", C.c_double * NLC_MAX_D),
    ]


class RnnDesc(C.Structure):
    _fields_ = [
        ("d", C.c_int32),
        ("nin", C.c_int32),
        ("hidden", C.c_int32),
        ("time_input", C.c_int32),
        ("time_div", C.c_double),
        ("state_mean", C.c_double * NLC_MAX_D),
        ("state_std", C.c_double * NLC_MAX_D),
        ("action_mean", C.c_double * NLC_MAX_NIN),
        ("action_std", C.c_double * NLC_MAX_NIN),
    ]


class ModelDesc(C.Structure):
    _fields_ = [
        ("d", C.c_int32),
        ("nin", C.c_int32),
        ("h", C.c_int32),
        ("ilt", IltDesc),
        ("time_div", C.c_double),
        ("state_mean", C.c_double * NLC_MAX_D),
        ("state_std", C.c_double * NLC_MAX_D),
        ("action_mean", C.c_double * NLC_MAX_NIN),
        ("action_std", C.c_double * NLC_MAX_NIN),
    ]


class MppiDesc(C.Structure):
    _fields_ = [
        ("K", C.c_int64),
        ("K_global", C.c_int64),
        ("k_offset", C.c_int64),
        ("T", C.c_int32),
        ("nu", C.c_int32),
        ("d", C.c_int32),
        ("B", C.c_int32),
        ("lambda_", C.c_double),
        ("u_scale", C.c_double),
        ("has_bounds", C.c_int32),
        ("u_min", C.c_double * NLC_MAX_NU),
        ("u_max", C.c_double * NLC_MAX_NU),
        ("u_init", C.c_double * NLC_MAX_NU),
        ("noise_mu", C.c_double * NLC_MAX_NU),
        ("noise_sigma", C.c_double * (NLC_MAX_NU * NLC_MAX_NU)),
        ("noise_sigma_inv", C.c_double * (NLC_MAX_NU * NLC_MAX_NU)),
        ("noise_chol", C.c_double * (NLC_MAX_NU * NLC_MAX_NU)),
        ("sample_null_action", C.c_int32),
        ("noise_abs_cost", C.c_int32),
        ("u_per_command", C.c_int32),
        ("dynamics", C.c_int32),
        ("env", C.c_int32),
        ("delay", C.c_int32),
        ("friction", C.c_int32),
        ("E", C.c_int32),
        ("cost_external", C.c_int32),
        ("ts_pred", C.c_double),
    ]


class MppiBuffers(C.Structure):
    _fields_ = [
        ("noise", C.c_void_p),
        ("perturbed", C.c_void_p),
        ("states", C.c_void_p),
        ("actions", C.c_void_p),
        ("cost_total", C.c_void_p),
        ("cost_nz", C.c_void_p),
        ("omega", C.c_void_p),
        ("partials", C.c_void_p),
        ("action", C.c_void_p),
        ("workspace", C.c_void_p),
    ]


_lib = None
_lib_lock = threading.Lock()


def load_library():
    """dlopen libnlc_hip.so and declare prototypes.  Raises if the extension has not been built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C neurallaplacecontrol_amd/csrc`). "
                "There is no CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        vp, i64, dbl, i32 = C.c_void_p, C.c_int64, C.c_double, C.c_int
        P = C.POINTER
        lib.nlc_abi_version.restype = i32
        lib.nlc_create.argtypes = [i32, P(vp)]
        lib.nlc_destroy.argtypes = [vp]
        lib.nlc_destroy.restype = None
        lib.nlc_last_error.argtypes = [vp]
        lib.nlc_last_error.restype = C.c_char_p
        lib.nlc_set_stream.argtypes = [vp, vp]
        lib.nlc_synchronize.argtypes = [vp]
        lib.nlc_device_info.argtypes = [vp, C.c_char_p, i32, P(i32), P(i32), P(dbl)]
        lib.nlc_ilt_rep_inputs.argtypes = [vp, P(IltDesc), vp, vp, i32, i64, i64, i32, vp]
        lib.nlc_ilt_reconstruct.argtypes = [vp, P(IltDesc), vp, vp, vp, i64, i32, vp]
        lib.nlc_ilt_reconstruct_backward.argtypes = [vp, P(IltDesc), vp, vp, vp, vp, i64, i32, vp, vp]
        lib.nlc_model_blob_size.argtypes = [P(ModelDesc)]
        lib.nlc_model_blob_size.restype = i64
        lib.nlc_set_model.argtypes = [vp, P(ModelDesc), vp, i64]
        lib.nlc_gru_encode.argtypes = [vp, vp, i64, i32, vp]
        lib.nlc_model_workspace_bytes.argtypes = [vp, i64]
        lib.nlc_model_workspace_bytes.restype = i64
        lib.nlc_model_forward.argtypes = [vp, vp, vp, vp, i64, i32, vp, vp]
        lib.nlc_model_forward_const_t.argtypes = [vp, vp, vp, dbl, i64, i32, vp, vp]
        lib.nlc_rnn_blob_size.argtypes = [P(RnnDesc)]
        lib.nlc_rnn_blob_size.restype = i64
        lib.nlc_set_rnn_model.argtypes = [vp, P(RnnDesc), vp, i64]
        lib.nlc_rnn_forward.argtypes = [vp, vp, vp, vp, i64, i32, vp, vp]
        lib.nlc_node_blob_size.argtypes = [P(NodeDesc)]
        lib.nlc_node_blob_size.restype = i64
        lib.nlc_set_node_model.argtypes = [vp, P(NodeDesc), vp, i64]
        lib.nlc_node_forward.argtypes = [vp, vp, vp, dbl, i64, vp]
        lib.nlc_env_step.argtypes = [vp, i32, i32, dbl, i32, i64, i32, i32, vp, vp, vp, vp, vp]
        lib.nlc_env_obs.argtypes = [vp, i32, i64, vp, vp]
        lib.nlc_rep_func.argtypes = [vp, vp, i64, vp, vp]
        lib.nlc_set_option.argtypes = [vp, C.c_char_p, dbl]
        lib.nlc_get_stat.argtypes = [vp, C.c_char_p, P(dbl)]
        lib.nlc_mppi_configure.argtypes = [vp, P(MppiDesc)]
        lib.nlc_mppi_workspace_bytes.argtypes = [vp]
        lib.nlc_mppi_workspace_bytes.restype = i64
        lib.nlc_mppi_set_U.argtypes = [vp, vp]
        lib.nlc_mppi_get_U.argtypes = [vp, vp]
        lib.nlc_mppi_rollout.argtypes = [vp, vp, i32, vp, P(MppiBuffers), i32, C.c_uint64, C.c_uint64]
        lib.nlc_mppi_weights.argtypes = [vp, P(MppiBuffers)]
        lib.nlc_mppi_finish.argtypes = [vp, vp, i32, i32, P(MppiBuffers), vp]
        lib.nlc_comm_unique_id.argtypes = [vp]
        lib.nlc_comm_init.argtypes = [vp, i32, i32, vp]
        lib.nlc_comm_destroy.argtypes = [vp]
        lib.nlc_comm_self_test.argtypes = [vp]
        lib.nlc_profile_enable.argtypes = [vp, i32]
        lib.nlc_profile_reset.argtypes = [vp]
        lib.nlc_profile_count.argtypes = [vp]
        lib.nlc_profile_read.argtypes = [vp, i32, C.c_char_p, i32, P(dbl), P(i64)]
        for name in SYMBOLS:
            getattr(lib, name)  # AttributeError here = header/library mismatch
        _lib = lib
        return lib


class NlcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libnlc_hip: {ERRORS.get(code, code)}: {msg}")
        self.code = code


# Options every ctx created from now on starts with (``nlc_set_option`` names; a planner's own ``planner_options`` are applied
# after them and win).  Set programmatically -- ``neurallaplacecontrol_amd.set_default_options({"gru_gemm": 1})`` turns the
# int8-sliced encoder on for every planner and model of the process; no environment variable is read.  The GPU suite's
# ``--nlc-planner-opt`` and its ``encoder_mode`` fixture go through here (tests/conftest.py).
_DEFAULT_OPTIONS = {}


def set_default_options(options=None):
    """Replace the process-wide default options of new ctxs; returns the previous ones.  ``None`` / ``{}`` clears them."""
    old = dict(_DEFAULT_OPTIONS)
    _DEFAULT_OPTIONS.clear()
    for k, v in (options or {}).items():
        _DEFAULT_OPTIONS[str(k)] = float(v)
    return old


class Ctx:
    """One ``nlc_ctx`` (= one process x one GPU).  Not thread-safe."""

    def __init__(self, device_index=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.nlc_create(int(device_index), C.byref(h))
        if rc != 0:
            raise NlcError(rc, (self.lib.nlc_last_error(None) or b"").decode())
        self.h = h
        self.device_index = int(device_index)
        for name, value in _DEFAULT_OPTIONS.items():
            self.set_option(name, value)

    def check(self, rc):
        if rc != 0:
            raise NlcError(rc, (self.lib.nlc_last_error(self.h) or b"").decode())

    def use_torch_stream(self):
        """Enqueue on torch's current stream for this device so torch ops and NLC kernels stay ordered."""
        import torch

        s = torch.cuda.current_stream(self.device_index).cuda_stream
        self.check(self.lib.nlc_set_stream(self.h, C.c_void_p(s)))

    def set_option(self, name, value):
        """Planner tuning knob of ``include/nlc.h`` (``nlc_set_option``)."""
        self.check(self.lib.nlc_set_option(self.h, name.encode(), float(value)))

    def get_stat(self, name):
        """Read-only planner counter of ``include/nlc.h`` (``nlc_get_stat``), as a float."""
        v = C.c_double()
        self.check(self.lib.nlc_get_stat(self.h, name.encode(), C.byref(v)))
        return v.value

    def comm_unique_id(self):
        """Rank 0: the NLC_COMM_ID_BYTES bytes every rank passes to ``comm_init`` (``nlc_comm_unique_id``)."""
        buf = C.create_string_buffer(COMM_ID_BYTES)
        rc = self.lib.nlc_comm_unique_id(buf)
        if rc != 0:
            raise NlcError(rc, (self.lib.nlc_last_error(None) or b"").decode())
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        """The library's own RCCL communicator over the ranks of a K-sharded planner (``nlc_comm_init``)."""
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {COMM_ID_BYTES} bytes")
        self.check(self.lib.nlc_comm_init(self.h, int(rank), int(world), C.c_char_p(bytes(unique_id))))

    def comm_self_test(self):
        """Collective check of the library's communicator (``nlc_comm_self_test``); raises NlcError on failure."""
        self.check(self.lib.nlc_comm_self_test(self.h))

    def comm_destroy(self):
        self.check(self.lib.nlc_comm_destroy(self.h))

    def device_info(self):
        name = C.create_string_buffer(128)
        cus, mhz, gib = C.c_int(), C.c_int(), C.c_double()
        self.check(self.lib.nlc_device_info(self.h, name, 128, C.byref(cus), C.byref(mhz), C.byref(gib)))
        return dict(name=name.value.decode(), num_cus=cus.value, clock_mhz=mhz.value, hbm_gib=gib.value)

    def profile(self, on=True):
        self.check(self.lib.nlc_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self.check(self.lib.nlc_profile_reset(self.h))

    def profile_read(self):
        out = {}
        for i in range(self.lib.nlc_profile_count(self.h)):
            name = C.create_string_buffer(64)
            ms, n = C.c_double(), C.c_int64()
            self.check(self.lib.nlc_profile_read(self.h, i, name, 64, C.byref(ms), C.byref(n)))
            out[name.value.decode()] = dict(total_ms=ms.value, launches=n.value)
        return out

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.nlc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def new_ctx(device=None):
    """A fresh ctx on `device` (torch.device / index / None = torch's current CUDA device)."""
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("neurallaplacecontrol_amd needs an AMD MI355X (no HIP device visible); there is no CPU path")
    if device is None:
        idx = torch.cuda.current_device()
    elif isinstance(device, int):
        idx = device
    else:
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return Ctx(idx)


def ptr(t):
    """Device/host pointer of a contiguous float64 tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    import torch

    assert t.dtype == torch.float64 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return C.c_void_p(t.data_ptr())


def ilt_desc(algo, terms, options=None):
    """Resolve the torchlaplace defaults recalled in SURVEY §A.3 (parity unpinned vs upstream)."""
    if algo == "cme":
        raise NotImplementedError(
            "ilt_algorithm='cme': the method's node / weight parameter sets (one per order; the reference only carries "
            "the LIST of orders, config.py:278-418) ship inside torchlaplace, which is absent here -- they are the result "
            "of a numerical optimisation and cannot be restated offline.  fourier, dehoog, fixed_tablot and stehfest run."
        )
    if algo not in ILT_ALGOS:
        raise NotImplementedError(
            f"ilt_algorithm={algo!r}: fourier, dehoog, fixed_tablot and stehfest are implemented on the HIP path"
        )
    o = {"fourier": dict(alpha=1.0e-3, scale=2.0), "dehoog": dict(alpha=1.0e-10, scale=2.0),
         "fixed_tablot": dict(alpha=1.0, scale=1.0), "stehfest": dict(alpha=1.0, scale=1.0)}[algo]
    if options:
        o.update(options)
    tol = o.get("tol")
    if tol is None:
        tol = 10.0 * o["alpha"]
    return IltDesc(ILT_ALGOS[algo], int(terms), float(o["alpha"]), float(tol), float(o["scale"]))
