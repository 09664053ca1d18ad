"""ctypes binding of libgpflowslim_hip.so (C ABI: include/gpflowslim_hip.h).

This is the only place where the Python mirror of the gpflowSlim API touches the device.
There is deliberately no CPU fallback: if the HIP library cannot be loaded, or no GPU is
visible, every compute entry point raises.
"""
import ctypes
import os
import threading

import numpy as np

GPS_MAX_DIMS = 32
GPS_MAX_NODES = 64
GPS_MAX_STACK = 4

# enum gps_kern_op
K_RBF, K_MATERN12, K_MATERN32, K_MATERN52, K_PERIODIC, K_WHITE, K_CONSTANT, K_EXPONENTIAL = 1, 2, 3, 4, 5, 6, 7, 8
K_SQDIST, K_EUCLID = 9, 10
K_ADD, K_MUL = 16, 17
K_NKN_LINROW, K_NKN_PRODUCT, K_NKN_ACT = 32, 33, 34

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int_p = ctypes.POINTER(ctypes.c_int)
_i64 = ctypes.c_int64


class KernNode(ctypes.Structure):
    """gps_kern_node_t"""
    _fields_ = [("op", ctypes.c_int32),
                ("n_dims", ctypes.c_int32),
                ("active_dims", ctypes.c_int32 * GPS_MAX_DIMS),
                ("variance", ctypes.c_double),
                ("period", ctypes.c_double),
                ("lengthscales", ctypes.c_double * GPS_MAX_DIMS)]


def lib_path():
    env = os.environ.get("GPFLOWSLIM_HIP_LIB")
    if env:
        return env
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.join(os.path.dirname(here), "lib", "libgpflowslim_hip.so")


_SIGNATURES = {
    "gps_create": [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)],
    "gps_destroy": [ctypes.c_void_p],
    "gps_release_buffers": [ctypes.c_void_p],
    "gps_device_info": [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, _c_int_p, ctypes.POINTER(_i64),
                        ctypes.c_char_p, ctypes.c_int],
    "gps_kmat": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, _c_double_p, _i64, _c_double_p, _i64,
                 _i64, ctypes.c_double, _c_double_p],
    "gps_potrf": [ctypes.c_void_p, _c_double_p, _i64, _c_double_p, _c_int_p],
    "gps_trsm_lower": [ctypes.c_void_p, _c_double_p, _i64, _c_double_p, _i64, ctypes.c_int],
    "gps_gpr_set_data": [ctypes.c_void_p, _c_double_p, _i64, _i64],
    "gps_gpr_lml": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, ctypes.c_double, _c_double_p, _i64,
                    _c_double_p, _c_int_p],
    "gps_gpr_lml_grad": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, ctypes.c_double, _c_double_p, _i64,
                         _c_double_p, _c_double_p, ctypes.c_int, _c_int_p, _c_double_p, _c_double_p, _c_int_p],
    "gps_gpr_predict": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, ctypes.c_double, _c_double_p,
                        _i64, _c_double_p, _i64, ctypes.c_int, ctypes.c_int, _c_double_p, _c_double_p, _c_int_p],
    "gps_conditional": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, _c_double_p, _i64, _i64,
                        ctypes.c_double, _c_double_p, _i64, _c_double_p, _i64, _c_double_p, ctypes.c_int,
                        ctypes.c_int, ctypes.c_int, _c_double_p, _c_double_p, _c_int_p],
    "gps_base_conditional": [ctypes.c_void_p, _c_double_p, _c_double_p, _c_double_p, _i64, _i64, _c_double_p,
                             _i64, _c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_double_p,
                             _c_double_p, _c_int_p],
    "gps_svgp_elbo": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, _c_double_p, _i64, _i64, ctypes.c_double,
                      _c_double_p, _i64, _c_double_p, _c_double_p, _i64, _c_double_p, ctypes.c_int, ctypes.c_int,
                      ctypes.c_double, ctypes.c_double, _c_double_p, _c_double_p, _c_double_p, _c_int_p],
    "gps_svgp_elbo_grad": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, _c_double_p, _i64, _i64, ctypes.c_double,
                           _c_double_p, _i64, _c_double_p, _c_double_p, _i64, _c_double_p, ctypes.c_int, ctypes.c_int,
                           ctypes.c_double, ctypes.c_double, _c_double_p, _c_double_p, ctypes.c_int, _c_int_p, _c_double_p,
                           _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_int_p],
    "gps_kmat_vjp": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, _c_double_p, _i64, _c_double_p, _i64, _i64,
                     _c_double_p, _c_double_p, ctypes.c_int, _c_int_p],
    "gps_kmat_input_vjp": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, _c_double_p, _i64, _c_double_p, _i64, _i64,
                           _c_double_p, _c_double_p],
    "gps_gauss_kl": [ctypes.c_void_p, _c_double_p, _i64, _c_double_p, _i64, _c_double_p, ctypes.c_int, _c_double_p,
                     _c_int_p],
    "gps_sgpr": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, _c_double_p, _i64, _c_double_p, _i64, _i64,
                 ctypes.c_double, ctypes.c_double, _c_double_p, _i64, _c_double_p, _i64, ctypes.c_int, _c_double_p,
                 _c_double_p, _c_double_p, _c_int_p],
    "gps_sgpr_grad": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, _c_double_p, _i64, _c_double_p, _i64, _i64,
                      ctypes.c_double, ctypes.c_double, _c_double_p, _i64, _c_double_p, _c_double_p, ctypes.c_int, _c_int_p,
                      _c_double_p, _c_double_p, _c_double_p, _c_int_p],
    "gps_fitc_grad": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, _c_double_p, _i64, _c_double_p, _i64, _i64,
                      ctypes.c_double, ctypes.c_double, _c_double_p, _i64, _c_double_p, _c_double_p, ctypes.c_int, _c_int_p,
                      _c_double_p, _c_double_p, _c_double_p, _c_int_p],
    "gps_fitc": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, _c_double_p, _i64, _c_double_p, _i64, _i64,
                 ctypes.c_double, ctypes.c_double, _c_double_p, _i64, _c_double_p, _i64, ctypes.c_int, _c_double_p,
                 _c_double_p, _c_double_p, _c_int_p],
    "gps_sparse_last_terms": [ctypes.c_void_p, _c_double_p],
    "gps_profile_enable": [ctypes.c_void_p, ctypes.c_int],
    "gps_profile_reset": [ctypes.c_void_p],
    "gps_profile_get": [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(_i64), _c_double_p, _c_double_p,
                        _c_double_p],
    "gps_last_stage_ms": [ctypes.c_void_p, _c_double_p],
    "gps_set_option": [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double],
    "gps_set_stream": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int],
    "gps_dist_begin": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, ctypes.c_double, _c_double_p, _i64,
                       ctypes.c_int, ctypes.c_int, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64)],
    "gps_dist_msg_doubles": [ctypes.c_void_p, _i64, ctypes.POINTER(_i64)],
    "gps_dist_set_comm": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    "gps_dist_set_comm_bufs": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int],
    "gps_dist_comm_bufs_needed": [ctypes.c_void_p, _c_int_p],
    "gps_device_bytes": [ctypes.c_void_p, ctypes.POINTER(_i64)],
    "gps_dist_solve_begin": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, _c_double_p, _i64],
    "gps_dist_solve_pack": [ctypes.c_void_p, _i64, ctypes.c_int],
    "gps_dist_solve_apply": [ctypes.c_void_p, _i64, ctypes.c_int],
    "gps_dist_solve_finish": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, _c_double_p, _c_double_p],
    "gps_dist_lml": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, ctypes.c_double, _c_double_p, _i64, _i64, ctypes.c_int,
                     ctypes.c_int, _c_double_p, _c_int_p],
    "gps_dist_predict": [ctypes.c_void_p, ctypes.POINTER(KernNode), ctypes.c_int, _c_double_p, _i64, ctypes.c_int, _c_double_p, _c_double_p],
    "gps_dist_panel_factor": [ctypes.c_void_p, _i64, ctypes.c_int],
    "gps_dist_unpack": [ctypes.c_void_p, _i64, ctypes.c_int],
    "gps_dist_update": [ctypes.c_void_p, _i64, _i64, _i64, ctypes.c_int],
    "gps_dist_set_bulk_stream": [ctypes.c_void_p, ctypes.c_void_p],
    "gps_dist_finish": [ctypes.c_void_p, _c_double_p, _c_int_p],
    "gps_comm_load": [ctypes.c_char_p],
    "gps_comm_version": [_c_int_p],
    "gps_comm_unique_id": [ctypes.c_void_p, ctypes.c_int],
    "gps_comm_init": [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int],
    "gps_comm_destroy": [ctypes.c_void_p],
    "gps_comm_abort": [ctypes.c_void_p],
    "gps_comm_exchange": [ctypes.c_void_p, ctypes.c_void_p, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int],
    "gps_comm_wait": [ctypes.c_void_p, ctypes.c_int],
    "gps_comm_allreduce": [ctypes.c_void_p, ctypes.c_void_p, _i64],
    "gps_comm_install_allreduce": [ctypes.c_void_p, ctypes.c_void_p, _i64],
    "gps_set_allreduce": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _i64],
    "gps_allreduce_doubles": [_i64, _i64, ctypes.POINTER(_i64)],
    "gps_diag_potrf_base_stamps": [ctypes.c_void_p, ctypes.c_int, _c_double_p],
    "gps_diag_mfma_f64": [ctypes.c_void_p, ctypes.c_int, _c_double_p, _c_int_p],
    "gps_diag_gemm_nt": [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _i64, _i64, _i64, _c_double_p, _c_double_p,
                         _c_double_p],
    "gps_diag_gemm_nt_batched": [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _i64, _i64, _i64, _i64, _c_double_p, _c_double_p,
                                 _c_double_p],
    "gps_diag_trsm512": [ctypes.c_void_p, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_double_p, _c_double_p],
    "gps_diag_trsm512_stamps": [ctypes.c_void_p, _i64, ctypes.c_int, ctypes.c_int, _c_double_p, ctypes.POINTER(ctypes.c_longlong), _i64],
    "gps_diag_trsm_leaf": [ctypes.c_void_p, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_double_p, _c_double_p],
    "gps_diag_set_cu_mask": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int],
    "gps_diag_gemm_timeline": [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _i64, _i64, _i64, ctypes.c_int,
                               ctypes.POINTER(ctypes.c_longlong), _i64, ctypes.POINTER(ctypes.c_int64), _c_double_p],
}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["gps_last_error", "gps_comm_load_error"])
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, _i64)      # gps_allreduce_fn

_lib = None
_lib_lock = threading.Lock()


def _share_hip_runtime_with_torch():
    """One HIP/HSA runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same
    SONAME as /opt/rocm's).  If this library pulled in /opt/rocm's copy first and torch then loaded its
    bundled one, the process would hold two HSA runtimes and the second would see no GPU.  So when a
    torch wheel with a bundled runtime is installed, load that copy first (without importing torch);
    our NEEDED libamdhip64.so.7 then resolves to it by SONAME, whichever of the two is imported first."""
    if os.environ.get("GPFLOWSLIM_NO_TORCH_RUNTIME"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass


def load_library():
    """Load the C-ABI library (no GPU needed for this step).  Raises if it is missing."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                "gpflowSlim (MI355X): HIP library not found at %s -- build it with "
                "`make -C gpflow-slim_amd/csrc` (or __graft_entry__.build()). There is no CPU fallback." % path)
        _share_hip_runtime_with_torch()
        lib = ctypes.CDLL(path)
        for name, argtypes in _SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                if os.environ.get("GPFLOWSLIM_HIP_LIB"):      # (an older build named for an A/B: entry points added since are simply absent)
                    continue
                raise
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        lib.gps_last_error.argtypes = [ctypes.c_void_p]
        lib.gps_last_error.restype = ctypes.c_char_p
        lib.gps_comm_load_error.argtypes = []
        lib.gps_comm_load_error.restype = ctypes.c_char_p
        _lib = lib
        return lib


def comm_load(path=None):
    """Open librccl for the native collectives.  Default: $GPFLOWSLIM_RCCL_LIB if set, else the copy a PyTorch wheel bundles, if
    there is one (a process that also imports torch then holds ONE RCCL), else the system's."""
    lib = load_library()
    if path is None:
        path = os.environ.get("GPFLOWSLIM_RCCL_LIB") or None
    if path is None:
        try:
            import importlib.util
            spec = importlib.util.find_spec("torch")
            if spec is not None and spec.submodule_search_locations:
                cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
                if os.path.exists(cand):
                    path = cand
        except Exception:
            path = None
    rc = lib.gps_comm_load(path.encode() if path else None)
    if rc != 0:
        msg = lib.gps_comm_load_error()
        raise RuntimeError("librccl could not be loaded (%d): %s" % (rc, msg.decode() if msg else ""))


def comm_unique_id():
    """128 opaque bytes from ncclGetUniqueId: rank 0 draws them, every rank passes them to Handle.comm_init."""
    lib = load_library()
    comm_load()
    buf = ctypes.create_string_buffer(128)
    rc = lib.gps_comm_unique_id(buf, 128)
    if rc != 0:
        raise RuntimeError("gps_comm_unique_id failed (%d)" % rc)
    return buf.raw


def comm_version():
    lib = load_library()
    comm_load()
    v = ctypes.c_int(0)
    if lib.gps_comm_version(ctypes.byref(v)) != 0:
        raise RuntimeError("gps_comm_version failed")
    return v.value


class NotPositiveDefiniteError(ValueError):
    """Analogue of TensorFlow's InvalidArgumentError from tf.cholesky (models/gpr.py:70)."""


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(_c_double_p)


def make_program(nodes):
    if len(nodes) == 0 or len(nodes) > GPS_MAX_NODES:
        raise ValueError("kernel program must have 1..%d nodes, got %d" % (GPS_MAX_NODES, len(nodes)))
    arr = (KernNode * len(nodes))()
    for i, nd in enumerate(nodes):
        arr[i] = nd
    return arr


def primitive_node(op, variance, dims=(), lengthscales=(), period=0.0):
    nd = KernNode()
    nd.op = op
    nd.variance = float(variance)
    nd.period = float(period)
    k = len(dims)
    if k > GPS_MAX_DIMS:
        raise ValueError("a primitive kernel supports at most %d active dims" % GPS_MAX_DIMS)
    nd.n_dims = k
    if k:
        nd.active_dims[:k] = [int(d) for d in dims]      # (slice assignment: one call instead of one per element -- this runs every step)
    ls = np.asarray(lengthscales, dtype=np.float64).ravel()
    if ls.size == 1 and k > 1:
        nd.lengthscales[:k] = [float(ls[0])] * k
    elif ls.size:
        m = min(ls.size, GPS_MAX_DIMS)
        nd.lengthscales[:m] = ls[:m].tolist()
    return nd


def op_node(op):
    nd = KernNode()
    nd.op = op
    return nd


class Handle(object):
    """One GPU, one stream, one set of resident buffers (gps_handle_t)."""

    def __init__(self, device=None):
        lib = load_library()
        if device is None:
            device = int(os.environ.get("GPFLOWSLIM_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        h = ctypes.c_void_p()
        rc = lib.gps_create(int(device), ctypes.byref(h))
        if rc != 0 or not h:
            raise RuntimeError("gpflowSlim (MI355X): gps_create(device=%d) failed with %d -- no usable GPU; "
                               "there is no CPU fallback" % (device, rc))
        self._lib = lib
        self._h = h
        self.device = int(device)
        self.resident_token = None     # identity of the data set held by gps_gpr_set_data
        self.dist_state = None         # what a PARTITIONED factor on this handle was computed from (gpflowSlim.distributed)
        self.factor_key = None         # what the resident Cholesky factor / alpha were computed from (models/gpr.py)

    @property
    def factor_key(self):
        return self._factor_key

    @factor_key.setter
    def factor_key(self, key):
        # The claims "a factor is resident" live on the HANDLE, next to the factor: every evaluation that touches the
        # factor buffers passes through this setter (models/gpr.py, every sparse / conditional entry below), and whatever
        # partitioned factor a distributed evaluation left behind is gone with it -- gpr_lml_distributed records its claim
        # (dist_state) AFTER resetting the key.
        self._factor_key = key
        self.dist_state = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gps_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.gps_last_error(self._h)
            raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    # ---- info / measurement
    def device_info(self):
        name = ctypes.create_string_buffer(256)
        arch = ctypes.create_string_buffer(256)
        ncu = ctypes.c_int(0)
        hbm = _i64(0)
        self._check(self._lib.gps_device_info(self._h, name, 256, ctypes.byref(ncu), ctypes.byref(hbm), arch, 256),
                    "gps_device_info")
        return {"name": name.value.decode(), "arch": arch.value.decode(), "n_cu": ncu.value, "hbm_bytes": hbm.value}

    def profile_enable(self, on):
        self._check(self._lib.gps_profile_enable(self._h, 1 if on else 0), "gps_profile_enable")

    def profile_reset(self):
        self._check(self._lib.gps_profile_reset(self._h), "gps_profile_reset")

    def profile_get(self, klass):
        n = _i64(0)
        ms, fl, by = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        self._check(self._lib.gps_profile_get(self._h, klass.encode(), ctypes.byref(n), ctypes.byref(ms),
                                              ctypes.byref(fl), ctypes.byref(by)), "gps_profile_get")
        return {"launches": n.value, "ms": ms.value, "flops": fl.value, "bytes": by.value}

    def last_stage_ms(self):
        out = np.zeros(5)
        self._check(self._lib.gps_last_stage_ms(self._h, _ptr(out)), "gps_last_stage_ms")
        return dict(zip(["kmat", "potrf", "trsv", "predict", "total"], out.tolist()))

    def release_buffers(self):
        """Give the device buffers back (the resident data set and factor are dropped with them)."""
        self._check(self._lib.gps_release_buffers(self._h), "gps_release_buffers")
        self.resident_token = None
        self.factor_key = None
        self.resident_shape = None

    def set_option(self, key, value):
        self._check(self._lib.gps_set_option(self._h, key.encode(), float(value)), "gps_set_option")

    def set_stream(self, hip_stream, external=True):
        """Run the library's kernels on the caller's HIP stream (integer handle, 0 = legacy default
        stream); external=False restores the handle's own stream."""
        self._check(self._lib.gps_set_stream(self._h, ctypes.c_void_p(hip_stream or 0), 1 if external else 0),
                    "gps_set_stream")

    # ---- block-column distributed factorisation (driven by gpflowSlim.distributed)
    def dist_begin(self, prog, noise_var, resid, nparts, part, nb):
        resid = _f64(resid)
        npan, mx = _i64(0), _i64(0)
        self._check(self._lib.gps_dist_begin(self._h, prog, len(prog), float(noise_var), _ptr(resid), resid.shape[1],
                                             int(nparts), int(part), int(nb), ctypes.byref(npan), ctypes.byref(mx)),
                    "gps_dist_begin")
        return npan.value, mx.value

    def dist_msg_doubles(self, j):
        out = _i64(0)
        self._check(self._lib.gps_dist_msg_doubles(self._h, int(j), ctypes.byref(out)), "gps_dist_msg_doubles")
        return out.value

    def dist_set_comm(self, ptr0, ptr1):
        self._check(self._lib.gps_dist_set_comm(self._h, ctypes.c_void_p(ptr0), ctypes.c_void_p(ptr1)), "gps_dist_set_comm")

    def dist_set_comm_bufs(self, ptrs):
        arr = (ctypes.c_void_p * len(ptrs))(*[ctypes.c_void_p(p) for p in ptrs])
        self._check(self._lib.gps_dist_set_comm_bufs(self._h, arr, len(ptrs)), "gps_dist_set_comm_bufs")

    def dist_comm_bufs_needed(self):
        out = ctypes.c_int(0)
        self._check(self._lib.gps_dist_comm_bufs_needed(self._h, ctypes.byref(out)), "gps_dist_comm_bufs_needed")
        return out.value

    def device_bytes(self):
        out = _i64(0)
        self._check(self._lib.gps_device_bytes(self._h, ctypes.byref(out)), "gps_device_bytes")
        return out.value

    def dist_solve_begin(self, prog, Xnew):
        Xnew = _f64(Xnew)
        self._check(self._lib.gps_dist_solve_begin(self._h, prog, len(prog), _ptr(Xnew), Xnew.shape[0]), "gps_dist_solve_begin")

    def dist_solve_pack(self, j, buf):
        self._check(self._lib.gps_dist_solve_pack(self._h, int(j), int(buf)), "gps_dist_solve_pack")

    def dist_solve_apply(self, j, buf):
        self._check(self._lib.gps_dist_solve_apply(self._h, int(j), int(buf)), "gps_dist_solve_apply")

    def dist_solve_finish(self, prog, n_new, r):
        mean = np.empty((n_new, max(r, 1)))
        var = np.empty(n_new)
        self._check(self._lib.gps_dist_solve_finish(self._h, prog, len(prog), _ptr(mean), _ptr(var)), "gps_dist_solve_finish")
        return mean[:, :r], var

    def dist_lml(self, prog, noise_var, resid, nb, lookahead, exchange_mode):
        """gps_dist_lml: the whole block-column factorisation driven inside the library (native communicator required)."""
        resid = _f64(resid)
        lml = ctypes.c_double(0)
        info = ctypes.c_int(0)
        self.resident_token_factor = None
        self._check(self._lib.gps_dist_lml(self._h, prog, len(prog), float(noise_var), _ptr(resid), resid.shape[1], int(nb), int(lookahead),
                                           int(exchange_mode), ctypes.byref(lml), ctypes.byref(info)), "gps_dist_lml")
        if info.value > 0:
            raise NotPositiveDefiniteError(
                "Cholesky decomposition was not successful: leading minor of order %d is not positive definite" % info.value)
        return lml.value

    def dist_predict(self, prog, Xnew, r, exchange_mode):
        Xnew = _f64(Xnew)
        n_new = Xnew.shape[0]
        mean = np.empty((n_new, max(r, 1)))
        var = np.empty(n_new)
        self._check(self._lib.gps_dist_predict(self._h, prog, len(prog), _ptr(Xnew) if n_new else None, n_new, int(exchange_mode),
                                               _ptr(mean) if n_new else None, _ptr(var) if n_new else None), "gps_dist_predict")
        return mean[:, :r], var

    def dist_panel_factor(self, j, buf):
        self._check(self._lib.gps_dist_panel_factor(self._h, int(j), int(buf)), "gps_dist_panel_factor")

    def dist_unpack(self, j, buf):
        self._check(self._lib.gps_dist_unpack(self._h, int(j), int(buf)), "gps_dist_unpack")

    def dist_update(self, j, c_lo, c_hi, lane=0):
        self._check(self._lib.gps_dist_update(self._h, int(j), int(c_lo), int(c_hi), int(lane)), "gps_dist_update")

    def dist_set_bulk_stream(self, hip_stream):
        self._check(self._lib.gps_dist_set_bulk_stream(self._h, ctypes.c_void_p(hip_stream) if hip_stream else None),
                    "gps_dist_set_bulk_stream")

    def dist_finish(self):
        lml = ctypes.c_double(0)
        info = ctypes.c_int(0)
        self._check(self._lib.gps_dist_finish(self._h, ctypes.byref(lml), ctypes.byref(info)), "gps_dist_finish")
        if info.value > 0:
            raise NotPositiveDefiniteError(
                "Cholesky decomposition was not successful: leading minor of order %d is not positive definite"
                % info.value)
        return lml.value

    def diag_mfma_f64(self, waves_per_simd=1):
        tf = ctypes.c_double(0)
        ok = ctypes.c_int(0)
        self._check(self._lib.gps_diag_mfma_f64(self._h, waves_per_simd, ctypes.byref(tf), ctypes.byref(ok)),
                    "gps_diag_mfma_f64")
        return tf.value, bool(ok.value)

    def diag_gemm_nt(self, op, lower, A, B, C):
        A, B = _f64(A), _f64(B)
        C = np.array(C, dtype=np.float64, order="C", copy=True)
        m, k = A.shape
        n = B.shape[0]
        assert B.shape[1] == k and C.shape == (m, n)
        self._check(self._lib.gps_diag_gemm_nt(self._h, op, int(lower), m, n, k, _ptr(A), _ptr(B), _ptr(C)),
                    "gps_diag_gemm_nt")
        return C

    def diag_gemm_nt_batched(self, op, tri, A, B, C):
        """A [batch, m, k], B [batch, n, k], C [batch, m, n]; tri: 0 none, 1 A upper, 2 A lower, 3 B lower triangular."""
        A, B = _f64(A), _f64(B)
        C = np.array(C, dtype=np.float64, order="C", copy=True)
        batch, m, k = A.shape
        n = B.shape[1]
        assert B.shape == (batch, n, k) and C.shape == (batch, m, n)
        self._check(self._lib.gps_diag_gemm_nt_batched(self._h, op, int(tri), batch, m, n, k, _ptr(A), _ptr(B), _ptr(C)),
                    "gps_diag_gemm_nt_batched")
        return C

    def diag_trsm_leaf(self, m, mode, upper=False, reps=50):
        """(microseconds per launch, scaled residual) of one 128-column solve leaf on m rows."""
        us, res = ctypes.c_double(0), ctypes.c_double(0)
        self._check(self._lib.gps_diag_trsm_leaf(self._h, int(m), int(mode), int(bool(upper)), int(reps),
                                                 ctypes.byref(us), ctypes.byref(res)), "gps_diag_trsm_leaf")
        return us.value, res.value

    def diag_trsm512(self, m, backward=False, panel=True, reps=20):
        """(microseconds per 512-column solve of m rows, max |one launch - launch by launch|)."""
        us, diff = ctypes.c_double(0), ctypes.c_double(0)
        self._check(self._lib.gps_diag_trsm512(self._h, int(m), int(bool(backward)), int(bool(panel)), int(reps),
                                               ctypes.byref(us), ctypes.byref(diff)), "gps_diag_trsm512")
        return us.value, diff.value

    def diag_trsm512_stamps(self, m, backward=False, reps=5):
        """(microseconds per solve, int64 array [m / 64, 32] of phase stamps) of the one-launch 512-column solve."""
        nb = int(m) // 32                # (64 or 32 rows per workgroup: the unused tail stays 0)
        buf = np.zeros((nb, 32), dtype=np.int64)
        us = ctypes.c_double(0)
        self._check(self._lib.gps_diag_trsm512_stamps(self._h, int(m), int(bool(backward)), int(reps), ctypes.byref(us),
                                                      buf.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), nb), "gps_diag_trsm512_stamps")
        return us.value, buf

    def diag_set_cu_mask(self, words):
        arr = (ctypes.c_uint32 * len(words))(*[int(w) & 0xffffffff for w in words])
        self._check(self._lib.gps_diag_set_cu_mask(self._h, arr, len(words)), "gps_diag_set_cu_mask")

    def diag_gemm_timeline(self, op, lower, m, n, k, reps=5, cap_blocks=1 << 17):
        """(ms per launch, stamps [nblocks, 6] = start, end (100 MHz ticks), HW_ID, XCC_ID, K-loop start, K-loop end) on random device data."""
        st = np.zeros((cap_blocks, 6), dtype=np.int64)
        nb, ms = ctypes.c_int64(), ctypes.c_double()
        self._check(self._lib.gps_diag_gemm_timeline(self._h, op, int(lower), m, n, k, reps,
                                                     st.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), cap_blocks,
                                                     ctypes.byref(nb), ctypes.byref(ms)), "gps_diag_gemm_timeline")
        return ms.value, st[:nb.value]

    # ---- kernels.K
    def kmat(self, prog, X, X2=None, diag_add=0.0):
        X = _f64(X)
        n, d = X.shape
        if X2 is None:
            out = np.empty((n, n))
            self._check(self._lib.gps_kmat(self._h, prog, len(prog), _ptr(X), n, None, 0, d, float(diag_add),
                                           _ptr(out)), "gps_kmat")
        else:
            X2 = _f64(X2)
            if X2.shape[1] != d:
                raise ValueError("X and X2 must have the same number of columns")
            m = X2.shape[0]
            out = np.empty((n, m))
            self._check(self._lib.gps_kmat(self._h, prog, len(prog), _ptr(X), n, _ptr(X2), m, d, float(diag_add),
                                           _ptr(out)), "gps_kmat")
        return out

    # ---- linear algebra on host matrices
    def potrf(self, A):
        A = _f64(A)
        n = A.shape[0]
        if A.shape != (n, n):
            raise ValueError("cholesky needs a square matrix")
        L = np.empty_like(A)
        info = ctypes.c_int(0)
        self._check(self._lib.gps_potrf(self._h, _ptr(A), n, _ptr(L), ctypes.byref(info)), "gps_potrf")
        if info.value > 0:
            raise NotPositiveDefiniteError(
                "Cholesky decomposition was not successful: leading minor of order %d is not positive definite"
                % info.value)
        return L

    def trsm_lower(self, L, B, trans=False):
        L = _f64(L)
        B = np.array(B, dtype=np.float64, order="C", copy=True)
        squeeze = B.ndim == 1
        if squeeze:
            B = B[:, None].copy()
        n = L.shape[0]
        if L.shape != (n, n) or B.shape[0] != n:
            raise ValueError("triangular solve: shape mismatch")
        self._check(self._lib.gps_trsm_lower(self._h, _ptr(L), n, _ptr(B), B.shape[1], 1 if trans else 0),
                    "gps_trsm_lower")
        return B[:, 0] if squeeze else B

    # ---- GPR
    def gpr_set_data(self, X, token):
        X = _f64(X)
        _need(X.ndim == 2 and X.shape[0] > 0 and X.shape[1] > 0, "GPR needs a non-empty X [N, D]")
        self.resident_shape = X.shape
        self._check(self._lib.gps_gpr_set_data(self._h, _ptr(X), X.shape[0], X.shape[1]), "gps_gpr_set_data")
        self.resident_token = token
        self.factor_key = None

    def gpr_lml(self, prog, noise_var, resid):
        resid = _f64(resid)
        shape = getattr(self, "resident_shape", None)
        _need(shape is not None and resid.ndim == 2 and resid.shape[0] == shape[0], "Y must have one row per row of X")
        lml = ctypes.c_double(0)
        info = ctypes.c_int(0)
        self._check(self._lib.gps_gpr_lml(self._h, prog, len(prog), float(noise_var), _ptr(resid), resid.shape[1],
                                          ctypes.byref(lml), ctypes.byref(info)), "gps_gpr_lml")
        if info.value > 0:
            raise NotPositiveDefiniteError(
                "Cholesky decomposition was not successful: leading minor of order %d is not positive definite"
                % info.value)
        return lml.value

    def gpr_lml_grad(self, prog, noise_var, resid):
        """(lml, grad_slots, grad_noise, K_y^-1 resid)  -- see gps_gpr_lml_grad in the header."""
        resid = _f64(resid)
        shape = getattr(self, "resident_shape", None)
        _need(shape is not None and resid.ndim == 2 and resid.shape[0] == shape[0] and resid.shape[1] > 0,
              "Y must have one row per row of X")
        n, r = shape[0], resid.shape[1]          # the C side reads and writes n = rows of the resident X
        lml = ctypes.c_double(0)
        info = ctypes.c_int(0)
        nslots = ctypes.c_int(0)
        cap = 160
        slots = np.zeros(cap)
        gnoise = ctypes.c_double(0)
        kinv_resid = np.empty((n, r))
        self._check(self._lib.gps_gpr_lml_grad(self._h, prog, len(prog), float(noise_var), _ptr(resid), r,
                                               ctypes.byref(lml), _ptr(slots), cap, ctypes.byref(nslots),
                                               ctypes.byref(gnoise), _ptr(kinv_resid), ctypes.byref(info)),
                    "gps_gpr_lml_grad")
        if info.value > 0:
            raise NotPositiveDefiniteError(
                "Cholesky decomposition was not successful: leading minor of order %d is not positive definite"
                % info.value)
        return lml.value, slots[:nslots.value].copy(), gnoise.value, kinv_resid

    def gpr_predict(self, prog, noise_var, resid, Xnew, full_cov=False, refactor=True):
        resid = _f64(resid)
        Xnew = _f64(Xnew)
        shape = getattr(self, "resident_shape", None)
        _need(Xnew.ndim == 2 and shape is not None and Xnew.shape[1] == shape[1],
              "Xnew must be [N*, %s]" % (shape[1] if shape else "D"))
        _need(resid.ndim == 2 and resid.shape[0] == shape[0], "Y must have one row per row of X")
        n_new = Xnew.shape[0]
        r = resid.shape[1]
        mean = np.empty((n_new, r))
        var = np.empty((n_new, n_new) if full_cov else (n_new,))
        if n_new == 0:                       # nothing to predict: the reference returns empty tensors
            return mean, var
        info = ctypes.c_int(0)
        self._check(self._lib.gps_gpr_predict(self._h, prog, len(prog), float(noise_var), _ptr(resid), r,
                                              _ptr(Xnew), n_new, 1 if full_cov else 0, 1 if refactor else 0,
                                              _ptr(mean), _ptr(var), ctypes.byref(info)), "gps_gpr_predict")
        if info.value > 0:
            raise NotPositiveDefiniteError(
                "Cholesky decomposition was not successful: leading minor of order %d is not positive definite"
                % info.value)
        return mean, var

    # ---- SGPR
    def sgpr(self, prog, Z, X, resid, jitter, noise_var, Xnew=None, full_cov=False, want_bound=True, fitc=False):
        """gps_sgpr (Titsias bound) or, with fitc=True, gps_fitc (Snelson & Ghahramani)."""
        Z, X, resid = _f64(Z), _f64(X), _f64(resid)
        m, d = Z.shape
        n, r = resid.shape
        _need(X.ndim == 2 and X.shape == (n, d) and m > 0 and n > 0, "sparse GP regression needs Z [M, D], X [N, D], Y [N, R]")
        bound = ctypes.c_double(0)
        info = ctypes.c_int(0)
        if Xnew is not None and np.shape(Xnew)[0] == 0:
            _need(np.ndim(Xnew) == 2 and np.shape(Xnew)[1] == d, "Xnew must be [N*, %d]" % d)
            return 0.0, np.empty((0, r)), np.empty((0, 0) if full_cov else (0,))
        if Xnew is not None:
            Xnew = _f64(Xnew)
            _need(Xnew.ndim == 2 and Xnew.shape[1] == d, "Xnew must be [N*, %d]" % d)
            n_new = Xnew.shape[0]
            mean = np.empty((n_new, r))
            var = np.empty((n_new, n_new) if full_cov else (n_new,))
            xp, mp_, vp = _ptr(Xnew), _ptr(mean), _ptr(var)
        else:
            n_new, mean, var, xp, mp_, vp = 0, None, None, None, None, None
        self.resident_token = None
        self.factor_key = None
        fn = self._lib.gps_fitc if fitc else self._lib.gps_sgpr
        self._check(fn(self._h, prog, len(prog), _ptr(Z), m, _ptr(X), n, d, float(jitter),
                       float(noise_var), _ptr(resid), r, xp, n_new, 1 if full_cov else 0,
                       ctypes.byref(bound) if want_bound else None, mp_, vp, ctypes.byref(info)),
                    "gps_fitc" if fitc else "gps_sgpr")
        if info.value > 0:
            raise NotPositiveDefiniteError("Cholesky decomposition was not successful (order %d)" % info.value)
        return bound.value, mean, var

    def sgpr_grad(self, prog, Z, X, resid, jitter, noise_var, want_grad_Z=True, fitc=False):
        """(bound, grad_slots, grad_noise, d/d mean(X) [N, R], grad_Z [M, D] or None) of the SGPR bound -- gps_sgpr_grad -- or,
        with fitc=True, of the FITC log-likelihood -- gps_fitc_grad; gradients with respect to the constrained values."""
        Z, X, resid = _f64(Z), _f64(X), _f64(resid)
        m, d = Z.shape
        n, r = resid.shape
        _need(X.ndim == 2 and X.shape == (n, d) and m > 0 and n > 0, "sparse GP regression needs Z [M, D], X [N, D], Y [N, R]")
        bound, gnoise = ctypes.c_double(0), ctypes.c_double(0)
        info, nslots = ctypes.c_int(0), ctypes.c_int(0)
        cap = 700
        slots = np.zeros(cap)
        g_mean = np.zeros((n, r))
        g_Z = np.zeros((m, d))
        self.resident_token = None
        self.factor_key = None
        fn = self._lib.gps_fitc_grad if fitc else self._lib.gps_sgpr_grad
        self._check(fn(self._h, prog, len(prog), _ptr(Z), m, _ptr(X), n, d, float(jitter), float(noise_var),
                       _ptr(resid), r, ctypes.byref(bound), _ptr(slots), cap, ctypes.byref(nslots),
                       ctypes.byref(gnoise), _ptr(g_mean), _ptr(g_Z) if want_grad_Z else None,
                       ctypes.byref(info)), "gps_fitc_grad" if fitc else "gps_sgpr_grad")
        if info.value > 0:
            raise NotPositiveDefiniteError("Cholesky decomposition was not successful (order %d)" % info.value)
        return bound.value, slots[:nslots.value].copy(), gnoise.value, g_mean, (g_Z if want_grad_Z else None)

    # ---- native collectives (csrc/comm_rccl.hip; driven by gpflowSlim.distributed.RcclComm)
    def comm_init(self, rank, world, unique_id):
        buf = ctypes.create_string_buffer(bytes(unique_id), len(unique_id))
        self._check(self._lib.gps_comm_init(self._h, int(rank), int(world), buf, len(unique_id)), "gps_comm_init")

    def comm_destroy(self):
        self._check(self._lib.gps_comm_destroy(self._h), "gps_comm_destroy")

    def comm_abort(self):
        """ncclCommAbort: leave the communicator without waiting for the peers (a failing rank's way out)."""
        self._check(self._lib.gps_comm_abort(self._h), "gps_comm_abort")

    def comm_exchange(self, dev_ptr, count, root, mode, slot):
        self._check(self._lib.gps_comm_exchange(self._h, ctypes.c_void_p(dev_ptr), int(count), int(root), int(mode), int(slot)),
                    "gps_comm_exchange")

    def comm_wait(self, slot):
        self._check(self._lib.gps_comm_wait(self._h, int(slot)), "gps_comm_wait")

    def comm_allreduce(self, dev_ptr, count):
        self._check(self._lib.gps_comm_allreduce(self._h, ctypes.c_void_p(dev_ptr), int(count)), "gps_comm_allreduce")

    def comm_install_allreduce(self, dev_ptr, capacity_doubles):
        self._check(self._lib.gps_comm_install_allreduce(self._h, ctypes.c_void_p(dev_ptr), int(capacity_doubles)),
                    "gps_comm_install_allreduce")

    def set_allreduce(self, callback, dev_ptr, capacity_doubles):
        """gps_set_allreduce: `callback` an ALLREDUCE_FN instance (kept alive by the caller) or None to remove it."""
        cb = ctypes.cast(callback, ctypes.c_void_p) if callback is not None else None
        self._check(self._lib.gps_set_allreduce(self._h, cb, None, ctypes.c_void_p(dev_ptr) if dev_ptr else None,
                                                int(capacity_doubles)), "gps_set_allreduce")

    def allreduce_doubles(self, m, r):
        out = _i64(0)
        self._check(self._lib.gps_allreduce_doubles(int(m), int(r), ctypes.byref(out)), "gps_allreduce_doubles")
        return out.value

    def sparse_last_terms(self):
        """sum log diag LB, tr(A A^T), sum c^2, Kdiag constant, sum log nu of the last sgpr() call."""
        out = np.zeros(5)
        self._check(self._lib.gps_sparse_last_terms(self._h, _ptr(out)), "gps_sparse_last_terms")
        return dict(sum_log_diag_LB=out[0], tr_AAT=out[1], sum_c2=out[2], kdiag=out[3], sum_log_nu=out[4])

    # ---- conditionals
    @staticmethod
    def _prep_q_sqrt(q_sqrt):
        if q_sqrt is None:
            return None, 0
        q_sqrt = np.asarray(q_sqrt, dtype=np.float64)
        if q_sqrt.ndim == 2:
            return _f64(q_sqrt), 2
        if q_sqrt.ndim == 3:
            return _f64(np.transpose(q_sqrt, (2, 0, 1))), 3       # [m,m,k] -> [k,m,m]
        raise ValueError("Bad dimension for q_sqrt: %s" % str(q_sqrt.ndim))

    @staticmethod
    def _check_q_sqrt(q, qnd, m, k):
        if q is None:
            return
        _need(q.shape == ((m, k) if qnd == 2 else (k, m, m)), "q_sqrt must be [M, K] or [M, M, K]")

    def _cond_outputs(self, n_new, k, full_cov):
        fmean = np.empty((n_new, k))
        fvar = np.empty((k, n_new, n_new) if full_cov else (n_new, k))
        return fmean, fvar

    def conditional(self, prog, Z, Xnew, f, jitter, q_sqrt=None, white=False, full_cov=False):
        Z, Xnew, f = _f64(Z), _f64(Xnew), _f64(f)
        m, d = Z.shape
        _need(Xnew.ndim == 2 and Xnew.shape[1] == d, "Xnew must be [N*, %d]" % d)
        _need(f.ndim == 2 and f.shape[0] == m and m > 0, "f must be [M, K] with one row per row of X")
        n_new, k = Xnew.shape[0], f.shape[1]
        q, qnd = self._prep_q_sqrt(q_sqrt)
        self._check_q_sqrt(q, qnd, m, k)
        fmean, fvar = self._cond_outputs(n_new, k, full_cov)
        if n_new == 0:
            return fmean, (np.empty((0, 0, k)) if full_cov else fvar)
        info = ctypes.c_int(0)
        self.resident_token = None
        self.factor_key = None
        self._check(self._lib.gps_conditional(self._h, prog, len(prog), _ptr(Z), m, d, float(jitter), _ptr(Xnew),
                                              n_new, _ptr(f), k, _ptr(q) if q is not None else None, qnd,
                                              1 if white else 0, 1 if full_cov else 0, _ptr(fmean), _ptr(fvar),
                                              ctypes.byref(info)), "gps_conditional")
        if info.value > 0:
            raise NotPositiveDefiniteError("Cholesky decomposition was not successful (order %d)" % info.value)
        if full_cov:
            fvar = np.ascontiguousarray(np.transpose(fvar, (1, 2, 0)))     # [n,n,k]  conditionals.py:119
        return fmean, fvar

    def svgp_elbo(self, prog, Z, X, yres, q_mu, q_sqrt, jitter, noise_var, white=True, scale=1.0):
        """(elbo, kl, sum of variational expectations) of models/svgp.py:108-125 for the Gaussian likelihood."""
        Z, X, yres, q_mu = _f64(Z), _f64(X), _f64(yres), _f64(q_mu)
        _need(Z.ndim == 2 and X.ndim == 2 and Z.shape[1] == X.shape[1], "Z [M, D] and X [N, D] must share D")
        m, d = Z.shape
        n = X.shape[0]
        _need(q_mu.ndim == 2 and q_mu.shape[0] == m, "q_mu must be [M, K]")
        k = q_mu.shape[1]
        _need(yres.shape == (n, k), "Y - mean must be [N, K] with K = number of latent functions")
        _need(m > 0 and n > 0 and k > 0, "empty SVGP problem")
        q, qnd = self._prep_q_sqrt(q_sqrt)
        _need(q is not None, "SVGP needs q_sqrt")
        self._check_q_sqrt(q, qnd, m, k)
        elbo, kl, ve = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        info = ctypes.c_int(0)
        self.resident_token = None
        self.factor_key = None
        self._check(self._lib.gps_svgp_elbo(self._h, prog, len(prog), _ptr(Z), m, d, float(jitter), _ptr(X), n,
                                            _ptr(yres), _ptr(q_mu), k, _ptr(q), qnd, 1 if white else 0,
                                            float(noise_var), float(scale), ctypes.byref(elbo), ctypes.byref(kl),
                                            ctypes.byref(ve), ctypes.byref(info)), "gps_svgp_elbo")
        if info.value > 0:
            raise NotPositiveDefiniteError("Kuu + jitter*I is not positive definite (leading minor of order %d)" % info.value)
        return elbo.value, kl.value, ve.value

    def svgp_elbo_grad(self, prog, Z, X, yres, q_mu, q_sqrt, jitter, noise_var, white=True, scale=1.0, want_grad_Z=False):
        """(elbo, grad_slots, grad_noise, grad_q_mu [M, K], grad_q_sqrt shaped like q_sqrt, d/d mean(X) [N, K]) -- gradients
        w.r.t. the constrained values, either parametrisation (gps_svgp_elbo_grad); want_grad_Z: a seventh item, the gradient
        with respect to the inducing inputs [M, D]."""
        Z, X, yres, q_mu = _f64(Z), _f64(X), _f64(yres), _f64(q_mu)
        _need(Z.ndim == 2 and X.ndim == 2 and Z.shape[1] == X.shape[1], "Z [M, D] and X [N, D] must share D")
        m, d = Z.shape
        n = X.shape[0]
        _need(q_mu.ndim == 2 and q_mu.shape[0] == m, "q_mu must be [M, K]")
        k = q_mu.shape[1]
        _need(yres.shape == (n, k), "Y - mean must be [N, K] with K = number of latent functions")
        _need(m > 0 and n > 0 and k > 0, "empty SVGP problem")
        q, qnd = self._prep_q_sqrt(q_sqrt)
        _need(q is not None, "SVGP needs q_sqrt")
        self._check_q_sqrt(q, qnd, m, k)
        elbo, gnoise = ctypes.c_double(0), ctypes.c_double(0)
        info, nslots = ctypes.c_int(0), ctypes.c_int(0)
        cap = 700
        slots = np.zeros(cap)
        g_qmu = np.zeros((m, k))
        g_q = np.zeros_like(q)
        g_mean = np.zeros((n, k))
        g_Z = np.zeros((m, d))
        self.resident_token = None
        self.factor_key = None
        self._check(self._lib.gps_svgp_elbo_grad(self._h, prog, len(prog), _ptr(Z), m, d, float(jitter), _ptr(X), n, _ptr(yres),
                                                 _ptr(q_mu), k, _ptr(q), qnd, 1 if white else 0, float(noise_var), float(scale),
                                                 ctypes.byref(elbo), _ptr(slots), cap, ctypes.byref(nslots), ctypes.byref(gnoise),
                                                 _ptr(g_qmu), _ptr(g_q), _ptr(g_mean), _ptr(g_Z) if want_grad_Z else None,
                                                 ctypes.byref(info)), "gps_svgp_elbo_grad")
        if info.value > 0:
            raise NotPositiveDefiniteError("Kuu + jitter*I is not positive definite (leading minor of order %d)" % info.value)
        if qnd == 3:
            g_q = np.ascontiguousarray(np.transpose(g_q, (1, 2, 0)))           # [k, m, m] -> [m, m, k]
        if want_grad_Z:
            return elbo.value, slots[:nslots.value].copy(), gnoise.value, g_qmu, g_q, g_mean, g_Z
        return elbo.value, slots[:nslots.value].copy(), gnoise.value, g_qmu, g_q, g_mean

    def kmat_input_vjp(self, prog, X, W, X2=None):
        """d/dX sum_ij W[i, j] k(X_i, X2_j)  [N, D]: reverse mode through kern.K(X, X2) with respect to the points X
        (X2 None: K(X, X), both arguments move)."""
        X, W = _f64(X), _f64(W)
        n, d = X.shape
        m = n
        if X2 is not None:
            X2 = _f64(X2)
            _need(X2.ndim == 2 and X2.shape[1] == d, "X2 must be [M, %d]" % d)
            m = X2.shape[0]
        _need(W.shape == (n, m), "W must be [N, M]")
        g = np.zeros((n, d))
        self.resident_token = None
        self.factor_key = None
        self._check(self._lib.gps_kmat_input_vjp(self._h, prog, len(prog), _ptr(X), n, _ptr(X2) if X2 is not None else None, m, d,
                                                 _ptr(W), _ptr(g)), "gps_kmat_input_vjp")
        return g

    def kmat_vjp(self, prog, X, W, X2=None):
        """sum_ij W[i, j] d k(X_i, X2_j) / d(kernel parameter slot): reverse mode through kern.K(X, X2)."""
        X, W = _f64(X), _f64(W)
        n, d = X.shape
        m = n
        if X2 is not None:
            X2 = _f64(X2)
            _need(X2.ndim == 2 and X2.shape[1] == d, "X2 must be [M, %d]" % d)
            m = X2.shape[0]
        _need(W.shape == (n, m), "W must be [N, M]")
        cap = 700
        slots = np.zeros(cap)
        ns = ctypes.c_int(0)
        self._check(self._lib.gps_kmat_vjp(self._h, prog, len(prog), _ptr(X), n, _ptr(X2) if X2 is not None else None, m, d,
                                           _ptr(W), _ptr(slots), cap, ctypes.byref(ns)), "gps_kmat_vjp")
        return slots[:ns.value].copy()

    def gauss_kl(self, q_mu, q_sqrt, K=None):
        """kullback_leiblers.py:26-105"""
        q_mu = _f64(q_mu)
        _need(q_mu.ndim == 2, "q_mu must be [M, K]")
        m, k = q_mu.shape
        q, qnd = self._prep_q_sqrt(q_sqrt)
        _need(q is not None, "gauss_kl needs q_sqrt")
        self._check_q_sqrt(q, qnd, m, k)
        if m == 0 or k == 0:
            return 0.0
        if K is not None:
            K = _f64(K)
            _need(K.shape == (m, m), "K must be [M, M]")
        out = ctypes.c_double(0)
        info = ctypes.c_int(0)
        self._check(self._lib.gps_gauss_kl(self._h, _ptr(K) if K is not None else None, m, _ptr(q_mu), k, _ptr(q),
                                           qnd, ctypes.byref(out), ctypes.byref(info)), "gps_gauss_kl")
        if info.value > 0:
            raise NotPositiveDefiniteError("K is not positive definite (leading minor of order %d)" % info.value)
        return out.value

    def base_conditional(self, Kmn, Kmm, Knn, f, q_sqrt=None, white=False, full_cov=False):
        Kmn, Kmm, Knn, f = _f64(Kmn), _f64(Kmm), _f64(Knn), _f64(f)
        _need(Kmn.ndim == 2 and f.ndim == 2, "Kmn must be [M, N] and f [M, K]")
        m, n_new = Kmn.shape
        k = f.shape[1]
        _need(Kmm.shape == (m, m) and f.shape[0] == m and m > 0, "Kmm must be [M, M] and f [M, K]")
        _need(Knn.shape == ((n_new, n_new) if full_cov else (n_new,)),
              "Knn must be [N, N] with full_cov and [N] without")
        q, qnd = self._prep_q_sqrt(q_sqrt)
        self._check_q_sqrt(q, qnd, m, k)
        fmean, fvar = self._cond_outputs(n_new, k, full_cov)
        if n_new == 0:
            return fmean, (np.empty((0, 0, k)) if full_cov else fvar)
        info = ctypes.c_int(0)
        self.resident_token = None
        self.factor_key = None
        self._check(self._lib.gps_base_conditional(self._h, _ptr(Kmn), _ptr(Kmm), _ptr(Knn), m, n_new, _ptr(f), k,
                                                   _ptr(q) if q is not None else None, qnd, 1 if white else 0,
                                                   1 if full_cov else 0, _ptr(fmean), _ptr(fvar),
                                                   ctypes.byref(info)), "gps_base_conditional")
        if info.value > 0:
            raise NotPositiveDefiniteError("Cholesky decomposition was not successful (order %d)" % info.value)
        if full_cov:
            fvar = np.ascontiguousarray(np.transpose(fvar, (1, 2, 0)))
        return fmean, fvar


def _need(cond, msg):
    """Shape errors surface as ValueError before any pointer reaches the library (the reference gets the
    equivalent InvalidArgumentError from the TF op that sees the mismatching shapes)."""
    if not cond:
        raise ValueError(msg)


_default = None
_default_lock = threading.Lock()


def get_handle():
    """Process-wide default handle (device = $GPFLOWSLIM_DEVICE, else $LOCAL_RANK, else 0)."""
    global _default
    with _default_lock:
        if _default is None:
            _default = Handle()
        return _default


def set_handle(handle):
    global _default
    with _default_lock:
        _default = handle
