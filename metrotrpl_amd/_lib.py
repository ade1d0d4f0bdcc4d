"""ctypes binding of the C ABI in include/metrotrpl_amd.h.

There is no CPU fallback: if the HIP library is missing, or no gfx950 device is
present, every compute entry point raises RuntimeError.
"""
import ctypes as C
import os
import subprocess
import threading



def configure_runtime(hw_queues=16):
    """Ask the HIP runtime for `hw_queues` hardware queues (GPU_MAX_HW_QUEUES; the runtime reads it at its first call,
    not at import).  The pipelined host path and the engines' side streams need more than the default 4: two streams on
    one queue serialise their launches (include/metrotrpl_amd.h, "context").  A value the caller has exported wins.
    Called when the package is imported, unless MTRPL_KEEP_HW_QUEUES=1 says to leave the process environment alone --
    it is a process-wide setting that every other HIP user of the process (PyTorch, RCCL) and every child process sees
    (the C library itself never touches the environment).  Returns the value in force, or None when
    nothing was set."""
    if os.environ.get("MTRPL_KEEP_HW_QUEUES") == "1":
        return os.environ.get("GPU_MAX_HW_QUEUES")
    return os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(hw_queues)))


configure_runtime()

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("MTRPL_LIB_PATH") or os.path.join(_PKG, "lib", "libmetrotrpl_amd.so")
CSRC = os.path.join(_PKG, "csrc")

OK, ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_DEVICE, ERR_NO_DEVICE = 0, 1, 2, 3, 4
MEAS = {"TRPL": 0, "TRTS": 1}
MODEL = {"std": 0, "traps": 1}
INI = {"density": 0, "fluence": 1}
PAR_SLOTS = ["n0", "p0", "mu_n", "mu_p", "ks", "Cn", "Cp", "Sf", "Sb", "tauN", "tauP", "eps", "Tm",
             "kC", "Nt", "tauE"]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class SolveDesc(C.Structure):
    """mtrpl_solve_desc"""
    _fields_ = [("par", C.c_double * 16), ("thickness", C.c_double), ("hmax", C.c_double),
                ("rtol", C.c_double), ("atol", C.c_double), ("min_y", C.c_double),
                ("ini", C.c_double * 3), ("nx", C.c_int32), ("nt", C.c_int32), ("meas", C.c_int32),
                ("model", C.c_int32), ("ini_mode", C.c_int32), ("reserved", C.c_int32),
                ("ini_off", C.c_int64), ("t_off", C.c_int64), ("out_off", C.c_int64)]


class LikeDesc(C.Structure):
    """mtrpl_like_desc"""
    _fields_ = [("sig_off", C.c_int64), ("t_off", C.c_int64), ("exp_off", C.c_int64),
                ("nt", C.c_int32), ("n_exp", C.c_int32), ("irf", C.c_int32), ("force_min_y", C.c_int32),
                ("scale_shift", C.c_double), ("sigma", C.c_double)]


SOLVE_DESC_DTYPE = np.dtype([("par", "f8", 16), ("thickness", "f8"), ("hmax", "f8"), ("rtol", "f8"),
                             ("atol", "f8"), ("min_y", "f8"), ("ini", "f8", 3), ("nx", "i4"),
                             ("nt", "i4"), ("meas", "i4"), ("model", "i4"), ("ini_mode", "i4"),
                             ("reserved", "i4"), ("ini_off", "i8"), ("t_off", "i8"), ("out_off", "i8")],
                            align=True)
LIKE_DESC_DTYPE = np.dtype([("sig_off", "i8"), ("t_off", "i8"), ("exp_off", "i8"), ("nt", "i4"),
                            ("n_exp", "i4"), ("irf", "i4"), ("force_min_y", "i4"), ("scale_shift", "f8"),
                            ("sigma", "f8")], align=True)
assert SOLVE_DESC_DTYPE.itemsize == C.sizeof(SolveDesc)
assert LIKE_DESC_DTYPE.itemsize == C.sizeof(LikeDesc)

# every symbol include/metrotrpl_amd.h declares: (restype, argtypes)
_vp = C.c_void_p
SIGNATURES = {
    "mtrpl_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "mtrpl_ctx_destroy": (C.c_int, [_vp]),
    "mtrpl_ctx_trim": (C.c_int, [_vp]),
    "mtrpl_host_alloc": (C.c_int, [_vp, C.c_int64, C.POINTER(_vp)]),
    "mtrpl_host_free": (C.c_int, [_vp, _vp]),
    "mtrpl_last_error": (C.c_char_p, [_vp]),
    "mtrpl_version": (C.c_char_p, []),
    "mtrpl_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "mtrpl_get_option": (C.c_int, [_vp, C.c_char_p, _dp]),
    "mtrpl_solve_batch": (C.c_int, [_vp, C.c_int64, _vp, _dp, C.c_int64, _dp, C.c_int64, _dp, C.c_int64,
                                    _ip, _ip]),
    "mtrpl_solve_batch_dev": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, _vp]),
    "mtrpl_solve_batch_dev2": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, _vp, C.c_int32]),
    "mtrpl_rhs_batch": (C.c_int, [_vp, C.c_int64, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp]),
    "mtrpl_E_field": (C.c_int, [_vp, C.c_int64, C.c_int32, _dp, _dp, C.c_double, C.c_double, C.c_double,
                                C.c_double, C.c_double, _dp]),
    "mtrpl_calculate_PL": (C.c_int, [_vp, C.c_int64, C.c_int32, C.c_double, _dp, _dp, C.c_double,
                                     C.c_double, C.c_double, _dp]),
    "mtrpl_calculate_TRTS": (C.c_int, [_vp, C.c_int64, C.c_int32, C.c_double, _dp, _dp, C.c_double,
                                       C.c_double, C.c_double, C.c_double, _dp]),
    "mtrpl_calculate_RR": (C.c_int, [_vp, C.c_int64, _dp, _dp, C.c_double, C.c_double, C.c_double, _dp]),
    "mtrpl_calculate_photoc": (C.c_int, [_vp, C.c_int64, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double, _dp]),
    "mtrpl_I_moment": (C.c_int, [_vp, C.c_int32, _dp, _dp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "mtrpl_newton_probe": (C.c_int, [_vp, C.c_int64, C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp]),
    "mtrpl_reduce_probe": (C.c_int, [_vp, C.c_int64, _vp, _vp]),
    "mtrpl_make_I_table": (C.c_int, [_vp, C.c_int32, _dp, _dp, _dp]),
    "mtrpl_irf_conv_len": (C.c_int64, [C.c_double, C.c_double]),
    "mtrpl_irf_convolve": (C.c_int, [_vp, C.c_int32, _dp, _dp, C.c_int32, _dp, C.c_double, C.c_int32,
                                     C.c_int64, _dp, _dp, _ip]),
    "mtrpl_convolve": (C.c_int, [_vp, C.c_int64, _dp, C.c_int32, _dp, _dp]),
    "mtrpl_interp_linear": (C.c_int, [_vp, C.c_int64, _dp, _dp, C.c_int64, _dp, _dp]),
    "mtrpl_irf_tables_upload": (C.c_int, [_vp, C.c_int32, _ip, _dp, _dp]),
    "mtrpl_loglik_batch_dev": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32,
                                         C.c_int32, _vp, _vp]),
    "mtrpl_pt_decide_dev": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mtrpl_pt_select_dev": (C.c_int, [_vp, C.c_int64, C.c_int32, C.c_int64, _vp, _vp, _vp, _vp, _vp]),
}


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    srcs.append(os.path.join(_REPO, "include", "metrotrpl_amd.h"))
    if not force and os.path.exists(LIB_PATH) and all(
            os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-o", LIB_PATH, os.path.join(CSRC, "mtrpl_capi.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None
_lock = threading.Lock()


def lib():
    """Load the shared library (raises RuntimeError if it was never built)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                                   "g.build()'` (needs hipcc). There is no CPU fallback.")
            # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7
            # (same SONAME as /opt/rocm's).  Whichever is loaded first serves every
            # later NEEDED entry, so let torch (device memory / streams / RCCL
            # plumbing for bench.py and likelihood.py) bind its runtime first.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
            L = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                if os.environ.get("MTRPL_LIB_PATH") and not hasattr(L, name):
                    continue  # (an older build of the library named explicitly for an A/B run: it has what it has)
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
            _lib = L
    return _lib


class Context:
    """mtrpl_ctx: one per process and GPU; blocking calls; not thread-safe.

    Pinned host memory handed out by `PinnedBuffer` belongs to the context (mtrpl_ctx_destroy frees it): while any such
    buffer is alive `close()` only marks the context closed -- every later call through it raises -- and the native
    context goes with the last buffer, so that an array a caller still holds never points at freed memory."""

    def __init__(self, device=0):
        self._h = _vp()
        self._pinned_live = 0      # PinnedBuffers alive
        self._closed = False
        rc = lib().mtrpl_ctx_create(int(device), C.byref(self._h))
        if rc != OK:
            self._h = None
            raise RuntimeError(f"mtrpl_ctx_create(device={device}) failed with code {rc}: "
                               "no MI355X (gfx950) device visible. There is no CPU fallback.")
        self.device = int(device)

    def _destroy(self):
        if getattr(self, "_h", None):
            lib().mtrpl_ctx_destroy(self._h)
            self._h = None

    def close(self):
        self._closed = True
        self.__dict__.pop("_pinned_free", None)  # (pooled buffers nobody holds: they go with the context)
        if getattr(self, "_pinned_live", 0) == 0:
            self._destroy()

    def __del__(self):
        try:
            # (a live PinnedBuffer holds a reference to its context: when this runs there is none)
            self._closed = True
            self._destroy()
        except Exception:
            pass

    @property
    def handle(self):
        if getattr(self, "_closed", False):
            raise RuntimeError("metrotrpl_amd: this context has been closed")
        return self._h

    def check(self, rc):
        if rc == OK:
            return
        msg = lib().mtrpl_last_error(self._h)
        msg = msg.decode() if msg else ""
        if rc == ERR_BAD_ARG:
            raise ValueError(msg)
        if rc == ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        raise RuntimeError(f"metrotrpl_amd error {rc}: {msg}")

    def set_option(self, key, value):
        self.check(lib().mtrpl_set_option(self.handle, key.encode(), float(value)))

    def get_option(self, key):
        v = C.c_double()
        self.check(lib().mtrpl_get_option(self.handle, key.encode(), C.byref(v)))
        return v.value

    def trim(self):
        """Free device memory the context no longer uses (mtrpl_ctx_trim: synchronises the device -- quiet points only)."""
        self.check(lib().mtrpl_ctx_trim(self.handle))


class PinnedBuffer:
    """`nbytes` of pinned host memory from the library (mtrpl_host_alloc) as the base object of numpy arrays
    (`np.asarray(buf)` is a float64 view that keeps `buf` alive).  mtrpl_solve_batch copies straight into such memory
    while later sub-launches compute.  Pinning is slow (page by page), so freed buffers go back to a small per-context
    free-list by size instead of to the runtime (at most `_KEEP_BYTES` in total; what does not fit is freed).

    The memory is the CONTEXT's: a live buffer keeps the native context alive past `Context.close()` (see there).
    Every byte a caller keeps this way stays page-locked -- results that are kept for long belong in ordinary arrays
    (`batch.solve_batch` returns those unless asked for `pinned=True`)."""
    _KEEP_BYTES = 256 << 20

    def __init__(self, ctx, nbytes):
        nbytes = max(8, int(nbytes) + (-int(nbytes)) % 8)
        self.ctx, self.nbytes, self.ptr = ctx, nbytes, None
        pool = ctx.__dict__.setdefault("_pinned_free", {})
        have = pool.get(nbytes)
        if have:
            self.ptr = have.pop()
        else:
            p = _vp()
            ctx.check(lib().mtrpl_host_alloc(ctx.handle, nbytes, C.byref(p)))
            self.ptr = p.value
        ctx._pinned_live += 1
        self.__array_interface__ = {"data": (self.ptr, False), "shape": (nbytes // 8,), "typestr": "<f8", "version": 3}

    def __del__(self):
        try:
            ctx = self.ctx
            if self.ptr is None or not getattr(ctx, "_h", None):
                return
            ctx._pinned_live -= 1
            if ctx._closed:
                if ctx._pinned_live == 0:
                    ctx._destroy()  # the last buffer of a closed context takes the native context with it
                return
            pool = ctx.__dict__.setdefault("_pinned_free", {})
            if sum(k * len(v) for k, v in pool.items()) + self.nbytes <= self._KEEP_BYTES:
                pool.setdefault(self.nbytes, []).append(self.ptr)
            else:
                lib().mtrpl_host_free(ctx._h, self.ptr)
        except Exception:
            pass


def pinned_f64(ctx, n):
    """float64 array of n elements in pinned host memory (contents undefined)."""
    return np.asarray(PinnedBuffer(ctx, 8 * int(n)))[:int(n)]


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def dptr(a):
    return a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


_STREAMS = {}


def shared_streams(dev, n):
    """`n` side streams of device `dev`, the SAME objects for every caller of this process.  The runtime multiplexes
    HIP streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, default 4): a process that creates fresh
    streams for every stage ends up with two concurrently used streams on one queue -- their launches serialise
    (measured: two of three chain groups of an ensemble sharing a queue, 4.2 ms per iteration instead of 2.4)."""
    import torch
    pool = _STREAMS.setdefault(str(dev), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]
