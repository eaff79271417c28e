"""ctypes binding of libhgmm_hip.so (the C ABI declared in include/hgmm.h).

There is NO CPU fallback anywhere in this package: if the HIP library is missing or no
gfx950 device is visible, loading / context creation raises ``HgmmError`` loudly.
"""
from __future__ import annotations

import ctypes as C
import contextlib
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhgmm_hip.so")

COV_TYPES = {"diag": 0, "spherical": 1}
VARIANTS = {"W": 0, "G": 1}
KERNEL_IDS = {"flat_estep": 0, "flat_fused": 1, "flat_mstep": 2, "tree_estep": 3,
              "tree_loglik": 4, "tree_reg": 5, "util_fill": 6, "full_pass": 7, "full_moments": 8,
              "kmeans_assign": 9, "kmeans_accum": 10, "allreduce": 11, "full_fused": 12}


class HgmmError(RuntimeError):
    pass


_lib = None
_lib_lock = threading.Lock()

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_vp = C.c_void_p


def _sig(lib, name, argtypes, restype=C.c_int):
    fn = getattr(lib, name)
    fn.argtypes = argtypes
    fn.restype = restype
    return fn


def load_library(path: str = LIB_PATH):
    """dlopen the HIP library and declare every entry point of include/hgmm.h."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(path):
            raise HgmmError(
                "HIP extension %s is missing; build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or python <package>/build.py). There is no CPU fallback." % path)
        # multi-process RCCL on this platform needs dmabuf IPC (the host driver does not offer the
        # legacy IPC handles); must be in the environment before the HSA runtime starts
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        ctx = _vp
        _sig(lib, "hgmm_version", [])
        _sig(lib, "hgmm_device_count", [C.POINTER(C.c_int)])
        _sig(lib, "hgmm_create", [C.c_int, C.POINTER(_vp)])
        _sig(lib, "hgmm_destroy", [ctx])
        _sig(lib, "hgmm_last_error", [ctx], C.c_char_p)
        _sig(lib, "hgmm_device_info", [ctx, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)])
        _sig(lib, "hgmm_synchronize", [ctx])
        _sig(lib, "hgmm_alloc", [ctx, C.c_size_t, C.POINTER(_vp)])
        _sig(lib, "hgmm_free", [ctx, _vp])
        _sig(lib, "hgmm_h2d", [ctx, _vp, _vp, C.c_size_t])
        _sig(lib, "hgmm_d2h", [ctx, _vp, _vp, C.c_size_t])
        _sig(lib, "hgmm_d2d", [ctx, _vp, _vp, C.c_size_t])
        _sig(lib, "hgmm_set_points_f32", [ctx, _vp, C.c_int64])
        _sig(lib, "hgmm_set_points_f64", [ctx, _vp, C.c_int64])
        _sig(lib, "hgmm_num_points", [ctx], C.c_int64)
        _sig(lib, "hgmm_points_create_f32", [ctx, _vp, C.c_int64, C.POINTER(_vp)])
        _sig(lib, "hgmm_points_create_f64", [ctx, _vp, C.c_int64, C.POINTER(_vp)])
        _sig(lib, "hgmm_points_bind", [ctx, _vp])
        _sig(lib, "hgmm_points_destroy", [ctx, _vp])
        _sig(lib, "hgmm_points_count", [_vp], C.c_int64)
        _sig(lib, "hgmm_points_download_f32", [ctx, _vp, _vp])
        _sig(lib, "hgmm_flat_estep", [ctx, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _f64p])
        _sig(lib, "hgmm_flat_estep_async", [ctx, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
        _sig(lib, "hgmm_flat_estep_dev", [ctx, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
        _sig(lib, "hgmm_flat_mstep_dev", [ctx, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp])
        _sig(lib, "hgmm_flat_predict_dev", [ctx, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp])
        _sig(lib, "hgmm_elementwise_f32", [ctx, C.c_int, C.c_int64, _vp, _vp, C.c_float, _vp])
        _sig(lib, "hgmm_host_scalars", [ctx, C.c_int, C.POINTER(_f64p), C.POINTER(_vp)])
        _sig(lib, "hgmm_event_record", [ctx, C.c_int])
        _sig(lib, "hgmm_event_wait", [ctx, C.c_int])
        _sig(lib, "hgmm_flat_predict", [ctx, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp])
        _sig(lib, "hgmm_flat_log_prob", [ctx, C.c_int, C.c_int, _vp, _vp, _vp])
        _sig(lib, "hgmm_pace_info", [ctx, _f64p, C.POINTER(C.c_int), C.POINTER(C.c_int)])
        _sig(lib, "hgmm_pace_reset", [ctx])
        _sig(lib, "hgmm_flat_mstep", [ctx, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp])
        _sig(lib, "hgmm_flat_train", [ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _vp, _vp, _vp,
                                      _vp, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)])
        _sig(lib, "hgmm_flat_train_begin", [ctx, C.c_int, C.c_int, C.c_int, C.c_float, _vp, _vp, _vp, C.c_int])
        _sig(lib, "hgmm_flat_train_step", [ctx, C.c_int])
        _sig(lib, "hgmm_flat_train_end", [ctx, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)])
        _sig(lib, "hgmm_flat_stats", [ctx, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _f64p, _f64p])
        _sig(lib, "hgmm_tree_build", [ctx, C.c_int, C.c_double, C.c_double, _vp, C.c_double, C.c_int,
                                      _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.POINTER(C.c_int)])
        _sig(lib, "hgmm_tree_set_nodes", [ctx, C.c_int, _vp, _vp, _vp])
        _sig(lib, "hgmm_tree_set_precision", [ctx, C.c_int])
        _sig(lib, "hgmm_tree_set_target", [ctx, _vp, C.c_int64])
        _sig(lib, "hgmm_tree_reg_estep", [ctx, _vp, _vp, C.c_double, C.c_double, _vp, _vp, _vp])
        _sig(lib, "hgmm_tree_reg_normal", [ctx, _vp, _vp, C.c_double, C.c_double, _vp])
        _sig(lib, "hgmm_tree_node_complexity", [ctx, _vp])
        _sig(lib, "hgmm_tree_estep", [ctx, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
        _sig(lib, "hgmm_tree_mstep", [ctx, C.c_int64, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                      _vp, _vp, _vp])
        _sig(lib, "hgmm_tree_loglik", [ctx, C.c_int64, _vp, _vp, _vp, C.c_int64, C.c_int64, _f64p])
        _sig(lib, "hgmm_tree_stats", [ctx, C.POINTER(C.c_uint64), C.POINTER(C.c_int)])
        _sig(lib, "hgmm_fullcov_fit", [ctx, C.c_int, C.c_double, C.c_double, _vp, C.c_double, C.c_int, _vp, _vp, _vp,
                                       _vp, _vp, C.c_int, C.POINTER(C.c_int)])
        _sig(lib, "hgmm_fullcov_estep", [ctx, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64p])
        _sig(lib, "hgmm_kmeans_plusplus", [ctx, C.c_int, C.c_int64, _vp, C.c_int, _vp, _vp])
        _sig(lib, "hgmm_kmeans_step", [ctx, C.c_int, _vp, C.c_int, _vp, _f64p, C.POINTER(C.c_int64)])
        _sig(lib, "hgmm_kmeans_labels", [ctx, _vp, _vp])
        _sig(lib, "hgmm_kmeans_lloyd", [ctx, C.c_int, _vp, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.POINTER(C.c_int), _vp, C.POINTER(C.c_int64)])
        _sig(lib, "hgmm_gauss_transform", [ctx, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_double, _vp])
        _sig(lib, "hgmm_comm_unique_id", [_vp])
        _sig(lib, "hgmm_comm_init_rank", [ctx, C.c_int, C.c_int, _vp])
        _sig(lib, "hgmm_comm_destroy", [ctx])
        _sig(lib, "hgmm_comm_init_host", [ctx, C.c_int, C.c_int, C.c_char_p])
        _sig(lib, "hgmm_comm_init_ipc", [ctx, C.c_int, C.c_int, C.c_char_p])
        _sig(lib, "hgmm_comm_allreduce_f64", [ctx, _vp, C.c_int, C.c_int])
        _sig(lib, "hgmm_profile_enable", [ctx, C.c_int])
        _sig(lib, "hgmm_profile_reset", [ctx])
        _sig(lib, "hgmm_profile_get", [ctx, C.c_int, _f64p, C.POINTER(C.c_int64)])
        _sig(lib, "hgmm_fullcov_phase_clocks", [ctx, C.c_int, C.POINTER(C.c_int64)])
        _sig(lib, "hgmm_util_fill_f32", [ctx, _vp, C.c_int64, C.c_float, C.c_int])
        _sig(lib, "hgmm_kmeans_center_f64", [_vp, C.c_int64, _vp, _vp, _vp])
        _sig(lib, "hgmm_tree_register", [ctx, _vp, _vp, C.c_double, C.c_double, C.c_int, C.c_double, _vp,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int), _vp])
        _sig(lib, "hgmm_comm_stats", [ctx, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)])
        _sig(lib, "hgmm_config_count", [])
        _sig(lib, "hgmm_config_name", [C.c_int], C.c_char_p)
        _sig(lib, "hgmm_config_set", [ctx, C.c_char_p, C.c_int])
        _sig(lib, "hgmm_config_get", [ctx, C.c_char_p, C.POINTER(C.c_int)])
        _i64p, _i32p = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
        _sig(lib, "hgmm_set_points_batch_f64", [ctx, C.c_int, C.POINTER(_vp), _i64p])
        _sig(lib, "hgmm_set_points_batch_f32", [ctx, C.c_int, C.POINTER(_vp), _i64p])
        _sig(lib, "hgmm_tree_build_batch", [ctx, C.c_int, _i64p, C.c_int, C.c_double, C.c_double, _vp, C.c_double, C.c_int,
                                            _vp, _vp, _vp, _vp, _vp, C.c_int, _vp])
        _sig(lib, "hgmm_tree_get_nodes_batch", [ctx, C.c_int, _vp, _vp, _vp])
        _sig(lib, "hgmm_tree_set_targets_batch", [ctx, C.c_int, C.POINTER(_vp), _i64p])
        _sig(lib, "hgmm_tree_set_targets_batch_f32", [ctx, C.c_int, C.POINTER(_vp), _i64p])
        _sig(lib, "hgmm_tree_register_batch", [ctx, C.c_int, _vp, _vp, C.c_double, C.c_double, C.c_int, C.c_double, _vp,
                                               _vp, _vp, _vp])
        _lib = lib
        return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (shape, a.shape))
    return a


EW_OPS = {"add": 0, "sub": 1, "rsub": 2, "mul": 3, "div": 4, "rdiv": 5, "sqrt": 6, "exp": 7, "log": 8, "max": 9, "min": 10}
# arrays up to this size are carved from the context's arena (no hipMalloc / hipFree, nothing waits when they die)
SMALL_BYTES = 128 << 10
SMALL_SLABS = 64
# dead arrays kept for reuse by the next array of the same size, in total at most this much memory (and at most 1/16
# of the device's: Context._freed_cap; an allocation that fails releases them and is tried again, Context._alloc)
FREED_CAP_BYTES = 16 << 30
# np.exp() of an array with at least this many elements stays a lazy view (the consumer fuses the exponential)
LAZY_EXP_ELEMS = 1 << 20
# ndarray methods a DeviceArray answers by downloading itself first
_HOST_METHODS = frozenset((
    "sum", "mean", "min", "max", "std", "var", "copy", "T", "reshape", "tolist", "flatten", "ravel", "any", "all",
    "round", "clip", "dot", "squeeze", "transpose", "item", "argmin", "argsort", "cumsum", "prod", "nonzero", "tobytes",
    "real", "imag", "take", "repeat", "itemsize", "strides", "base", "flags"))
# ... and the ones that would silently modify (or alias) that throw-away copy: refused
_HOST_MUTATORS = frozenset(("fill", "flat", "view", "sort", "put", "itemset", "resize", "setflags", "partition"))


class DeviceArray:
    """A dense array living in HBM (what a CuPy ndarray was to the reference).

    ``np.asarray(d)`` / ``d.get()`` copy it to the host.  ``d.exp()`` is a lazy view used by
    ``m_step(X, log_resp.exp())`` so the exponential is fused into the moment kernel.

    float32 arrays do elementwise arithmetic with scalars and with arrays of their own shape ON THE DEVICE
    (``+ - * /``, ``np.sqrt / np.exp / np.log / np.maximum / np.minimum``): small kernels on the context's stream,
    nothing waits -- so a caller's own EM loop, ``inv_cov = 1 / (xp.sqrt(covariances + 1e-6) + eps)`` included
    (gmm_impl.py:134), never brings the parameters to the host.  Everything else an ndarray can do (indexing,
    ``.sum()``, comparisons, broadcasting against other shapes) is answered from a host copy."""

    __array_priority__ = 1000

    def __init__(self, ctx, shape, dtype, is_log=False, _base=None):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.is_log = is_log
        self._base = _base
        self._slab = None
        if _base is None:
            self._slab, self.ptr = ctx._alloc_small(self.nbytes)
            if self._slab is None:
                self.ptr = ctx._alloc(self.nbytes)
            self._owner = True
        else:
            self.ptr = _base.ptr
            self._owner = False

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def ndim(self):
        return len(self.shape)

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        self.ctx._d2h(out, self.ptr)
        if self._base is not None and self.is_log and not self._base.is_log:
            np.exp(out, out=out)
        return out

    def get_rows(self, lo, hi):
        """Download rows [lo, hi) of a 2-D (or entries of a 1-D) array without moving the rest."""
        lo, hi = int(lo), int(hi)
        if not (0 <= lo <= hi <= self.shape[0]):
            raise IndexError("row range out of bounds")
        row_elems = int(np.prod(self.shape[1:])) if len(self.shape) > 1 else 1
        out = np.empty((hi - lo,) + self.shape[1:], dtype=self.dtype)
        if out.size:
            off = lo * row_elems * self.dtype.itemsize
            self.ctx._d2h(out, C.c_void_p(self.ptr.value + off))
        if self._base is not None and self.is_log and not self._base.is_log:
            np.exp(out, out=out)
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype, copy=False)

    def exp(self):
        """Lazy exp view (keeps log values in HBM; consumers fuse the exponential)."""
        return DeviceArray(self.ctx, self.shape, self.dtype, is_log=True, _base=self)

    def argmax(self, axis=None):
        return self.get().argmax(axis=axis)

    def astype(self, dtype, copy=True, **kw):
        """Same dtype: the array itself stays where it is (as ``cupy.ndarray.astype(..., copy=False)`` would
        leave it); another dtype is answered from a host copy."""
        if np.dtype(dtype) == self.dtype and self._base is None:
            if not copy:
                return self
            dup = DeviceArray(self.ctx, self.shape, self.dtype)      # copy=True (the default): a fresh array, as NumPy / CuPy give
            self.ctx._d2d(dup.ptr, self.ptr, self.nbytes)
            return dup
        return self.get().astype(dtype, **kw)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, idx):
        return self.get()[idx]

    def __iter__(self):
        return iter(self.get())

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s%s)" % (self.shape, self.dtype, ", lazy exp" if self._base is not None else "")

    def __getattr__(self, name):
        # (reached only for names the class does not define)
        if name in _HOST_MUTATORS:
            raise TypeError("DeviceArray.%s would act on a throw-away host copy; use .get() for a host array" % name)
        if name in _HOST_METHODS:
            return getattr(self.get(), name)
        raise AttributeError(name)

    # -- elementwise arithmetic --------------------------------------------------------------
    def _on_device(self):
        return self.dtype == np.float32 and self._base is None and self.size > 0

    def _ew(self, op, other=None):
        """out = self op other on the device, or NotImplemented when this pair is not a device case."""
        if not self._on_device():
            return NotImplemented
        b_ptr, scalar = None, 0.0
        keep = None
        if other is None:
            pass
        elif isinstance(other, DeviceScalar):
            scalar = float(other)
        elif isinstance(other, DeviceArray):
            if not other._on_device() or other.shape != self.shape or other.ctx is not self.ctx:
                return NotImplemented
            b_ptr = other.ptr
        elif isinstance(other, np.ndarray) and other.ndim > 0:
            if other.shape != self.shape:
                return NotImplemented
            keep = self.ctx.to_device(np.ascontiguousarray(other, dtype=np.float32))
            b_ptr = keep.ptr
        elif isinstance(other, (int, float, np.integer, np.floating)) or (isinstance(other, np.ndarray) and other.ndim == 0):
            scalar = float(other)
        else:
            return NotImplemented
        out = DeviceArray(self.ctx, self.shape, np.float32)
        self.ctx._check(self.ctx.lib.hgmm_elementwise_f32(self.ctx.h, EW_OPS[op], self.size, self.ptr, b_ptr,
                                                          C.c_float(scalar), out.ptr))
        return out

    def _host_op(self, fn, other, reverse=False):
        a, b = np.asarray(self), (np.asarray(other) if isinstance(other, DeviceArray) else other)
        return fn(b, a) if reverse else fn(a, b)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method == "__call__" and not kwargs:
            name = ufunc.__name__
            if len(inputs) == 1 and name in ("sqrt", "exp", "log"):
                if name == "exp" and self._base is None and self.size >= LAZY_EXP_ELEMS:
                    return self.exp()                       # m_step(X, xp.exp(log_resp)), gmm_impl.py:132
                r = self._ew(name)
                if r is not NotImplemented:
                    return r
            elif len(inputs) == 2:
                pair = {"add": ("add", "add"), "subtract": ("sub", "rsub"), "multiply": ("mul", "mul"),
                        "true_divide": ("div", "rdiv"), "divide": ("div", "rdiv"), "maximum": ("max", "max"),
                        "minimum": ("min", "min")}.get(name)
                if pair is not None:
                    if inputs[0] is self:
                        r = self._ew(pair[0], inputs[1])
                    else:
                        r = self._ew(pair[1], inputs[0])
                    if r is not NotImplemented:
                        return r
        host = [np.asarray(x) if isinstance(x, DeviceArray) else x for x in inputs]
        if "out" in kwargs and any(isinstance(o, DeviceArray) for o in kwargs["out"]):
            raise TypeError("a DeviceArray cannot be the out= of a NumPy ufunc")
        return getattr(ufunc, method)(*host, **kwargs)

    def free(self):
        if self._owner and self.ptr:
            if self._slab is not None:
                self.ctx._free_small(self._slab)
            else:
                self.ctx._free(self.ptr, self.nbytes)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _binary(op, rop, fn):
    def fwd(self, other):
        r = self._ew(op, other)
        return self._host_op(fn, other) if r is NotImplemented else r

    def rev(self, other):
        r = self._ew(rop, other)
        return self._host_op(fn, other, reverse=True) if r is NotImplemented else r
    return fwd, rev


for _n, _op, _rop, _fn in (("add", "add", "add", np.add), ("sub", "sub", "rsub", np.subtract),
                           ("mul", "mul", "mul", np.multiply), ("truediv", "div", "rdiv", np.true_divide)):
    _f, _r = _binary(_op, _rop, _fn)
    setattr(DeviceArray, "__%s__" % _n, _f)
    setattr(DeviceArray, "__r%s__" % _n, _r)
for _n, _fn in (("lt", np.less), ("le", np.less_equal), ("gt", np.greater), ("ge", np.greater_equal),
                ("eq", np.equal), ("ne", np.not_equal), ("pow", np.power)):
    setattr(DeviceArray, "__%s__" % _n, (lambda fn: lambda self, other: self._host_op(fn, other))(_fn))
DeviceArray.__neg__ = lambda self: self * -1.0
DeviceArray.__abs__ = lambda self: np.abs(np.asarray(self))
DeviceArray.__hash__ = lambda self: id(self)


class DeviceScalar(DeviceArray):
    """A float64 scalar that a kernel writes and the host reads when somebody looks at it (what a 0-d CuPy array was
    to the reference: ``e_step`` returns ``xp.mean(log_prob_norm)`` without synchronising, gmm_impl.py:114-116).
    It lives in pinned host memory the device writes directly (hgmm_host_scalars) with an event recorded behind its
    producer: ``float(s)``, arithmetic, comparisons and ``np.float32(s)`` wait for THAT kernel only -- an M-step
    enqueued after the E-step keeps running while the caller's loop looks at the E-step's mean."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.shape = (1,)
        self.dtype = np.dtype(np.float64)
        self.is_log = False
        self._base = None
        self._slab = None
        self._owner = False
        self._value = None
        self._marked = False
        self._slot, self.ptr = ctx._scalar_take(self)

    def _mark(self):
        """Call right after the producing kernel has been enqueued."""
        self.ctx._check(self.ctx.lib.hgmm_event_record(self.ctx.h, self._slot))
        self._marked = True

    def item(self):
        if self._value is None:
            if self._marked:
                self.ctx._check(self.ctx.lib.hgmm_event_wait(self.ctx.h, self._slot))
            else:
                self.ctx.synchronize()
            self._value = float(self.ctx._scalar_host[self._slot])
        return self._value

    def get(self):
        return np.array([self.item()])

    def __float__(self):
        return self.item()

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.item(), dtype=dtype or np.float64)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        host = [x.item() if isinstance(x, DeviceScalar) else (np.asarray(x) if isinstance(x, DeviceArray) else x)
                for x in inputs]
        return getattr(ufunc, method)(*host, **kwargs)

    def __repr__(self):
        return "DeviceScalar(%r)" % self.item()

    def __format__(self, spec):
        return format(self.item(), spec)

    def __abs__(self):
        return abs(self.item())

    def __neg__(self):
        return -self.item()

    def free(self):
        pass


_SCALAR_OPS = {
    "__add__": lambda a, b: a + b, "__radd__": lambda a, b: b + a, "__sub__": lambda a, b: a - b,
    "__rsub__": lambda a, b: b - a, "__mul__": lambda a, b: a * b, "__rmul__": lambda a, b: b * a,
    "__truediv__": lambda a, b: a / b, "__rtruediv__": lambda a, b: b / a,
    "__lt__": lambda a, b: a < b, "__le__": lambda a, b: a <= b, "__gt__": lambda a, b: a > b,
    "__ge__": lambda a, b: a >= b, "__eq__": lambda a, b: a == b, "__ne__": lambda a, b: a != b}
# a DeviceArray operand answers through ITS operator with the roles swapped (its device kernels take a scalar)
_SCALAR_SWAP = {"__add__": "__radd__", "__radd__": "__add__", "__sub__": "__rsub__", "__rsub__": "__sub__",
                "__mul__": "__rmul__", "__rmul__": "__mul__", "__truediv__": "__rtruediv__",
                "__rtruediv__": "__truediv__", "__lt__": "__gt__", "__le__": "__ge__", "__gt__": "__lt__",
                "__ge__": "__le__", "__eq__": "__eq__", "__ne__": "__ne__"}


def _scalar_op(name):
    """Operators of a DeviceScalar.  The value takes part as a NumPy float32 scalar -- what ``xp.mean(log_prob_norm)``
    of a float32 array is in the reference (gmm_impl.py:114-116), so ``abs(change) < tol`` of a caller's loop rounds
    as it does there; ``float(s)`` keeps the kernel's float64.  Operands that are not numbers: arrays take the array
    path, anything else gets Python's NotImplemented protocol (``s == None`` is False, ``s in [None, 1.0]`` works)."""
    fn = _SCALAR_OPS[name]

    def f(self, other):
        if isinstance(other, DeviceScalar):
            return fn(np.float32(self.item()), np.float32(other.item()))
        if isinstance(other, DeviceArray):
            return getattr(other, _SCALAR_SWAP[name])(float(np.float32(self.item())))
        if isinstance(other, (bool, int, float, np.generic, np.ndarray)):
            return fn(np.float32(self.item()), other)
        return NotImplemented
    f.__name__ = name
    return f


for _n in _SCALAR_OPS:
    setattr(DeviceScalar, _n, _scalar_op(_n))
DeviceScalar.__hash__ = lambda self: id(self)


class Context:
    """One engine context = one GPU (one rank).  Thin, explicit wrapper over the C ABI."""

    def __init__(self, device_id: int = 0):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.hgmm_create(int(device_id), C.byref(h))
        if rc != 0:
            msg = self.lib.hgmm_last_error(None)
            raise HgmmError("hgmm_create(device=%d) failed (%d): %s -- this package has no CPU fallback"
                            % (device_id, rc, msg.decode() if msg else "?"))
        self.h = h
        self.device_id = device_id
        self.nranks, self.rank = 1, 0
        self._points_handles = set()          # clouds created through points_create and not destroyed yet
        self._arena = None                    # (pointer, free slab indices) of the small-array pool
        self._freed = {}                      # size -> pointers of dead arrays kept for reuse (_free)
        self._freed_bytes = 0
        self._freed_cap_bytes = None
        self._scalar_host = None              # pinned, device-visible doubles (DeviceScalar)
        self._scalar_dev = None
        self._scalar_next = 0
        self._scalar_owner = [None] * 64

    # -- plumbing ---------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0 and not getattr(self, "h", None):
            raise HgmmError("this Context has been closed")
        if rc != 0:
            msg = self.lib.hgmm_last_error(self.h)
            raise HgmmError("hgmm call failed (%d): %s" % (rc, msg.decode() if msg else "?"))

    def close(self):
        if getattr(self, "h", None):
            for ref in self._scalar_owner:            # scalars nobody has looked at yet keep their value
                sc = ref() if ref is not None else None
                if sc is not None:
                    try:
                        sc.item()
                    except Exception:
                        pass
            self._scalar_host = None
            for hv in list(self._points_handles):     # resident clouds nobody freed: they must go before the context
                self.lib.hgmm_points_destroy(self.h, _vp(hv))
            self._points_handles.clear()
            if self._arena is not None:               # (arrays carved from it die with the context)
                self.lib.hgmm_free(self.h, self._arena[0])
                self._arena = None
            for lst in self._freed.values():
                for ptr in lst:
                    self.lib.hgmm_free(self.h, ptr)
            self._freed, self._freed_bytes = {}, 0
            self.lib.hgmm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _alloc(self, nbytes):
        nbytes = int(max(nbytes, 4))
        cached = self._freed.get(nbytes)
        if cached:                                   # an array of exactly this size died earlier: take its memory
            self._freed_bytes -= nbytes
            return cached.pop()
        p = _vp()
        rc = self.lib.hgmm_alloc(self.h, nbytes, C.byref(p))
        if rc != 0 and self._freed_bytes:
            self.trim()                              # out of memory with dead arrays still cached: release them, once more
            rc = self.lib.hgmm_alloc(self.h, nbytes, C.byref(p))
        self._check(rc)
        return p

    def trim(self):
        """Give the memory of dead arrays (kept for reuse, see _free) back to the device."""
        for lst in self._freed.values():
            for ptr in lst:
                self.lib.hgmm_free(self.h, ptr)
        self._freed, self._freed_bytes = {}, 0

    def _freed_cap(self):
        """Dead arrays are kept up to 1/16 of the device's memory (18 GB of an MI355X's 288), FREED_CAP_BYTES at most."""
        if self._freed_cap_bytes is None:
            try:
                self._freed_cap_bytes = min(FREED_CAP_BYTES, self.device_info()["hbm_bytes"] // 16)
            except HgmmError:
                self._freed_cap_bytes = FREED_CAP_BYTES
        return self._freed_cap_bytes

    def _free(self, p, nbytes=None):
        """Arrays that die are kept for the next array of the same size (a caller's loop allocates the same shapes
        over and over: labels per frame, log_resp per iteration) -- up to FREED_CAP_BYTES; hipFree beyond that.  Everything
        that touches the memory is ordered on the context's one stream, so handing it out again needs no waiting,
        while hgmm_free has to drain the stream first."""
        if not getattr(self, "h", None):
            return
        if nbytes is not None:
            nbytes = int(max(nbytes, 4))
            lst = self._freed.setdefault(nbytes, [])
            if self._freed_bytes + nbytes <= self._freed_cap() and len(lst) < 4:
                lst.append(p)
                self._freed_bytes += nbytes
                return
        self.lib.hgmm_free(self.h, p)

    def _alloc_small(self, nbytes):
        """-> (slab index, pointer) from the context's arena, or (None, None) when the request is large or the
        arena is in use to the last slab.  Everything that touches the arrays is ordered on the context's one
        stream, so a slab can be handed out again as soon as its array died."""
        if nbytes > SMALL_BYTES:
            return None, None
        if self._arena is None:
            self._arena = (self._alloc(SMALL_BYTES * SMALL_SLABS), list(range(SMALL_SLABS - 1, -1, -1)))
        base, free = self._arena
        if not free:
            return None, None
        i = free.pop()
        return i, C.c_void_p(base.value + i * SMALL_BYTES)

    def _free_small(self, slab):
        if self._arena is not None:
            self._arena[1].append(slab)

    def _scalar_take(self, owner):
        """The next of 64 host-visible scalar slots (round robin); a scalar still alive in the slot is read first."""
        if self._scalar_host is None:
            hp, dp = _f64p(), _vp()
            self._check(self.lib.hgmm_host_scalars(self.h, 64, C.byref(hp), C.byref(dp)))
            self._scalar_host = np.ctypeslib.as_array(hp, shape=(64,))
            self._scalar_dev = dp.value
        slot = self._scalar_next
        self._scalar_next = (slot + 1) % 64
        old = self._scalar_owner[slot]
        old = old() if old is not None else None
        if old is not None:
            old.item()
        self._scalar_owner[slot] = weakref.ref(owner)
        return slot, C.c_void_p(self._scalar_dev + 8 * slot)

    def _d2h(self, host, dev_ptr):
        self._check(self.lib.hgmm_d2h(self.h, _ptr(host), dev_ptr, host.nbytes))

    def _d2d(self, dst_ptr, src_ptr, nbytes):
        self._check(self.lib.hgmm_d2d(self.h, dst_ptr, src_ptr, int(nbytes)))

    # -- per-context options (hgmm_config_*: each starts from HGMM_<NAME> in the environment at creation) -----------------
    def config_names(self):
        return [self.lib.hgmm_config_name(i).decode() for i in range(self.lib.hgmm_config_count())]

    def config_get(self, name):
        v = C.c_int()
        self._check(self.lib.hgmm_config_get(self.h, name.encode(), C.byref(v)))
        return v.value

    def config_set(self, name, value):
        self._check(self.lib.hgmm_config_set(self.h, name.encode(), int(value)))
        return self

    def config(self, **options):
        """``with ctx.config(tree_overlap=0): ...`` -- options set for the block, restored afterwards."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            old = {k: self.config_get(k) for k in options}
            try:
                for k, v in options.items():
                    self.config_set(k, v)
                yield self
            finally:
                for k, v in old.items():
                    self.config_set(k, v)
        return scope()

    def synchronize(self):
        self._check(self.lib.hgmm_synchronize(self.h))

    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int()
        mem = C.c_int64()
        self._check(self.lib.hgmm_device_info(self.h, name, 256, C.byref(cus), C.byref(mem)))
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": mem.value}

    def empty(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)

    def to_device(self, a):
        a = np.ascontiguousarray(a)
        d = DeviceArray(self, a.shape, a.dtype)
        self._check(self.lib.hgmm_h2d(self.h, d.ptr, _ptr(a), a.nbytes))
        return d

    # -- points -----------------------------------------------------------------------
    def set_points(self, X):
        """Upload the point cloud [N,3]; float64 input keeps full precision for the HGMM path
        (the reference's CPU twin works on Open3D float64 points), the flat EM path uses the
        float32 cast the reference applies (gmm_waymo/src/gmm.py:73)."""
        X = np.asarray(X)
        if X.ndim != 2 or X.shape[1] != 3:
            raise ValueError("points must have shape [N,3], got %s" % (X.shape,))
        if X.dtype == np.float64:
            Xc = np.ascontiguousarray(X)
            self._check(self.lib.hgmm_set_points_f64(self.h, _ptr(Xc), Xc.shape[0]))
        else:
            Xc = np.ascontiguousarray(X, dtype=np.float32)
            self._check(self.lib.hgmm_set_points_f32(self.h, _ptr(Xc), Xc.shape[0]))
        self.n = Xc.shape[0]
        # (this is the context's OWN cloud and becomes the bound one; clouds held through points_create handles --
        #  DevicePoints -- stay resident and are re-bound when they are used)
        return self

    @property
    def num_points(self):
        return int(self.lib.hgmm_num_points(self.h))

    # several resident clouds (hgmm_points_*): a handle owns its device copy, bind is a pointer swap
    def points_create(self, X):
        """Upload [N,3] points into a handle of their own (float64 input keeps full precision for the HGMM / KMeans
        kernels).  -> opaque handle for points_bind / points_destroy; the context's bound cloud does not change."""
        X = np.asarray(X)
        if X.ndim != 2 or X.shape[1] != 3:
            raise ValueError("points must have shape [N,3], got %s" % (X.shape,))
        h = _vp()
        if X.dtype == np.float64:
            Xc = np.ascontiguousarray(X)
            self._check(self.lib.hgmm_points_create_f64(self.h, _ptr(Xc), Xc.shape[0], C.byref(h)))
            self._points_handles.add(h.value)
            return h
        else:
            Xc = np.ascontiguousarray(X, dtype=np.float32)
            self._check(self.lib.hgmm_points_create_f32(self.h, _ptr(Xc), Xc.shape[0], C.byref(h)))
        self._points_handles.add(h.value)
        return h

    def points_download(self, handle=None):
        """The float32 rows [N,3] of a resident cloud (None: the bound one) as a NumPy array."""
        n = int(self.lib.hgmm_points_count(handle)) if handle else self.num_points
        out = np.empty((n, 3), dtype=np.float32)
        self._check(self.lib.hgmm_points_download_f32(self.h, handle, _ptr(out)))
        return out

    def points_bind(self, handle):
        """Every following call works on this cloud (None: the cloud of the last set_points)."""
        self._check(self.lib.hgmm_points_bind(self.h, handle))
        self.n = self.num_points

    def points_destroy(self, handle):
        if getattr(self, "h", None) and handle and handle.value in self._points_handles:
            self._points_handles.discard(handle.value)
            self._check(self.lib.hgmm_points_destroy(self.h, handle))

    # -- flat EM ----------------------------------------------------------------------
    @staticmethod
    def _flat_args(mu, inv_or_cov, w, cov_type):
        mu = _f32(mu)
        J = mu.shape[0]
        if mu.shape != (J, 3):
            raise ValueError("means must be [J,3]")
        ic = _f32(inv_or_cov, (J, 3) if cov_type == "diag" else (J,))
        w = _f32(w, (J,))
        return J, mu, ic, w

    def _flat_dev_args(self, mu, inv_or_cov, w, cov_type):
        """The parameters as float32 DeviceArrays of this context (host arrays among them are uploaded)."""
        def dev(a, shape):
            if isinstance(a, DeviceArray) and a.ctx is self and a.dtype == np.float32 and a._base is None:
                if shape is not None and a.shape != tuple(shape):
                    raise ValueError("expected shape %s, got %s" % (shape, a.shape))
                return a
            # anything else -- host arrays, float64 / lazy-view / foreign-context DeviceArrays -- goes through the host
            return self.to_device(_f32(np.asarray(a), shape))
        mu = dev(mu, None)
        J = mu.shape[0]
        if mu.shape != (J, 3):
            raise ValueError("means must be [J,3]")
        return J, mu, dev(inv_or_cov, (J, 3) if cov_type == "diag" else (J,)), dev(w, (J,))

    def flat_estep(self, inv_std, mu, w, cov_type="diag", variant="W", want_log_resp=True,
                   want_lpn=False, want_argmax=False, out=None, lazy_mean=False):
        """``out``: optional pre-allocated DeviceArray [N,J] float32 to write log_resp into.
        ``lazy_mean``: do not wait for the kernel -- the mean log-normaliser comes back as a ``DeviceScalar``
        that is downloaded when it is looked at (hgmm_flat_estep_async)."""
        on_device = any(isinstance(a, DeviceArray) for a in (inv_std, mu, w))
        if on_device:
            J, mu, inv_std, w = self._flat_dev_args(mu, inv_std, w, cov_type)
        else:
            J, mu, inv_std, w = self._flat_args(mu, inv_std, w, cov_type)
        n = self.num_points
        if out is not None and (out.shape != (n, J) or out.dtype != np.float32):
            raise ValueError("out must be a float32 DeviceArray of shape %s" % ((n, J),))
        lr = out if out is not None else (self.empty((n, J), np.float32) if want_log_resp else None)
        lpn = self.empty((n,), np.float32) if want_lpn else None
        am = self.empty((n,), np.int32) if want_argmax else None
        if lazy_mean or on_device:
            ms = DeviceScalar(self)
            if on_device:
                self._check(self.lib.hgmm_flat_estep_dev(
                    self.h, COV_TYPES[cov_type], VARIANTS[variant], J, mu.ptr, inv_std.ptr, w.ptr,
                    None if lr is None else lr.ptr, None if lpn is None else lpn.ptr, None if am is None else am.ptr,
                    ms.ptr))
            else:
                self._check(self.lib.hgmm_flat_estep_async(
                    self.h, COV_TYPES[cov_type], VARIANTS[variant], J, _ptr(mu), _ptr(inv_std), _ptr(w),
                    None if lr is None else lr.ptr, None if lpn is None else lpn.ptr, None if am is None else am.ptr,
                    ms.ptr))
            ms._mark()
            return (ms if lazy_mean else float(ms)), lr, lpn, am
        mean = C.c_double()
        self._check(self.lib.hgmm_flat_estep(
            self.h, COV_TYPES[cov_type], VARIANTS[variant], J, _ptr(mu), _ptr(inv_std), _ptr(w),
            None if lr is None else lr.ptr, None if lpn is None else lpn.ptr, None if am is None else am.ptr, C.byref(mean)))
        return mean.value, lr, lpn, am

    def flat_log_prob(self, inv_std, mu, cov_type="diag", out=None):
        J, mu, inv_std, _ = self._flat_args(mu, inv_std, np.ones(len(mu), np.float32), cov_type)
        if out is not None and (out.shape != (self.num_points, J) or out.dtype != np.float32):
            raise ValueError("out must be a float32 DeviceArray of shape %s" % ((self.num_points, J),))
        out = out if out is not None else self.empty((self.num_points, J), np.float32)
        self._check(self.lib.hgmm_flat_log_prob(self.h, COV_TYPES[cov_type], J, _ptr(mu), _ptr(inv_std), out.ptr))
        return out

    def pace_info(self):
        """(rate in GB/s the next paced N x J writer offers its rows at, how often the controller has lowered it, how
        often a probe upwards has held)."""
        t, n, u = C.c_double(), C.c_int(), C.c_int()
        self._check(self.lib.hgmm_pace_info(self.h, C.byref(t), C.byref(n), C.byref(u)))
        return t.value, n.value, u.value

    def pace_reset(self):
        """Forget the store pacer's learnt rate and ceiling (hgmm_pace_reset)."""
        self._check(self.lib.hgmm_pace_reset(self.h))

    def flat_predict(self, inv_std, mu, w, cov_type="diag", variant="W"):
        if any(isinstance(a, DeviceArray) for a in (inv_std, mu, w)):
            J, mu, inv_std, w = self._flat_dev_args(mu, inv_std, w, cov_type)
            lab = self.empty((self.num_points,), np.int32)
            self._check(self.lib.hgmm_flat_predict_dev(self.h, COV_TYPES[cov_type], VARIANTS[variant], J,
                                                       mu.ptr, inv_std.ptr, w.ptr, lab.ptr))
            return lab
        J, mu, inv_std, w = self._flat_args(mu, inv_std, w, cov_type)
        lab = self.empty((self.num_points,), np.int32)
        self._check(self.lib.hgmm_flat_predict(self.h, COV_TYPES[cov_type], VARIANTS[variant], J,
                                               _ptr(mu), _ptr(inv_std), _ptr(w), lab.ptr))
        return lab

    def flat_mstep(self, resp, cov_type="diag", variant="W", centre_hint=None, device_out=False):
        """``device_out``: the new parameters stay in HBM (float32 DeviceArrays w [J], mu [J,3], cov [J,3] / [J]);
        nothing is downloaded and nothing waits (hgmm_flat_mstep_dev)."""
        if not isinstance(resp, DeviceArray):
            resp = self.to_device(np.ascontiguousarray(resp, dtype=np.float32))
        n, J = resp.shape
        if n != self.num_points:
            raise ValueError("resp has %d rows, context holds %d points" % (n, self.num_points))
        if device_out:
            hint = None
            if centre_hint is not None:
                hint = centre_hint if isinstance(centre_hint, DeviceArray) else self.to_device(_f32(centre_hint, (J, 3)))
                if hint.shape != (J, 3) or hint.dtype != np.float32 or hint.ctx is not self or hint._base is not None:
                    raise ValueError("centre_hint must be a float32 [J,3] array")
            w = DeviceArray(self, (J,), np.float32)
            mu = DeviceArray(self, (J, 3), np.float32)
            cov = DeviceArray(self, (J, 3) if cov_type == "diag" else (J,), np.float32)
            self._check(self.lib.hgmm_flat_mstep_dev(self.h, COV_TYPES[cov_type], VARIANTS[variant], J, resp.ptr,
                                                     1 if resp.is_log else 0, None if hint is None else hint.ptr,
                                                     w.ptr, mu.ptr, cov.ptr))
            return w, mu, cov
        if isinstance(centre_hint, DeviceArray):
            centre_hint = centre_hint.get()
        hint = None if centre_hint is None else _f32(centre_hint, (J, 3))
        w = np.empty(J, np.float32)
        mu = np.empty((J, 3), np.float32)
        cov = np.empty((J, 3) if cov_type == "diag" else (J,), np.float32)
        self._check(self.lib.hgmm_flat_mstep(self.h, COV_TYPES[cov_type], VARIANTS[variant], J, resp.ptr,
                                             1 if resp.is_log else 0, _ptr(hint), _ptr(w), _ptr(mu), _ptr(cov)))
        return w, mu, cov

    def flat_train(self, max_iter, tol, mu, cov, w, cov_type="diag", variant="W"):
        J, mu, cov, w = self._flat_args(mu, cov, w, cov_type)
        mu, cov, w = mu.copy(), cov.copy(), w.copy()
        inv = np.empty_like(cov)
        lls = np.zeros(max(int(max_iter), 1), np.float32)
        n_it, conv = C.c_int(), C.c_int()
        self._check(self.lib.hgmm_flat_train(self.h, COV_TYPES[cov_type], VARIANTS[variant], J, int(max_iter),
                                             float(tol), _ptr(mu), _ptr(cov), _ptr(w), _ptr(inv), _ptr(lls),
                                             C.byref(n_it), C.byref(conv)))
        return inv, mu, w, cov, lls[:n_it.value].copy(), bool(conv.value)

    def flat_train_begin(self, tol, mu, cov, w, cov_type="diag", variant="W", lls_capacity=1024):
        J, mu, cov, w = self._flat_args(mu, cov, w, cov_type)
        self._train_shape = (J, cov_type, lls_capacity)
        self._check(self.lib.hgmm_flat_train_begin(self.h, COV_TYPES[cov_type], VARIANTS[variant], J,
                                                   float(tol), _ptr(mu), _ptr(cov), _ptr(w), int(lls_capacity)))

    def flat_train_step(self, iters=1):
        self._check(self.lib.hgmm_flat_train_step(self.h, int(iters)))

    def flat_train_end(self):
        J, cov_type, cap = self._train_shape
        mu = np.empty((J, 3), np.float32)
        cov = np.empty((J, 3) if cov_type == "diag" else (J,), np.float32)
        inv = np.empty_like(cov)
        w = np.empty(J, np.float32)
        lls = np.zeros(cap, np.float32)
        n_it, conv = C.c_int(), C.c_int()
        self._check(self.lib.hgmm_flat_train_end(self.h, _ptr(mu), _ptr(cov), _ptr(w), _ptr(inv), _ptr(lls),
                                                 C.byref(n_it), C.byref(conv)))
        return inv, mu, w, cov, lls[:min(n_it.value, cap)].copy(), bool(conv.value), n_it.value

    def flat_stats(self, inv_std, mu, w, cov_type="diag", variant="W"):
        J, mu, inv_std, w = self._flat_args(mu, inv_std, w, cov_type)
        stats = np.empty((J, 7), np.float64)
        s, n = C.c_double(), C.c_double()
        self._check(self.lib.hgmm_flat_stats(self.h, COV_TYPES[cov_type], VARIANTS[variant], J, _ptr(mu),
                                             _ptr(inv_std), _ptr(w), _ptr(stats), C.byref(s), C.byref(n)))
        return stats, s.value, n.value

    # -- HGMM ---------------------------------------------------------------------------
    def tree_build(self, L, ls, ld, init_mu, sig2, max_iters_per_level=1000, q_capacity=None, want_leaf=True):
        """``want_leaf=False``: the node tables only, like the reference's buildGMMTree (hgmm_gpu.py:466-548 returns the
        nodes; its currentIdx stays on the device) -- the N-long leaf assignment is neither un-sorted nor downloaded
        (4 MB per million points through pageable memory: 0.4 ms)."""
        T = 8 * (8 ** L - 1) // 7
        init_mu = np.ascontiguousarray(init_mu, dtype=np.float64)
        if init_mu.shape != (T, 3):
            raise ValueError("init_mu must be [%d,3]" % T)
        n = self.num_points
        pi = np.empty(T)
        mu = np.empty((T, 3))
        cov = np.empty((T, 3, 3))
        leaf = np.empty(n, np.int32) if want_leaf else None
        iters = np.zeros(L, np.int32)
        qcap = int(q_capacity or L * max_iters_per_level)
        q = np.zeros(qcap)
        qlen = C.c_int()
        self._check(self.lib.hgmm_tree_build(self.h, int(L), float(ls), float(ld), _ptr(init_mu), float(sig2),
                                             int(max_iters_per_level), _ptr(pi), _ptr(mu), _ptr(cov), _ptr(leaf),
                                             _ptr(iters), _ptr(q), qcap, C.byref(qlen)))
        return pi, mu, cov, leaf, iters, q[:qlen.value].copy()

    def tree_set_precision(self, dtype):
        """``np.float64`` (default, the reference CPU twin's type) or ``np.float32`` (its GPU file's, hgmm_gpu.py:472-484):
        the level log-likelihood behind tree_build's stop rule evaluates its Gaussians in float32 on large clouds."""
        dt = np.dtype(dtype)
        if dt not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("tree precision must be float64 or float32, not %s" % dt)
        self._check(self.lib.hgmm_tree_set_precision(self.h, 1 if dt == np.dtype(np.float32) else 0))
        self.tree_dtype = dt
        return self

    def tree_set_nodes(self, L, pi, mu, cov):
        T = 8 * (8 ** L - 1) // 7
        pi = np.ascontiguousarray(pi, dtype=np.float64).reshape(T)
        mu = np.ascontiguousarray(mu, dtype=np.float64).reshape(T, 3)
        cov = np.ascontiguousarray(cov, dtype=np.float64).reshape(T, 3, 3)
        self._tree_T = T
        self._check(self.lib.hgmm_tree_set_nodes(self.h, int(L), _ptr(pi), _ptr(mu), _ptr(cov)))

    def tree_set_target(self, target):
        t = np.ascontiguousarray(target, dtype=np.float64)
        if t.ndim != 2 or t.shape[1] != 3:
            raise ValueError("target must be [M,3]")
        self._check(self.lib.hgmm_tree_set_target(self.h, _ptr(t), t.shape[0]))

    def tree_reg_estep(self, T, rot=None, t=None, scale=1.0, lambda_c=0.01):
        rot = None if rot is None else np.ascontiguousarray(rot, dtype=np.float64).reshape(3, 3)
        t = None if t is None else np.ascontiguousarray(t, dtype=np.float64).reshape(3)
        m0 = np.empty(T)
        m1 = np.empty((T, 3))
        m2 = np.empty((T, 3, 3))
        self._check(self.lib.hgmm_tree_reg_estep(self.h, _ptr(rot), _ptr(t), float(scale), float(lambda_c),
                                                 _ptr(m0), _ptr(m1), _ptr(m2)))
        return m0, m1, m2

    def tree_reg_normal(self, rot=None, t=None, scale=1.0, lambda_c=0.01):
        """E-step + normal equations of one registration iteration on the device (hgmm_tree_reg_normal).
        -> (AtA[6,6], Atb[6], btb) of the reference's stacked twist system (hgmm_gpu.py:729-752)."""
        rot = None if rot is None else np.ascontiguousarray(rot, dtype=np.float64).reshape(3, 3)
        t = None if t is None else np.ascontiguousarray(t, dtype=np.float64).reshape(3)
        out = np.empty(28)
        self._check(self.lib.hgmm_tree_reg_normal(self.h, _ptr(rot), _ptr(t), float(scale), float(lambda_c), _ptr(out)))
        ata = np.zeros((6, 6))
        ata[np.triu_indices(6)] = out[:21]
        ata = ata + np.triu(ata, 1).T
        return ata, out[21:27].copy(), float(out[27])

    def tree_register(self, rot, t, scale=1.0, lambda_c=0.01, max_iter=20, tol=1.0e-4, q_prev=None, want_trace=False):
        """Up to ``max_iter`` registration iterations inside the library (hgmm_tree_register).
        -> (rot[3,3], t[3], iterations done, q of the last one or ``q_prev``, status, trace or None);
        status 0: budget used up, 1: stopped by ``tol``, 2: the next iteration needs the host M-step."""
        rot = np.array(rot, dtype=np.float64).reshape(3, 3)
        t = np.array(t, dtype=np.float64).reshape(3)
        q = np.array([np.nan if q_prev is None else float(q_prev)])
        trace = np.zeros((max(int(max_iter), 1), 13)) if want_trace else None
        done, status = C.c_int(), C.c_int()
        self._check(self.lib.hgmm_tree_register(self.h, _ptr(rot), _ptr(t), float(scale), float(lambda_c), int(max_iter),
                                                float(tol), _ptr(q), C.byref(done), C.byref(status), _ptr(trace)))
        q_out = None if np.isnan(q[0]) else float(q[0])
        return rot, t, done.value, q_out, status.value, (None if trace is None else trace[:done.value])

    # -- batched HGMM: B independent clouds / scan pairs per launch set (hgmm_tree_*_batch) ------------------------------
    @staticmethod
    def _cloud_list(clouds, what):
        """-> (arrays kept alive, pointer table, counts, all_float32).  A list of float32 clouds stays float32 (the library
        widens on the device: exactly the float64 values, half the bytes, no host pass); anything else becomes float64."""
        raw = [np.asarray(getattr(a, "points", a)) for a in clouds]
        if not raw:
            raise ValueError("%s: no clouds" % what)
        all32 = all(a.dtype == np.float32 for a in raw)
        arrs = [np.ascontiguousarray(a, dtype=np.float32 if all32 else np.float64) for a in raw]
        for a in arrs:
            if a.ndim != 2 or a.shape[1] != 3 or a.shape[0] < 1:
                raise ValueError("%s: every cloud must be [N,3] with N >= 1, got %s" % (what, a.shape))
        ptrs = (_vp * len(arrs))(*[a.ctypes.data for a in arrs])
        counts = (C.c_int64 * len(arrs))(*[a.shape[0] for a in arrs])
        return arrs, ptrs, counts, all32

    def set_points_batch(self, clouds):
        """B clouds [N_b,3] become ONE resident cloud, cloud after cloud, each uploaded from its own array
        (hgmm_set_points_batch_f64 / _f32).  -> the arrays that were uploaded (their lengths are the forest's counts)."""
        arrs, ptrs, counts, all32 = self._cloud_list(clouds, "set_points_batch")
        entry = self.lib.hgmm_set_points_batch_f32 if all32 else self.lib.hgmm_set_points_batch_f64
        self._check(entry(self.h, len(arrs), ptrs, counts))
        self.n = int(sum(a.shape[0] for a in arrs))
        self._batch_counts = [a.shape[0] for a in arrs]
        return arrs

    def tree_build_batch(self, counts, L, ls, ld, init_mu, sig2, max_iters_per_level=1000, want_tables=True,
                         want_trace=False, q_capacity=None):
        """``B = len(counts)`` trees on the resident cloud's consecutive pieces in the same launches (hgmm_tree_build_batch);
        each bitwise what :meth:`tree_build` gives for that piece alone.  ``init_mu`` [B,T,3].
        -> (pi [B,T], mu [B,T,3], cov [B,T,3,3]) or (None, None, None), iters [B,L], list of q traces or None."""
        B = len(counts)
        T = 8 * (8 ** L - 1) // 7
        init_mu = np.ascontiguousarray(init_mu, dtype=np.float64)
        if init_mu.shape != (B, T, 3):
            raise ValueError("init_mu must be [%d,%d,3]" % (B, T))
        cnt = (C.c_int64 * B)(*[int(v) for v in counts])
        pi = np.empty((B, T)) if want_tables else None
        mu = np.empty((B, T, 3)) if want_tables else None
        cov = np.empty((B, T, 3, 3)) if want_tables else None
        iters = np.zeros((B, L), np.int32)
        qcap = int(q_capacity or L * min(max_iters_per_level, 4096)) if want_trace else 0
        q = np.zeros((B, qcap)) if want_trace else None
        qlen = np.zeros(B, np.int32) if want_trace else None
        self._check(self.lib.hgmm_tree_build_batch(self.h, B, cnt, int(L), float(ls), float(ld), _ptr(init_mu), float(sig2),
                                                   int(max_iters_per_level), _ptr(pi), _ptr(mu), _ptr(cov), _ptr(iters),
                                                   _ptr(q), qcap, _ptr(qlen)))
        traces = [q[b, :qlen[b]].copy() for b in range(B)] if want_trace else None
        return (pi, mu, cov), iters, traces

    def tree_get_nodes_batch(self, b, L):
        T = 8 * (8 ** L - 1) // 7
        pi, mu, cov = np.empty(T), np.empty((T, 3)), np.empty((T, 3, 3))
        self._check(self.lib.hgmm_tree_get_nodes_batch(self.h, int(b), _ptr(pi), _ptr(mu), _ptr(cov)))
        return pi, mu, cov

    def tree_set_targets_batch(self, targets):
        arrs, ptrs, counts, all32 = self._cloud_list(targets, "tree_set_targets_batch")
        entry = self.lib.hgmm_tree_set_targets_batch_f32 if all32 else self.lib.hgmm_tree_set_targets_batch
        self._check(entry(self.h, len(arrs), ptrs, counts))
        return arrs

    def tree_register_batch(self, rot, t, scale=1.0, lambda_c=0.01, max_iter=20, tol=1.0e-4, q_prev=None, want_trace=False):
        """Up to ``max_iter`` registration iterations of every (tree b, target b) pair of the resident forest in the same
        launches (hgmm_tree_register_batch), each bitwise :meth:`tree_register` on that pair.
        -> (rot [B,3,3], t [B,3], iterations [B], q [B] (NaN: none), status [B], traces or None)."""
        rot = np.array(rot, dtype=np.float64).reshape(-1, 3, 3)
        B = rot.shape[0]
        t = np.array(t, dtype=np.float64).reshape(B, 3)
        q = np.full(B, np.nan) if q_prev is None else np.array(q_prev, dtype=np.float64).reshape(B)
        iters, status = np.zeros(B, np.int32), np.zeros(B, np.int32)
        trace = np.zeros((B, max(int(max_iter), 1), 13)) if want_trace else None
        self._check(self.lib.hgmm_tree_register_batch(self.h, B, _ptr(rot), _ptr(t), float(scale), float(lambda_c),
                                                      int(max_iter), float(tol), _ptr(q), _ptr(iters), _ptr(status),
                                                      _ptr(trace)))
        traces = [trace[b, :iters[b]] for b in range(B)] if want_trace else None
        return rot, t, iters, q, status, traces

    @staticmethod
    def _node_tables(pi, mu, cov):
        pi = np.ascontiguousarray(pi, dtype=np.float64).reshape(-1)
        T = len(pi)
        mu = np.ascontiguousarray(mu, dtype=np.float64).reshape(T, 3)
        cov = np.ascontiguousarray(cov, dtype=np.float64).reshape(T, 3, 3)
        return T, pi, mu, cov

    def tree_estep(self, pi, mu, cov, parent_idx):
        T, pi, mu, cov = self._node_tables(pi, mu, cov)
        par = np.ascontiguousarray(parent_idx, dtype=np.int32)
        if par.shape != (self.num_points,):
            raise ValueError("parent_idx must have one entry per point")
        m0, m1, m2 = np.empty(T), np.empty((T, 3)), np.empty((T, 3, 3))
        cur = np.empty(self.num_points, np.int32)
        self._check(self.lib.hgmm_tree_estep(self.h, T, _ptr(pi), _ptr(mu), _ptr(cov), _ptr(par), _ptr(m0), _ptr(m1),
                                             _ptr(m2), _ptr(cur)))
        return m0, m1, m2, cur

    def tree_mstep(self, m0, m1, m2, j_begin, j_end, n_points, ld, pi, mu, cov):
        T, pi, mu, cov = self._node_tables(pi, mu, cov)
        pi, mu, cov = pi.copy(), mu.copy(), cov.copy()
        m0 = np.ascontiguousarray(m0, dtype=np.float64).reshape(T)
        m1 = np.ascontiguousarray(m1, dtype=np.float64).reshape(T, 3)
        m2 = np.ascontiguousarray(m2, dtype=np.float64).reshape(T, 3, 3)
        self._check(self.lib.hgmm_tree_mstep(self.h, T, _ptr(m0), _ptr(m1), _ptr(m2), int(j_begin), int(j_end),
                                             float(n_points), float(ld), _ptr(pi), _ptr(mu), _ptr(cov)))
        return pi, mu, cov

    def tree_loglik(self, pi, mu, cov, j_begin, j_end):
        T, pi, mu, cov = self._node_tables(pi, mu, cov)
        q = C.c_double()
        self._check(self.lib.hgmm_tree_loglik(self.h, T, _ptr(pi), _ptr(mu), _ptr(cov), int(j_begin), int(j_end),
                                              C.byref(q)))
        return q.value

    def tree_stats(self):
        """(pdf evaluations done by the level log-likelihood kernels since the node table was last set up, flags)."""
        pairs, flags = C.c_uint64(), C.c_int()
        self._check(self.lib.hgmm_tree_stats(self.h, C.byref(pairs), C.byref(flags)))
        return int(pairs.value), int(flags.value)

    def tree_node_complexity(self, T):
        out = np.empty(T)
        self._check(self.lib.hgmm_tree_node_complexity(self.h, _ptr(out)))
        return out

    # -- flat full-covariance EM -----------------------------------------------------------
    def fullcov_fit(self, J, ls, ld, init_mu, sig2, max_iters=1000):
        init_mu = np.ascontiguousarray(init_mu, dtype=np.float64)
        if init_mu.shape != (J, 3):
            raise ValueError("init_mu must be [%d,3]" % J)
        pi, mu, cov = np.empty(J), np.empty((J, 3)), np.empty((J, 3, 3))
        labels = np.empty(self.num_points, np.int32)
        q = np.zeros(int(max_iters))
        qlen = C.c_int()
        self._check(self.lib.hgmm_fullcov_fit(self.h, int(J), float(ls), float(ld), _ptr(init_mu), float(sig2),
                                              int(max_iters), _ptr(pi), _ptr(mu), _ptr(cov), _ptr(labels), _ptr(q),
                                              int(max_iters), C.byref(qlen)))
        return pi, mu, cov, labels, q[:qlen.value].copy()

    def fullcov_estep(self, pi, mu, cov):
        pi = np.ascontiguousarray(pi, dtype=np.float64)
        J = len(pi)
        mu = np.ascontiguousarray(mu, dtype=np.float64).reshape(J, 3)
        cov = np.ascontiguousarray(cov, dtype=np.float64).reshape(J, 3, 3)
        m0, m1, m2 = np.empty(J), np.empty((J, 3)), np.empty((J, 3, 3))
        labels = np.empty(self.num_points, np.int32)
        q = C.c_double()
        self._check(self.lib.hgmm_fullcov_estep(self.h, J, _ptr(pi), _ptr(mu), _ptr(cov), _ptr(m0), _ptr(m1), _ptr(m2),
                                                _ptr(labels), C.byref(q)))
        return m0, m1, m2, labels, q.value

    # -- KMeans initialiser (float64 points) -------------------------------------------------
    def kmeans_plusplus(self, k, first_id, rand_vals):
        """Greedy k-means++ seeding on the resident cloud.  rand_vals[(k-1), n_trials] are the
        host-drawn uniforms of steps 1..k-1 in draw order.  Returns (ids[k] int64, centres[k,3])."""
        k = int(k)
        if k > 1:
            rand_vals = np.ascontiguousarray(rand_vals, dtype=np.float64).reshape(k - 1, -1)
        n_trials = rand_vals.shape[1] if k > 1 else 1
        ids = np.empty(k, np.int64)
        centres = np.empty((k, 3))
        self._check(self.lib.hgmm_kmeans_plusplus(self.h, k, int(first_id), _ptr(rand_vals) if k > 1 else None,
                                                  int(n_trials), _ptr(ids), _ptr(centres)))
        return ids, centres

    def kmeans_step(self, centres, reset_labels=False):
        """One Lloyd assignment: returns (sums[k,3], counts[k], inertia, n_changed)."""
        centres = np.ascontiguousarray(centres, dtype=np.float64).reshape(-1, 3)
        k = len(centres)
        sums = np.empty((k, 4))
        inertia = C.c_double()
        changed = C.c_int64()
        self._check(self.lib.hgmm_kmeans_step(self.h, k, _ptr(centres), int(bool(reset_labels)), _ptr(sums),
                                              C.byref(inertia), C.byref(changed)))
        return sums[:, :3].copy(), sums[:, 3].copy(), inertia.value, changed.value

    def kmeans_lloyd(self, centres, max_iter, tol_abs, reset_labels=True):
        """Device-resident Lloyd loop.  Returns (centres, n_iter, strict, pending) where ``pending`` is None
        or, when an iteration met an empty cluster, that iteration's (sums[k,3], counts[k], n_changed)."""
        centres = np.array(centres, dtype=np.float64).reshape(-1, 3)
        k = len(centres)
        n_iter, strict, needs = C.c_int(), C.c_int(), C.c_int()
        sums = np.empty((k, 4))
        changed = C.c_int64()
        self._check(self.lib.hgmm_kmeans_lloyd(self.h, k, _ptr(centres), int(max_iter), float(tol_abs),
                                               int(bool(reset_labels)), C.byref(n_iter), C.byref(strict),
                                               C.byref(needs), _ptr(sums), C.byref(changed)))
        pending = (sums[:, :3].copy(), sums[:, 3].copy(), changed.value) if needs.value else None
        return centres, n_iter.value, bool(strict.value), pending

    def kmeans_labels(self, with_distances=False):
        labels = np.empty(self.num_points, np.int32)
        d2 = np.empty(self.num_points) if with_distances else None
        self._check(self.lib.hgmm_kmeans_labels(self.h, _ptr(labels), _ptr(d2) if with_distances else None))
        return (labels, d2) if with_distances else labels

    # -- L2 GMMReg ---------------------------------------------------------------------------
    def gauss_transform(self, centres, points, weights, h):
        """out[k, i] = sum_j weights[k, j] exp(-|points_i - centres_j|^2 / h^2); 1-D weights -> 1-D out."""
        centres = np.ascontiguousarray(centres, dtype=np.float64).reshape(-1, 3)
        points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        one = w.ndim == 1
        w2 = w.reshape(1, -1) if one else w
        if w2.shape[1] != len(centres):
            raise ValueError("weights must have one column per centre")
        out = np.empty((w2.shape[0], len(points)))
        self._check(self.lib.hgmm_gauss_transform(self.h, _ptr(centres), len(centres), _ptr(points), len(points),
                                                  _ptr(w2), w2.shape[0], float(h), _ptr(out)))
        return out[0] if one else out

    # -- multi-GPU ------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.hgmm_comm_unique_id(buf)
        if rc != 0:
            raise HgmmError("ncclGetUniqueId failed (%d)" % rc)
        return buf.raw

    def comm_init(self, nranks, rank, unique_id: bytes):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self.lib.hgmm_comm_init_rank(self.h, int(nranks), int(rank), buf))
        self.nranks, self.rank = int(nranks), int(rank)

    def comm_init_host(self, nranks, rank, name: str):
        """Host shared-memory communicator (tests on a single-GPU box; see include/hgmm.h)."""
        self._check(self.lib.hgmm_comm_init_host(self.h, int(nranks), int(rank), name.encode()))
        self.nranks, self.rank = int(nranks), int(rank)

    def comm_init_ipc(self, nranks, rank, name: str):
        """One-shot peer exchange over mapped peer memory (hipIpc; the ranks are processes of one node, see
        include/hgmm.h): every all-reduce is one kernel per rank, all ranks end with bitwise the same sums."""
        self._check(self.lib.hgmm_comm_init_ipc(self.h, int(nranks), int(rank), name.encode()))
        self.nranks, self.rank = int(nranks), int(rank)

    def comm_stats(self):
        """-> (all-reduces this context has enqueued on its communicator, level-iterations of hgmm_tree_build enqueued
        behind a level's stop): the cost of keeping the host ahead of a device-side stop rule under a communicator."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.lib.hgmm_comm_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def comm_destroy(self):
        self._check(self.lib.hgmm_comm_destroy(self.h))
        self.nranks, self.rank = 1, 0

    def allreduce(self, values, op="sum"):
        a = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._check(self.lib.hgmm_comm_allreduce_f64(self.h, _ptr(a), a.size, 1 if op == "max" else 0))
        return a

    # -- profiling ----------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self.lib.hgmm_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._check(self.lib.hgmm_profile_reset(self.h))

    def util_fill(self, arr, value=0.0, nontemporal=True, mode=0, grid_mult=0):
        flags = (1 if nontemporal else 0) | (int(mode) << 8) | (int(grid_mult) << 16)
        self._check(self.lib.hgmm_util_fill_f32(self.h, arr.ptr, arr.size, float(value), flags))

    def fullcov_phase_clocks(self, enable):
        """Arm (True) the phase clocks of the one-pass full-covariance kernels, or disarm them (False) and return the last
        launch's cycles [8 waves][4: phase A, phase B, phase C, barrier wait] (hgmm_fullcov_phase_clocks)."""
        if enable:
            self._check(self.lib.hgmm_fullcov_phase_clocks(self.h, 1, None))
            return None
        out = np.zeros((8, 4), np.int64)
        self._check(self.lib.hgmm_fullcov_phase_clocks(self.h, 0, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def profile_get(self, kernel):
        ms, n = C.c_double(), C.c_int64()
        self._check(self.lib.hgmm_profile_get(self.h, KERNEL_IDS[kernel], C.byref(ms), C.byref(n)))
        return ms.value, n.value


_default_ctx = None


def default_context() -> Context:
    """The context the module-level API works on: the calling THREAD's (``use_context``, what a replica worker runs
    under) or else the process-wide one on the rank's GPU (LOCAL_RANK, else device 0)."""
    global _default_ctx
    tl = getattr(_thread_ctx, "ctx", None)
    if tl is not None and getattr(tl, "h", None):
        return tl
    if _default_ctx is None or not getattr(_default_ctx, "h", None):
        with _default_ctx_lock:                       # two threads of a plain ThreadPoolExecutor must not both create one
            if _default_ctx is None or not getattr(_default_ctx, "h", None):
                _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_ctx


def set_default_context(ctx):
    global _default_ctx
    _default_ctx = ctx


_thread_ctx = threading.local()
_default_ctx_lock = threading.Lock()


@contextlib.contextmanager
def use_context(ctx):
    """Inside the block ``default_context()`` of THIS thread is ``ctx``: every module-level function of the mirrors
    (``train_gmm``, ``buildGMMTree``, ``registration_gmmreg`` ...) that is not handed a context runs on it.  One thread
    per context is how hgmm_amd.replicas fans independent pairs / frames out over the GPUs."""
    prev = getattr(_thread_ctx, "ctx", None)
    _thread_ctx.ctx = ctx
    try:
        yield ctx
    finally:
        _thread_ctx.ctx = prev
