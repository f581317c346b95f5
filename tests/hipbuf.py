"""Device buffers for the GPU tests without importing torch: the handful of calls the tests make on torch tensors (zeros / full / empty /
from_numpy(...).to(...), .data_ptr(), .cpu().numpy()), implemented on hipMalloc / hipMemcpy through ctypes.  `from tests import hipbuf as torch`
keeps the test bodies as they were.  (On a cold image the first `import torch` costs minutes — tests/conftest.py — and the tests of the C ABI only
ever used torch as an allocator.)"""
import ctypes as C

import numpy as np

int64, int32, uint8 = np.int64, np.int32, np.uint8
_hip = None


def _lib():
    global _hip
    if _hip is None:
        for name in ("/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("libamdhip64.so not found")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    return _hip


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: hipError {rc}")


class _Host:
    def __init__(self, a):
        self._a = a

    def numpy(self):
        return self._a


class Tensor:
    """a flat device array of a numpy dtype"""

    def __init__(self, n, dtype):
        self.dtype = np.dtype(dtype)
        self.n = int(n)
        self._ptr = C.c_void_p()
        _check(_lib().hipMalloc(C.byref(self._ptr), max(self.n * self.dtype.itemsize, 8)), "hipMalloc")

    def data_ptr(self):
        return self._ptr.value

    def numel(self):
        return self.n

    def _upload(self, a):
        a = np.ascontiguousarray(a)
        _check(_lib().hipMemcpy(self._ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, 1), "hipMemcpy H2D")
        return self

    def cpu(self):
        out = np.empty(self.n, dtype=self.dtype)
        _check(_lib().hipDeviceSynchronize(), "hipDeviceSynchronize")
        _check(_lib().hipMemcpy(out.ctypes.data_as(C.c_void_p), self._ptr, out.nbytes, 2), "hipMemcpy D2H")
        return _Host(out)

    def __del__(self):
        try:
            if self._ptr:
                _lib().hipFree(self._ptr)
        except Exception:
            pass


def _count(shape):
    return int(np.prod(shape)) if isinstance(shape, (tuple, list)) else int(shape)


def zeros(shape, dtype=int64, device=None):
    t = Tensor(_count(shape), dtype)
    _check(_lib().hipMemset(t._ptr, 0, max(t.n * t.dtype.itemsize, 8)), "hipMemset")
    _check(_lib().hipDeviceSynchronize(), "hipDeviceSynchronize")
    return t


def empty(shape, dtype=int64, device=None):
    return zeros(shape, dtype, device)


def full(shape, value, dtype=int64, device=None):
    n = _count(shape)
    return Tensor(n, dtype)._upload(np.full(n, value, dtype=dtype))


class _Pending:
    def __init__(self, a):
        self._a = np.ascontiguousarray(a)

    def to(self, device=None):
        return Tensor(self._a.size, self._a.dtype)._upload(self._a.reshape(-1))


def from_numpy(a):
    return _Pending(a)


class cuda:  # torch.cuda.synchronize()
    @staticmethod
    def synchronize():
        _check(_lib().hipDeviceSynchronize(), "hipDeviceSynchronize")
