"""Oracle (test infrastructure): ctypes front-end of oracle/pointops_oracle.c (numpy in / out)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpointops_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib():
    if not os.path.exists(_SO):
        build()
    return ctypes.CDLL(_SO)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def knn_query(nsample, xyz, offset, new_xyz=None, new_offset=None):
    if new_xyz is None:
        new_xyz, new_offset = xyz, offset
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    new_xyz = np.ascontiguousarray(new_xyz, dtype=np.float32)
    offset = np.ascontiguousarray(offset, dtype=np.int32)
    new_offset = np.ascontiguousarray(new_offset, dtype=np.int32)
    m = new_xyz.shape[0]
    idx = np.zeros((m, nsample), dtype=np.int32)
    dist2 = np.zeros((m, nsample), dtype=np.float32)
    _lib().oracle_knn_query(ctypes.c_int(m), ctypes.c_int(nsample), _ptr(xyz), _ptr(new_xyz), _ptr(offset),
                            _ptr(new_offset), _ptr(idx), _ptr(dist2))
    return idx, dist2


def grouping_forward(inp, idx):
    inp = np.ascontiguousarray(inp, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    m, ns = idx.shape
    c = inp.shape[1]
    out = np.zeros((m, ns, c), dtype=np.float32)
    _lib().oracle_grouping_forward(ctypes.c_int(m), ctypes.c_int(ns), ctypes.c_int(c), _ptr(inp), _ptr(idx), _ptr(out))
    return out


def interpolation_forward(inp, idx, weight):
    inp = np.ascontiguousarray(inp, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    n, k = idx.shape
    c = inp.shape[1]
    out = np.zeros((n, c), dtype=np.float32)
    _lib().oracle_interpolation_forward(ctypes.c_int(n), ctypes.c_int(c), ctypes.c_int(k), _ptr(inp), _ptr(idx),
                                        _ptr(weight), _ptr(out))
    return out


def grouping_backward(grad_out, idx, n):
    """grad_in (n, c) of grouping_forward; idx must not hold -1 (the reference kernel would write out of bounds:
    its Python wrapper routes -1 through an appended zero row, functions/grouping.py:41-63)."""
    grad_out = np.ascontiguousarray(grad_out, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    assert (idx >= 0).all()
    m, ns, c = grad_out.shape
    grad_in = np.zeros((n, c), dtype=np.float32)
    _lib().oracle_grouping_backward(ctypes.c_int(m), ctypes.c_int(ns), ctypes.c_int(c), _ptr(grad_out), _ptr(idx),
                                    _ptr(grad_in))
    return grad_in


def interpolation_backward(grad_out, idx, weight, m):
    grad_out = np.ascontiguousarray(grad_out, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    assert (idx >= 0).all()
    n, c = grad_out.shape
    k = idx.shape[1]
    grad_in = np.zeros((m, c), dtype=np.float32)
    _lib().oracle_interpolation_backward(ctypes.c_int(n), ctypes.c_int(c), ctypes.c_int(k), _ptr(grad_out), _ptr(idx),
                                         _ptr(weight), _ptr(grad_in))
    return grad_in
