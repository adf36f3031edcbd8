"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes front-end to ``libmsda_oracle.so`` (the plain-C restatement in
``msda_oracle.c``; see that file's header for the reference file:line each
routine follows).  Importable only from ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg; the product package never imports it.

Inputs/outputs are numpy arrays (or anything ``np.ascontiguousarray`` accepts,
including CPU torch tensors via ``.numpy()``), fp32 or fp64, laid out exactly as
the reference op's tensors (value[N,S,M,D], shapes[L,2] int64, level_start[L]
int64, loc[N,Lq,M,L,P,2], attn[N,Lq,M,L,P]).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmsda_oracle.so")
_lib = None


def build(force=False):
    """Compile the C oracle with gcc (no-op when the .so is newer than the source)."""
    src = os.path.join(_HERE, "msda_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libmsda_oracle.so"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.msda_oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads():
    return int(_load().msda_oracle_num_threads())


def set_num_threads(n):
    _load().msda_oracle_set_num_threads(ctypes.c_int(int(n)))


def _prep(value, shapes, level_start, loc, attn):
    value = np.ascontiguousarray(value)
    dt = value.dtype
    if dt not in (np.float32, np.float64):
        raise TypeError("oracle handles float32/float64 only, got %s" % dt)
    loc = np.ascontiguousarray(loc, dtype=dt)
    attn = np.ascontiguousarray(attn, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    level_start = np.ascontiguousarray(level_start, dtype=np.int64)
    N, S, M, D = value.shape
    _, Lq, M2, L, P, two = loc.shape
    assert M2 == M and two == 2 and attn.shape == (N, Lq, M, L, P)
    assert shapes.shape == (L, 2) and level_start.shape == (L,)
    assert int((shapes[:, 0] * shapes[:, 1]).sum()) == S
    suf = "f32" if dt == np.float32 else "f64"
    return value, shapes, level_start, loc, attn, (N, S, M, D, L, Lq, P), suf


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def forward(value, shapes, level_start, loc, attn):
    """-> out[N, Lq, M*D]   (reference: ms_deform_im2col_cuda.cuh:237-299)."""
    value, shapes, level_start, loc, attn, dims, suf = _prep(value, shapes, level_start, loc, attn)
    N, S, M, D, L, Lq, P = dims
    out = np.empty((N, Lq, M * D), dtype=value.dtype)
    fn = getattr(_load(), "msda_oracle_forward_" + suf)
    fn(_p(value), _p(shapes), _p(level_start), _p(loc), _p(attn),
       *[ctypes.c_int(x) for x in dims], _p(out))
    return out


def backward(grad_out, value, shapes, level_start, loc, attn):
    """-> (grad_value, grad_loc, grad_attn)   (reference: ms_deform_im2col_cuda.cuh:87-159, 301-403)."""
    value, shapes, level_start, loc, attn, dims, suf = _prep(value, shapes, level_start, loc, attn)
    N, S, M, D, L, Lq, P = dims
    grad_out = np.ascontiguousarray(grad_out, dtype=value.dtype)
    assert grad_out.shape == (N, Lq, M * D)
    gv = np.empty_like(value)
    gl = np.empty_like(loc)
    ga = np.empty_like(attn)
    fn = getattr(_load(), "msda_oracle_backward_" + suf)
    fn(_p(grad_out), _p(value), _p(shapes), _p(level_start), _p(loc), _p(attn),
       *[ctypes.c_int(x) for x in dims], _p(gv), _p(gl), _p(ga))
    return gv, gl, ga
