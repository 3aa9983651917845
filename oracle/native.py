"""ctypes front-end of oracle/rnnt_oracle.c (numpy float64 in / out).

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "liboracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "rnnt_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "build/liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(ctypes.c_void_p)


I64 = ctypes.c_int64
F64 = ctypes.c_double


def logsumexp(x):
    """Row-wise logsumexp of a 2-D array (training/lib/csrc/logsumexp.cu:65-105)."""
    x, px = _d(x)
    rows, n = x.shape
    out = np.empty(rows, dtype=np.float64)
    lib().oracle_logsumexp(px, I64(rows), I64(n), I64(n), out.ctypes.data_as(ctypes.c_void_p))
    return out


def _norm_idx(eos_idx, star_idx):
    # training/lib/src/rnnt_ext/transducer/loss.py:184-194
    return (-1 if eos_idx is None else int(eos_idx)), (-2 if star_idx is None else int(star_idx))


def transducer_forward(x, label, f_len, y_len, blank_idx, batch_offset=None, max_f_len=None,
                       delay_penalty=0.0, eos_penalty=0.0, eos_idx=None, star_lam=0.0,
                       star_idx=None, denom=None):
    """alpha, beta, loss, denom.  x: [B,T,U,V] padded or [rows,V] packed (batch_offset given).

    ``star_lam`` is log(star_penalty), as passed to the kernel
    (training/lib/src/rnnt_ext/transducer/loss.py:145).
    """
    packed = batch_offset is not None
    x, px = _d(x)
    V = x.shape[-1]
    label, pl = _i32(label)
    f_len, pf = _i32(f_len)
    y_len, py = _i32(y_len)
    B = label.shape[0]
    max_glen = label.shape[1] + 1
    max_flen = int(max_f_len) if packed else x.shape[1]
    if denom is None:
        denom = logsumexp(x.reshape(-1, V)).reshape(x.shape[:-1])
    denom, pd = _d(denom)
    if packed:
        bo, pbo = _i64(batch_offset)
    else:
        bo, pbo = None, ctypes.c_void_p(0)
    eos_idx, star_idx = _norm_idx(eos_idx, star_idx)
    alpha = np.full((B, max_flen, max_glen), np.nan)
    beta = np.full((B, max_flen, max_glen), np.nan)
    loss = np.empty(B)
    lib().oracle_transducer_forward(
        px, pd, pl, pf, py, pbo, I64(B), I64(max_flen), I64(max_glen), I64(V), F64(delay_penalty),
        I64(blank_idx), F64(eos_penalty), I64(eos_idx), F64(star_lam), I64(star_idx),
        ctypes.c_int(int(packed)), alpha.ctypes.data_as(ctypes.c_void_p),
        beta.ctypes.data_as(ctypes.c_void_p), loss.ctypes.data_as(ctypes.c_void_p))
    return alpha, beta, loss, denom


def transducer_backward(x, denom, loss_grad, alpha, beta, label, f_len, y_len, blank_idx,
                        batch_offset=None, delay_penalty=0.0, eos_penalty=0.0, eos_idx=None,
                        star_lam=0.0, star_idx=None):
    packed = batch_offset is not None
    x, px = _d(x)
    V = x.shape[-1]
    denom, pd = _d(denom)
    loss_grad, pg = _d(loss_grad)
    alpha, pa = _d(np.nan_to_num(alpha))
    beta, pb = _d(np.nan_to_num(beta))
    label, pl = _i32(label)
    f_len, pf = _i32(f_len)
    y_len, py = _i32(y_len)
    B, max_flen, max_glen = alpha.shape
    if packed:
        bo, pbo = _i64(batch_offset)
    else:
        bo, pbo = None, ctypes.c_void_p(0)
    eos_idx, star_idx = _norm_idx(eos_idx, star_idx)
    gx = np.empty_like(x)
    lib().oracle_transducer_backward(
        px, pd, pg, pa, pb, pf, py, pl, pbo, I64(B), I64(max_flen), I64(max_glen), I64(V),
        F64(delay_penalty), I64(blank_idx), F64(eos_penalty), I64(eos_idx), F64(star_lam),
        I64(star_idx), ctypes.c_int(int(packed)), gx.ctypes.data_as(ctypes.c_void_p))
    return gx


def lstm_fwd(R, gates, c0, y0, hard=False):
    """Returns (activated gates [T,B,4H], c [T+1,B,H], y [T+1,B,H])."""
    R, pR = _d(R)
    g = np.array(gates, dtype=np.float64, order="C", copy=True)
    T, B, H4 = g.shape
    H = H4 // 4
    c = np.zeros((T + 1, B, H))
    y = np.zeros((T + 1, B, H))
    c[0] = c0
    y[0] = y0
    lib().oracle_lstm_fwd(pR, g.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p),
                          y.ctypes.data_as(ctypes.c_void_p), I64(T), I64(B), I64(H),
                          ctypes.c_int(int(hard)))
    return g, c, y


def lstm_bwd(R, gates_act, c, delta, hard=False):
    """Returns dG [T,B,4H] (and the modified partials)."""
    R, pR = _d(R)
    g, pg = _d(gates_act)
    c, pc = _d(c)
    partials = np.array(delta, dtype=np.float64, order="C", copy=True)
    T, B, H = partials.shape
    dG = np.empty((T, B, 4 * H))
    lib().oracle_lstm_bwd(pR, pg, pc, partials.ctypes.data_as(ctypes.c_void_p),
                          dG.ctypes.data_as(ctypes.c_void_p), I64(T), I64(B), I64(H),
                          ctypes.c_int(int(hard)))
    return dG, partials
