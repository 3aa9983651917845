"""rnnt_ext.cuda.lstm — training/lib/csrc/lstm.cu:411-456 (pybind signatures),
:353-405 (checks + dispatch).  Ops mutate caller-allocated tensors in place."""
import torch

from caiman_asr_amd import _lib


_WORK = {}


def _workspace(B, H, dtype, device, backward):
    """Scratch for the MFMA path (tiled weights + operand ring), cached per shape: the kernels of
    one stream run in order, so consecutive layers can reuse it."""
    if dtype not in (torch.float16, torch.bfloat16) or H % 32 != 0:
        return None
    n = _lib.lib().caiman_lstm_workspace_elems(B, H, int(backward))
    key = (device, dtype, torch.cuda.current_stream(device).cuda_stream)
    buf = _WORK.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(n, dtype=dtype, device=device)
        _WORK[key] = buf
    return buf


def _step_bytes(B, H, es, backward):
    """Algorithmic HBM bytes of ONE recurrent step: the recurrent weights
    (not resident across kernel boundaries) + every per-step operand once."""
    if backward:  # R + dG[t+1] in + gates + c[t], c[t+1] + delta + dG[t] out ; dC (f32) in/out
        return 4 * H * H * es + B * (4 * H + 4 * H + 2 * H + H + 4 * H) * es + 2 * B * H * 4
    return 4 * H * H * es + B * (H + 4 * H + H + 4 * H + 2 * H) * es  # R + h in + gates in/out + c in/out + y out


def _dims(c):
    # lstm.cu:226-228: c is [T+1, B, H] or [T+1, H]
    if c.dim() == 3:
        return c.size(1), c.size(2)
    return 1, c.size(1)


def _fwd(R, gates, c, y, hard):
    for t, n in ((R, "R"), (gates, "gates"), (c, "c"), (y, "y")):
        _lib.check_input(t, n)
    if not (R.dtype == gates.dtype == c.dtype == y.dtype):
        raise RuntimeError("R, gates, c and y must share one dtype")
    T = gates.size(0)
    B, H = _dims(c)
    if R.shape != (4 * H, H) or gates.numel() != T * B * 4 * H or c.size(0) != T + 1 or y.shape != c.shape:
        raise RuntimeError(f"inconsistent LSTM shapes R{list(R.shape)} gates{list(gates.shape)} "
                           f"c{list(c.shape)} y{list(y.shape)}")
    work = _workspace(B, H, gates.dtype, gates.device, False)
    with _lib.timed("lstm_fwd", T, T * _step_bytes(B, H, gates.element_size(), False)):
        _lib.check(_lib.lib().caiman_lstm_fused_fwd(
            _lib.ptr(R), _lib.ptr(gates), _lib.ptr(c), _lib.ptr(y), _lib.ptr(work) if work is not None else None,
            T, B, H, _lib.dtype_tag(gates.dtype), int(hard), _lib.stream()))


def _bwd(R, gates, c, delta, dG, hard):
    if not delta.is_cuda:
        raise RuntimeError("delta must be a CUDA tensor")
    for t, n in ((R, "R"), (gates, "gates"), (c, "c"), (dG, "dG")):
        _lib.check_input(t, n)
    T = delta.size(0)
    B, H = _dims(c)
    if T == 0:
        return
    d = delta if delta.dim() == 3 else delta.unsqueeze(1)
    if d.stride(2) != 1 and H > 1:
        d = d.contiguous()  # the reference always copies (lstm.cu:394-396)
    dC = torch.empty((B, H), dtype=_lib.acc_dtype(gates.dtype), device=gates.device)
    work = _workspace(B, H, gates.dtype, gates.device, True)
    with _lib.timed("lstm_bwd", T, T * _step_bytes(B, H, gates.element_size(), True)):
        _lib.check(_lib.lib().caiman_lstm_fused_bwd(
            _lib.ptr(R), _lib.ptr(gates), _lib.ptr(c), _lib.ptr(d), d.stride(0), d.stride(1), _lib.ptr(dG),
            _lib.ptr(dC), _lib.ptr(work) if work is not None else None, T, B, H, _lib.dtype_tag(gates.dtype),
            int(hard), _lib.stream()))


def lstm_fused_fwd_soft(R, gates, c, y):
    """Compute the LSTM forward pass with soft activation functions."""
    _fwd(R, gates, c, y, False)


def lstm_fused_fwd_hard(R, gates, c, y):
    """Compute the LSTM forward pass with hard activation functions."""
    _fwd(R, gates, c, y, True)


def lstm_fused_bwd_soft(R, gates, c, delta, dG):
    """Compute the LSTM backward pass with soft activation functions."""
    _bwd(R, gates, c, delta, dG, False)


def lstm_fused_bwd_hard(R, gates, c, delta, dG):
    """Compute the LSTM backward pass with hard activation functions."""
    _bwd(R, gates, c, delta, dG, True)
