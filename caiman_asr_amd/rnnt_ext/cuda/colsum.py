"""Column sums on the device: bias gradients of the LSTM and linear layers (csrc/colsum.hip).

The reference gets them from `dG.sum([0, 1])` (training/lib/src/rnnt_ext/custom_lstm/lstm.py:57); torch's generic
reduction reaches well under 1 TB/s on these [T*B, 4H] shapes, the streaming kernel here runs at HBM speed."""
import torch

from caiman_asr_amd import _lib


def colsum(x: torch.Tensor) -> torch.Tensor:
    """x [rows, cols] -> [cols], or x [batch, rows, cols] -> [batch, cols]; same dtype as x, fp32 accumulation."""
    squeeze = x.dim() == 2
    x3 = x.unsqueeze(0) if squeeze else x
    if x3.dim() != 3:
        raise RuntimeError("colsum expects a 2-D or 3-D tensor")
    batch, rows, cols = x3.shape
    ok = (x3.is_cuda and x3.dtype in (torch.float16, torch.bfloat16) and cols % 8 == 0 and rows > 0 and batch <= 65535
          and x3.stride(2) == 1 and x3.stride(1) == cols and (batch == 1 or x3.stride(0) % 8 == 0)
          and x3.data_ptr() % 16 == 0)
    if not ok:   # shapes / types outside the kernel's contract (fp32 runs, odd widths): torch's device reduction
        out = x3.sum(1)
        return out[0] if squeeze else out
    lib = _lib.lib()
    splits = int(lib.caiman_colsum_splits(batch, rows, cols))
    partial = torch.empty((batch, splits, cols), dtype=torch.float32, device=x.device)
    out = torch.empty((batch, cols), dtype=x.dtype, device=x.device)
    _lib.check(lib.caiman_colsum(_lib.ptr(x3), batch, rows, cols, x3.stride(0) if batch > 1 else rows * cols, _lib.ptr(out),
                                 _lib.ptr(partial), splits, _lib.dtype_tag(x.dtype), _lib.stream()))
    return out[0] if squeeze else out
