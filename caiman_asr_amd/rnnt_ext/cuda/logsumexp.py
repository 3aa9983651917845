"""rnnt_ext.cuda.logsumexp — training/lib/csrc/logsumexp.cu:247-259 (pybind signature)."""
import torch

from caiman_asr_amd import _lib


def logsumexp(input: torch.Tensor, max_threads: int = 128, promote: bool = False) -> torch.Tensor:
    """LogSumExp along the last dimension of a rank-2 tensor with no memory overhead.

    Checks follow training/lib/csrc/logsumexp.cu:191-195.  Unlike the reference (which
    launches on the default stream, logsumexp.cu:146-176) this runs on the current stream.
    """
    if not input.is_cuda:
        raise RuntimeError("input must be a CUDA tensor")
    if input.dim() != 2:
        raise RuntimeError("input must be a 2D tensor")
    if input.size(1) > 0 and input.stride(1) != 1:
        raise RuntimeError("input must be contiguous in the last dimension")
    if input.size(0) > 1 and input.stride(0) < input.size(1):
        raise RuntimeError("input tensor must not alias itself")
    rows, n = input.shape
    out_dtype = _lib.acc_dtype(input.dtype) if promote else input.dtype
    out = torch.empty((rows,), dtype=out_dtype, device=input.device)
    stride = input.stride(0) if rows > 1 else max(n, 1)
    with _lib.timed("logsumexp"):
        _lib.check(_lib.lib().caiman_logsumexp(
            _lib.ptr(input), rows, n, stride, _lib.dtype_tag(input.dtype), _lib.ptr(out),
            _lib.dtype_tag(out_dtype), int(max_threads), _lib.stream()))
    return out
