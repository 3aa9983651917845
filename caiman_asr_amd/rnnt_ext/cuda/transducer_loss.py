"""rnnt_ext.cuda.transducer_loss — training/lib/csrc/transducer_loss.cu:594-600 (pybind),
:396-501 (forward host code), :503-590 (backward host code)."""
import torch

from caiman_asr_amd import _lib


def forward(x, denom, label, aud_len, txt_len, batch_offset, dp_lam, max_flen, blank_idx, eos_lam,
            eos_idx, star_lam, star_idx, packed):
    """-> [alpha, beta, loss]; alpha/beta [B, max_flen, max_glen], loss [B], accumulate dtype."""
    for t, n in ((x, "x"), (denom, "denom"), (label, "label"), (aud_len, "aud_len"),
                 (txt_len, "txt_len")):
        _lib.check_input(t, n)
    if packed:
        _lib.check_input(batch_offset, "batch_offset")
    batch = label.size(0)
    max_glen = label.size(1) + 1
    V = x.size(-1)
    acc = _lib.acc_dtype(x.dtype)
    if denom.dtype != acc:
        raise RuntimeError(f"denom must be {acc} for x of {x.dtype}")
    if label.dtype != torch.int32 or aud_len.dtype != torch.int32 or txt_len.dtype != torch.int32:
        raise RuntimeError("label, aud_len and txt_len must be int32")
    if packed and batch_offset.dtype != torch.int64:
        raise RuntimeError("batch_offset must be int64")
    alpha = torch.empty((batch, max_flen, max_glen), dtype=acc, device=x.device)
    beta = torch.empty((batch, max_flen, max_glen), dtype=acc, device=x.device)
    loss = torch.empty((batch,), dtype=acc, device=x.device)
    with _lib.timed("loss_fwd"):
        _lib.check(_lib.lib().caiman_transducer_loss_forward(
            _lib.ptr(x), _lib.ptr(denom), _lib.ptr(label), _lib.ptr(aud_len), _lib.ptr(txt_len),
            _lib.ptr(batch_offset) if packed else None, batch, int(max_flen), max_glen, V, float(dp_lam),
            int(blank_idx), float(eos_lam), int(eos_idx), float(star_lam), int(star_idx), int(bool(packed)),
            _lib.dtype_tag(x.dtype), _lib.ptr(alpha), _lib.ptr(beta), _lib.ptr(loss), _lib.stream()))
    return [alpha, beta, loss]


def backward(x, denom, loss_grad, alpha, beta, aud_len, txt_len, label, batch_offset, dp_lam,
             max_flen, blank_idx, eos_lam, eos_idx, star_lam, star_idx, packed):
    """-> x_grad, same shape / dtype as x."""
    for t, n in ((x, "x"), (denom, "denom"), (label, "label"), (loss_grad, "loss_grad"),
                 (alpha, "alpha"), (beta, "beta"), (aud_len, "aud_len"), (txt_len, "txt_len")):
        _lib.check_input(t, n)
    if packed:
        _lib.check_input(batch_offset, "batch_offset")
    batch = label.size(0)
    max_glen = label.size(1) + 1
    V = x.size(-1)
    acc = _lib.acc_dtype(x.dtype)
    if loss_grad.dtype != acc:
        loss_grad = loss_grad.to(acc)
    x_grad = torch.empty_like(x)
    total_rows = x.numel() // V if V > 0 else 0
    with _lib.timed("loss_bwd"):
        _lib.check(_lib.lib().caiman_transducer_loss_backward(
            _lib.ptr(x), _lib.ptr(denom), _lib.ptr(loss_grad), _lib.ptr(alpha), _lib.ptr(beta),
            _lib.ptr(aud_len), _lib.ptr(txt_len), _lib.ptr(label),
            _lib.ptr(batch_offset) if packed else None, batch, int(max_flen), max_glen, V, total_rows,
            float(dp_lam), int(blank_idx), float(eos_lam), int(eos_idx), float(star_lam), int(star_idx),
            int(bool(packed)), _lib.dtype_tag(x.dtype), _lib.ptr(x_grad), _lib.stream()))
    return x_grad


COLSUM_ROWS_PER_BLOCK = 64


def colsum_supported(x) -> bool:
    """The fused column sums need 16-byte aligned rows and at most 16 column chunks per lane of an eight-wave workgroup
    (V <= 65536 for 16-bit logits; four waves up to 5 chunks per lane, csrc/transducer_loss.hip); anything else takes the
    plain backward and a reduction."""
    V, es = x.shape[-1], x.element_size()
    return (x.is_cuda and V > 0 and (V * es) % 16 == 0 and x.data_ptr() % 16 == 0
            and ((V * es // 16 + 7) // 8 + 63) // 64 <= 16)


def backward_colsum(x, denom, loss_grad, alpha, beta, aud_len, txt_len, label, batch_offset, dp_lam,
                    max_flen, blank_idx, eos_lam, eos_idx, star_lam, star_idx, packed):
    """`backward` that also returns the column sums of x_grad ([V], fp32): the bias gradient of the projection that
    produced x, taken inside the same pass (include/caiman_rnnt.h, caiman_transducer_loss_backward_colsum) instead of
    by a separate reduction over the whole gradient."""
    if not colsum_supported(x):
        x_grad = backward(x, denom, loss_grad, alpha, beta, aud_len, txt_len, label, batch_offset, dp_lam, max_flen,
                          blank_idx, eos_lam, eos_idx, star_lam, star_idx, packed)
        return x_grad, x_grad.reshape(-1, x.size(-1)).float().sum(0)
    for t, n in ((x, "x"), (denom, "denom"), (label, "label"), (loss_grad, "loss_grad"),
                 (alpha, "alpha"), (beta, "beta"), (aud_len, "aud_len"), (txt_len, "txt_len")):
        _lib.check_input(t, n)
    if packed:
        _lib.check_input(batch_offset, "batch_offset")
    batch = label.size(0)
    max_glen = label.size(1) + 1
    V = x.size(-1)
    acc = _lib.acc_dtype(x.dtype)
    if loss_grad.dtype != acc:
        loss_grad = loss_grad.to(acc)
    x_grad = torch.empty_like(x)
    total_rows = x.numel() // V if V > 0 else 0
    nblk = (total_rows + COLSUM_ROWS_PER_BLOCK - 1) // COLSUM_ROWS_PER_BLOCK
    if total_rows == 0:
        return x_grad, torch.zeros(V, dtype=torch.float32, device=x.device)
    lead = (nblk * V + 7) // 8 * 8     # keeps the row descriptors behind the partial sums 32-byte aligned
    ws = torch.empty(lead + 8 * total_rows, dtype=torch.float32, device=x.device)
    partial = ws[:nblk * V].view(nblk, V)
    with _lib.timed("loss_bwd"):
        _lib.check(_lib.lib().caiman_transducer_loss_backward_colsum(
            _lib.ptr(x), _lib.ptr(denom), _lib.ptr(loss_grad), _lib.ptr(alpha), _lib.ptr(beta),
            _lib.ptr(aud_len), _lib.ptr(txt_len), _lib.ptr(label),
            _lib.ptr(batch_offset) if packed else None, batch, int(max_flen), max_glen, V, total_rows,
            float(dp_lam), int(blank_idx), float(eos_lam), int(eos_idx), float(star_lam), int(star_idx),
            int(bool(packed)), _lib.dtype_tag(x.dtype), _lib.ptr(x_grad), _lib.ptr(ws), COLSUM_ROWS_PER_BLOCK,
            _lib.stream()))
    return x_grad, partial.sum(0)
