"""rnnt_ext.transducer.loss — RNN-T transducer loss module + autograd function on the gfx950 C-ABI library.

Drop-in for the public surface of training/lib/src/rnnt_ext/transducer/loss.py:43-253 — `TransducerLoss(packed_input)`
with the same `forward` keywords and defaults, `TransducerLossFunc.apply` with the same positional order — so the
reference's callers (training/caiman_asr_train/rnnt/loss.py:68-127) and tests read unchanged.  The body is this
repo's own: one immutable record carries the call's scalars from forward to backward, the sentinel / sanity rules
live in one helper, and the compute is caiman_logsumexp + caiman_transducer_loss_forward / _backward.
"""
import math
from dataclasses import dataclass
from typing import List, Optional

import torch

import caiman_asr_amd.rnnt_ext.cuda.logsumexp as logsumexp_cu
import caiman_asr_amd.rnnt_ext.cuda.transducer_loss as transducer_loss_cu

_NO_EOS, _NO_STAR = -1, -2   # sentinels the kernels understand as "feature off" (transducer_loss.cu:396-501)

# The backward kernel can sum the gradient rows it writes (caiman_transducer_loss_backward_colsum): that is the bias
# gradient of the projection that produced the logits, which autograd would otherwise obtain by one more pass over the
# whole gradient tensor (5.3 GB at B = 32).  The sums are left here for the projection's backward to pick up
# (train_utils/overlap.py::_LinearTransposedBackward); nothing changes for a consumer that does not look.
FUSE_BIAS_GRADIENT = True
_latest_colsum = None   # (data_ptr, shape, dtype, version counter, column sums) of the most recent x_grad


def take_bias_gradient(dy: torch.Tensor):
    """Column sums of `dy` if `dy` IS the gradient tensor the last loss backward produced -- same storage, shape and
    dtype, and NOT written since (its version counter has not moved: a tensor hook that rescales the logits' gradient
    in place, or autograd accumulating a second consumer's gradient into it, would leave the sums stale) -- else None.
    An entry is good for one backward pass: the next loss forward clears it."""
    global _latest_colsum
    ent, _latest_colsum = _latest_colsum, None
    if (ent is not None and ent[0] == dy.data_ptr() and ent[1] == tuple(dy.shape) and ent[2] == dy.dtype
            and ent[3] == dy._version):
        return ent[4]
    return None


# The projection that produced the logits may have computed their row normalisers already (csrc/joint_gemm.hip: log-sum-exp
# in the GEMM epilogue, train_utils/overlap.py).  Same hand-over discipline as the bias gradient above: the entry names
# the tensor by storage, shape, dtype and version counter, is good for ONE loss forward, and nothing changes for a caller
# that never offers.
_latest_row_lse = None


def offer_row_lse(logits: torch.Tensor, lse: torch.Tensor):
    """Called by the projection's forward.  The entry holds a WEAK reference to the logits' storage: once that memory is
    released (a projection whose output never reached a loss: evaluation, an exception), the entry is dead even if the
    allocator hands the same address, shape and dtype to a later tensor."""
    global _latest_row_lse
    import weakref

    _latest_row_lse = (logits.data_ptr(), tuple(logits.shape), logits.dtype, logits._version, lse,
                       weakref.ref(logits.untyped_storage()))


def clear_row_lse():
    """Every projection forward starts with this (also the library path, which offers nothing)."""
    global _latest_row_lse
    _latest_row_lse = None


def take_row_lse(x: torch.Tensor):
    """log-sum-exp of every row of `x` if the projection that wrote `x` left it here (and `x` was not written since)."""
    global _latest_row_lse
    ent, _latest_row_lse = _latest_row_lse, None
    if (ent is not None and ent[0] == x.data_ptr() and ent[1] == tuple(x.shape) and ent[2] == x.dtype
            and ent[3] == x._version and ent[4].numel() == x.numel() // x.shape[-1]):
        st = ent[5]()
        if st is not None and st.data_ptr() == x.untyped_storage().data_ptr():
            return ent[4]
    return None


@dataclass(frozen=True)
class _LossCall:
    """Scalars of one loss evaluation, in the units the kernels take them."""
    delay_penalty: float
    max_f_len: int
    blank_idx: int
    eos_penalty: float
    eos_idx: int
    log_star_penalty: float
    star_idx: int
    packed: bool

    def kernel_args(self):
        return (self.delay_penalty, self.max_f_len, self.blank_idx, self.eos_penalty, self.eos_idx,
                self.log_star_penalty, self.star_idx, self.packed)


def _special_indices(blank_idx: int, eos_idx: Optional[int], star_idx: Optional[int]):
    """None -> the kernels' sentinels; blank, EOS and star must be three different classes."""
    eos = _NO_EOS if eos_idx is None else int(eos_idx)
    star = _NO_STAR if star_idx is None else int(star_idx)
    for name, idx in (("eos_idx", eos), ("star_idx", star)):
        if idx == blank_idx:
            raise AssertionError(f"{name} must be different from blank_idx")
    if star == eos:
        raise AssertionError("star_idx must be different from eos_idx")
    return eos, star


class TransducerLoss(torch.nn.Module):
    """Per-utterance transducer loss (Graves 2012) with delay / EOS / star penalties.

    packed_input: logits arrive as [sum_b T_b*(U_b+1), V] (don't-care cells removed) instead of [B, T, U+1, V];
    `batch_offset` (cumulative cell counts) and `max_f_len` are then required.
    """

    def __init__(self, packed_input: bool = False):
        super().__init__()
        self.packed_input = packed_input
        self.dummy_batch_offset = torch.empty(0)
        # The reference checks its length tensors on the host in every call (three device syncs per step).  Same
        # checks by default; a training loop that has validated its loader clears this after the first step.
        self.validate_lengths = True

    def _check_lengths(self, label, f_len, y_len, max_f_len):
        lo_f, lo_y, hi_y, hi_f = (int(v) for v in torch.stack(
            [f_len.min(), y_len.min(), y_len.max(), f_len.max()]).tolist())    # one sync for all four
        if lo_f < 1:
            raise AssertionError("f_len must be non-negative")
        if lo_y < 0:
            raise AssertionError("y_len must be non-negative")
        if hi_y > label.size(1):
            raise AssertionError("y_len must be less than label length")
        if self.packed_input and max_f_len != hi_f:
            raise AssertionError(f"max_f_len ({max_f_len}) must equal f_len.max() ({hi_f})")

    def forward(self, x: torch.Tensor, label: torch.Tensor, f_len: torch.Tensor, y_len: torch.Tensor, blank_idx: int,
                eos_idx: Optional[int] = None, star_idx: Optional[int] = None,
                batch_offset: Optional[torch.Tensor] = None, max_f_len: Optional[int] = None,
                debug_list: Optional[List[torch.Tensor]] = None, delay_penalty: float = 0.0, eos_penalty: float = 0.0,
                star_penalty: float = 1.0) -> torch.Tensor:
        """-> loss [B].  x: logits (padded 4-D or packed 2-D); label [B, Umax] int32; f_len / y_len [B] int32;
        `debug_list == []` receives [alpha, beta]."""
        if x.dim() not in (2, 4):
            raise AssertionError("Shape (B, T, U, H) or (*, H)")
        if self.packed_input:
            if batch_offset is None or max_f_len is None:
                raise Exception("Please specify batch_offset and max_f_len when packing is enabled")
        else:
            batch_offset, max_f_len = self.dummy_batch_offset, x.size(1)
        if self.validate_lengths:
            self._check_lengths(label, f_len, y_len, max_f_len)
        return TransducerLossFunc.apply(x, label, f_len, y_len, batch_offset, delay_penalty, max_f_len, blank_idx,
                                        eos_penalty, eos_idx, math.log(star_penalty), star_idx, debug_list,
                                        self.packed_input)


class TransducerLossFunc(torch.autograd.Function):
    """apply(x, label, f_len, y_len, batch_offset, delay_penalty, max_f_len, blank_idx, eos_penalty, eos_idx,
    star_penalty (already a log), star_idx, debug_list, packed_input) -> loss [B]; gradient flows to x only."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, label, f_len, y_len, batch_offset, delay_penalty, max_f_len, blank_idx, eos_penalty, eos_idx,
                star_penalty, star_idx, debug_list, packed_input):
        global _latest_colsum
        _latest_colsum = None   # an untaken entry of an earlier pass must not meet a recycled allocation
        eos, star = _special_indices(blank_idx, eos_idx, star_idx)
        call = _LossCall(float(delay_penalty), int(max_f_len), int(blank_idx), float(eos_penalty), eos,
                         float(star_penalty), star, bool(packed_input))
        if call.packed:
            rows = x
        else:
            if not x.is_contiguous():
                raise AssertionError("activations must be contiguous or packed")
            rows = x.view(-1, x.shape[-1])
        denom = take_row_lse(x)
        if denom is None:
            denom = logsumexp_cu.logsumexp(rows, 128, True)    # log-normaliser of every lattice cell
        denom = denom.view(x.shape[:-1])
        alpha, beta, loss = transducer_loss_cu.forward(x, denom, label, f_len, y_len, batch_offset, *call.kernel_args())
        if debug_list is not None and len(debug_list) == 0:
            debug_list.extend((alpha, beta))
        ctx.save_for_backward(x, denom, alpha, beta, f_len, y_len, label, batch_offset)
        ctx.call = call
        return loss

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, loss_grad):
        global _latest_colsum
        x, denom, alpha, beta, f_len, y_len, label, batch_offset = ctx.saved_tensors
        if FUSE_BIAS_GRADIENT and transducer_loss_cu.colsum_supported(x):
            x_grad, colsum = transducer_loss_cu.backward_colsum(x, denom, loss_grad.contiguous(), alpha, beta, f_len, y_len,
                                                                label, batch_offset, *ctx.call.kernel_args())
            _latest_colsum = (x_grad.data_ptr(), tuple(x_grad.shape), x_grad.dtype, x_grad._version, colsum)
        else:
            x_grad = transducer_loss_cu.backward(x, denom, loss_grad.contiguous(), alpha, beta, f_len, y_len, label,
                                                 batch_offset, *ctx.call.kernel_args())
        return (x_grad,) + (None,) * 13
