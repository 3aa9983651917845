"""TransducerJoint — same constructor / call signature as apex.contrib.transducer.TransducerJoint
as used by the reference (training/caiman_asr_train/rnnt/model.py:228-238,425-434), backed by
csrc/joint.hip."""
import torch

from caiman_asr_amd import _lib


class _JointFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, g, f_len, g_len, batch_offset, packed_batch, pack_output, relu, dropout_p, seed):
        for t, n in ((f, "f"), (g, "g")):
            _lib.check_input(t, n)
        if f.dtype != g.dtype:
            raise RuntimeError(f"f and g must share a dtype, got {f.dtype} and {g.dtype}")
        B, T, H = f.shape
        U = g.shape[1]
        f_len = f_len.to(torch.int32).contiguous()
        g_len = g_len.to(torch.int32).contiguous()
        if pack_output:
            if batch_offset is None:
                raise RuntimeError("Please specify batch_offset when packing is enabled")
            batch_offset = batch_offset.to(torch.int64).contiguous()
            rows = int(packed_batch)
            out = torch.empty((rows, H), dtype=f.dtype, device=f.device)
        else:
            rows = B * T * U
            out = torch.empty((B, T, U, H), dtype=f.dtype, device=f.device)
        with _lib.timed("joint_fwd"):
            _lib.check(_lib.lib().caiman_joint_forward(
                _lib.ptr(f), _lib.ptr(g), _lib.ptr(f_len), _lib.ptr(g_len),
                _lib.ptr(batch_offset) if pack_output else None, B, T, U, H, rows, int(pack_output), int(relu),
                float(dropout_p), int(seed), _lib.dtype_tag(f.dtype), _lib.ptr(out), _lib.stream()))
        mode = 1 if relu else (2 if dropout_p > 0 else 0)
        ctx.save_for_backward(out if mode else None, f_len, g_len, batch_offset if pack_output else None)
        ctx.meta = (B, T, U, H, pack_output, mode, 1.0 / (1.0 - dropout_p) if dropout_p > 0 else 1.0)
        return out

    @staticmethod
    def backward(ctx, dh):
        out, f_len, g_len, batch_offset = ctx.saved_tensors
        B, T, U, H, pack_output, mode, scale = ctx.meta
        dh = dh.contiguous()
        df = torch.empty((B, T, H), dtype=dh.dtype, device=dh.device)
        dg = torch.empty((B, U, H), dtype=dh.dtype, device=dh.device)
        with _lib.timed("joint_bwd"):
            _lib.check(_lib.lib().caiman_joint_backward(
                _lib.ptr(dh), _lib.ptr(out) if mode else None, _lib.ptr(f_len), _lib.ptr(g_len),
                _lib.ptr(batch_offset) if pack_output else None, B, T, U, H, int(pack_output), mode, float(scale),
                _lib.dtype_tag(dh.dtype), _lib.ptr(df), _lib.ptr(dg), _lib.stream()))
        return df, dg, None, None, None, None, None, None, None, None


class TransducerJoint(torch.nn.Module):
    """h[b,t,u,:] = dropout(relu(f[b,t,:] + g[b,u,:])), optionally packed.

    Arguments (apex-compatible): pack_output, relu, dropout, dropout_prob.  `opt`,
    `fwd_tile_size` and `probe_mask` of apex are accepted and ignored.
    """

    def __init__(self, pack_output=False, relu=False, dropout=False, opt=1, fwd_tile_size=4,
                 dropout_prob=0.0, probe_mask=False):
        super().__init__()
        self.pack_output = pack_output
        self.relu = relu
        self.dropout = dropout
        self.dropout_prob = dropout_prob

    def forward(self, f, g, f_len, g_len, batch_offset=None, packed_batch=0):
        p = self.dropout_prob if (self.dropout and self.training) else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0
        return _JointFunc.apply(f, g, f_len, g_len, batch_offset, packed_batch, self.pack_output, self.relu, p, seed)
