"""torch-CPU restatement of the RNN-T network + loss (float64 by default).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows
training/caiman_asr_train/rnnt/model.py:297-447 (forward wiring), rnn.py:165-208 (LSTM stack),
model.py:35-49 (StackTime), loss.py:112-125 (batch mean) and uses oracle/rnnt_oracle.c for the
transducer loss.  Pinned by tests/golden/rnnt_*.npz, which were produced by the reference's own
RNNT class (oracle/gen_golden.py).
"""
import math

import numpy as np
import torch

from oracle import native

# ---- storage rounding ---------------------------------------------------------------------------------------------
# The HIP path computes in fp32 but STORES activations, weights images and activation gradients in the autocast type
# (bf16).  `storage=torch.bfloat16` makes this restatement round at the same points -- forward values where the HIP
# path writes a 16-bit tensor (rf), gradients where its backward pass writes one (rb) -- while all arithmetic between
# two storage points stays float64.  A comparison of the bf16 HIP run against THIS oracle then measures the kernels,
# not the 8-bit mantissa pushed through ten LSTM layers (which a comparison against the unrounded oracle has to allow
# for).  Rounding points (file:line of the HIP side): LSTM operand images + bias sum (csrc/lstm_images.hip), projection
# output = pre-activations (proj_gemm.hip epilogue / torch.addmm out dtype), h and c rows (lstm.hip: `cv`, `yv` of the
# cell update; c is rounded for the NEXT step, tanh(c) uses the unrounded value), dG = gradient of the pre-activations
# (lstm.hip backward epilogue: vI..vO), delta = dX of a layer (proj_gemm / matmul out dtype), autocast linears
# (joint_enc / joint_pred / joint_fc: weight, bias, output and output gradient), joint activations (joint.hip).  Kept
# unrounded where the HIP path keeps fp32: dC and dh inside the recurrence, the loss lattice, parameter gradients.
_STORAGE = None


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt):
        return x.to(dt).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _RoundBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(g.dtype), None


def rf(x):
    """value as the HIP path stores it (gradient passes unchanged)"""
    return x if _STORAGE is None else _RoundFwd.apply(x, _STORAGE)


def rb(x):
    """gradient as the HIP path stores it (value passes unchanged)"""
    return x if _STORAGE is None else _RoundBwd.apply(x, _STORAGE)


def rfb(x):
    return rb(rf(x))


class _Cell(torch.autograd.Function):
    """One LSTM cell update (training/lib/csrc/lstm.cu:85-135) whose backward is the reference's pointwise backward ON THE
    STORED VALUES (lstm.cu:137-212: derivatives are taken on the ACTIVATED gates and on c as they sit in memory): with a
    16-bit storage type the gates i, f, g, o and the new c are rounded before the backward pass reads them (csrc/lstm.hip
    writes `v` = the activated gates and `cv`; the backward kernels read them back), while the forward value of h uses the
    unrounded c.  storage None: plain autograd of the same formulas."""

    @staticmethod
    def forward(ctx, z, c_prev, storage):
        H = z.shape[1] // 4
        i, f, g, o = torch.sigmoid(z[:, :H]), torch.sigmoid(z[:, H:2 * H]), torch.tanh(z[:, 2 * H:3 * H]), torch.sigmoid(z[:, 3 * H:])
        c_new = i * g + f * c_prev
        h = o * torch.tanh(c_new)

        def r(t):
            return t if storage is None else t.to(storage).to(t.dtype)

        ctx.save_for_backward(r(i), r(f), r(g), r(o), c_prev, r(c_new))
        return h, c_new

    @staticmethod
    def backward(ctx, dh, dc_next):
        i, f, g, o, c_prev, c_cur = ctx.saved_tensors
        ct = torch.tanh(c_cur)
        dc = dh * o * (1 - ct * ct) + dc_next
        dz = torch.cat([dc * g * (1 - i) * i, dc * c_prev * (1 - f) * f, dc * i * (1 - g * g), dh * ct * (1 - o) * o], 1)
        return dz, dc * f, None


def _linear(x, w, b):
    """autocast nn.Linear: 16-bit weight, bias and output; 16-bit output gradient"""
    return rfb(x @ rf(w).t() + rf(b))


def _lstm_stack(sd, prefix, x, num_layers, state=None):
    """Plain-python multi-layer LSTM (gate order i,f,g,o).  `x` arrives as the layer below stored
    it; the gradient this stack returns for it is rounded (it is a dX GEMM's output)."""
    T, B, _ = x.shape
    hs, cs = [], []
    for l in range(num_layers):
        W, R = rf(sd[f"{prefix}.weight_ih_l{l}"]), rf(sd[f"{prefix}.weight_hh_l{l}"])
        b = rf(sd[f"{prefix}.bias_ih_l{l}"] + sd[f"{prefix}.bias_hh_l{l}"])
        H = R.shape[1]
        h = x.new_zeros(B, H) if state is None else state[0][l]
        c = x.new_zeros(B, H) if state is None else state[1][l]
        pre_all = rf(rb(x) @ W.t() + b)
        outs = []
        for t in range(T):
            pre = rb(pre_all[t] + h @ R.t())          # its gradient is dG[t], a stored row
            h_new, c_new = _Cell.apply(pre, c, _STORAGE)
            h = rf(h_new)
            c = rf(c_new)
            outs.append(h)
        x = torch.stack(outs, 0)
        hs.append(h)
        cs.append(c)
    return x, (torch.stack(hs), torch.stack(cs))


def stack_time(x, lens, factor):
    T, B, H = x.shape
    seq = [x]
    for i in range(1, factor):
        tmp = torch.zeros_like(x)
        tmp[:-i] = x[i:]
        seq.append(tmp)
    return torch.cat(seq, 2)[::factor], (lens + factor - 1) // factor


def encode(sd, cfg, x, x_lens):
    x, _ = _lstm_stack(sd, "encoder.pre_rnn.lstm", rf(x), cfg["enc_pre_rnn_layers"])
    x, lens = stack_time(x, x_lens, cfg["enc_stack_time_factor"])
    x, _ = _lstm_stack(sd, "encoder.post_rnn.lstm", x, cfg["enc_post_rnn_layers"])
    return _linear(rb(x).transpose(0, 1), sd["joint_enc.weight"], sd["joint_enc.bias"]), lens


def predict(sd, cfg, y, state=None, add_sos=True):
    e = sd["prediction.embed.weight"][y]  # [B,U,H]
    if add_sos:
        e = torch.cat([e.new_zeros(e.shape[0], 1, e.shape[2]), e], 1)
    g, st = _lstm_stack(sd, "prediction.dec_rnn.lstm", rf(e).transpose(0, 1), cfg["pred_rnn_layers"], state)
    return _linear(rb(g).transpose(0, 1), sd["joint_pred.weight"], sd["joint_pred.bias"]), st


def joint(sd, f, g):
    h = rfb(torch.relu(f.unsqueeze(2) + g.unsqueeze(1)))
    return _linear(h, sd["joint_net.2.weight"], sd["joint_net.2.bias"])


def forward(sd, cfg, x, x_lens, y, y_lens):
    """-> padded logits [B, T', U+1, V], f_lens."""
    f, f_lens = encode(sd, cfg, x, x_lens)
    g, _ = predict(sd, cfg, y)
    return joint(sd, f, g), f_lens


class OracleTransducerLoss(torch.autograd.Function):
    """Per-utterance loss on PADDED logits via oracle/rnnt_oracle.c (float64)."""

    @staticmethod
    def forward(ctx, x, label, f_len, y_len, blank, mods):
        xn = x.detach().double().numpy()
        a, b, loss, denom = native.transducer_forward(xn, label.numpy(), f_len.numpy(), y_len.numpy(), blank, **mods)
        ctx.pack = (xn, a, b, denom, label.numpy(), f_len.numpy(), y_len.numpy(), blank, mods)
        return torch.from_numpy(loss).to(x.dtype)

    @staticmethod
    def backward(ctx, gl):
        xn, a, b, denom, label, f_len, y_len, blank, mods = ctx.pack
        g = native.transducer_backward(xn, denom, gl.double().numpy(), a, b, label, f_len, y_len, blank, **mods)
        return torch.from_numpy(g).to(gl.dtype), None, None, None, None, None


def loss_and_grads(sd_np, cfg, x, x_lens, y, y_lens, blank, delay_penalty=0.0, eos_penalty=0.0, eos_idx=None,
                   star_penalty=1.0, star_idx=None, dtype=torch.float64, storage=None):
    """Batch-mean loss and d loss / d every parameter (dict name -> ndarray).  `storage` (e.g. torch.bfloat16): round
    stored activations / weight images / activation gradients where the HIP path stores them (top of this file)."""
    global _STORAGE
    sd = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in sd_np.items()}
    _STORAGE = storage
    try:
        return _loss_and_grads(sd, cfg, x, x_lens, y, y_lens, blank, delay_penalty, eos_penalty, eos_idx, star_penalty,
                               star_idx, dtype)
    finally:
        _STORAGE = None


def _loss_and_grads(sd, cfg, x, x_lens, y, y_lens, blank, delay_penalty, eos_penalty, eos_idx, star_penalty, star_idx, dtype):
    logits, f_lens = forward(sd, cfg, torch.as_tensor(x, dtype=dtype), torch.as_tensor(x_lens),
                             torch.as_tensor(y), torch.as_tensor(y_lens))
    mods = dict(delay_penalty=delay_penalty, eos_penalty=eos_penalty, eos_idx=eos_idx,
                star_lam=math.log(star_penalty), star_idx=star_idx)
    per_utt = OracleTransducerLoss.apply(logits, torch.as_tensor(y).int(), f_lens.int(),
                                         torch.as_tensor(y_lens).int(), blank, mods)
    loss = per_utt.mean()
    loss.backward()
    return float(loss.detach()), {k: v.grad.numpy() for k, v in sd.items() if v.grad is not None}, logits.detach().numpy()


def greedy_decode(sd, cfg, x, x_lens, blank, max_symbols_per_step=30, max_symbol_per_sample=None):
    """Sequential per-utterance greedy search: the batched loop of
    training/caiman_asr_train/rnnt/batched_greedy.py:59-199 unrolled for one stream at a time
    (same stop rules, same never-reset-on-blank emission counter, :122-139,168-199).
    -> (tokens, frames, confidences) per utterance."""
    f, f_lens = encode(sd, cfg, x, x_lens)
    toks, frames, confs = [], [], []
    for b in range(f.shape[0]):
        tk, fr, cf = [], [], []
        g, st = predict(sd, cfg, torch.zeros(1, 0, dtype=torch.long))  # SOS: zero embedding, zero state
        t, per_step, total = 0, 0, 0
        last = int(f_lens[b]) - 1
        while True:
            logits = joint(sd, f[b:b + 1, t:t + 1], g[:, -1:])[0, 0, 0]
            lp = torch.log_softmax(logits, -1)
            k = int(torch.argmax(lp))
            if t == last and k == blank:
                break
            if max_symbols_per_step is not None and t == last and per_step >= max_symbols_per_step:
                break
            if max_symbol_per_sample is not None and total >= max_symbol_per_sample:
                break
            nonblank = k != blank
            if nonblank:
                tk.append(k)
                fr.append(t)
                cf.append(float(lp[k].exp()))
                total += 1
            advance = not nonblank
            if max_symbols_per_step is not None:
                per_step += int(nonblank)
                advance = advance or per_step >= max_symbols_per_step
                if not (per_step < max_symbols_per_step or t == last):
                    per_step = 0
            t = min(t + int(advance), last)
            if nonblank:
                g, st = predict(sd, cfg, torch.tensor([[k]]), st, add_sos=False)
        toks.append(tk)
        frames.append(fr)
        confs.append(cf)
    return toks, frames, confs
