"""Shared synthetic-data builders for the parity tests (mirrors the shapes used by
training/lib/tests/transducer/test_loss.py:13-46 `mock_data`)."""
import numpy as np


def mock_lattice(batch_size, time_dim, vocab=10, max_decode_length=9, seed=0, packed=False,
                 eos_idx=None, star_idx=None, full=False):
    """Random logits + labels.  Returns dict with x (padded [B,T,U+1,V] or packed [rows,V]),
    label [B,U], f_len, y_len, batch_offset, blank."""
    rng = np.random.default_rng(seed)
    blank = vocab - 1
    U = max_decode_length - 1
    label = rng.integers(0, blank - 1, size=(batch_size, U)).astype(np.int32)
    f_len = rng.integers(1, time_dim + 1, size=batch_size).astype(np.int32)
    f_len[0] = time_dim
    y_len = rng.integers(0, U + 1, size=batch_size).astype(np.int32)
    y_len[-1] = U
    if full:
        f_len[:] = time_dim
        y_len[:] = U
    if eos_idx is not None:
        for i in range(batch_size):
            if y_len[i] > 0:
                label[i, y_len[i] - 1] = eos_idx
    if star_idx is not None:
        mask = rng.random(label.shape) < 0.25
        mask[0, 0] = True
        label = np.where(mask, star_idx, label).astype(np.int32)
    xp = rng.standard_normal((batch_size, time_dim, U + 1, vocab))
    batch_offset = np.cumsum(f_len.astype(np.int64) * (y_len.astype(np.int64) + 1))
    out = dict(label=label, f_len=f_len, y_len=y_len, blank=blank, batch_offset=batch_offset,
               max_f_len=int(f_len.max()), x_padded=xp)
    rows = []
    for b in range(batch_size):
        rows.append(xp[b, : f_len[b], : y_len[b] + 1].reshape(-1, vocab))
    out["x_packed"] = np.concatenate(rows, 0)
    out["x"] = out["x_packed"] if packed else xp
    return out


def unpack(packed_rows, f_len, y_len, T, U1, fill=0.0):
    """[rows, ...] packed -> [B, T, U1, ...] padded."""
    B = len(f_len)
    out = np.full((B, T, U1) + packed_rows.shape[1:], fill, dtype=packed_rows.dtype)
    off = 0
    for b in range(B):
        n = int(f_len[b]) * (int(y_len[b]) + 1)
        out[b, : f_len[b], : y_len[b] + 1] = packed_rows[off : off + n].reshape(
            (f_len[b], y_len[b] + 1) + packed_rows.shape[1:])
        off += n
    return out


class OracleRNNT:
    """Minimal `encode / predict / joint` object backed by the CPU oracle (oracle/model.py), so that host-side
    decoders can be exercised without a GPU.  Test infrastructure only."""

    def __init__(self, sd, cfg):
        import torch

        self.sd = {k: torch.as_tensor(v) for k, v in sd.items()}
        self.cfg = cfg
        self.training = False

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        self.training = mode
        return self

    def encode(self, x, x_lens, enc_state=None):
        from oracle import model as om

        f, lens = om.encode(self.sd, self.cfg, x, x_lens)
        return f, lens, None

    def predict(self, y, pred_state=None, add_sos=True, special_sos=None):
        import torch
        from oracle import model as om

        if y is None:  # one zero-embedding step
            B = 1 if pred_state is None else pred_state[0].size(1)
            g, st = om.predict(self.sd, self.cfg, torch.zeros(B, 0, dtype=torch.long), pred_state, add_sos=True)
        else:
            g, st = om.predict(self.sd, self.cfg, y, pred_state, add_sos=add_sos)
        return g, st, None

    def joint(self, f, g, *a, **k):
        from oracle import model as om

        return om.joint(self.sd, f, g)


class OracleBeamStep:
    """CPU stand-in for the HIP expansion round of caiman_asr_amd.rnnt.beam_native (prediction + joint via the
    oracle-backed model, log-softmax / EOS correction / top-k via torch): lets the native search object be driven
    through its C-ABI without a GPU.  Test infrastructure only."""

    def __init__(self, decoder, model):
        import torch

        self.dec, self.model, self.torch = decoder, model, torch
        self.states = {}

    def __call__(self, frames2d, rows, y_last, state_in, state_out, n_slots):
        import numpy as np

        torch, k = self.torch, self.dec.beam_width
        f = frames2d[torch.from_numpy(np.asarray(rows, dtype=np.int64))].unsqueeze(1)
        sc, tk, bl = [], [], []
        for i in range(len(y_last)):
            if y_last[i] < 0:
                g, st, _ = self.model.predict(None, None, add_sos=False)
            else:
                g, st, _ = self.model.predict(torch.tensor([[int(y_last[i])]]), self.states[int(state_in[i])], add_sos=False)
            self.states[int(state_out[i])] = st
            log_p = self.dec._joint_step(f[i:i + 1], g)
            s, t = log_p.topk(min(k, log_p.shape[1]), dim=1)
            sc.append(s[0].numpy()); tk.append(t[0].numpy().astype(np.int32)); bl.append(float(log_p[0, self.dec.blank_idx]))
        return (np.ascontiguousarray(np.stack(sc), dtype=np.float32), np.ascontiguousarray(np.stack(tk), dtype=np.int32),
                np.asarray(bl, dtype=np.float32))
