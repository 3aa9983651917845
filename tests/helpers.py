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
