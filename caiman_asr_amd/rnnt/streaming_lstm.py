"""LSTM stacks advanced a few timesteps at a time for THOUSANDS of rows (live streams).

The training-side step kernel (csrc/lstm.hip) streams a layer's recurrent weights once per 32-row tile -- right for
a batch of 32..128 utterances, 500x too much traffic for 16 000 streams.  With that many rows a timestep of a layer
is a well-shaped library GEMM, gates = [x_t | h_{t-1}] · [W_ih | W_hh]^T + (b_ih + b_hh), followed by the cell
kernel of the beam round (`caiman_beam_lstm_cell`, include/caiman_beam.h), which also writes the next layer's
input row.  State lives in pools [layers, 1 + rows, hidden] (row 0 unused here), h in the compute dtype, c in f32,
and is updated in place.  Same arithmetic as the module path (training/lib/csrc/lstm.cu:99-123, i,f,g,o order)."""
import os
from typing import Optional

import torch

from caiman_asr_amd import _lib

# One launch per layer-step: the 4-gate GEMM on MFMA with the cell update, the state scatter and the next layer's input
# row as its epilogue (csrc/proj_gemm.hip, caiman_lstm_step_gemm) instead of library GEMM + cell kernel.
# CAIMAN_DECODE_FUSED_LSTM=0 restores the two-launch form (A/B, and the fallback for hidden sizes % 128 != 0).
FUSED_STEP = os.environ.get("CAIMAN_DECODE_FUSED_LSTM", "1") != "0"
# Measured on one box (bench_decode.py): 2 000 beam streams, tick p50 16.5 - 18.4 ms (two launches) -> 12.1 - 13.6 ms (fused:
# a round is latency-bound, two launches fewer per layer); 16 000 greedy streams 12.8 -> 14.9 ms (at that many rows the
# library's big-tile GEMM beats the 128 x 128 tiles of the fused kernel by more than the cell pass costs).  Hence a row limit.
FUSED_MAX_ROWS = int(os.environ.get("CAIMAN_DECODE_FUSED_MAX_ROWS", "6000"))


def _pad128(n: int) -> int:
    return (n + 127) // 128 * 128


def fused_step_ok(hidden: int, cd, n_rows: int = 0) -> bool:
    return FUSED_STEP and hidden % 128 == 0 and cd in (torch.float16, torch.bfloat16) and n_rows <= FUSED_MAX_ROWS


def fused_layer_weights(lstm, l: int, cd):
    """-> (W [4H, Ip + H] with rows ordered [unit][gate] and the input part zero-padded to Ip = pad128(I), bias [4H] in
    the same order, Ip): the operand images of caiman_lstm_step_gemm for layer l."""
    W_ih, W_hh = getattr(lstm, f"weight_ih_l{l}").detach(), getattr(lstm, f"weight_hh_l{l}").detach()
    H, I = W_hh.shape[1], W_ih.shape[1]
    Ip = _pad128(I)
    W = torch.zeros(4 * H, Ip + H, device=W_ih.device, dtype=torch.float32)
    W[:, :I] = W_ih
    W[:, Ip:] = W_hh
    W = W.view(4, H, Ip + H).transpose(0, 1).reshape(4 * H, Ip + H).to(cd).contiguous()
    b = (getattr(lstm, f"bias_ih_l{l}") + getattr(lstm, f"bias_hh_l{l}")).detach().float().view(4, H).t().reshape(4 * H)
    return W, b.to(cd).contiguous(), Ip


def lstm_step_gemm(X, W, b, n, H, c_pool_l, h_pool_l, h_pool_next, s_in, s_out, nxt, tag, st):
    """X [>= n, K] rows [x | h_prev]; pools / slots / nxt as caiman_beam_lstm_cell (s_in / s_out: device pointers)."""
    _lib.check(_lib.lib().caiman_lstm_step_gemm(_lib.ptr(X), X.shape[1], _lib.ptr(W), _lib.ptr(b), n, H, W.shape[1],
                                                _lib.ptr(c_pool_l), _lib.ptr(h_pool_l),
                                                None if h_pool_next is None else _lib.ptr(h_pool_next), s_in, s_out,
                                                _lib.ptr(nxt), nxt.shape[1], tag, st))


class LargeBatchLSTM:
    def __init__(self, rnn_module, n_rows: int):
        """rnn_module: caiman_asr_amd.rnnt.rnn.LSTM (plain stack: no batch norm, soft activations)."""
        if rnn_module.batch_norm or getattr(rnn_module.lstm, "hard", False):
            raise NotImplementedError("large-batch streaming covers the plain LSTM stacks of the shipped configs")
        self.lstm = rnn_module.lstm
        self.L = rnn_module.num_layers
        self.B = n_rows
        self.H = self.lstm.weight_hh_l0.shape[1]
        self.I = self.lstm.weight_ih_l0.shape[1]
        self.w = {}
        self.h = self.c = None
        self.Ip = _pad128(self.I)      # fused step: layer 0's input columns padded to the GEMM's K granularity

    def _weights(self, cd):
        w = self.w.get(cd)
        if w is None:
            w = {}
            with torch.no_grad():
                for l in range(self.L):
                    if fused_step_ok(self.H, cd, self.B):
                        w[f"W{l}"], w[f"b{l}"], _ = fused_layer_weights(self.lstm, l, cd)
                        continue
                    w[f"W{l}"] = torch.cat([getattr(self.lstm, f"weight_ih_l{l}"), getattr(self.lstm, f"weight_hh_l{l}")], 1) \
                        .detach().to(cd).contiguous()
                    w[f"b{l}"] = (getattr(self.lstm, f"bias_ih_l{l}") + getattr(self.lstm, f"bias_hh_l{l}")).detach().to(cd)
            self.w[cd] = w
        return w

    def _ensure(self, dev, cd):
        if self.h is None or self.h.dtype != cd:
            B, H, L = self.B, self.H, self.L
            old = self.h
            self.h = torch.zeros(L, B + 1, H, device=dev, dtype=cd) if old is None else old.to(cd)
            if self.c is None:
                self.c = torch.zeros(L, B + 1, H, device=dev, dtype=torch.float32)
            self.iota = torch.arange(B, device=dev, dtype=torch.int32)
            I0 = self.Ip if fused_step_ok(H, cd, B) else self.I
            self.X = [torch.empty(B, (I0 if l == 0 else H) + H, device=dev, dtype=cd) for l in range(L)]
            self.top = torch.empty(B, H, device=dev, dtype=cd)
            self.gates = None if fused_step_ok(H, cd, B) else torch.empty(B, 4 * H, device=dev, dtype=cd)

    def set_state(self, h: Optional[torch.Tensor], c: Optional[torch.Tensor]):
        """(h, c) each [L, B, H] as the module path carries them, or None for zeros."""
        if h is not None:
            self._ensure(h.device, h.dtype)
            self.h[:, 1:] = h
            self.c[:, 1:] = c.float()

    def state(self):
        return self.h[:, 1:], self.c[:, 1:]

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [T, B, I] -> top-layer outputs [T, B, H]; the carried state advances by T steps."""
        T, B, I = x.shape
        assert B == self.B and I == self.I
        cd = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else self.lstm.weight_hh_l0.dtype
        self._ensure(x.device, cd)
        w = self._weights(cd)
        fused = fused_step_ok(self.H, cd, B)
        x = x.to(cd)
        if fused and self.Ip != I:
            x = torch.nn.functional.pad(x, (0, self.Ip - I))     # zero columns meet zero weight columns
            I = self.Ip
        x = x.contiguous()
        lib, tag, st, H, L = _lib.lib(), _lib.dtype_tag(cd), _lib.stream(), self.H, self.L
        iota = _lib.ptr(self.iota)
        out = torch.empty(T, B, H, device=x.device, dtype=cd)
        for t in range(T):
            # X0[i] = [x_t[i] | h_0[i]]: the gather kernel with x_t as the "embedding table" and ids 0..B-1
            _lib.check(lib.caiman_beam_gather_inputs(_lib.ptr(x[t]), I, _lib.ptr(self.h[0]), H, iota, iota, B,
                                                     _lib.ptr(self.X[0]), self.X[0].shape[1], tag, st))
            for l in range(L):
                last = l + 1 == L
                nxt = out[t] if last else self.X[l + 1]
                if fused:
                    lstm_step_gemm(self.X[l], w[f"W{l}"], w[f"b{l}"], B, H, self.c[l], self.h[l],
                                   None if last else self.h[l + 1], iota, iota, nxt, tag, st)
                    continue
                torch.addmm(w[f"b{l}"], self.X[l], w[f"W{l}"].t(), out=self.gates)
                _lib.check(lib.caiman_beam_lstm_cell(_lib.ptr(self.gates), H, _lib.ptr(self.c[l]), _lib.ptr(self.h[l]),
                                                     None if last else _lib.ptr(self.h[l + 1]), iota, iota, B, _lib.ptr(nxt),
                                                     H if last else self.X[l + 1].shape[1], tag, st))
        return out


class LargeBatchPredictor:
    """Prediction network (embedding -> LSTM stack -> joint_pred) stepped for thousands of rows, some of which keep
    their state: rows with `emitted` false write their new state to a sink row of the pools, so no masked copies of
    the [layers, rows, hidden] state tensors are needed (the greedy loop's `torch.where(emitted, new, old)`)."""

    def __init__(self, model, n_rows: int):
        rnn = model.prediction["dec_rnn"]
        if rnn.batch_norm or getattr(rnn.lstm, "hard", False):
            raise NotImplementedError("large-batch prediction covers the plain LSTM stacks of the shipped configs")
        self.model, self.lstm, self.L, self.B = model, rnn.lstm, rnn.num_layers, n_rows
        self.H = model.pred_n_hid
        self.w = {}
        self.h = None

    def _weights(self, cd):
        w = self.w.get(cd)
        if w is None:
            m = self.model
            with torch.no_grad():
                w = dict(embed=m.prediction["embed"].weight.detach().to(cd).contiguous(),
                         Wp=m.joint_pred.weight.detach().to(cd).contiguous(), bp=m.joint_pred.bias.detach().to(cd))
                E = w["embed"].shape[1]
                for l in range(self.L):
                    if fused_step_ok(self.H, cd, self.B) and E % 128 == 0:
                        w[f"W{l}"], w[f"b{l}"], _ = fused_layer_weights(self.lstm, l, cd)
                        continue
                    w[f"W{l}"] = torch.cat([getattr(self.lstm, f"weight_ih_l{l}"), getattr(self.lstm, f"weight_hh_l{l}")], 1) \
                        .detach().to(cd).contiguous()
                    w[f"b{l}"] = (getattr(self.lstm, f"bias_ih_l{l}") + getattr(self.lstm, f"bias_hh_l{l}")).detach().to(cd)
            self.w[cd] = w
        return w

    def _ensure(self, dev, cd):
        if self.h is None or self.h.dtype != cd:
            B, H, L = self.B, self.H, self.L
            E = self.model.prediction["embed"].weight.shape[1]
            self.h = torch.zeros(L, B + 2, H, device=dev, dtype=cd)        # row 0: zero start state, row B + 1: sink
            self.c = torch.zeros(L, B + 2, H, device=dev, dtype=torch.float32)
            self.iota = torch.arange(B, device=dev, dtype=torch.int32)
            self.X = [torch.empty(B, (E if l == 0 else H) + H, device=dev, dtype=cd) for l in range(L)]
            self.top = torch.empty(B, H, device=dev, dtype=cd)
            self.gates = torch.empty(B, 4 * H, device=dev, dtype=cd)
            self.g = torch.empty(B, self.model.joint_pred.weight.shape[0], device=dev, dtype=cd)

    @torch.no_grad()
    def step(self, y: Optional[torch.Tensor], emitted: Optional[torch.Tensor], dev=None) -> torch.Tensor:
        """y [B] token ids (None: the start-of-sequence step, zero embedding from the zero state, all rows);
        emitted [B] bool: rows that take the new state.  -> g [B, Hj] (meaningful where emitted)."""
        cd = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else self.model.joint_pred.weight.dtype
        dev = dev if dev is not None else y.device
        self._ensure(dev, cd)
        w = self._weights(cd)
        lib, tag, st, H, L, B = _lib.lib(), _lib.dtype_tag(cd), _lib.stream(), self.H, self.L, self.B
        if y is None:
            y32 = torch.full((B,), -1, dtype=torch.int32, device=dev)
            s_in = torch.full((B,), -1, dtype=torch.int32, device=dev)     # pool row 0: zeros
            s_out = self.iota
        else:
            y32 = y.to(torch.int32)
            s_in = self.iota
            s_out = torch.where(emitted, self.iota, torch.full_like(self.iota, B))
        E = w["embed"].shape[1]
        _lib.check(lib.caiman_beam_gather_inputs(_lib.ptr(w["embed"]), E, _lib.ptr(self.h[0]), H, _lib.ptr(y32),
                                                 _lib.ptr(s_in), B, _lib.ptr(self.X[0]), self.X[0].shape[1], tag, st))
        fused = fused_step_ok(H, cd, B) and E % 128 == 0
        for l in range(L):
            last = l + 1 == L
            nxt = self.top if last else self.X[l + 1]
            if fused:
                lstm_step_gemm(self.X[l], w[f"W{l}"], w[f"b{l}"], B, H, self.c[l], self.h[l], None if last else self.h[l + 1],
                               _lib.ptr(s_in), _lib.ptr(s_out), nxt, tag, st)
                continue
            torch.addmm(w[f"b{l}"], self.X[l], w[f"W{l}"].t(), out=self.gates)
            _lib.check(lib.caiman_beam_lstm_cell(_lib.ptr(self.gates), H, _lib.ptr(self.c[l]), _lib.ptr(self.h[l]),
                                                 None if last else _lib.ptr(self.h[l + 1]), _lib.ptr(s_in), _lib.ptr(s_out), B,
                                                 _lib.ptr(nxt), nxt.shape[1], tag, st))
        torch.addmm(w["bp"], self.top, w["Wp"].t(), out=self.g)
        return self.g
