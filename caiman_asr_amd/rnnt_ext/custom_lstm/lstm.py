"""rnnt_ext.custom_lstm.lstm — multi-layer LSTM with torch.nn.LSTM-compatible parameter names.

Interface mirror of training/lib/src/rnnt_ext/custom_lstm/lstm.py (Function :11-144,
Layer :161-254, CustomLSTM :257-399).  The all-timestep input GEMM and the weight-gradient
GEMMs are plain library GEMMs (torch -> hipBLASLt); the sequential part runs in the gfx950
kernels of csrc/lstm.hip through `rnnt_ext.cuda.lstm`.
"""
import math
from typing import Optional, Tuple

import torch
from torch import Tensor as Ten

import caiman_asr_amd.rnnt_ext.cuda.lstm as lstm_cu


def _state_rows(first: Ten, steps: int, width: int, dtype) -> Ten:
    """[steps + 1, B, width] buffer of one recurrent state, row 0 = the initial state"""
    rows = torch.empty((steps + 1, first.shape[0], width), dtype=dtype, device=first.device)
    rows[0].copy_(first)
    return rows


class Function(torch.autograd.Function):
    """One LSTM layer over a whole sequence (the operator behind rnnt_ext.custom_lstm.lstm.Function of the reference,
    training/lib/src/rnnt_ext/custom_lstm/lstm.py:11-144): (y0, c0, x, W, R, bW, bR) -> (y[1..T], c[1..T]).  The input
    projection of all timesteps is one library GEMM whose output type (the autocast type) is the type of everything the
    recurrence stores; the recurrence itself is `lstm_fused_fwd / _bwd` (csrc/lstm.hip).  Gradients go to x and the four
    parameters -- not to the initial state (truncated back-propagation through time) -- and the parameter gradients
    leave their GEMMs / sums as fp32 whatever the storage type."""

    @staticmethod
    @torch.autograd.function.once_differentiable
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, lstm_fused_fwd, lstm_fused_bwd, y0: Ten, c0: Ten, x: Ten, W: Ten, R: Ten,
                bW: Ten, bR: Ten) -> Tuple[Ten, Ten]:
        steps, hidden = x.shape[0], R.shape[1]
        x2 = x.flatten(0, 1)
        pre = torch.addmm(bW + bR, x2, W.t()).view(steps, x.shape[1], 4 * hidden)   # x_t W^T + b for every t at once
        store = pre.dtype
        y_rows, c_rows = _state_rows(y0, steps, hidden, store), _state_rows(c0, steps, hidden, store)
        R_store = R.to(store)
        lstm_fused_fwd(R_store, pre, c_rows, y_rows)        # pre becomes the activated gates, rows 1.. of y / c are filled
        ctx.x_needs_grad = x.requires_grad
        ctx.save_for_backward(W, R_store, x2, y_rows, c_rows, pre)
        ctx.kernel = lstm_fused_bwd
        return y_rows[1:], c_rows[1:]

    @staticmethod
    @torch.autograd.function.once_differentiable
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, delta: Ten, *_):
        W, R_store, x2, y_rows, c_rows, gates = ctx.saved_tensors
        if delta.dtype != R_store.dtype:
            raise AssertionError("gradient of the layer output must have the storage type of the layer")
        dG = torch.empty(gates.shape, dtype=gates.dtype, device=gates.device)
        ctx.kernel(R_store, gates, c_rows, delta, dG)
        dG2 = dG.view(-1, dG.shape[-1])
        low = dG2.is_cuda and dG2.dtype in (torch.float16, torch.bfloat16)
        f32 = dict(out_dtype=torch.float32) if low else {}
        db = dG2.sum(0, dtype=torch.float32 if low else dG2.dtype).unsqueeze(0)
        dx = torch.mm(dG2, W.to(dG2.dtype)).view(delta.shape[0], delta.shape[1], -1) if ctx.x_needs_grad else None
        dw = torch.mm(dG2.t(), x2.detach().to(dG2.dtype), **f32)
        dr = torch.mm(dG2.t(), y_rows[:-1].view(-1, y_rows.shape[-1]), **f32)
        return None, None, None, None, dx, dw, dr, db, db


class HardLayer(torch.nn.Module):
    def forward(self, *args, **kwargs):
        return Function.apply(lstm_cu.lstm_fused_fwd_hard, lstm_cu.lstm_fused_bwd_hard, *args, **kwargs)


class SoftLayer(torch.nn.Module):
    def forward(self, *args, **kwargs):
        return Function.apply(lstm_cu.lstm_fused_fwd_soft, lstm_cu.lstm_fused_bwd_soft, *args, **kwargs)


class Layer(torch.nn.Module):
    """A single LSTM layer (soft or hard activations, optional recurrent-weight dropout)."""

    def __init__(self, input_size: int, hidden_size: int, hard: bool = False, rw_dropout: float = 0.0,
                 dtype=None, device=None):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.hard = hard
        self.layer_fun = HardLayer() if hard else SoftLayer()
        self.rw_dropout = rw_dropout
        self.drop_fun = torch.nn.Dropout(p=rw_dropout) if rw_dropout != 0.0 else torch.nn.Identity()
        kw = {"dtype": dtype, "device": device}
        self.weight_ih = torch.nn.Parameter(torch.empty(4 * hidden_size, input_size, **kw))
        self.weight_hh = torch.nn.Parameter(torch.empty(4 * hidden_size, hidden_size, **kw))
        self.bias_ih = torch.nn.Parameter(torch.empty(4 * hidden_size, **kw))
        self.bias_hh = torch.nn.Parameter(torch.empty(4 * hidden_size, **kw))
        rsh = 1.0 / math.sqrt(hidden_size)
        with torch.no_grad():
            for param in self.parameters():
                param.uniform_(-rsh, rsh)

    def forward(self, x: Ten, state: Tuple[Ten, Ten]) -> Tuple[Ten, Ten]:
        """x [T,B,I], state (y0, c0) each [B,H] -> (y1..yT, c1..cT)."""
        return self.layer_fun(*state, x, self.weight_ih, self.drop_fun(self.weight_hh), self.bias_ih,
                              self.bias_hh)

    def extra_repr(self):
        return (f"input_size={self.input_size:.>4}, hidden_size={self.hidden_size:.>4}, "
                f"hard={self.hard}, rw_dropout={self.rw_dropout}")


class CustomLSTM(torch.nn.Module):
    """Partial drop-in for torch.nn.LSTM with optional hard activations; weight names match
    torch.nn.LSTM so state_dicts / checkpoints transfer (lstm.py:323-327)."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int = 1, dropout: float = 0.0,
                 hard: bool = False, quantize: bool = False, rw_dropout: float = 0.0, dtype=None,
                 device=None):
        super().__init__()
        assert not quantize, "Cuda CustomLSTM does not support quantization"
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.bl_dropout = dropout
        self.quantize = quantize
        self.hard = hard
        self.rw_dropout = rw_dropout
        self.drop_function = torch.nn.Dropout(p=dropout) if dropout != 0.0 else torch.nn.Identity()
        self.pipeline_layers = True  # layer-pipelined schedule for f16/bf16 stacks (custom_lstm/stack.py)
        kw = dict(hidden_size=hidden_size, hard=hard, rw_dropout=rw_dropout, dtype=dtype, device=device)
        self.layers = [Layer(input_size, **kw)]
        self.layers.extend(Layer(hidden_size, **kw) for _ in range(num_layers - 1))
        for i, layer in enumerate(self.layers):
            self.register_parameter(name=f"weight_ih_l{i}", param=layer.weight_ih)
            self.register_parameter(name=f"weight_hh_l{i}", param=layer.weight_hh)
            self.register_parameter(name=f"bias_ih_l{i}", param=layer.bias_ih)
            self.register_parameter(name=f"bias_hh_l{i}", param=layer.bias_hh)

    def _apply(self, fn, *a, **k):
        # `self.layers` is a plain list (as in the reference), so the Layer modules are not
        # children; keep their Parameter objects identical to the registered ones after
        # .to()/.cuda()/.half() replace them.
        out = super()._apply(fn, *a, **k)
        for i, layer in enumerate(self.layers):
            for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                layer._parameters[n] = self._parameters[f"{n}_l{i}"]
        return out

    def train(self, mode: bool = True):
        super().train(mode)
        for layer in self.layers:
            layer.train(mode)
        return self

    def forward(self, input: Ten, state: Optional[Tuple[Ten, Ten]] = None
                ) -> Tuple[Ten, Tuple[Ten, Ten], Tuple[Ten, Ten]]:
        """-> (output [T,B,H], (h_n, c_n) [L,B,H], (all_h, all_c) [L,T,B,H])."""
        from caiman_asr_amd.rnnt_ext.custom_lstm import stack

        gate_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else input.dtype
        if self.pipeline_layers and self.rw_dropout == 0.0 and stack.eligible(input, self.hidden_size, self.num_layers,
                                                                              gate_dtype):
            # whole-stack layer pipeline (same kernels and GEMM operands, different schedule)
            L = self.num_layers
            if state is None:
                h0 = c0 = None           # the function clears the first row of its own buffers
            else:
                h0, c0 = state[0].detach(), state[1].detach()
            params = []
            for layer in self.layers:
                params += [layer.weight_ih, layer.weight_hh, layer.bias_ih, layer.bias_hh]
            y, all_h, all_c = stack.StackFunction.apply(input, h0, c0, self.hard, float(self.bl_dropout), self.training,
                                                        *params)
            return y, (all_h[:, -1], all_c[:, -1]), (all_h, all_c)

        # one layer after the other (fp32 / fp64 inputs, recurrent-weight dropout, stacks the pipeline does not take)
        depth, (steps, batch) = self.num_layers, input.shape[:2]
        every_h = every_c = None
        feed = input
        for idx, layer in enumerate(self.layers):
            if state is None:
                first = tuple(torch.zeros((batch, self.hidden_size), device=input.device, dtype=input.dtype) for _ in range(2))
            else:
                first = (state[0][idx].detach(), state[1][idx].detach())
            h, c = layer(feed if idx == 0 else self.drop_function(feed), first)
            if every_h is None:      # allocated in the type the first layer stores its rows in
                every_h = h.new_empty((depth, steps, batch, self.hidden_size))
                every_c = c.new_empty((depth, steps, batch, self.hidden_size))
            every_h[idx], every_c[idx] = h, c
            feed = h
        return feed, (every_h[:, -1], every_c[:, -1]), (every_h, every_c)
