"""LSTM stack wrapper: forget-gate bias init, weight scaling, optional batch norm, final dropout.
Interface mirror of training/caiman_asr_train/rnnt/rnn.py:20-208."""
import torch


def rnn(input_size, hidden_size, num_layers, batch_norm, forget_gate_bias=1.0, dropout=0.0, **kwargs):
    return LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, batch_norm=batch_norm,
                dropout=dropout, forget_gate_bias=forget_gate_bias, **kwargs)


class LSTM(torch.nn.Module):
    """`custom_lstm=True` selects the gfx950 CustomLSTM (the path this package exists for);
    `custom_lstm=False` selects torch.nn.LSTM (library kernels).  There is no CPU LSTM here:
    the reference's TorchScript fallback (`gpu_unavailable`, `quantize`) is not provided."""

    def __init__(self, input_size, hidden_size, num_layers, batch_norm, dropout, forget_gate_bias,
                 weights_init_scale=1.0, hidden_hidden_bias_scale=0.0, **kwargs):
        super().__init__()
        self.num_layers = num_layers
        self.batch_norm = batch_norm
        custom = kwargs.get("custom_lstm", False)
        if kwargs.get("quantize", False):
            raise ValueError("quantize=True needs the reference's legacy TorchScript LSTM (qtorch); "
                             "it is outside the MI355X hot path and not provided")
        if custom and kwargs.get("gpu_unavailable", False):
            raise ValueError("gpu_unavailable=True selects a CPU LSTM in the reference; this build has "
                             "no CPU fallback for the RNN-T kernels")
        rw_dropout = kwargs.get("rw_dropout", 0.0)
        if custom:
            from caiman_asr_amd.rnnt_ext.custom_lstm.lstm import CustomLSTM

        def make(isz, layers, drop):
            if custom:
                return CustomLSTM(input_size=isz, hidden_size=hidden_size, num_layers=layers, dropout=drop,
                                  rw_dropout=rw_dropout)
            return torch.nn.LSTM(input_size=isz, hidden_size=hidden_size, num_layers=layers, dropout=drop)

        if batch_norm:
            self.lstms = torch.nn.ModuleList(
                [make(input_size if i == 0 else hidden_size, 1, 0.0) for i in range(num_layers)])
            self.batch_norms = torch.nn.ModuleList(
                [torch.nn.BatchNorm1d(num_features=hidden_size) for _ in range(num_layers)])
            self.dropouts = (torch.nn.ModuleList([torch.nn.Dropout(dropout) for _ in range(num_layers)])
                             if dropout else None)
        else:
            self.lstm = make(input_size, num_layers, dropout)
            self.dropout = torch.nn.Dropout(dropout) if dropout else None

        # rnn.py:150-161
        for name, v in self.named_parameters():
            if "weight" in name or "bias" in name:
                v.data *= float(weights_init_scale)
        if forget_gate_bias is not None:
            for name, v in self.named_parameters():
                if "bias_ih" in name:
                    v.data[hidden_size:2 * hidden_size].fill_(forget_gate_bias)
                if "bias_hh" in name:
                    v.data[hidden_size:2 * hidden_size] *= float(hidden_hidden_bias_scale)
        self.using_custom_lstm = custom

    def forward(self, x, h=None):
        """x [T,B,I]; h = (h0, c0) each [L,B,H] or None -> (out, (h_n, c_n), all_states|None)."""
        if self.batch_norm:
            h_fl, c_fl = [], []
            for layer in range(self.num_layers):
                st = None if h is None else (h[0][layer].unsqueeze(0), h[1][layer].unsqueeze(0))
                x, (h_f, c_f), *_ = self.lstms[layer](x, st)
                x = self.batch_norms[layer](x.permute(1, 2, 0)).permute(2, 0, 1)
                if self.dropouts:
                    x = self.dropouts[layer](x)
                h_fl.append(h_f[0])
                c_fl.append(c_f[0])
            return x, (torch.stack(h_fl, 0), torch.stack(c_fl, 0)), None
        if self.using_custom_lstm:
            x, h, all_h = self.lstm(x, h)
        else:
            x, h = self.lstm(x, h)
            all_h = None
        if self.dropout:
            x = self.dropout(x)
        return x, h, all_h
