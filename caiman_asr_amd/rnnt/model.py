"""RNNT network: stacked-LSTM encoder with time stacking, LSTM prediction network, joint.

Interface mirror of training/caiman_asr_train/rnnt/model.py (`StackTime` :35-49, `RNNT`
:52-491, `label_collate` :494-519): same constructor keywords (so `RNNT(n_classes,
**yaml['rnnt'])` works on the unchanged training/configs YAMLs), same method names, same
parameter names / state_dict keys (checked against the reference's model_schema in tests).
"""
from itertools import chain
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from caiman_asr_amd.rnnt.joint import TransducerJoint
from caiman_asr_amd.rnnt.rnn import rnn
from caiman_asr_amd.rnnt.state import (EncoderState, PredNetState, RNNTState, get_pred_net_state,
                                       maybe_get_last_nonpadded)


class StackTime(nn.Module):
    """out[t'] = cat(x[f*t'], ..., x[f*t'+f-1]) on the feature axis (zeros past the end);
    lens' = ceil(lens / f).  One reshape instead of the reference's f shifted copies."""

    def __init__(self, factor):
        super().__init__()
        self.factor = int(factor)

    def forward(self, x, x_lens):
        T, B, H = x.shape
        f = self.factor
        pad = (-T) % f
        if pad:
            x = torch.cat([x, x.new_zeros(pad, B, H)], 0)
        x = x.view((T + pad) // f, f, B, H).transpose(1, 2).reshape((T + pad) // f, B, f * H)
        return x, (x_lens.int() + f - 1) // f


class RNNT(nn.Module):
    """Recurrent Neural Network Transducer.  Argument meaning: reference model.py:55-77."""

    def __init__(self, n_classes, in_feats, enc_n_hid, enc_batch_norm, pred_batch_norm, enc_pre_rnn_layers,
                 enc_post_rnn_layers, enc_stack_time_factor, enc_dropout, pred_dropout, joint_dropout,
                 pred_n_hid, pred_rnn_layers, joint_n_hid, forget_gate_bias, custom_lstm=False, quantize=False,
                 enc_rw_dropout=0.0, pred_rw_dropout=0.0, hidden_hidden_bias_scale=0.0, weights_init_scale=1.0,
                 enc_lr_factor=1.0, pred_lr_factor=1.0, joint_enc_lr_factor=1.0, joint_pred_lr_factor=1.0,
                 joint_net_lr_factor=1.0, joint_apex_transducer=None, joint_apex_relu_dropout=False,
                 enc_freeze=False, gpu_unavailable=False):
        super().__init__()
        if joint_apex_relu_dropout and not joint_apex_transducer:
            raise ValueError("Can't have joint_apex_relu_dropout=True without bool(joint_apex_transducer)==True")
        if joint_apex_transducer is not None:
            assert joint_apex_transducer in {"pack", "not_pack"}
        self._module_to_lr_factor = {
            "encoder": enc_lr_factor, "prediction": pred_lr_factor, "joint_enc": joint_enc_lr_factor,
            "joint_pred": joint_pred_lr_factor, "joint_net": joint_net_lr_factor}
        self.pred_n_hid = pred_n_hid
        self.enc_stack_time_factor = enc_stack_time_factor
        self.encoder_pipe = True    # one layer pipeline across pre_rnn / StackTime / post_rnn when the stacks allow it
        self.joint_fc_nt_backward = True   # input gradient of the joint projection against a [K, N] weight copy (10 % faster GEMM)
        self.pred_in_encoder_pipe = False  # opt-in: the prediction network's LSTM steps in the same launches (measured: no gain)

        common = dict(forget_gate_bias=forget_gate_bias, custom_lstm=custom_lstm, quantize=quantize,
                      hidden_hidden_bias_scale=hidden_hidden_bias_scale, weights_init_scale=weights_init_scale,
                      gpu_unavailable=gpu_unavailable)
        self.encoder = nn.ModuleDict({
            "pre_rnn": rnn(input_size=in_feats, hidden_size=enc_n_hid, num_layers=enc_pre_rnn_layers,
                           batch_norm=enc_batch_norm, rw_dropout=enc_rw_dropout, dropout=enc_dropout,
                           tensor_name="pre_rnn", **common),
            "stack_time": StackTime(enc_stack_time_factor),
            "post_rnn": rnn(input_size=enc_stack_time_factor * enc_n_hid, hidden_size=enc_n_hid,
                            num_layers=enc_post_rnn_layers, batch_norm=enc_batch_norm, rw_dropout=enc_rw_dropout,
                            dropout=enc_dropout, tensor_name="post_rnn", **common),
        })
        self.encoder.requires_grad_(not enc_freeze)
        self.prediction = nn.ModuleDict({
            "embed": nn.Embedding(n_classes - 1, pred_n_hid),  # blank is never an input
            "dec_rnn": rnn(input_size=pred_n_hid, hidden_size=pred_n_hid, num_layers=pred_rnn_layers,
                           batch_norm=pred_batch_norm, rw_dropout=pred_rw_dropout, dropout=pred_dropout,
                           tensor_name="dec_rnn", **common),
        })
        self.joint_pred = nn.Linear(pred_n_hid, joint_n_hid)
        self.joint_enc = nn.Linear(enc_n_hid, joint_n_hid)
        self.joint_net = nn.Sequential(nn.ReLU(inplace=True), nn.Dropout(p=joint_dropout),
                                       nn.Linear(joint_n_hid, n_classes))
        self.relu_drop = self.joint_net[:2]
        self.joint_fc = self.joint_net[-1]
        self.joint_apex_transducer = joint_apex_transducer
        if joint_apex_transducer is not None:
            pack_output = joint_apex_transducer == "pack"
            if joint_apex_relu_dropout:
                self.apex_joint = TransducerJoint(pack_output=pack_output, relu=True, dropout=True,
                                                  dropout_prob=joint_dropout)
            else:
                self.apex_joint = TransducerJoint(pack_output=pack_output)

    # ---- forward pieces -------------------------------------------------------
    def enc_pred(self, x, x_lens, y, y_lens, pred_net_state: Optional[PredNetState] = None,
                 enc_state: Optional[EncoderState] = None):
        if self.encoder_pipe and self.pred_in_encoder_pipe and x.is_cuda:
            out = self._enc_pred_one_pipeline(x, x_lens, y, y_lens, pred_net_state, enc_state)
            if out is not None:
                return out
        return self.enc_pred_static(x, x_lens, y, y_lens, self.encode, self.predict,
                                    pred_net_state=pred_net_state, enc_state=enc_state)

    def _enc_pred_one_pipeline(self, x, x_lens, y, y_lens, pred_net_state, enc_state):
        """Encoder and prediction network in the same LSTM launches (encoder_pipe.py); None if not covered."""
        y = label_collate(y)
        from caiman_asr_amd.train_utils.overlap import embedding

        emb = embedding(self.prediction["embed"], y)            # predict(): embedding, SOS row in front
        Bn, _, E = emb.shape
        if pred_net_state is None:
            start = torch.zeros((Bn, 1, E), device=emb.device, dtype=emb.dtype)
        else:
            start = self.prediction["embed"](pred_net_state.last_token).to(device=emb.device, dtype=emb.dtype)
        pred_in = torch.cat([start, emb], dim=1).transpose(0, 1).contiguous()
        merged = self._encode_one_pipeline(x, x_lens, enc_state, pred_in=pred_in,
                                           pred_state=pred_net_state.next_to_last_pred_state if pred_net_state else None)
        if merged is None:
            return None
        yt, lens2, all_pre, all_post, yp, all_pred = merged
        dec = self.prediction["dec_rnn"]
        if dec.dropout:
            yp = dec.dropout(yp)
        f = self._joint_in(self.joint_enc, yt.transpose(0, 1))
        g = self._joint_in(self.joint_pred, yp.transpose(0, 1))
        g_lens = y_lens + 1
        new_enc = EncoderState(pre_rnn=maybe_get_last_nonpadded(all_pre, x_lens),
                               post_rnn=maybe_get_last_nonpadded(all_post, lens2))
        new_pred = get_pred_net_state(y, all_pred, y_lens, g_lens)
        state = RNNTState(enc_state=new_enc, pred_net_state=new_pred) if new_pred is not None else None
        return (f, lens2), (g, g_lens), state

    @staticmethod
    def enc_pred_static(x, x_lens, y, y_lens, encode, predict, pred_net_state: Optional[PredNetState] = None,
                        enc_state: Optional[EncoderState] = None):
        y = label_collate(y)
        f, x_lens, new_enc_state = encode(x, x_lens, enc_state=enc_state)
        g, _, all_pred_hid = predict(
            y, pred_state=(pred_net_state.next_to_last_pred_state if pred_net_state else None), add_sos=True,
            special_sos=pred_net_state.last_token if pred_net_state else None)
        g_lens = y_lens + 1
        new_pred = get_pred_net_state(y, all_pred_hid, y_lens, g_lens)
        rnnt_state = None
        if new_enc_state is not None and new_pred is not None:
            rnnt_state = RNNTState(enc_state=new_enc_state, pred_net_state=new_pred)
        return (f, x_lens), (g, g_lens), rnnt_state

    def forward(self, x, x_lens, y, y_lens, pred_net_state: Optional[PredNetState] = None, batch_offset=None,
                enc_state: Optional[EncoderState] = None, packed_batch: Optional[int] = None):
        """`packed_batch` (= batch_offset[-1], known on the host from get_packing_meta_data) is an
        optional extra over the reference signature that avoids a device sync in `joint`."""
        (f, x_lens), (g, g_lens), new_state = self.enc_pred(x, x_lens, y, y_lens, pred_net_state=pred_net_state,
                                                            enc_state=enc_state)
        out = self.joint(f, g, x_lens, g_lens, batch_offset, packed_batch=packed_batch)
        return out, x_lens, new_state

    def _encode_one_pipeline(self, x, x_lens, enc_state, pred_in=None, pred_state=None):
        """pre_rnn -> StackTime -> post_rnn as ONE layer pipeline (rnnt_ext/custom_lstm/encoder_pipe.py), or None when
        the configuration is not covered (then the stacks run one after the other).  With `pred_in` [U+1, B, E] the
        prediction network's LSTM steps ride in the same launches (None is returned if it cannot)."""
        pre, post = self.encoder["pre_rnn"], self.encoder["post_rnn"]
        if not (self.encoder_pipe and getattr(pre, "using_custom_lstm", False) and not pre.batch_norm
                and not post.batch_norm):
            return None
        from caiman_asr_amd.rnnt_ext.custom_lstm import encoder_pipe as ep

        a, b, f = pre.lstm, post.lstm, self.enc_stack_time_factor
        gate_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
        if not (ep.eligible(x, a.hidden_size, a.num_layers, b.num_layers, gate_dtype, f) and a.hidden_size == b.hidden_size
                and a.hard == b.hard and a.rw_dropout == 0.0 and b.rw_dropout == 0.0 and a.bl_dropout == b.bl_dropout
                and a.pipeline_layers and b.pipeline_layers):
            return None
        p = None
        if pred_in is not None:
            dec = self.prediction["dec_rnn"]
            if not getattr(dec, "using_custom_lstm", False) or dec.batch_norm:
                return None
            p = dec.lstm
            if not (p.hidden_size % 32 == 0 and p.hidden_size <= a.hidden_size and p.hard == a.hard and p.rw_dropout == 0.0
                    and p.pipeline_layers and a.num_layers + b.num_layers + p.num_layers <= 8
                    and pred_in.shape[1] == x.shape[1]):
                return None
        # (batches the batch-tile LSTM kernels do not take -- B > 32 at H = 1536 / 768 -- are cut into 32-row slices inside the
        # library's wave calls, csrc/lstm.hip::res_batch_slice: everything else of the pipeline runs at the full batch)
        y, all_pre, all_post, yp, all_pred = ep.encoder_pipe(
            x, a, b, f, enc_state.pre_rnn if enc_state else None, enc_state.post_rnn if enc_state else None,
            pred=p, xp=pred_in, pred_state=pred_state)
        if post.dropout:
            y = post.dropout(y)
        return y, (x_lens.int() + f - 1) // f, all_pre, all_post, yp, all_pred

    def encode(self, x, x_lens, enc_state: Optional[EncoderState] = None):
        """x [T,B,I], x_lens [B] -> f [B,T',Hj], lens', EncoderState|None."""
        merged = self._encode_one_pipeline(x, x_lens, enc_state)
        if merged is not None:
            y, lens2, all_pre, all_post, _, _ = merged
            pre_last = maybe_get_last_nonpadded(all_pre, x_lens)
            post_last = maybe_get_last_nonpadded(all_post, lens2)
            return self._joint_in(self.joint_enc, y.transpose(0, 1)), lens2, EncoderState(pre_rnn=pre_last, post_rnn=post_last)
        x, _, all_pre = self.encoder["pre_rnn"](x, enc_state.pre_rnn if enc_state else None)
        pre_last = maybe_get_last_nonpadded(all_pre, x_lens)
        x, x_lens = self.encoder["stack_time"](x, x_lens)
        x, _, all_post = self.encoder["post_rnn"](x, enc_state.post_rnn if enc_state else None)
        post_last = maybe_get_last_nonpadded(all_post, x_lens)
        x = self._joint_in(self.joint_enc, x.transpose(0, 1))
        new_state = None
        if all_pre is not None and all_post is not None:
            new_state = EncoderState(pre_rnn=pre_last, post_rnn=post_last)
        return x, x_lens, new_state

    def _joint_in(self, lin, x):
        """joint_enc / joint_pred (torch.nn.Linear in the reference, model.py:82-83 there): on the GPU in training the same
        product with fp32 parameter gradients (train_utils/overlap.py::linear_f32_grads); plain module call otherwise."""
        if x.is_cuda and torch.is_grad_enabled() and (lin.weight.requires_grad or x.requires_grad):
            from caiman_asr_amd.train_utils.overlap import linear_f32_grads

            return linear_f32_grads(x, lin.weight, lin.bias)
        # contiguous rows: torch's linear fuses the bias into the GEMM only then (else: a second rounding in the 16-bit type)
        return lin(x.contiguous() if x.is_cuda else x)

    def predict(self, y, pred_state=None, add_sos: bool = True, special_sos=None):
        """y [B,U] (or None for a single zero-embedding step) -> g [B,U+1,Hj], (h,c), all states."""
        if y is not None:
            from caiman_asr_amd.train_utils.overlap import embedding

            y = embedding(self.prediction["embed"], y)    # the module's lookup; on the GPU the table's gradient by one kernel
        else:
            B = 1 if pred_state is None else pred_state[0].size(1)
            y = torch.zeros((B, 1, self.pred_n_hid), device=self.joint_enc.weight.device,
                            dtype=self.joint_enc.weight.dtype)
        if add_sos:
            B, U, H = y.shape
            if special_sos is None:
                start = torch.zeros((B, 1, H), device=y.device, dtype=y.dtype)
            else:
                start = self.prediction["embed"](special_sos).to(device=y.device, dtype=y.dtype)
            y = torch.cat([start, y], dim=1).contiguous()
        y = y.transpose(0, 1)
        g, hid, all_hid = self.prediction["dec_rnn"](y, pred_state)
        g = self._joint_in(self.joint_pred, g.transpose(0, 1))
        return g, hid, all_hid

    def joint(self, f, g, f_len=None, g_len=None, batch_offset=None, packed_batch: Optional[int] = None):
        """f [B,T,H], g [B,U+1,H] -> logits [B,T,U+1,V], or packed [rows,V] when the joint packs."""
        if self.joint_apex_transducer is None or f_len is None or g_len is None:
            h = self.relu_drop(self.torch_transducer_joint(f, g, f_len, g_len))
        else:
            assert batch_offset is not None
            if packed_batch is None:
                packed_batch = batch_offset[-1].item()
            h = self.apex_joint(f, g, f_len, g_len, batch_offset=batch_offset, packed_batch=packed_batch)
            if not self.apex_joint.relu:
                h = self.relu_drop(h)
        if self.training and h.is_cuda and torch.is_grad_enabled() and h.dtype in (torch.float16, torch.bfloat16) \
                and self.joint_fc_nt_backward:
            from caiman_asr_amd.train_utils.overlap import linear_transposed_backward

            return linear_transposed_backward(h, self.joint_fc.weight, self.joint_fc.bias)
        return self.joint_fc(h)

    @staticmethod
    def torch_transducer_joint(f, g, f_len=None, g_len=None):
        return f.unsqueeze(2) + g.unsqueeze(1)

    # ---- optimiser / checkpoint surface ------------------------------------------
    def param_groups(self, lr, return_module_name=False):
        out = []
        for name, lr_factor in self._module_to_lr_factor.items():
            res = {"params": self._chain_params(getattr(self, name)), "lr": lr * lr_factor}
            if return_module_name:
                res["module_name"] = name
            out.append(res)
        return out

    def _chain_params(self, *layers):
        return chain(*[layer.parameters() for layer in layers])

    def state_dict(self, *args, **kwargs):
        """joint_fc.* are aliases of joint_net.2.* and are dropped (model.py:464-473)."""
        sd = super().state_dict(*args, **kwargs)
        prefix = kwargs.get("prefix", "")
        sd.pop(prefix + "joint_fc.weight", None)
        sd.pop(prefix + "joint_fc.bias", None)
        return sd

    def load_state_dict(self, state_dict, strict=True):
        state_dict = dict(state_dict)
        if "joint_net.2.weight" in state_dict:
            state_dict["joint_fc.weight"] = state_dict["joint_net.2.weight"].detach().clone()
        if "joint_net.2.bias" in state_dict:
            state_dict["joint_fc.bias"] = state_dict["joint_net.2.bias"].detach().clone()
        return super().load_state_dict(state_dict, strict=strict)


def label_collate(labels):
    """List of label index lists -> padded LongTensor [B, Umax]; tensors pass through as int64."""
    if isinstance(labels, torch.Tensor):
        return labels.type(torch.int64)
    if not isinstance(labels, (list, tuple)):
        raise ValueError(f"`labels` should be a list or tensor not {type(labels)}")
    max_len = max(len(label) for label in labels)
    cat = np.zeros((len(labels), max_len), dtype=np.int32)
    for e, lab in enumerate(labels):
        cat[e, :len(lab)] = lab
    return torch.LongTensor(cat)
