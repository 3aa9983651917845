"""Stage by stage: where does the HIP bf16 forward of the golden model's prediction network leave the bf16-storage oracle?
(tools/bf16_residual_table.py found g = joint_pred(prediction) 6e-3 apart while the oracle in fp32 arithmetic is bit-identical
to the oracle in f64.)  Prints, per stage, the fraction of elements that differ and the largest difference in bf16 ulps."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from caiman_asr_amd.rnnt.model import RNNT  # noqa: E402
from oracle import model as omodel  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "rnnt_mfma.npz"))
sd = {k[3:]: g[k] for k in g.files if k.startswith("sd.")}
cfg = dict(json.loads(str(g["cfg"])), custom_lstm=True, joint_apex_transducer="pack", joint_apex_relu_dropout=True)
V = int(g["n_classes"])
dev = torch.device("cuda:0")
m = RNNT(n_classes=V, **cfg)
m.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
m = m.to(dev).eval()
y = torch.tensor(g["y"], device=dev)
sdt = {k: torch.tensor(v, dtype=torch.float64) for k, v in sd.items()}


def report(name, hip, ref):
    hip = hip.detach().double().cpu().numpy()
    ref = ref.double().numpy()
    ulp = np.maximum(np.abs(ref), 1e-30)
    ulp = 2.0 ** (np.floor(np.log2(ulp)) - 7)          # bf16: 8 significant bits
    d = np.abs(hip - ref) / ulp
    print(f"{name:44s} differ {np.mean(d > 0):7.4f}  max {d.max():6.2f} ulp  max-abs/range {np.abs(hip - ref).max() / np.abs(ref).max():.2e}")


omodel._STORAGE = torch.bfloat16
with torch.no_grad():
    e = sdt["prediction.embed.weight"][torch.as_tensor(g["y"])]
    e = torch.cat([e.new_zeros(e.shape[0], 1, e.shape[2]), e], 1)
    x_in = omodel.rf(e).transpose(0, 1)
    o1, _ = omodel._lstm_stack(sdt, "prediction.dec_rnn.lstm", x_in, 1)
    o2, _ = omodel._lstm_stack(sdt, "prediction.dec_rnn.lstm", x_in, 2)
    gp = omodel._linear(o2.transpose(0, 1), sdt["joint_pred.weight"], sdt["joint_pred.bias"])
    with torch.autocast("cuda", dtype=torch.bfloat16):
        emb = m.prediction["embed"](y)
        pin = torch.cat([torch.zeros_like(emb[:, :1]), emb], 1).transpose(0, 1).contiguous()
        report("prediction input (embedding rows, as bf16)", pin.to(torch.bfloat16), x_in)
        dec = m.prediction["dec_rnn"]
        out, _, all_hid = dec(pin, None)
        lstm = dec.lstm if hasattr(dec, "lstm") else dec
        if all_hid is not None:
            report("LSTM layer 0 output h", all_hid[0][0], o1)
        report("LSTM layer 1 output h (stack output)", out, o2)
        report("g = joint_pred(stack output)", m.joint_pred(out.transpose(0, 1)), gp)
        # one layer alone through the stack path, and through the per-layer operator
        from caiman_asr_amd.rnnt_ext.custom_lstm.lstm import CustomLSTM
        one = CustomLSTM(64, 64, 1).to(dev)
        with torch.no_grad():
            for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
                getattr(one, n).copy_(torch.tensor(sd["prediction.dec_rnn.lstm." + n]))
        for pipe in (True, False):
            one.pipeline_layers = pipe
            report(f"single layer 0, pipeline_layers={pipe}", one(pin)[0], o1)
omodel._STORAGE = None

# ---- the library's small bf16 GEMM against an fp32 product rounded once
import torch.nn.functional as F  # noqa: E402

xb = out.transpose(0, 1).to(torch.bfloat16).contiguous()
wb, bb = m.joint_pred.weight.to(torch.bfloat16), m.joint_pred.bias.to(torch.bfloat16)
ref32 = (xb.float().reshape(-1, 64) @ wb.float().t() + bb.float()).to(torch.bfloat16).view(*xb.shape[:-1], -1)
report("fp32 product rounded once vs oracle", ref32, gp)
for flag in (True, False):
    torch.backends.cuda.matmul.allow_bf16_reduced_precision_reduction = flag
    report(f"F.linear bf16, reduced_precision_reduction={flag}", F.linear(xb, wb, bb), gp)
    report(f"torch.addmm bf16, same flag", torch.addmm(bb, xb.reshape(-1, 64), wb.t()).view_as(ref32), gp)
    report(f"torch.mm bf16 + bias in fp32, same flag", (torch.mm(xb.reshape(-1, 64), wb.t()).float() + bb.float()).to(torch.bfloat16).view_as(ref32), gp)
print("TunableOp / hipblaslt env:", {k: v for k, v in os.environ.items() if "BLAS" in k or "TUNABLE" in k})
