"""Weight-resident LSTM chunk kernels vs the per-timestep launches: agreement, run-to-run determinism (a stale
hand-off would show up as a difference between two resident runs) and time of a pipelined stack.

    python tools/lstm_resident_bench.py [--layers 8] [--hidden 1024] [--batch 32] [--steps 256]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caiman_asr_amd import _lib  # noqa: E402
from caiman_asr_amd.rnnt_ext.custom_lstm.lstm import CustomLSTM  # noqa: E402

DEV = "cuda"


def run(m, x, h0, c0, w, mode, backward=True):
    lib = _lib.lib()
    lib.caiman_lstm_resident_mode(mode)
    m.zero_grad()
    xin = x.detach().clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y, (hn, cn), _ = m(xin, (h0, c0))
    out = [y.float(), hn.float(), cn.float()]
    if backward:
        (y.float() * w).sum().backward()
        out += [xin.grad.clone()] + [p.grad.clone() for p in m.parameters()]
    torch.cuda.synchronize()
    lib.caiman_lstm_resident_mode(1)
    return out


def agreement(T, B, I, H, L, dropout=0.0):
    torch.manual_seed(T + H)
    m = CustomLSTM(I, H, L, dropout=dropout, device=DEV)
    x = torch.randn(T, B, I, device=DEV)
    h0 = torch.randn(L, B, H, device=DEV) * 0.3
    c0 = torch.randn(L, B, H, device=DEV) * 0.3
    w = torch.randn(T, B, H, device=DEV)
    res = {}
    for name, mode in (("step", 0), ("res1", 1), ("res2", 1)):
        torch.manual_seed(7)
        res[name] = run(m, x, h0, c0, w, mode)
    det = all(torch.equal(a, b) for a, b in zip(res["res1"], res["res2"]))
    err = max(((a - b).abs().max() / (b.abs().max() + 1e-6)).item() for a, b in zip(res["res1"], res["step"]))
    same_y = torch.equal(res["res1"][0], res["step"][0])
    return {"T": T, "B": B, "H": H, "L": L, "dropout": dropout, "deterministic": det, "max_rel_err_vs_step": err,
            "y_bit_equal": same_y}


def timing(T, B, I, H, L, reps):
    torch.manual_seed(0)
    m = CustomLSTM(I, H, L, device=DEV)
    x = torch.randn(T, B, I, device=DEV)
    h0 = torch.zeros(L, B, H, device=DEV)
    c0 = torch.zeros(L, B, H, device=DEV)
    out = {}
    lib = _lib.lib()
    for name, mode in (("step", 0), ("resident", 1)):
        lib.caiman_lstm_resident_mode(mode)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            for _ in range(3):
                m(x, (h0, c0))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                m(x, (h0, c0))
            torch.cuda.synchronize()
        out[name + "_ms"] = (time.perf_counter() - t0) * 1e3 / reps
        xg = x.clone().requires_grad_(True)
        for i in range(3 + reps):
            if i == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            m.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y, _, _ = m(xg, (h0, c0))
            y.float().sum().backward()
        torch.cuda.synchronize()
        out[name + "_fwd_bwd_ms"] = (time.perf_counter() - t0) * 1e3 / reps
    # phase timers of one workgroup (mode 2): where a timestep's time goes
    import ctypes
    lib.caiman_lstm_resident_mode(2)
    buf = (ctypes.c_uint32 * 10)()
    lib.caiman_lstm_resident_profile(buf)
    lib.caiman_lstm_resident_profile_bwd2((ctypes.c_uint32 * 16)())
    for _ in range(reps):
        m.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y, _, _ = m(xg, (h0, c0))
        y.float().sum().backward()
    lib.caiman_lstm_resident_profile(buf)
    buf2 = (ctypes.c_uint32 * 16)()
    lib.caiman_lstm_resident_profile_bwd2(buf2)
    v2 = list(buf2)
    if v2[7]:   # the 2-D split backward kernel served the backward launches
        names2 = ("wait_quarter", "gather_mfma", "partials_out_drain", "wait_group", "partials_in_epilogue", "drain_barrier")
        out["bwd2_us_per_timestep"] = {k: round(v2[i] * 0.01 / v2[7], 3) for i, k in enumerate(names2)}
        out["bwd2_timesteps"] = v2[7]
        names3 = ("dma_issue", "s0_wait", "s0_mfma", "s1_wait", "s1_mfma", "later_waits", "later_mfma")
        out["bwd2_gather_split_us"] = {k: round(v2[8 + i] * 0.01 / v2[7], 3) for i, k in enumerate(names3)}
    v = list(buf)
    names = ("wait", "operand_to_lds", "mfma_cell", "drain_barrier")
    for base, tag in ((0, "fwd"), (5, "bwd")):
        n = max(v[base + 4], 1)
        out[tag + "_us_per_timestep"] = {k: round(v[base + i] * 0.01 / n, 3) for i, k in enumerate(names)}
        out[tag + "_timesteps"] = v[base + 4]
    lib.caiman_lstm_resident_mode(1)
    out.update({"T": T, "B": B, "H": H, "L": L})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--skip-agreement", action="store_true")
    a = ap.parse_args()
    lib = _lib.lib()
    if not a.skip_agreement:
        for cfg in [(40, 3, 16, 64, 2, 0.0), (70, 32, 48, 128, 3, 0.0), (150, 17, 64, 256, 4, 0.0), (150, 8, 64, 512, 3, 0.3),
                    (200, 32, 256, 1024, 8, 0.0), (200, 32, 256, 768, 2, 0.2)]:
            print(json.dumps(agreement(*cfg)), flush=True)
            print(json.dumps({"failures": lib.caiman_lstm_resident_failures()}), flush=True)
    print(json.dumps(timing(a.steps, a.batch, a.hidden, a.hidden, a.layers, a.reps)), flush=True)
    print(json.dumps({"failures": lib.caiman_lstm_resident_failures()}), flush=True)


if __name__ == "__main__":
    main()
