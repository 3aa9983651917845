"""Size-independent properties at the BASELINE sizes (base-85M: H = 1024, V = 8704, B = 32, LibriSpeech-like
lattices) where the CPU oracle would take minutes: flow conservation of the loss gradient, alpha/beta
agreement, pack == no-pack, state-passing equivalence of the MFMA LSTM and schedule invariance of the layer
pipeline."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _lattice(B=32, seed=0):
    rng = np.random.default_rng(seed)
    dur = np.clip(rng.normal(6.0, 2.0, size=B), 1.0, 9.0)
    f_len = np.ceil(np.ceil(dur * 100 / 3) / 2).astype(np.int32)
    y_len = np.maximum(1, np.round(3.3 * dur)).astype(np.int32)
    return f_len, y_len


def test_transducer_loss_properties_at_base_vocab():
    from caiman_asr_amd.rnnt_ext.transducer.loss import TransducerLoss

    V, blank = 8704, 8703
    f_len, y_len = _lattice()
    B, U = len(f_len), int(y_len.max())
    g = torch.Generator(device=DEV).manual_seed(1)
    label = torch.randint(0, blank, (B, U), device=DEV, dtype=torch.int32, generator=g)
    bo = torch.tensor(np.cumsum(f_len.astype(np.int64) * (y_len + 1)), device=DEV)
    rows = int(bo[-1])
    x = (torch.randn(rows, V, device=DEV, generator=g) * 2).to(torch.bfloat16).requires_grad_(True)
    fl, yl = torch.tensor(f_len, device=DEV), torch.tensor(y_len, device=DEV)
    dbg = []
    loss = TransducerLoss(packed_input=True)(x, label, fl, yl, blank, batch_offset=bo, max_f_len=int(f_len.max()),
                                             debug_list=dbg, delay_penalty=0.01)
    loss.sum().backward()
    alpha, beta = dbg
    assert torch.isfinite(loss).all() and (loss > 0).all()
    # (1) flow conservation: for every lattice cell the gradient sums to zero over the vocabulary
    rs = x.grad.float().sum(-1)
    assert rs.abs().max().item() < 3e-2  # bf16 rounding of 8704 terms of magnitude <= 1
    # occupancy: sum of the positive parts per utterance along any anti-diagonal is <= 1; total mass sanity:
    # (2) alpha / beta agreement: alpha(T-1,U) + null(T-1,U) == beta(0,0) == -loss
    denom = torch.logsumexp(x.detach().float(), -1)
    off = 0
    for b in range(B):
        T, U1 = int(f_len[b]), int(y_len[b]) + 1
        last = off + (T - 1) * U1 + U1 - 1
        null = x.detach()[last, blank].float() - denom[last]
        assert torch.allclose(alpha[b, T - 1, U1 - 1] + null, -loss[b], rtol=2e-4, atol=2e-3)
        assert torch.allclose(beta[b, 0, 0], -loss[b])
        off += T * U1
    # (3) pack == no-pack on the same logits (padded copy of a few utterances)
    sel = [0, 5, 17]
    Tm, Um = int(f_len[sel].max()), int(y_len[sel].max())
    xp = torch.zeros(len(sel), Tm, Um + 1, V, device=DEV, dtype=torch.bfloat16)
    starts = np.concatenate([[0], bo.cpu().numpy()])
    for i, b in enumerate(sel):
        T, U1 = int(f_len[b]), int(y_len[b]) + 1
        xp[i, :T, :U1] = x.detach()[starts[b]:starts[b] + T * U1].view(T, U1, V)
    lp = TransducerLoss()(xp, label[sel][:, :Um].contiguous(), fl[sel], yl[sel], blank, delay_penalty=0.01)
    # delay penalty depends on T only; labels beyond y_len are ignored
    assert torch.allclose(lp, loss.detach()[sel], rtol=1e-5)


def test_mfma_lstm_state_passing_and_schedule_invariance_at_base_width():
    from caiman_asr_amd.rnnt_ext.custom_lstm.lstm import CustomLSTM

    torch.manual_seed(0)
    H, L, B, T = 1024, 6, 32, 70
    m = CustomLSTM(2048, H, L, device=DEV)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(0.5)
    x = torch.randn(T, B, 2048, device=DEV)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        full, (hf, cf), (ah, ac) = m(x)
        a, sa, _ = m(x[:33])
        b, (hb, cb), _ = m(x[33:], sa)
        m.pipeline_layers = False
        ref, (hr, cr), _ = m(x)
        from caiman_asr_amd import _lib
        prev = _lib.lib().caiman_lstm_resident_mode(0)
        try:
            m.pipeline_layers = True
            full0, (hf0, cf0), _ = m(x)
        finally:
            _lib.lib().caiman_lstm_resident_mode(prev)
    # concat(A, B) == A then B | state  (training/tests/rnnt/test_model.py:107-296); the carried state is
    # rounded to the input dtype (fp32 here) exactly like the hand-over inside one call is rounded to bf16
    assert torch.allclose(torch.cat([a, b]).float(), full.float(), atol=2e-2)
    assert torch.allclose(hb.float(), hf.float(), atol=2e-2) and torch.allclose(cb.float(), cf.float(), atol=4e-2)
    # the layer pipeline only re-orders launches: with the per-timestep kernels on both sides it is bit-identical to
    # the layer-by-layer schedule; the weight-resident chunk kernels (default) sum the recurrent product in another
    # order and use the hardware exp / rcp: equal to the storage type's resolution
    assert torch.equal(ref, full0) and torch.equal(hr, hf0) and torch.equal(cr, cf0)
    assert torch.allclose(ref.float(), full.float(), atol=1e-2 * ref.float().abs().max().item())
    assert torch.allclose(cr.float(), cf.float(), atol=1e-2 * cr.float().abs().max().item())
    assert ah.shape == (L, T, B, H) and torch.equal(ah[-1], full)


def test_base_85m_step_at_128_utterances_per_gpu_runs_on_the_batch_tile_kernels():
    """BASELINE.json configs[2] per-GPU shape: base-85M, B = 128 (short utterances keep the lattice small), one bf16
    training step.  The recurrence must be served by the weight-resident batch-tile kernels (4 tiles of 32 rows), and the
    step must agree with the same step on the per-timestep kernels to the storage type's resolution."""
    import json
    import os

    from caiman_asr_amd import _lib
    from caiman_asr_amd.rnnt.loss import ApexTransducerLoss, get_packing_meta_data
    from caiman_asr_amd.rnnt.model import RNNT

    lib = _lib.lib()
    cfg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rnnt_cfg_base.json")))
    cfg = dict(cfg, enc_dropout=0.0, pred_dropout=0.0, joint_dropout=0.0)
    V, B = 8704, 128
    torch.manual_seed(3)
    m = RNNT(n_classes=V, **cfg).to(DEV).train()
    rng = np.random.default_rng(1)
    T1 = 40
    x_lens = rng.integers(20, T1 + 1, size=B)
    x_lens[0] = T1
    y_lens = rng.integers(2, 7, size=B)
    x = torch.tensor(rng.standard_normal((T1, B, 240)).astype(np.float32), device=DEV)
    y = torch.tensor(rng.integers(0, V - 1, size=(B, 6)), device=DEV)
    xl, yl = torch.tensor(x_lens), torch.tensor(y_lens)
    meta = get_packing_meta_data(xl, yl, 2, device=DEV)
    loss_fn = ApexTransducerLoss(blank_idx=V - 1, eos_idx=None, star_idx=None, packed_input=True)
    names = ("encoder.pre_rnn.lstm.weight_hh_l0", "encoder.post_rnn.lstm.weight_ih_l0", "prediction.dec_rnn.lstm.weight_hh_l1",
             "joint_net.2.weight")
    out = []
    for mode in (1, 0):
        prev = lib.caiman_lstm_resident_mode(mode)
        try:
            n0 = lib.caiman_lstm_resident_launches()
            m.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                logits, out_lens, _ = m(x, xl.to(DEV), y, yl.to(DEV), batch_offset=meta["batch_offset"],
                                        packed_batch=meta["packed_batch"])
                loss = loss_fn(logits, out_lens, y, yl.to(DEV), meta["batch_offset"], meta["max_f_len"])
            loss.backward()
            torch.cuda.synchronize()
            launches = lib.caiman_lstm_resident_launches() - n0
            grads = {n: p.grad.float().clone() for n, p in m.named_parameters() if n in names}
            assert len(grads) == len(names)
            out.append((float(loss.detach()), grads, launches))
        finally:
            lib.caiman_lstm_resident_mode(prev)
    (l1, g1, n1), (l0, g0, n0_) = out
    assert n1 > 0 and n0_ == 0, (n1, n0_)
    assert lib.caiman_lstm_resident_failures() == 0
    assert np.isfinite(l1) and abs(l1 - l0) <= 5e-3 * abs(l0), (l1, l0)
    for n in names:
        scale = float(g0[n].abs().max()) + 1e-12
        assert float((g1[n] - g0[n]).abs().max()) <= 6e-2 * scale, n


@pytest.mark.parametrize("size,V", [("base", 8704), ("large", 17408)])
def test_step_at_128_per_gpu_matches_the_bf16_storage_oracle(size, V):
    """("large": the same check on large-196M, BASELINE.json configs[3] at the reference's 128 utterances per GPU
    (docs/src/training/training_times.md:8): H = 1536 / 768 have weight-resident kernels for up to 32 rows only, so the
    library's wave calls run them on four 32-row slices of the batch -- csrc/lstm.hip::res_batch_slice -- and every launch
    of the step must still be a resident one.)
    BASELINE.json configs[2] per-GPU shape against the ORACLE (not against another kernel of this library): base-85M,
    B = 128, T = 40 frames, one bf16 training step on the weight-resident batch-tile kernels vs oracle.model.loss_and_grads
    rounded where the HIP path stores 16-bit values (`storage=torch.bfloat16`): loss and four gradients, one per
    sub-network, within 2e-2 of the tensor's range; the unrounded oracle as a loose second check.  (At this size the oracle
    runs in float32 arithmetic: it is itself one draw of the rounding-flip noise that oracle/bounds.py calibrates on the small
    models -- two fp32 implementations with the same bf16 rounding points differ by 1-1.5e-2 on the deepest gradient,
    profiles/r04_bf16_residual.md -- so the flat 2e-2 stays here.)"""
    import json
    import os

    from caiman_asr_amd import _lib
    from caiman_asr_amd.rnnt.loss import ApexTransducerLoss, get_packing_meta_data
    from caiman_asr_amd.rnnt.model import RNNT
    from oracle import model as omodel

    lib = _lib.lib()
    cfg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", f"rnnt_cfg_{size}.json")))
    cfg = dict(cfg, enc_dropout=0.0, pred_dropout=0.0, joint_dropout=0.0)
    B, T1 = 128, (40 if size == "base" else 24)      # (large: 24 frames keep the CPU oracle of the 196 M model near a minute)
    torch.manual_seed(5)
    m = RNNT(n_classes=V, **cfg).to(DEV).train()
    sd = {k: v.detach().float().cpu().numpy() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(2)
    x_lens = rng.integers(min(24, T1 - 8), T1 + 1, size=B)
    x_lens[0] = T1
    y_lens = rng.integers(1, 4, size=B)
    x = rng.standard_normal((T1, B, 240)).astype(np.float32)
    y = rng.integers(0, V - 1, size=(B, 3))
    xd, yd = torch.tensor(x, device=DEV), torch.tensor(y, device=DEV)
    xl, yl = torch.tensor(x_lens), torch.tensor(y_lens)
    meta = get_packing_meta_data(xl, yl, 2, device=DEV)
    loss_fn = ApexTransducerLoss(blank_idx=V - 1, eos_idx=None, star_idx=None, packed_input=True)
    names = ("encoder.pre_rnn.lstm.weight_hh_l0", "encoder.post_rnn.lstm.weight_ih_l0", "prediction.dec_rnn.lstm.weight_hh_l1",
             "joint_net.2.weight")
    n0 = lib.caiman_lstm_resident_launches()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits, out_lens, _ = m(xd, xl.to(DEV), yd, yl.to(DEV), batch_offset=meta["batch_offset"],
                                packed_batch=meta["packed_batch"])
        loss = loss_fn(logits, out_lens, yd, yl.to(DEV), meta["batch_offset"], meta["max_f_len"])
    loss.backward()
    torch.cuda.synchronize()
    assert lib.caiman_lstm_resident_launches() > n0 and lib.caiman_lstm_resident_failures() == 0
    got = {n: p.grad.double().cpu().numpy() for n, p in m.named_parameters() if n in names}
    del logits
    torch.cuda.empty_cache()
    for storage, loss_tol, grad_tol in ((torch.bfloat16, 2e-3, 2e-2), (None, 5e-3, 1e-1)):
        o_loss, o_grads, _ = omodel.loss_and_grads(sd, cfg, x, x_lens, y, y_lens, V - 1, dtype=torch.float32, storage=storage)
        assert abs(loss.item() - o_loss) <= loss_tol * abs(o_loss), (storage, loss.item(), o_loss)
        for n in names:
            r = o_grads[n]
            err = np.abs(got[n] - r).max() / (np.abs(r).max() + 1e-12)
            assert err <= grad_tol, (storage, n, err)
