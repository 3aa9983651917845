"""GPU parity at model level: the HIP-backed RNNT vs (a) outputs of the reference's own RNNT class
(tests/golden, from oracle/gen_golden.py) and (b) the CPU oracle's loss and parameter gradients."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def load(tag):
    g = np.load(os.path.join(GOLD, f"rnnt_{tag}.npz"))
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd.")}
    return g, sd, json.loads(str(g["cfg"]))


def build(tag, **over):
    from caiman_asr_amd.rnnt.model import RNNT

    g, sd, cfg = load(tag)
    cfg = dict(cfg, custom_lstm=True, **over)
    m = RNNT(n_classes=int(g["n_classes"]), **cfg)
    m.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    return g, sd, cfg, m.to(DEV)


@pytest.mark.parametrize("tag", ["tiny", "mfma"])
def test_forward_matches_reference_fp32(tag):
    g, sd, cfg, m = build(tag)
    m.eval()
    x, xl = torch.tensor(g["x"], device=DEV), torch.tensor(g["x_lens"], device=DEV)
    y, yl = torch.tensor(g["y"], device=DEV), torch.tensor(g["y_lens"], device=DEV)
    with torch.no_grad():
        f, fl, _ = m.encode(x, xl)
        gg, _, _ = m.predict(y)
        logits, out_lens, _ = m(x, xl, y, yl)
    assert np.array_equal(fl.cpu().numpy(), g["f_lens"])
    assert np.allclose(f.cpu().numpy(), g["f"], atol=5e-5)
    assert np.allclose(gg.cpu().numpy(), g["g"], atol=5e-5)
    assert np.allclose(logits.cpu().numpy(), g["logits"], atol=3e-4)


def test_forward_bf16_autocast_close_to_reference():
    g, sd, cfg, m = build("mfma")
    m.eval()
    x, xl = torch.tensor(g["x"], device=DEV), torch.tensor(g["x_lens"], device=DEV)
    y, yl = torch.tensor(g["y"], device=DEV), torch.tensor(g["y_lens"], device=DEV)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        logits, _, _ = m(x, xl, y, yl)
    assert logits.dtype == torch.bfloat16
    ref = g["logits"]
    err = np.abs(logits.float().cpu().numpy() - ref)
    # 5 stacked LSTM layers in bf16 storage: loose, but far below the logit scale (~10)
    assert err.max() < 0.12 * np.abs(ref).max() and err.mean() < 0.02 * np.abs(ref).max()


@pytest.mark.parametrize("tag,joint", [("tiny", "pack"), ("tiny", "not_pack"), ("mfma", "pack")])
def test_train_step_loss_and_all_parameter_grads_match_oracle(tag, joint):
    from caiman_asr_amd.rnnt.loss import ApexTransducerLoss, LossModifiers, get_packing_meta_data
    from oracle import model as omodel

    g, sd, cfg, m = build(tag, joint_apex_transducer=joint, joint_apex_relu_dropout=True)
    m.train()  # all dropout probabilities are 0 in the golden config
    V = int(g["n_classes"])
    x, xl = torch.tensor(g["x"], device=DEV), torch.tensor(g["x_lens"], device=DEV)
    y, yl = torch.tensor(g["y"], device=DEV), torch.tensor(g["y_lens"], device=DEV)
    meta = get_packing_meta_data(xl, yl, 2)
    logits, out_lens, _ = m(x, xl, y, yl, batch_offset=meta["batch_offset"])
    loss_fn = ApexTransducerLoss(blank_idx=V - 1, eos_idx=None, star_idx=None, packed_input=(joint == "pack"))
    loss = loss_fn(logits, out_lens, y, yl, meta["batch_offset"], meta["max_f_len"],
                   LossModifiers(delay_penalty=0.01, eos_penalty=0.0, star_penalty=1.0))
    loss.backward()
    ref_loss, ref_grads, ref_logits = omodel.loss_and_grads(sd, cfg, g["x"], g["x_lens"], g["y"], g["y_lens"], V - 1,
                                                            delay_penalty=0.01)
    assert loss.item() == pytest.approx(ref_loss, rel=2e-5)
    if joint == "pack":
        assert logits.dim() == 2 and logits.shape[0] == int(g["batch_offset"][-1])
    got = dict(m.named_parameters())
    for name, rg in ref_grads.items():
        gg = got[name].grad
        assert gg is not None, name
        scale = max(np.abs(rg).max(), 1e-3)
        assert np.allclose(gg.cpu().numpy(), rg, atol=3e-4 * scale + 2e-6), name


def test_joint_pack_equals_broadcast_add_and_padding_is_minus_one():
    # training/tests/rnnt/test_model.py:34-64
    from caiman_asr_amd.rnnt.joint import TransducerJoint

    torch.manual_seed(0)
    B, T, U, H = 3, 7, 5, 24
    f = torch.randn(B, T, H, device=DEV, requires_grad=True)
    g = torch.randn(B, U, H, device=DEV, requires_grad=True)
    f_len = torch.tensor([7, 4, 6], device=DEV)
    g_len = torch.tensor([5, 5, 2], device=DEV)
    ref = f.unsqueeze(2) + g.unsqueeze(1)
    out = TransducerJoint(pack_output=False)(f, g, f_len, g_len)
    bo = torch.cumsum(f_len * g_len, 0)
    packed = TransducerJoint(pack_output=True)(f, g, f_len, g_len, batch_offset=bo, packed_batch=int(bo[-1]))
    off = 0
    for b in range(B):
        t, u = int(f_len[b]), int(g_len[b])
        assert torch.equal(out[b, :t, :u], ref[b, :t, :u])
        assert (out[b, t:] == -1).all() and (out[b, :, u:] == -1).all()
        assert torch.equal(packed[off:off + t * u].view(t, u, H), ref[b, :t, :u])
        off += t * u
    # gradients: sum over the valid region only
    w = torch.randn_like(packed)
    packed.backward(w)
    f2, g2 = f.detach().clone().requires_grad_(True), g.detach().clone().requires_grad_(True)
    ref2 = f2.unsqueeze(2) + g2.unsqueeze(1)
    off, tot = 0, 0
    for b in range(B):
        t, u = int(f_len[b]), int(g_len[b])
        tot = tot + (ref2[b, :t, :u] * w[off:off + t * u].view(t, u, H)).sum()
        off += t * u
    tot.backward()
    assert torch.allclose(f.grad, f2.grad, atol=1e-5) and torch.allclose(g.grad, g2.grad, atol=1e-5)


def test_joint_relu_dropout_semantics():
    from caiman_asr_amd.rnnt.joint import TransducerJoint

    torch.manual_seed(1)
    B, T, U, H = 2, 6, 4, 64
    f = torch.randn(B, T, H, device=DEV, requires_grad=True)
    g = torch.randn(B, U, H, device=DEV, requires_grad=True)
    fl, gl = torch.tensor([6, 6], device=DEV), torch.tensor([4, 4], device=DEV)
    j = TransducerJoint(pack_output=False, relu=True, dropout=True, dropout_prob=0.3)
    j.eval()  # dropout off in eval (test_model.py:67-104)
    assert torch.equal(j(f, g, fl, gl), torch.relu(f.unsqueeze(2) + g.unsqueeze(1)))
    j.train()
    out = j(f, g, fl, gl)
    ref = torch.relu(f.unsqueeze(2) + g.unsqueeze(1))
    kept = out != 0
    assert torch.allclose(out[kept], ref[kept] / 0.7, rtol=1e-6)
    frac = 1 - kept[ref > 0].float().mean().item()
    assert 0.25 < frac < 0.35  # ~30 % of the positive activations dropped
    out.sum().backward()
    # d/df = sum_u mask/(1-p)
    assert torch.allclose(f.grad, (kept.float() / 0.7).sum(2), atol=1e-5)


def test_lamb_step_matches_python_restatement():
    from caiman_asr_amd.train_utils.optimizer import FusedLAMB

    torch.manual_seed(0)
    shapes = [(300, 70), (70,), (1, 5), (2000, 33)]
    params = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    groups = [dict(params=params[:2], lr=4e-3), dict(params=params[2:], lr=4e-3 * 0.343)]
    p0 = [p.detach().clone().double() for p in params]
    opt = FusedLAMB(groups, lr=4e-3, betas=(0.9, 0.999), eps=1e-9, weight_decay=1e-2, max_grad_norm=1.0,
                    ema_decay=0.999)
    m = [torch.zeros_like(p) for p in p0]
    v = [torch.zeros_like(p) for p in p0]
    ema = [p.clone() for p in p0]
    lrs = [4e-3, 4e-3, 4e-3 * 0.343, 4e-3 * 0.343]
    for step in range(1, 4):
        grads = [torch.randn(s, device=DEV) * (3.0 if step == 1 else 0.01) for s in shapes]
        for p, gr in zip(params, grads):
            p.grad.copy_(gr)
        opt.step(zero_grad=True)
        gd = [gr.double() for gr in grads]
        gnorm = torch.sqrt(sum((x ** 2).sum() for x in gd))
        clip = gnorm / 1.0 if gnorm > 1.0 else 1.0
        for i in range(len(p0)):
            sg = gd[i] / clip
            m[i] = 0.9 * m[i] + 0.1 * sg
            v[i] = 0.999 * v[i] + 0.001 * sg * sg
            upd = (m[i] / (1 - 0.9 ** step)) / (torch.sqrt(v[i] / (1 - 0.999 ** step)) + 1e-9) + 1e-2 * p0[i]
            ratio = p0[i].norm() / upd.norm()
            p0[i] = p0[i] - lrs[i] * ratio * upd
            ema[i] = 0.999 * ema[i] + 0.001 * p0[i]
        assert opt.grad_norm.item() == pytest.approx(gnorm.item(), rel=1e-5)
        for i, p in enumerate(params):
            assert torch.allclose(p.double(), p0[i], atol=2e-6, rtol=2e-5), (step, i)
            assert (p.grad == 0).all()
        for p, e in opt.ema_tensors().items():
            i = [id(q) for q in params].index(id(p))
            assert torch.allclose(e.double(), ema[i], atol=2e-6, rtol=2e-5)
    # a non-finite gradient leaves parameters and moments untouched
    before = [p.detach().clone() for p in params]
    params[0].grad[0, 0] = float("nan")
    opt.step()
    assert opt.last_step_applied.item() == 0
    for p, b in zip(params, before):
        assert torch.equal(p, b)


def test_batch_split_step_equals_plain_step():
    # training/tests/rnnt/test_batch_split.py:103-144 (loss and every gradient, rtol/atol 1e-4)
    from argparse import Namespace

    from caiman_asr_amd.rnnt.loss import ApexTransducerLoss, LossModifiers
    from caiman_asr_amd.train_utils.batch_splitting import train_step_batch_split
    from caiman_asr_amd.train_utils.core import train_step

    mods = LossModifiers(delay_penalty=0.01, eos_penalty=0.0, star_penalty=1.0)
    res = []
    for split in (1, 2):
        g, sd, cfg, m = build("tiny", joint_apex_transducer="pack", joint_apex_relu_dropout=True)
        m.train()
        V = int(g["n_classes"])
        # 4 utterances: repeat the golden batch of 3 with one duplicate
        idx = [0, 1, 2, 1]
        x = torch.tensor(g["x"][:, idx], device=DEV)
        xl, y, yl = torch.tensor(g["x_lens"][idx]), torch.tensor(g["y"][idx], device=DEV), torch.tensor(g["y_lens"][idx])
        loss_fn = ApexTransducerLoss(blank_idx=V - 1, eos_idx=None, star_idx=None, packed_input=True)
        args = Namespace(grad_accumulation_batches=1, batch_split_factor=split, no_amp=True, num_gpus=1)
        fn = train_step if split == 1 else train_step_batch_split
        loss, nan, _ = fn(m, loss_fn, args, x, xl, y, yl, None, None, mods)
        assert not nan
        res.append((loss, {n: p.grad.clone() for n, p in m.named_parameters()}))
    assert res[0][0] == pytest.approx(res[1][0], rel=1e-5)
    for n, ga in res[0][1].items():
        assert torch.allclose(ga, res[1][1][n], rtol=1e-4, atol=1e-5), n


def test_beam_expander_topk_and_pruning():
    from caiman_asr_amd.rnnt.beam import BeamExpander

    g, sd, cfg, m = build("mfma")
    m.eval()
    V = int(g["n_classes"])
    x, xl = torch.tensor(g["x"], device=DEV), torch.tensor(g["x_lens"], device=DEV)
    with torch.no_grad():
        f, _, _ = m.encode(x, xl)
    be = BeamExpander(m, V - 1)
    N = 5
    frames = f[0, :N].unsqueeze(1).contiguous()          # 5 hypotheses on 5 frames of utterance 0
    first = be.expand(frames, None, None)                 # SOS step
    assert len(first) == N
    lp, _ = be.log_probs(frames, None, None)
    ref = torch.log_softmax(m.joint(frames, m.predict(None, None, add_sos=False)[0].expand(N, -1, -1))[:, 0, 0].float() / 1.4, -1)
    assert torch.allclose(lp, ref, atol=1e-5)
    for i, e in enumerate(first):
        s, t = ref[i].topk(4)
        keep = s >= s.max() - 1.5
        assert torch.equal(e.tokens, t[keep].cpu()) and torch.allclose(e.scores, s[keep].cpu(), atol=1e-5)
        assert e.blank_logp == pytest.approx(ref[i, V - 1].item(), abs=1e-5)
        assert e.pred_state[0].shape == (cfg["pred_rnn_layers"], 1, cfg["pred_n_hid"])
    # second expansion from the carried states with explicit last tokens
    y_last = torch.stack([e.tokens[:1] for e in first]).to(DEV)
    y_last = torch.where(y_last == V - 1, torch.zeros_like(y_last), y_last)
    st = (torch.cat([e.pred_state[0] for e in first], 1), torch.cat([e.pred_state[1] for e in first], 1))
    second = be.expand(frames, y_last, st)
    assert len(second) == N and all(1 <= len(e.tokens) <= 4 for e in second)


def test_optimizer_state_and_ema_checkpoint_roundtrip(tmp_path):
    from argparse import Namespace

    from caiman_asr_amd.export.checkpointer import Checkpointer, ema_state_dict
    from caiman_asr_amd.train_utils.optimizer import build_optimizer

    g, sd, cfg, m = build("tiny")
    args = Namespace(lr=4e-3, weight_decay=1e-2, beta1=0.9, beta2=0.999, clip_norm=1.0, ema=0.9)
    opt = build_optimizer(args, m)
    for p in m.parameters():
        p.grad.normal_()
    opt.step()
    ema_sd = ema_state_dict(m, opt)
    assert list(ema_sd) == list(m.state_dict())
    w, e = m.joint_enc.weight, ema_sd["joint_enc.weight"]
    w0 = torch.tensor(sd["joint_enc.weight"], device=DEV)
    assert torch.allclose(e, 0.9 * w0 + 0.1 * w, atol=1e-6) and not torch.equal(e, w)
    ck = Checkpointer(str(tmp_path), "RNN-T")
    ck.save(m, ema_sd, opt, 1, 7, 0.5, {}, 1.0)
    g2, sd2, cfg2, m2 = build("tiny")
    opt2 = build_optimizer(args, m2)
    meta = {"best_wer": 1.0, "step": 0}
    ck.load(ck.last_checkpoint(), m2, None, opt2, meta)
    assert meta["step"] == 7 and int(opt2._step.item()) == 1
    assert torch.equal(opt2.flat_m, opt.flat_m) and torch.equal(opt2.flat_v, opt.flat_v)
    assert torch.equal(opt2.flat_ema, opt.flat_ema) and torch.equal(opt2.flat_p, opt.flat_p)
    # identical next step after resume
    for p, q in zip(m.parameters(), m2.parameters()):
        p.grad.fill_(0.01)
        q.grad.fill_(0.01)
    opt.step()
    opt2.step()
    assert torch.equal(opt2.flat_p, opt.flat_p)


@pytest.mark.parametrize("T1, factor", [(75, 2), (64, 2), (33, 2), (70, 3)])
def test_encoder_one_pipeline_equals_two_stacks(T1, factor):
    """pre_rnn -> StackTime -> post_rnn as ONE layer pipeline (encoder_pipe.py): same outputs, final states and
    gradients as the stack-after-stack schedule (bf16 rounding of differently chunked GEMMs aside)."""
    from caiman_asr_amd.rnnt.model import RNNT

    torch.manual_seed(0)
    kw = dict(n_classes=29, in_feats=48, enc_n_hid=64, enc_pre_rnn_layers=2, enc_post_rnn_layers=3,
              enc_stack_time_factor=factor, enc_dropout=0.0, enc_batch_norm=False, pred_n_hid=32, pred_rnn_layers=1,
              pred_dropout=0.0, pred_batch_norm=False, joint_n_hid=48, joint_dropout=0.0, forget_gate_bias=1.0,
              custom_lstm=True)
    m = RNNT(**kw).to(DEV)
    B = 5
    x = torch.randn(T1, B, 48, device=DEV)
    lens = torch.tensor([T1, T1 - 3, T1 // 2, 7, T1], device=DEV)
    outs = []
    for pipe in (False, True):
        m.encoder_pipe = pipe
        m.zero_grad()
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            f, f_lens, st = m.encode(xi, lens)
        w = torch.linspace(0.5, 1.5, f.numel(), device=DEV).view_as(f)
        (f.float() * w).sum().backward()
        grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None and n.startswith("encoder")}
        outs.append((f.float(), f_lens, st, xi.grad.clone(), grads))
    (f0, l0, s0, dx0, g0), (f1, l1, s1, dx1, g1) = outs
    assert torch.equal(l0, l1) and f0.shape == f1.shape
    assert torch.allclose(f0, f1, atol=2e-2 * float(f0.abs().max()), rtol=0)
    for a, b in ((s0.pre_rnn, s1.pre_rnn), (s0.post_rnn, s1.post_rnn)):
        for u, v in zip(a, b):
            assert torch.allclose(u.float(), v.float(), atol=2e-2, rtol=0)
    assert torch.allclose(dx0, dx1, atol=3e-2 * float(dx0.abs().max()), rtol=0)
    assert set(g0) == set(g1) and len(g0) == 20
    for n in g0:
        assert torch.allclose(g0[n], g1[n], atol=3e-2 * float(g0[n].abs().max()) + 1e-6, rtol=0), n


def test_encoder_one_pipeline_with_dropout_trains():
    from caiman_asr_amd.rnnt.model import RNNT

    torch.manual_seed(0)
    m = RNNT(n_classes=29, in_feats=48, enc_n_hid=64, enc_pre_rnn_layers=2, enc_post_rnn_layers=3, enc_stack_time_factor=2,
             enc_dropout=0.3, enc_batch_norm=False, pred_n_hid=32, pred_rnn_layers=1, pred_dropout=0.0, pred_batch_norm=False,
             joint_n_hid=48, joint_dropout=0.0, forget_gate_bias=1.0, custom_lstm=True).to(DEV).train()
    m.encoder_pipe = True
    x = torch.randn(70, 4, 48, device=DEV, requires_grad=True)
    lens = torch.tensor([70, 60, 33, 70], device=DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        f1, _, _ = m.encode(x, lens)
        f2, _, _ = m.encode(x, lens)
    assert not torch.equal(f1, f2)                       # fresh masks per call
    f1.float().square().mean().backward()
    assert torch.isfinite(x.grad).all() and x.grad.abs().sum() > 0
    assert all(torch.isfinite(p.grad).all() for n, p in m.named_parameters() if n.startswith("encoder"))
    m.eval()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        e1, _, _ = m.encode(x, lens)
        e2, _, _ = m.encode(x, lens)
    assert torch.equal(e1, e2)


@pytest.mark.parametrize("pred_layers, pred_hid, U", [(2, 32, 37), (1, 64, 5), (2, 64, 70)])
def test_prediction_network_rides_in_the_encoder_launches(pred_layers, pred_hid, U):
    """enc_pred with the prediction LSTM's steps in the encoder pipeline's launches (slots of their own hidden size)
    == encoder and prediction network run separately: outputs, carried states, gradients."""
    from caiman_asr_amd.rnnt.model import RNNT

    torch.manual_seed(1)
    m = RNNT(n_classes=29, in_feats=48, enc_n_hid=64, enc_pre_rnn_layers=2, enc_post_rnn_layers=3, enc_stack_time_factor=2,
             enc_dropout=0.0, enc_batch_norm=False, pred_n_hid=pred_hid, pred_rnn_layers=pred_layers, pred_dropout=0.0,
             pred_batch_norm=False, joint_n_hid=48, joint_dropout=0.0, forget_gate_bias=1.0, custom_lstm=True).to(DEV)
    T1, B = 75, 5
    x = torch.randn(T1, B, 48, device=DEV)
    lens = torch.tensor([T1, T1 - 3, T1 // 2, 7, T1], device=DEV)
    y = torch.randint(0, 28, (B, U), device=DEV)
    y_lens = torch.tensor([U, max(U - 2, 1), 1, max(U // 2, 1), U], device=DEV)
    outs = []
    for pipe in (False, True):
        m.encoder_pipe = m.pred_in_encoder_pipe = pipe
        m.zero_grad()
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            (f, f_lens), (g, g_lens), st = m.enc_pred(xi, lens, y, y_lens)
        (f.float().square().sum() + (g.float() * torch.linspace(0.5, 1.5, g.numel(), device=DEV).view_as(g)).sum()).backward()
        grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        outs.append((f.float(), g.float(), f_lens, g_lens, st, xi.grad.clone(), grads))
    (f0, g0, fl0, gl0, s0, dx0, gr0), (f1, g1, fl1, gl1, s1, dx1, gr1) = outs
    assert torch.equal(fl0, fl1) and torch.equal(gl0, gl1)
    assert torch.allclose(f0, f1, atol=2e-2 * float(f0.abs().max()), rtol=0)
    assert torch.allclose(g0, g1, atol=2e-2 * float(g0.abs().max()), rtol=0)
    for a, b in zip(s0.pred_net_state.next_to_last_pred_state, s1.pred_net_state.next_to_last_pred_state):
        assert torch.allclose(a.float(), b.float(), atol=2e-2, rtol=0)
    assert torch.equal(s0.pred_net_state.last_token, s1.pred_net_state.last_token)
    assert torch.allclose(dx0, dx1, atol=3e-2 * float(dx0.abs().max()), rtol=0)
    assert set(gr0) == set(gr1)
    for n in gr0:
        assert torch.allclose(gr0[n], gr1[n], atol=3e-2 * float(gr0[n].abs().max()) + 1e-6, rtol=0), n


@pytest.mark.parametrize("T1, factor, dt, drop", [(75, 2, torch.bfloat16, 0.0), (70, 3, torch.bfloat16, 0.2), (33, 2, torch.float16, 0.0),
                                                  (64, 2, torch.float16, 0.2)])
def test_grouped_projection_and_fused_weight_images_match_the_library_path(T1, factor, dt, drop):
    """H = 128 is the smallest width the grouped projection kernel takes (N = 512, K = 128 / factor * 128): the encoder
    pipeline with the hand-written chunk GEMMs + one-launch weight images against the same pipeline on the library calls
    (CAIMAN_PROJ_GEMM=0 / CAIMAN_LSTM_IMAGES=0 behaviour), ragged lengths, StackTime factor 2 and 3, both 16-bit types,
    with and without inter-layer dropout (same seed -> same masks)."""
    from caiman_asr_amd.rnnt.model import RNNT
    from caiman_asr_amd.rnnt_ext.custom_lstm import encoder_pipe as ep
    from caiman_asr_amd.rnnt_ext.custom_lstm import stack as stk

    torch.manual_seed(0)
    m = RNNT(n_classes=29, in_feats=48, enc_n_hid=128, enc_pre_rnn_layers=2, enc_post_rnn_layers=3, enc_stack_time_factor=factor,
             enc_dropout=drop, enc_batch_norm=False, pred_n_hid=32, pred_rnn_layers=1, pred_dropout=0.0, pred_batch_norm=False,
             joint_n_hid=48, joint_dropout=0.0, forget_gate_bias=1.0, custom_lstm=True).to(DEV).train()
    m.encoder_pipe = True
    B = 5
    x = torch.randn(T1, B, 48, device=DEV)
    lens = torch.tensor([T1, T1 - 3, T1 // 2, 7, T1], device=DEV)
    saved = (ep.PROJ, ep.IMAGES, stk.IMAGES)
    outs = []
    try:
        for proj, images in ((False, False), (True, True), (False, True)):
            ep.PROJ, ep.IMAGES, stk.IMAGES = proj, images, images
            m.zero_grad()
            torch.manual_seed(123)
            xi = x.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=dt):
                f, f_lens, _ = m.encode(xi, lens)
            w = torch.linspace(0.5, 1.5, f.numel(), device=DEV).view_as(f)
            (f.float() * w).sum().backward()
            grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None and n.startswith("encoder")}
            outs.append((f.detach().float(), xi.grad.clone(), grads))
    finally:
        ep.PROJ, ep.IMAGES, stk.IMAGES = saved
    (f0, dx0, g0), (f1, dx1, g1), (f2, dx2, g2) = outs
    # the one-launch weight images are bit-identical to the separately built ones: so is everything downstream
    assert torch.equal(f0, f2) and torch.equal(dx0, dx2)
    for n in g0:
        assert torch.equal(g0[n], g2[n]), n
    # the hand-written GEMM sums in another order than the library's: equal to the storage type's resolution
    tol = 2e-2 if dt == torch.bfloat16 else 4e-3
    assert torch.allclose(f0, f1, atol=tol * float(f0.abs().max()), rtol=0)
    assert torch.allclose(dx0, dx1, atol=1.5 * tol * float(dx0.abs().max()), rtol=0)
    assert len(g0) == 20
    for n in g0:
        assert torch.allclose(g0[n], g1[n], atol=1.5 * tol * float(g0[n].abs().max()) + 1e-6, rtol=0), n


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("back", [0, 1])
def test_last_state_gather_kernel_equals_advanced_indexing(dtype, back):
    """caiman_lstm_last_states (one launch per stack) against the reference's selection (rsp.py:108-130: two advanced-indexing
    gathers): same rows, including views that skip the initial-state row, int32 / int64 lengths and the wrap of index -1."""
    from caiman_asr_amd.rnnt.state import get_last_nonpadded_states

    L, T, B, H = 3, 17, 9, 64
    g = torch.Generator().manual_seed(L * T + back)
    full_h = torch.randn(L, T + 1, B, H, generator=g).to(dtype).to(DEV)
    full_c = torch.randn(L, T + 1, B, H, generator=g).to(dtype).to(DEV)
    h, c = full_h[:, 1:], full_c[:, 1:]
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0], lens[1] = T, 1                      # lens 1 with back = 1 selects index -1 = the last step
    for lens_d in (lens.to(DEV), lens.int().to(DEV), lens):
        got = get_last_nonpadded_states((h, c), lens_d, back)
        idx = lens.long().to(DEV) - 1 - back
        cols = torch.arange(B, device=DEV)
        assert torch.equal(got[0], h[:, idx, cols, :]) and torch.equal(got[1], c[:, idx, cols, :])
        assert got[0].is_contiguous() and got[0].shape == (L, B, H)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embedding_gradient_kernel_equals_autograd(dtype):
    """caiman_embedding_grad (one launch, fixed summation order, accumulating into `.grad`) against torch's own embedding
    backward: repeated tokens, tokens nobody emitted, more tokens than one staging block, an existing gradient to add to."""
    from caiman_asr_amd.train_utils import overlap

    V, E, B, U = 300, 192, 13, 170           # 2210 tokens: three staging blocks of 1024
    g = torch.Generator().manual_seed(V + E)
    emb = torch.nn.Embedding(V, E).to(DEV)
    idx = torch.randint(0, V // 2, (B, U), generator=g).to(DEV)      # the upper half of the table is never looked up
    w = torch.randn(B, U, E, generator=g).to(DEV).to(dtype)
    for trial in range(2):                    # the second pass accumulates on top of the first
        out = overlap.embedding(emb, idx)
        assert out.grad_fn is not None and type(out.grad_fn).__name__.startswith("_EmbeddingDirectGrad")
        (out.to(dtype) * w).float().sum().backward()
    got = emb.weight.grad.clone()
    ref_mod = torch.nn.Embedding(V, E).to(DEV)
    ref_mod.load_state_dict(emb.state_dict())
    for trial in range(2):
        (ref_mod(idx).to(dtype) * w).float().sum().backward()
    ref = ref_mod.weight.grad
    assert torch.equal(got[V // 2:], torch.zeros_like(got[V // 2:]))
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5 * ref.abs().max().item())
    # bit-identical from run to run
    emb.weight.grad = None
    for trial in range(2):
        (overlap.embedding(emb, idx).to(dtype) * w).float().sum().backward()
    assert torch.equal(emb.weight.grad, got)


def test_embedding_gradient_kernel_wide_rows_and_foreign_tokens():
    """Rows wider than one pass of the kernel's column registers (E > 2048: two passes over the token list), a token id
    outside the table (ignored, as nothing may be written for it) and a single position."""
    import ctypes  # noqa: F401

    from caiman_asr_amd import _lib

    V, E = 40, 2100
    g = torch.Generator().manual_seed(9)
    tokens = torch.randint(0, V, (300,), generator=g)
    tokens[17], tokens[250] = V + 5, -3
    dy = torch.randn(300, E, generator=g)
    grad = torch.randn(V, E, generator=g)
    ref = grad.clone().double()
    for n in range(300):
        if 0 <= int(tokens[n]) < V:
            ref[int(tokens[n])] += dy[n].double()
    td, dd, gd = tokens.to(DEV), dy.to(DEV), grad.to(DEV)
    lib = _lib.lib()
    _lib.check(lib.caiman_embedding_grad(_lib.ptr(td), 300, _lib.ptr(dd), _lib.dtype_tag(torch.float32), V, E, _lib.ptr(gd),
                                         _lib.stream()))
    torch.cuda.synchronize()
    assert torch.allclose(gd.cpu().double(), ref, rtol=0, atol=1e-4)
    one = torch.zeros(V, E, device=DEV)
    _lib.check(lib.caiman_embedding_grad(_lib.ptr(td[3:4]), 1, _lib.ptr(dd[3:4]), _lib.dtype_tag(torch.float32), V, E, _lib.ptr(one),
                                         _lib.stream()))
    torch.cuda.synchronize()
    assert torch.equal(one[int(tokens[3])], dd[3]) and float(one.abs().sum()) == float(dd[3].abs().sum())


def test_slab_accumulate_adds_the_slabs_in_order():
    """caiman_slab_accumulate: dst += slab 0 + slab 1 + ... (fp32, fixed order): against the same sum in torch, bit for bit."""
    from caiman_asr_amd import _lib

    g = torch.Generator().manual_seed(4)
    slabs = torch.randn(5, 3, 1028, generator=g).to(DEV)          # n = 3084 floats per slab: not a multiple of the block size
    dst = torch.randn(3, 1028, generator=g).to(DEV)
    ref = slabs[0].clone()
    for s_ in range(1, 5):
        ref = ref + slabs[s_]
    ref = dst + ref
    _lib.check(_lib.lib().caiman_slab_accumulate(_lib.ptr(slabs), 5, slabs[0].numel(), _lib.ptr(dst), _lib.stream()))
    torch.cuda.synchronize()
    assert torch.equal(dst, ref)
    with pytest.raises(RuntimeError):
        _lib.check(_lib.lib().caiman_slab_accumulate(_lib.ptr(slabs), 5, 3083, _lib.ptr(dst), _lib.stream()))
