"""CPU tests of the host-side logic against golden data produced by the reference's own Python
(oracle/gen_golden.py): constructor / state_dict schema, LR policy, shape ops, packing metadata."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cfg(name):
    return json.load(open(os.path.join(GOLD, f"rnnt_cfg_{name}.json")))


@pytest.mark.parametrize("name,n_classes,n_params", [("base", 8704, 84690432), ("large", 17408, 195590400)])
def test_state_dict_schema_matches_reference(name, n_classes, n_params):
    # training/caiman_asr_train/export/model_schema/{base,large}.json; 84.69 M / 195.59 M parameters
    from caiman_asr_amd.rnnt import config
    from caiman_asr_amd.rnnt.model import RNNT

    kw = config.validate_and_fill(RNNT, _cfg(name), optional=["n_classes"], deprecated=["hard_activation_functions"])
    kw.pop("n_classes", None)
    with torch.device("meta"):
        m = RNNT(n_classes=n_classes, **kw)
    schema = json.load(open(os.path.join(GOLD, f"schema_{name}.json")))
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert mine == schema
    assert list(mine) == list(schema)  # same order
    assert sum(p.numel() for p in m.parameters()) == n_params
    groups = m.param_groups(4e-3, return_module_name=True)
    assert [g["module_name"] for g in groups] == ["encoder", "prediction", "joint_enc", "joint_pred", "joint_net"]
    assert groups[-1]["lr"] == pytest.approx(4e-3 * _cfg(name)["joint_net_lr_factor"])


def test_unknown_config_key_is_rejected():
    from caiman_asr_amd.rnnt import config
    from caiman_asr_amd.rnnt.model import RNNT

    bad = dict(_cfg("base"), not_a_key=1)
    with pytest.raises(AssertionError, match="Unknown parameter"):
        config.validate_and_fill(RNNT, bad)
    ok = config.validate_and_fill(RNNT, dict(_cfg("base"), hard_activation_functions=False),
                                  deprecated=["hard_activation_functions"], optional=["n_classes"])
    assert "hard_activation_functions" not in ok


def test_yaml_loader_expands_anchors(tmp_path):
    from caiman_asr_amd.rnnt import config

    p = tmp_path / "c.yaml"
    p.write_text("a: &x {k: 1}\nb: *x\nc:\n  !!merge <<: *x\n  j: 2\nrnnt: {}\n")
    cfg = config.load(str(p))
    assert cfg["b"] == {"k": 1} and cfg["c"] == {"k": 1, "j": 2}
    cfg["a"]["k"] = 5
    assert cfg["b"]["k"] == 1  # deep copy, not an alias
    empty = tmp_path / "e.yaml"
    empty.write_text("")
    with pytest.raises(ValueError):
        config.load(str(empty))


def test_lr_policy_matches_reference():
    from caiman_asr_amd.train_utils.lr import lr_policy

    g = json.load(open(os.path.join(GOLD, "lr_policy.json")))

    class Opt:
        param_groups = [dict(lr=0.0), dict(lr=0.0)]

    for row in g["rows"]:
        lr_policy(Opt, g["initial_lr"], g["min_lr"], row[0], g["warmup"], g["hold"], g["half_life"])
        assert [pg["lr"] for pg in Opt.param_groups] == row[1:]  # exact: same float expression


def test_stack_time_and_frame_splicing_match_reference():
    from caiman_asr_amd.data.features import stack_subsample_frames
    from caiman_asr_amd.rnnt.model import StackTime

    g = np.load(os.path.join(GOLD, "shape_ops.npz"))
    x, lens = torch.tensor(g["x"]), torch.tensor(g["lens"])
    for fac in (2, 3):
        o, l = StackTime(fac)(x, lens)
        assert np.array_equal(o.numpy(), g[f"stack{fac}"]) and np.array_equal(l.numpy(), g[f"stack{fac}_lens"])
    a, alens = torch.tensor(g["a"]), torch.tensor(g["alens"])
    for s, ss in ((3, 3), (1, 1), (2, 1), (3, 2)):
        o, l = stack_subsample_frames(a, alens, s, ss)
        assert np.array_equal(o.numpy(), g[f"splice_{s}_{ss}"]), (s, ss)
        assert np.array_equal(l.numpy(), g[f"splice_{s}_{ss}_lens"])


def test_packing_metadata_matches_reference():
    from caiman_asr_amd.rnnt.loss import get_packing_meta_data

    g = np.load(os.path.join(GOLD, "rnnt_tiny.npz"))
    meta = get_packing_meta_data(torch.tensor(g["x_lens"]), torch.tensor(g["y_lens"]), 2)
    assert np.array_equal(meta["batch_offset"].numpy(), g["batch_offset"])
    assert meta["max_f_len"] == int(g["max_f_len"]) and meta["packed_batch"] == int(g["batch_offset"][-1])


def test_spec_augment_mask_statistics():
    from caiman_asr_amd.data.features import SpecAugment

    sa = SpecAugment(freq_masks=2, min_freq=0, max_freq=20, time_masks=10, min_time=0, max_time=0.03)
    lens = torch.tensor([400, 200, 1000])
    gen = torch.Generator().manual_seed(0)
    m = sa.make_mask((3, 80, 1000), lens, torch.device("cpu"), generator=gen)
    # at most 2 bands of <=20 bins; at most 10 spans of <= round(0.03*len) frames
    for b in range(3):
        assert m[b].all(1).sum() <= 40
        assert m[b].all(0).sum() <= 10 * round(0.03 * int(lens[b]))
    x = torch.randn(3, 80, 1000)
    y, _ = sa.calculate_features(x, lens)
    assert ((y == 0) | (y == x)).all()
    assert SpecAugment().calculate_features(x, lens)[0].equal(x)


def test_model_refuses_cpu_tensors_and_cpu_fallbacks():
    from caiman_asr_amd.rnnt.model import RNNT

    cfg = json.loads(str(np.load(os.path.join(GOLD, "rnnt_tiny.npz"))["cfg"]))
    cfg["custom_lstm"] = True
    m = RNNT(n_classes=30, **cfg)
    with pytest.raises(RuntimeError, match="CUDA"):
        m.encode(torch.randn(5, 2, cfg["in_feats"]), torch.tensor([5, 4]))
    with pytest.raises(ValueError, match="no CPU fallback"):
        RNNT(n_classes=30, **dict(cfg, gpu_unavailable=True))
    with pytest.raises(ValueError, match="quantize"):
        RNNT(n_classes=30, **dict(cfg, quantize=True))


def test_norm_blend_ratio_schedule():
    # training/caiman_asr_train/data/dali/mel_normalization.py:85-101
    from caiman_asr_amd.data.frontend import MelFeatNormalizer, NormType, norm_ramp_params

    m, s = torch.zeros(80), torch.ones(80)
    nz = MelFeatNormalizer(m, s, 100, 200, 0.25, NormType.BLENDED_STATS)
    assert [nz._calc_ratio(k) for k in (0, 100, 150, 200, 999)] == [0.25, 0.25, 0.625, 1.0, 1.0]
    assert MelFeatNormalizer(m, s, None, None, 0.25, NormType.DATASET_STATS)._calc_ratio(5) == 1.0
    assert MelFeatNormalizer(None, None, None, None, 0.25, NormType.UTTERANCE_STATS)._calc_ratio(5) == 0.0
    assert norm_ramp_params(NormType.BLENDED_STATS, 1632, 18000, 10880) == (30512, 35512)
    assert norm_ramp_params(NormType.DATASET_STATS, 1, 2, 3) == (None, None)


def test_mel_filterbank_tables():
    from caiman_asr_amd.data.frontend import hann_window, mel_filterbank
    from oracle import frontend as of

    w = mel_filterbank(16000, 512, 80)
    assert w.shape == (80, 257) and (w >= 0).all()
    # two independent constructions (vectorised product table vs per-bin oracle loops) agree
    assert np.allclose(w, of.mel_weights(16000, 512, 80), atol=1e-12)
    # Slaney area normalisation: each triangle integrates to ~1 over Hz
    area = w.sum(1) * (16000 / 512)
    assert np.allclose(area[5:], 1.0, atol=0.12)
    assert np.allclose(hann_window(400), of._hann_dali(400))


def test_encoder_pipeline_schedule_respects_dependencies():
    """encoder_pipe._schedule (host logic of the whole-encoder layer pipeline): every (layer, chunk) once, each after
    the chunks it depends on, never more slots per launch than the kernels accept."""
    import importlib

    ep = importlib.import_module("caiman_asr_amd.rnnt_ext.custom_lstm.encoder_pipe")
    assert ep._chunk(1024) in (24, ep.CH) and ep._chunk(1536) in (32, ep.CH)
    for fine in (True, False):
      for CH in (24, 32):
        ep.FINE = fine
        for T1, La, Lb, f, Tp, Lp in [(430, 2, 6, 2, 0, 0), (75, 2, 3, 2, 0, 0), (33, 1, 1, 2, 0, 0), (70, 2, 3, 3, 0, 0),
                                      (557, 2, 6, 2, 0, 0), (430, 2, 4, 2, 58, 2), (1, 2, 6, 2, 1, 2), (64, 3, 5, 1, 0, 0)]:
            T2 = -(-T1 // f)
            CHb = ep._post_chunk(f, CH)
            nA, nB, nP = -(-T1 // CH), -(-T2 // CHb), -(-Tp // CH) if Lp else 0
            ticks = ep._schedule(nA, nB, La, Lb, f, nP, Lp, CH)
            when = {}
            for t, tick in enumerate(ticks):
                assert 1 <= len(tick) <= 8
                for lk in tick:
                    assert lk not in when
                    when[lk] = t
            assert len(when) == La * nA + Lb * nB + Lp * nP
            for (l, k), t in when.items():
                if k > 0:
                    assert when[(l, k - 1)] < t                              # time order within a layer
                if 0 < l < La or La < l < La + Lb or l > La + Lb:
                    assert when[(l - 1, k)] < t                              # the layer below, same chunk
                if l == La:                                                  # first post layer: the pre chunks it stacks
                    lo, hi = f * k * CHb, min(f * (k * CHb + min(CHb, T2 - k * CHb)), T1)
                    for c in range(lo // CH, -(-hi // CH)):
                        assert when[(La - 1, c)] < t
    ep.FINE = True


# ---- train-step control pieces (round 2): schedules, RSP controller, optimiser wrapper ------------------------------
def test_step_schedule_latches_on_step_or_wer():
    # training/caiman_asr_train/train_utils/schedule.py:57-114; wiring setup/train.py:212-229
    from argparse import Namespace

    from caiman_asr_amd.train_utils.schedule import (ConstantSchedule, StepSchedule, build_delay_penalty_scheduler,
                                                     build_star_scheduler)

    s = StepSchedule(0.75, 1.0, wer_threshold=0.2)
    assert s.value() == 0.75 and s.step(10, hints={"wer": 0.5}) == 0.75
    assert s.step(11, hints={"wer": 0.19}) == 1.0
    assert s.step(12, hints={"wer": 0.9}) == 1.0          # latched
    with pytest.raises(ValueError):
        StepSchedule(0.0, 1.0)
    with pytest.raises(ValueError):
        StepSchedule(0.0, 1.0, wer_threshold=0.3).step(1)   # WER expected in hints
    t = StepSchedule(0.0, 0.01, toggle_step=100, wer_threshold=0.3)
    assert t.step(99) == 0.0 and t.step(100) == 0.01
    assert ConstantSchedule(0.3).step(5) == 0.3
    dp = build_delay_penalty_scheduler(Namespace(delay_penalty="wer_schedule", dp_initial_value=0.0, dp_final_value=0.01,
                                                 dp_toggle_step=None, dp_wer_threshold=0.3))
    assert dp.step(1, hints={"wer": 1.0}) == 0.0 and dp.step(2, hints={"wer": 0.29}) == 0.01
    assert build_delay_penalty_scheduler(Namespace(delay_penalty="0.005")).value() == 0.005
    star = build_star_scheduler(Namespace())
    assert star.value() == 0.75 and star.step(3, hints={"wer": 0.1}) == 1.0


def test_rsp_controller():
    # training/caiman_asr_train/train_utils/rsp.py:17-104; statistics as training/tests/train_utils/test_rsp.py
    import random
    from argparse import Namespace

    from caiman_asr_amd.train_utils import rsp

    assert not rsp.is_random_state_passing_on([1]) and not rsp.is_random_state_passing_on([3, 0, 0])
    assert rsp.is_random_state_passing_on([10, 0, 1])
    random.seed(0)
    trials = 10 ** 4
    draws = [rsp.generate_batch_history([10, 0, 2]) for _ in range(trials)]
    assert set(draws) == {1, 3} and draws.count(1) / trials == pytest.approx(10 / 12, abs=0.015)
    args = Namespace(rsp_delay=None, warmup_steps=1632, hold_steps=18000, half_life_steps=10880, training_steps=10 ** 5,
                     rsp_seq_len_freq=[99, 0, 1])
    cfg = {"rnnt": {"custom_lstm": True, "enc_batch_norm": False, "pred_batch_norm": False}}
    rsp.rsp_config_checks(args, cfg)
    assert args.rsp_delay == 1632 + 18000 + 3 * 10880
    with pytest.raises(AssertionError):
        rsp.rsp_config_checks(Namespace(rsp_seq_len_freq=[1, 1], rsp_delay=0), {"rnnt": dict(cfg["rnnt"], custom_lstm=False)})
    with pytest.raises(AssertionError):
        rsp.rsp_config_checks(Namespace(rsp_seq_len_freq=[0, 0], rsp_delay=0), cfg)
    state = object.__new__(rsp.RNNTState)
    a = Namespace(rsp_seq_len_freq=[1, 1], rsp_delay=100)
    assert rsp.rsp_end_step(state, False, 99, a, 5) == (None, 4, False)          # before the delay
    assert rsp.rsp_end_step(state, False, 100, a, 5) == (state, 4, True)
    assert rsp.rsp_end_step(state, True, 100, a, 5)[0] is None                    # NaN: the state may be the cause
    carry, counter, on = rsp.rsp_end_step(state, False, 100, a, 1)               # history used up: reset + redraw
    assert carry is None and counter in (1, 2) and on
    off = Namespace(rsp_seq_len_freq=[1], rsp_delay=0)
    assert rsp.rsp_end_step(None, False, 5, off, 1) == (None, 1, False)


def test_optimizer_wrapper_lower_bound_logic():
    # training/caiman_asr_train/train_utils/optimizer.py:31-48 with a scripted scaler
    from argparse import Namespace

    from caiman_asr_amd.train_utils.optimizer import OptimizerWrapper

    class FakeScaler:
        def __init__(self):
            self.scale, self.updates, self.steps = 1024.0, [], 0

        def step(self, opt):
            self.steps += 1

        def update(self, new=None):
            self.updates.append(new)
            self.scale = new if new is not None else self.scale / 2

        def get_scale(self):
            return self.scale

    class FakeOpt:
        param_groups = [{"lr": 0.5}]
        n = 0

        def step(self):
            FakeOpt.n += 1

        def zero_grad(self):
            pass

    sc = FakeScaler()
    w = OptimizerWrapper(Namespace(no_amp=False), FakeOpt(), sc, lower_bound=300.0)
    w.step()          # 512
    w.step()          # 256 < 300 -> override queued
    assert w.scale == 300.0
    w.step()          # update(300)
    assert sc.updates == [None, None, 300.0] and sc.scale == 300.0 and w.scale is None and sc.steps == 3
    assert w.learning_rate == 0.5
    plain = OptimizerWrapper(Namespace(no_amp=True), FakeOpt(), None)
    plain.step()
    assert FakeOpt.n == 1


def test_fused_decode_weight_images_reproduce_the_concatenated_gemm():
    """rnnt/streaming_lstm.py::fused_layer_weights (operand images of caiman_lstm_step_gemm): rows ordered [unit][gate],
    input columns zero-padded to a multiple of 128.  [x | 0 | h] against the image must give the gate pre-activations of
    [x | h] against [W_ih | W_hh] (training/lib/csrc/lstm.cu:259-271), permuted to [unit][gate]."""
    from caiman_asr_amd.rnnt.streaming_lstm import fused_layer_weights

    torch.manual_seed(0)
    I, H, n = 240, 128, 5
    lstm = torch.nn.LSTM(I, H, 2)
    for l, (i_in, ip) in enumerate(((I, 256), (H, 128))):
        W, b, Ip = fused_layer_weights(lstm, l, torch.float32)
        assert Ip == ip and W.shape == (4 * H, Ip + H) and b.shape == (4 * H,)
        x, h = torch.randn(n, i_in), torch.randn(n, H)
        X = torch.zeros(n, Ip + H)
        X[:, :i_in], X[:, Ip:] = x, h
        got = X @ W.t() + b                                               # [n, (unit, gate)]
        ref = (x @ getattr(lstm, f"weight_ih_l{l}").t() + h @ getattr(lstm, f"weight_hh_l{l}").t()
               + getattr(lstm, f"bias_ih_l{l}") + getattr(lstm, f"bias_hh_l{l}"))   # [n, (gate, unit)]
        assert torch.allclose(got.view(n, H, 4), ref.view(n, 4, H).transpose(1, 2), atol=1e-5)


def test_storage_oracle_rounds_values_forward_and_gradients_backward():
    """oracle/model.py storage mode: rf rounds the value and passes the gradient, rb passes the value and rounds the gradient;
    the cell's backward with storage=None is plain autograd of the same formulas; storage=bf16 changes the result by no more
    than bf16 resolution allows."""
    from oracle import model as om

    om._STORAGE = torch.bfloat16
    try:
        x = torch.tensor([1.0 + 2.0 ** -10, -3.0 - 2.0 ** -9], dtype=torch.float64, requires_grad=True)
        y = om.rf(x)
        assert torch.equal(y.detach(), x.detach().to(torch.bfloat16).double()) and not torch.equal(y.detach(), x.detach())
        g = torch.tensor([1.0 + 2.0 ** -10, 2.0 + 2.0 ** -8], dtype=torch.float64)
        y.backward(g)
        assert torch.equal(x.grad, g)
        x.grad = None
        z = om.rb(x)
        assert torch.equal(z.detach(), x.detach())
        z.backward(g)
        assert torch.equal(x.grad, g.to(torch.bfloat16).double()) and not torch.equal(x.grad, g)
    finally:
        om._STORAGE = None
    torch.manual_seed(1)
    z = torch.randn(3, 16, dtype=torch.float64, requires_grad=True)
    c = torch.randn(3, 4, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda a, b: om._Cell.apply(a, b, None), (z, c))
    h0, c0 = om._Cell.apply(z, c, None)
    h1, c1 = om._Cell.apply(z, c, torch.bfloat16)
    assert torch.equal(h0, h1) and torch.equal(c0, c1)             # forward values are not touched by the storage type
    g0 = torch.autograd.grad((h0.sum() + c0.sum()), (z, c))
    g1 = torch.autograd.grad((h1.sum() + c1.sum()), (z, c))
    for a, b in zip(g0, g1):
        assert not torch.equal(a, b) and torch.allclose(a, b, atol=3e-2)


def test_profile_kernel_names_map_to_their_rows():
    import importlib.util
    import os as _os

    spec = importlib.util.spec_from_file_location("profile_summary", _os.path.join(_os.path.dirname(GOLD), "..", "tools", "profile_summary.py"))
    ps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ps)
    assert ps.short("_ZN6caiman12_GLOBAL__N_124lstm_fwd_resident_bt_dmaIDF16bLb0ELi32EEEvNS0_8FwdSlotsIT_EEiPjS5_S5_") == "lstm_fwd_resident_bt_dma"
    assert ps.short("_ZN6caiman12_GLOBAL__N_120lstm_fwd_resident_btIDF16bLb0ELi32EEEv") == "lstm_fwd_resident_bt"
    assert ps.short("_ZN6caiman12_GLOBAL__N_121lstm_fwd_resident_dmaIDF16bLb0ELi48ELb0ELi2EEEv") == "lstm_fwd_resident_dma"
    assert ps.short("_ZN6caiman12_GLOBAL__N_116proj_gemm_kernelIDF16bLi128ELi128ELi2ELi8ELi1ELb1EEEvNS0_9ProjBatchE") == "proj_gemm_kernel[cell]"
    assert ps.short("_ZN6caiman12_GLOBAL__N_116proj_gemm_kernelIDF16bLi128ELi128ELi2ELi8ELi1ELb0EEEvNS0_9ProjBatchE") == "proj_gemm_kernel"
    assert ps.short("Cijk_Ailk_Bjlk_BBS_BH_Bias_HA_S_SAV_UserArgs_MT256x256x32") == "library_gemm"
    assert ps.short("void at::native::vectorized_elementwise_kernel<4>") is None


def test_weight_gradient_plan_and_cost_model_are_host_logic():
    """caiman_wgrad_tn_plan / _estimate_us (csrc/joint_wgrad.hip) run on the host: slice counts follow the 256-CU round
    quantisation, the estimate carries the per-round fixed cost, unsupported shapes return 0 / a negative estimate."""
    import ctypes

    import torch

    from caiman_asr_amd import _lib

    lib = _lib.lib()
    bf16 = _lib.dtype_tag(torch.bfloat16)
    per = ctypes.c_int64(0)

    def plan(M, N, K, P=1):
        s = lib.caiman_wgrad_tn_plan(M, N, K, P, bf16, ctypes.byref(per))
        return s, per.value

    assert plan(304000, 8704, 768) == (5, 60800)            # joint projection: 102 tiles x 5 = 510 workgroups, two rounds
    assert lib.caiman_joint_fc_wgrad_plan(304000, 8704, 768, bf16, ctypes.byref(per)) == 5 and per.value == 60800
    s, rows = plan(304000, 17408, 1024)                      # large-196M: 272 tiles, one slice would be 1.06 rounds
    assert s >= 8 and s * rows <= 304000 and rows % 128 == 0 and 304000 - s * rows < 128 * s   # slices: pairs of 64-row tiles
    assert plan(8896, 4096, 1024, 6) == (2, 4352)            # six LSTM layers: 384 tiles x 2 = 3 rounds, the last 192 rows ride in the last slice
    for M, N, K, P in [(200, 512, 512, 1), (4096, 500, 512, 1), (4096, 512, 240, 1), (4096, 512, 512, 0)]:
        assert plan(M, N, K, P)[0] == 0
        assert lib.caiman_wgrad_tn_estimate_us(M, N, K, P, bf16) < 0
    assert lib.caiman_wgrad_tn_plan(4096, 512, 512, 1, _lib.dtype_tag(torch.float32), ctypes.byref(per)) == 0
    est = lib.caiman_wgrad_tn_estimate_us
    # measured 3.05 ms / 597 us / 153 us (tools/joint_gemm_bench.py, tools/wgrad_tn_bench.py, 8-phase kernel): the model within 20 %
    assert 2450 < est(304000, 8704, 768, 1, bf16) < 3650
    assert 480 < est(8896, 4096, 1024, 6, bf16) < 715
    assert 125 < est(17792, 4096, 1024, 1, bf16) < 185
    # five layers fill 2.5 rounds: the model prices them above the library's 0.8 PF/s, and the caller keeps the library
    flops = 2.0 * 5 * 8896 * 4096 * 1024
    assert est(8896, 4096, 1024, 5, bf16) * 1e-6 > flops / 0.8e15


def test_round_buckets_and_mask_geometry_arithmetic_are_host_logic():
    """Host-side pieces of the round-4 launch sweep: the row-count buckets of the captured beam round (every count maps to a
    bucket that holds it, at most 25 % of padding above 1 024 rows, and the bucket list the capture loop walks contains
    it) and SpecAugment's arithmetic on given draws (widths and starts inside the reference's ranges, adaptive counts)."""
    import torch

    from caiman_asr_amd.data.features import SpecAugment
    from caiman_asr_amd.rnnt.beam_native import _rows_bucket

    walked, b = [], 64
    while b <= 8192:
        walked.append(b)
        b = b * 2 if b < 512 else b + 256
    for n in list(range(1, 700)) + [1023, 1024, 1025, 2000, 4095, 4097, 8000]:
        nb = _rows_bucket(n)
        assert nb >= n and nb in walked, (n, nb)
        assert nb == 64 or nb < 2 * n
        if n > 1024:
            assert nb - n < 256 and nb <= 1.25 * n
    sa = SpecAugment(freq_masks=2, min_freq=0, max_freq=20, time_masks=0.04, min_time=0, max_time=0.03)
    B, F, T = 5, 80, 1000
    lens = torch.tensor([1000.0, 500.0, 250.0, 100.0, 12.0])
    r = torch.rand(B, 2 * sa.freq_masks + 2 * sa._time_slots(T), generator=torch.Generator().manual_seed(3))
    f0, fw, t0, tw = sa.geometry_from_draws(r, lens, F, T)
    assert f0.shape == fw.shape == (B, 2) and t0.shape == tw.shape == (B, 41)
    assert bool(((fw >= 0) & (fw <= 20) & (f0 >= 0) & (f0 + fw <= F)).all())
    assert bool(((tw >= 0) & (t0 >= 0) & (t0 + tw <= T)).all())
    for b_ in range(B):
        n_masks, max_w = round(float(lens[b_]) * 0.04), round(float(lens[b_]) * 0.03)
        assert bool((tw[b_, n_masks:] == 0).all()) and bool((tw[b_] <= max_w).all())
    assert bool((r.min() >= 0) and (r.max() < 1))
