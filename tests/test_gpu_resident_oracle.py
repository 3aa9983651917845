"""GPU parity of the weight-resident LSTM chunk kernels (csrc/lstm.hip: lstm_fwd_resident / lstm_bwd_resident) DIRECTLY
against the f64 CPU oracle (oracle/rnnt_oracle.c, a restatement of training/lib/csrc/lstm.cu:85-346), through the
C-ABI wave calls the layer pipelines use.  The resident kernels are the default path of the benched configuration;
tests/test_gpu_lstm.py compares them with the per-timestep kernels, this file anchors them to the oracle itself.

Tolerance: inputs and every stored value are bf16 (8 significant bits, half-ulp 2^-9 relative).  One timestep adds a
K-term fp32 dot product of bf16 operands (error ~ 2^-9 * sqrt(K) * |h| * |r| ~ 2^-9 since |r| ~ 1/sqrt(K)) and one
rounding of each stored value; the recurrence is contractive (|f| < 1), so the drift over T steps stays a small
multiple of the bf16 resolution.  Bounds below: max error <= 16 half-ulps of the value range, mean error <= 1 half-ulp.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF16_HALF_ULP = 2.0 ** -9


def _il(a, H):
    """[..., 4H] gate-major (reference layout, lstm.cu:99-102) -> [..., H, 4] interleaved, flattened."""
    return a.reshape(*a.shape[:-1], 4, H).transpose(-1, -2).reshape(a.shape)


def _ref(a, H):
    """inverse of _il."""
    return a.reshape(*a.shape[:-1], H, 4).transpose(-1, -2).reshape(a.shape)


def _resident_fwd_bwd(R, gates_ref, c0, y0, delta, hard, expect_resident=True):
    """One LSTM layer, all T timesteps in ONE wave call each way (-> one resident launch).  Inputs: torch bf16 CPU
    tensors in the reference layouts.  Returns (activated gates, c [T+1], y [T+1], dG) in the reference layouts."""
    from caiman_asr_amd import _lib

    lib = _lib.lib()
    T, B, H4 = gates_ref.shape
    H = H4 // 4
    dt = torch.bfloat16
    tag = _lib.dtype_tag(dt)
    st = _lib.stream()
    bp = (B + 31) // 32 * 32
    Rd = R.to(DEV).contiguous()
    G = _il(gates_ref, H).to(DEV).contiguous()
    C = torch.zeros(T + 1, B, H, dtype=dt, device=DEV)
    Y = torch.zeros(T + 1, B, H, dtype=dt, device=DEV)
    C[0], Y[0] = c0.to(DEV), y0.to(DEV)
    wt = torch.empty(4 * H * H, dtype=dt, device=DEV)
    ring = torch.empty(2 * bp * H, dtype=dt, device=DEV)
    n0 = lib.caiman_lstm_resident_launches()
    _lib.check(lib.caiman_lstm_prepare(_lib.ptr(Rd), _lib.ptr(Y[0]), _lib.ptr(wt), _lib.ptr(ring), None, B, H, tag, 0, 1, st))
    slot = _lib.FwdSlot(wt.data_ptr(), G.data_ptr(), C.data_ptr(), Y.data_ptr(), ring.data_ptr(), 0, T, None, 0, 0.0, 0)
    arr = (_lib.FwdSlot * 1)(slot)
    _lib.check(lib.caiman_lstm_wave_fwd(ctypes.cast(arr, ctypes.c_void_p), 1, T, B, H, tag, int(hard), 1, 0, st))
    # backward
    D = delta.to(DEV).contiguous()
    dG = torch.empty_like(G)
    wtb = torch.empty(4 * H * H, dtype=dt, device=DEV)
    ringb = torch.empty(2 * bp * 4 * H, dtype=dt, device=DEV)
    dC = torch.empty(B * H, dtype=torch.float32, device=DEV)
    dbias = torch.zeros(4 * H, dtype=torch.float32, device=DEV)
    _lib.check(lib.caiman_lstm_prepare(_lib.ptr(Rd), None, _lib.ptr(wtb), _lib.ptr(ringb), _lib.ptr(dC), B, H, tag, 1, 1, st))
    thi = T - 1
    bslot = _lib.BwdSlot(wtb.data_ptr(), G[thi].data_ptr(), C[thi].data_ptr(), D[thi].data_ptr(), D.stride(0), D.stride(1),
                         dG[thi].data_ptr(), ringb.data_ptr(), dC.data_ptr(), thi & 1, T, 0, 0.0, 0, 0, 0, dbias.data_ptr())
    barr = (_lib.BwdSlot * 1)(bslot)
    _lib.check(lib.caiman_lstm_wave_bwd(ctypes.cast(barr, ctypes.c_void_p), 1, T, B, H, tag, int(hard), 1, 0, st))
    torch.cuda.synchronize()
    launches = lib.caiman_lstm_resident_launches() - n0
    # one launch each way; batches the batch-tile kernels do not take go out as 32-row slices of the B <= 32 kernels
    slices = (B + 31) // 32
    fwd = 1 if B <= 32 or (H in (256, 512, 1024) and B <= 128) else slices
    bwd = 1 if B <= 32 or (H in (512, 1024) and B <= 128) else slices
    assert (launches == fwd + bwd) == expect_resident, launches
    assert lib.caiman_lstm_resident_failures() == 0
    return (_ref(G.cpu(), H), C.cpu(), Y.cpu(), _ref(dG.cpu(), H), _ref(dbias.cpu(), H))


@pytest.mark.parametrize("T,B,H", [(40, 32, 128), (40, 32, 1024), (33, 7, 256), (24, 32, 768),
                                   # batch tiles of 32 rows (lstm_fwd_resident_bt / lstm_bwd_resident2_bt): ragged, 2 and 4 tiles
                                   (12, 33, 512), (10, 64, 1024), (6, 128, 1024), (10, 100, 512),
                                   # H = 1536 (large-196M encoder): DMA-gather forward kernel, 2-D split backward with 3 stages
                                   (20, 32, 1536), (17, 7, 1536),
                                   # 32-row slices of the B <= 32 kernels (res_batch_slice): ragged last slice at H = 1536,
                                   # the whole-row backward kernel (H = 768), batch-tile forward + sliced backward
                                   # (H = 256), more rows than the batch-tile kernels hold (H = 512)
                                   (5, 100, 1536), (8, 64, 768), (7, 70, 256), (5, 160, 512)])
@pytest.mark.parametrize("hard", [False, True])
def test_resident_kernels_match_the_f64_oracle(T, B, H, hard):
    from oracle import native

    g = torch.Generator().manual_seed(1000 * T + H + int(hard))
    dt = torch.bfloat16
    R = (torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dt)
    gates = torch.randn(T, B, 4 * H, generator=g).to(dt)
    c0 = (torch.randn(B, H, generator=g) * 0.5).to(dt)
    y0 = (torch.randn(B, H, generator=g) * 0.5).to(dt)
    delta = torch.randn(T, B, H, generator=g).to(dt)
    ga, c, y, dG, dbias = _resident_fwd_bwd(R, gates, c0, y0, delta, hard)
    og, oc, oy = native.lstm_fwd(R.double().numpy(), gates.double().numpy(), c0.double().numpy(), y0.double().numpy(), hard=hard)

    def check(name, got, ref, max_ulps, mean_ulps):
        got = got.double().numpy()
        scale = max(1.0, np.abs(ref).max())
        err = np.abs(got - ref)
        if hard:   # a clamp decided the other way on a rounded pre-activation moves a whole element: allow a few
            assert np.mean(err > max_ulps * BF16_HALF_ULP * scale) < 2e-3, (name, err.max())
        else:
            assert err.max() <= max_ulps * BF16_HALF_ULP * scale, (name, err.max() / (BF16_HALF_ULP * scale))
        assert err.mean() <= mean_ulps * BF16_HALF_ULP * scale, (name, err.mean() / (BF16_HALF_ULP * scale))

    check("gates", ga, og, 16, 1.0)
    check("y", y, oy, 16, 1.0)
    check("c", c, oc, 32, 2.0)
    # backward from the GPU's own (rounded) forward state, so that only the backward kernel is compared
    odG, _ = native.lstm_bwd(R.double().numpy(), ga.double().numpy(), c.double().numpy(), delta.double().numpy(), hard=hard)
    check("dG", dG, odG, 24, 1.5)
    ob = odG.sum((0, 1))
    assert np.allclose(dbias.double().numpy(), ob, atol=3e-2 * max(1.0, np.abs(ob).max())), "fused bias gradient"


def test_wide_layers_go_out_in_chip_sized_slot_groups():
    """H = 1536: a layer takes 48 workgroups, so a pipeline tick of 7 layers does not fit the 256 CUs as one resident grid;
    the wave calls split the slots into consecutive launches (5 + 2).  Every slot must equal the same layer run alone,
    bit for bit, and the launch count must show the split."""
    from caiman_asr_amd import _lib

    lib = _lib.lib()
    T, B, H, n = 6, 8, 1536, 7
    dt = torch.bfloat16
    tag, st = _lib.dtype_tag(dt), _lib.stream()
    g = torch.Generator().manual_seed(77)
    bp = 32
    per_launch = 256 // 48
    R = [(torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dt).to(DEV) for _ in range(n)]
    G0 = [torch.randn(T, B, 4 * H, generator=g).to(dt).to(DEV) for _ in range(n)]
    D = [torch.randn(T, B, H, generator=g).to(dt).to(DEV) for _ in range(n)]

    def run(group):
        """forward + backward of the layers in `group` in one wave call each way -> per layer (G, Y, dG, dbias)"""
        k = len(group)
        G = [G0[i].clone() for i in group]
        C = [torch.zeros(T + 1, B, H, dtype=dt, device=DEV) for _ in group]
        Y = [torch.zeros(T + 1, B, H, dtype=dt, device=DEV) for _ in group]
        wt = [torch.empty(4 * H * H, dtype=dt, device=DEV) for _ in group]
        ring = [torch.zeros(2 * bp * H, dtype=dt, device=DEV) for _ in group]
        for j, i in enumerate(group):
            _lib.check(lib.caiman_lstm_prepare(_lib.ptr(R[i]), _lib.ptr(Y[j][0]), _lib.ptr(wt[j]), _lib.ptr(ring[j]), None, B, H, tag, 0, 1, st))
        arr = (_lib.FwdSlot * k)(*[_lib.FwdSlot(wt[j].data_ptr(), G[j].data_ptr(), C[j].data_ptr(), Y[j].data_ptr(), ring[j].data_ptr(),
                                                0, T, None, 0, 0.0, 0) for j in range(k)])
        n0 = lib.caiman_lstm_resident_launches()
        _lib.check(lib.caiman_lstm_wave_fwd(ctypes.cast(arr, ctypes.c_void_p), k, T, B, H, tag, 0, 1, 0, st))
        dG = [torch.empty_like(G[j]) for j in range(k)]
        wtb = [torch.empty(4 * H * H, dtype=dt, device=DEV) for _ in group]
        ringb = [torch.zeros(2 * bp * 4 * H, dtype=dt, device=DEV) for _ in group]
        dC = [torch.zeros(B * H, dtype=torch.float32, device=DEV) for _ in group]
        db = [torch.zeros(4 * H, dtype=torch.float32, device=DEV) for _ in group]
        for j, i in enumerate(group):
            _lib.check(lib.caiman_lstm_prepare(_lib.ptr(R[i]), None, _lib.ptr(wtb[j]), _lib.ptr(ringb[j]), _lib.ptr(dC[j]), B, H, tag, 1, 1, st))
        thi = T - 1
        barr = (_lib.BwdSlot * k)(*[_lib.BwdSlot(wtb[j].data_ptr(), G[j][thi].data_ptr(), C[j][thi].data_ptr(), D[i][thi].data_ptr(),
                                                 D[i].stride(0), D[i].stride(1), dG[j][thi].data_ptr(), ringb[j].data_ptr(),
                                                 dC[j].data_ptr(), thi & 1, T, 0, 0.0, 0, 0, 0, db[j].data_ptr())
                                    for j, i in enumerate(group)])
        _lib.check(lib.caiman_lstm_wave_bwd(ctypes.cast(barr, ctypes.c_void_p), k, T, B, H, tag, 0, 1, 0, st))
        torch.cuda.synchronize()
        assert lib.caiman_lstm_resident_failures() == 0
        return [(G[j], Y[j], dG[j], db[j]) for j in range(k)], lib.caiman_lstm_resident_launches() - n0

    together, launches = run(list(range(n)))
    groups = (n + per_launch - 1) // per_launch
    assert launches == 2 * groups, launches       # forward + backward, each split into `groups` resident launches
    for i in range(n):
        alone, one = run([i])
        assert one == 2
        for name, a, b in zip(("gates", "y", "dG", "dbias"), together[i], alone[0]):
            assert torch.equal(a, b), (i, name)
    assert not torch.equal(together[0][1], together[1][1])   # the slots are different layers


def test_resident_geometry_guard_keeps_other_shapes_on_the_step_kernels():
    """Shapes outside the resident kernels' geometry (a hidden size without a resident kernel: 96 = 3 k-steps) must be
    served by the per-timestep kernels, with the same oracle bound."""
    from oracle import native

    T, B, H = 12, 40, 96
    g = torch.Generator().manual_seed(5)
    dt = torch.bfloat16
    R = (torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dt)
    gates = torch.randn(T, B, 4 * H, generator=g).to(dt)
    c0 = torch.zeros(B, H, dtype=dt)
    y0 = torch.zeros(B, H, dtype=dt)
    delta = torch.randn(T, B, H, generator=g).to(dt)
    from caiman_asr_amd import _lib

    would = bool(_lib.lib().caiman_lstm_resident_would_run(B, H, 1))
    assert not would
    ga, c, y, dG, _ = _resident_fwd_bwd(R, gates, c0, y0, delta, False, expect_resident=False)
    og, oc, oy = native.lstm_fwd(R.double().numpy(), gates.double().numpy(), c0.double().numpy(), y0.double().numpy())
    assert np.abs(y.double().numpy() - oy).max() <= 16 * BF16_HALF_ULP


@pytest.mark.parametrize("H", [512, 1024])
def test_split_and_whole_row_backward_kernels_agree_and_are_deterministic(H):
    """The 2-D split backward kernel (H = 512, 1024: a workgroup gathers a quarter of the dG row, K-quarter partial sums
    meet in a second hand-off) against the round-1 kernel (whole row per workgroup): same arithmetic up to the fp32
    summation order of the recurrent product; two runs of the split kernel must agree bit for bit (a stale hand-off
    would show up as a difference); B < 32 exercises the clamped batch rows."""
    from caiman_asr_amd import _lib

    lib = _lib.lib()
    T, B = 37, 23
    g = torch.Generator().manual_seed(H)
    dt = torch.bfloat16
    R = (torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dt)
    gates = torch.randn(T, B, 4 * H, generator=g).to(dt)
    c0 = (torch.randn(B, H, generator=g) * 0.5).to(dt)
    y0 = (torch.randn(B, H, generator=g) * 0.5).to(dt)
    delta = torch.randn(T, B, H, generator=g).to(dt)
    prev = lib.caiman_lstm_resident_bwd_split(1)
    try:
        a1 = _resident_fwd_bwd(R, gates, c0, y0, delta, False)
        a2 = _resident_fwd_bwd(R, gates, c0, y0, delta, False)
        lib.caiman_lstm_resident_bwd_split(0)
        b = _resident_fwd_bwd(R, gates, c0, y0, delta, False)
    finally:
        lib.caiman_lstm_resident_bwd_split(prev)
    for x1, x2 in zip(a1, a2):
        assert torch.equal(x1, x2)
    for i, (x1, x2) in enumerate(zip(a1, b)):
        scale = float(x2.abs().max()) + 1e-6
        assert torch.allclose(x1.float(), x2.float(), atol=(8e-3 if i < 4 else 2e-2) * scale, rtol=0), i


@pytest.mark.parametrize("T,B,H", [(21, 32, 1024), (9, 64, 512)])
def test_layer_to_xcd_placement_option_does_not_change_the_results(T, B, H):
    """caiman_lstm_resident_xcd_roles(1) launches the resident kernels as a flat grid whose workgroup -> layer mapping puts
    a layer on one XCD; the hand-off protocol is placement-independent and the arithmetic identical: bit-equal outputs."""
    from caiman_asr_amd import _lib

    lib = _lib.lib()
    g = torch.Generator().manual_seed(T * H)
    dt = torch.bfloat16
    R = (torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dt)
    gates = torch.randn(T, B, 4 * H, generator=g).to(dt)
    c0 = (torch.randn(B, H, generator=g) * 0.5).to(dt)
    y0 = (torch.randn(B, H, generator=g) * 0.5).to(dt)
    delta = torch.randn(T, B, H, generator=g).to(dt)
    prev = lib.caiman_lstm_resident_xcd_roles(0)
    try:
        a = _resident_fwd_bwd(R, gates, c0, y0, delta, False)
        lib.caiman_lstm_resident_xcd_roles(1)
        b = _resident_fwd_bwd(R, gates, c0, y0, delta, False)
    finally:
        lib.caiman_lstm_resident_xcd_roles(prev)
    for x1, x2 in zip(a, b):
        assert torch.equal(x1, x2)


@pytest.mark.parametrize("T,B,H,hard", [(9, 128, 1024, False), (11, 70, 512, False), (7, 33, 1024, True)])
def test_double_buffered_batch_tile_forward_kernel_is_bit_identical_to_the_register_staged_one(T, B, H, hard):
    """lstm_fwd_resident_bt_dma (operands of tile-step i + 1 by LDS-DMA under the MFMAs of tile-step i) must reproduce
    lstm_fwd_resident_bt exactly -- same MFMA order, same roundings -- including ragged last tiles and repeated launches
    (the prefetch decision depends on timing; the results must not)."""
    from caiman_asr_amd import _lib

    lib = _lib.lib()
    g = torch.Generator().manual_seed(T * B + H)
    dt = torch.bfloat16
    R = (torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dt)
    gates = torch.randn(T, B, 4 * H, generator=g).to(dt)
    c0 = (torch.randn(B, H, generator=g) * 0.5).to(dt)
    y0 = (torch.randn(B, H, generator=g) * 0.5).to(dt)
    delta = torch.randn(T, B, H, generator=g).to(dt)
    outs = []
    for mode in (1, 0, 1, 1):
        prev = lib.caiman_lstm_resident_bt_dma(mode)
        try:
            outs.append(_resident_fwd_bwd(R, gates, c0, y0, delta, hard))
        finally:
            lib.caiman_lstm_resident_bt_dma(prev)
    for other in outs[1:]:
        for name, a, b in zip(("gates", "c", "y", "dG", "dbias"), outs[0], other):
            assert torch.equal(a, b), name


def test_resident_launches_next_to_a_busy_side_stream_complete_and_agree():
    """A REAL contended launch (round-2 verdict: the time-out path was only ever tested by injecting a failure count): while
    a side stream keeps the chip busy with long GEMMs and short kernels, resident launches on the main stream find part of
    the CUs taken -- the workgroups already placed spin on their peers until the other kernel's workgroups retire (a
    resident workgroup fills a CU's register file, so the two never share a CU).  The launches must complete without a
    hand-off time-out and reproduce the uncontended results bit for bit, for the single-tile and the batch-tile kernels."""
    from caiman_asr_amd import _lib

    lib = _lib.lib()
    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)
    small = torch.randn(1 << 16, device=DEV)
    for T, B, H in ((24, 32, 1024), (10, 96, 512)):
        g = torch.Generator().manual_seed(T + B + H)
        dt = torch.bfloat16
        R = (torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dt)
        gates = torch.randn(T, B, 4 * H, generator=g).to(dt)
        c0 = (torch.randn(B, H, generator=g) * 0.5).to(dt)
        y0 = (torch.randn(B, H, generator=g) * 0.5).to(dt)
        delta = torch.randn(T, B, H, generator=g).to(dt)
        quiet = _resident_fwd_bwd(R, gates, c0, y0, delta, False)
        torch.cuda.synchronize()
        for rep in range(3):
            with torch.cuda.stream(side):
                for _ in range(6):
                    a @ a                      # ~1 ms each: every CU busy when the resident launch arrives
                    small.add_(1.0)
            busy = _resident_fwd_bwd(R, gates, c0, y0, delta, False)
            side.synchronize()
            for name, x, y in zip(("gates", "c", "y", "dG", "dbias"), quiet, busy):
                assert torch.equal(x, y), (T, B, H, rep, name)
    assert lib.caiman_lstm_resident_failures() == 0
