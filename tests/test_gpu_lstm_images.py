"""caiman_lstm_weight_images (csrc/lstm_images.hip): every 16-bit operand image of the LSTM parameters in one launch.
Bit-exact against the images the separate paths build: torch permute / transpose + cast for the input weights and the bias
(rounding of the same fp32 values), and caiman_lstm_prepare's own tiling kernels for the recurrent weights."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,K", [(32, 240), (64, 64), (512, 512), (1024, 2048), (96, 36)])
def test_images_equal_the_separately_built_ones(H, K, dt):
    from caiman_asr_amd import _lib
    from caiman_asr_amd.rnnt_ext.custom_lstm.encoder_pipe import _perm_cast, _perm_cast_t

    lib = _lib.lib()
    g = torch.Generator().manual_seed(H * 7 + K)
    layers = []
    for _ in range(3):   # several layers in one launch, the middle one without the N-major image
        W = torch.randn(4 * H, K, generator=g).to(DEV)
        R = torch.randn(4 * H, H, generator=g).to(DEV)
        bW, bR = torch.randn(4 * H, generator=g).to(DEV), torch.randn(4 * H, generator=g).to(DEV)
        layers.append((W, R, bW, bR))
    outs = []
    for i, (W, R, bW, bR) in enumerate(layers):
        Wt = torch.full((K, 4 * H), 3.0, dtype=dt, device=DEV)
        Wn = torch.full((4 * H, K), 3.0, dtype=dt, device=DEV) if i != 1 else None
        bias = torch.empty(4 * H, dtype=dt, device=DEV)
        Rf = torch.empty(4 * H * H, dtype=dt, device=DEV)
        Rb = torch.empty(4 * H * H, dtype=dt, device=DEV)
        outs.append((Wt, Wn, bias, Rf, Rb))
    arr = (_lib.LstmImages * 3)(*[
        _lib.LstmImages(W.data_ptr(), R.data_ptr(), bW.data_ptr(), bR.data_ptr(), Wt.data_ptr(),
                        Wn.data_ptr() if Wn is not None else None, bias.data_ptr(), Rf.data_ptr(), Rb.data_ptr(), H, K)
        for (W, R, bW, bR), (Wt, Wn, bias, Rf, Rb) in zip(layers, outs)])
    _lib.check(lib.caiman_lstm_weight_images(ctypes.cast(arr, ctypes.c_void_p), 3, _lib.dtype_tag(dt), _lib.stream()))
    torch.cuda.synchronize()
    tag, st = _lib.dtype_tag(dt), _lib.stream()
    for (W, R, bW, bR), (Wt, Wn, bias, Rf, Rb) in zip(layers, outs):
        assert torch.equal(Wt, _perm_cast_t(W, H, dt))
        if Wn is not None:
            assert torch.equal(Wn, _perm_cast(W, H, dt))
        assert torch.equal(bias, _perm_cast(bW + bR, H, dt))
        Rp = R.to(dt).contiguous()
        B = 4
        h0 = torch.zeros(B, H, dtype=dt, device=DEV)
        ring = torch.empty(2 * 32 * 4 * H, dtype=dt, device=DEV)
        dC = torch.empty(B * H, dtype=torch.float32, device=DEV)
        ref_f, ref_b = torch.empty_like(Rf), torch.empty_like(Rb)
        _lib.check(lib.caiman_lstm_prepare(_lib.ptr(Rp), _lib.ptr(h0), _lib.ptr(ref_f), _lib.ptr(ring), None, B, H, tag, 0, 1, st))
        _lib.check(lib.caiman_lstm_prepare(_lib.ptr(Rp), None, _lib.ptr(ref_b), _lib.ptr(ring), _lib.ptr(dC), B, H, tag, 1, 1, st))
        torch.cuda.synchronize()
        assert torch.equal(Rf, ref_f), "forward fragment image"
        assert torch.equal(Rb, ref_b), "backward fragment image"


def test_bad_geometry_is_refused():
    from caiman_asr_amd import _lib

    W = torch.zeros(4 * 48, 64, device=DEV)
    arr = (_lib.LstmImages * 1)(_lib.LstmImages(W.data_ptr(), W.data_ptr(), None, None, None, None, None, None, None, 48, 64))
    with pytest.raises(RuntimeError):
        _lib.check(_lib.lib().caiman_lstm_weight_images(ctypes.cast(arr, ctypes.c_void_p), 1, _lib.dtype_tag(torch.bfloat16),
                                                         _lib.stream()))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,K", [(32, 240), (64, 36), (512, 1024), (96, 7)])
def test_grad_deliver_adds_the_unpermuted_gradients(H, K, dt):
    """caiman_lstm_grad_deliver: dst[(gate*H + unit)*cols + c] += src[(unit*4 + gate)*cols + c] for the four parameters of a
    layer in one launch, 16-bit weight gradients and fp32 bias sums -- against the torch expression it replaces."""
    from caiman_asr_amd import _lib

    g = torch.Generator().manual_seed(H + K)
    gW = torch.randn(4 * H, K, generator=g).to(dt).to(DEV)
    gR = torch.randn(4 * H, H, generator=g).to(dt).to(DEV)
    gB = torch.randn(4 * H, generator=g).to(DEV)
    dst = [torch.randn(4 * H, K, generator=g).to(DEV), torch.randn(4 * H, H, generator=g).to(DEV),
           torch.randn(4 * H, generator=g).to(DEV), torch.randn(4 * H, generator=g).to(DEV)]
    ref = [d.clone() for d in dst]
    for r, s in zip(ref, (gW, gR, gB, gB)):
        r.view(4, H, *r.shape[1:]).add_(s.view(H, 4, *s.shape[1:]).transpose(0, 1))
    items = (_lib.GradItem * 4)(*[_lib.GradItem(s.data_ptr(), d.data_ptr(), H, d.shape[1] if d.dim() == 2 else 1,
                                                int(s.dtype == torch.float32), 0) for d, s in zip(dst, (gW, gR, gB, gB))])
    _lib.check(_lib.lib().caiman_lstm_grad_deliver(ctypes.cast(items, ctypes.c_void_p), 4, _lib.dtype_tag(dt), _lib.stream()))
    torch.cuda.synchronize()
    for d, r in zip(dst, ref):
        assert torch.equal(d, r)
