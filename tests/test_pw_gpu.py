"""GPU parity of the contractions on pre-arranged operands against the float64 oracle:
  asr_arrange_weights + asr_tap_gemm_pw   (fp32 MFMA, weights in fragment order: the 3x3 conv path of the layers Winograd does not take)
Bars: outputs 1e-4 abs at unit-scale operands (fp32 chains; north_star's bar for logits is 1e-3); the pw kernel must
also agree with asr_tap_gemm to fp32 rounding, and repeated launches must be bitwise identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def conv_ref(x, w, b=None):
    """float64 reference from the oracle (numpy), not from a GPU library"""
    from oracle import nn as onn
    y = onn.conv2d_same(x.double().cpu().numpy(), w.double().cpu().numpy())
    if b is not None:
        y = y + b.double().cpu().numpy()
    return torch.from_numpy(y).to(x.device)


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 9, 7, 20, 24), (1, 40, 25, 64, 64), (2, 16, 12, 32, 136), (1, 33, 50, 128, 32),
                                            (2, 12, 10, 4, 32), (1, 21, 100, 36, 100)])
def test_prearranged_conv_forward_and_data_gradient(B, H, W, cin, cout):
    from asr_dfcnn_transformer_amd import ops
    g = torch.Generator(device='cuda').manual_seed(0)
    x = ops.Plane(B, H, W, cin); x.set_interior(torch.randn(B, H, W, cin, device='cuda', generator=g))
    w = torch.randn(3, 3, cin, cout, device='cuda', generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device='cuda', generator=g) * 0.1
    sc = torch.rand(cout, device='cuda', generator=g) + 0.5
    sh = torch.randn(cout, device='cuda', generator=g) * 0.1
    a, y, y1 = ops.Plane(B, H, W, cout), ops.Plane(B, H, W, cout), ops.Plane(B, H, W, cout)
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, cout, ntaps=9, B=B, H=H, W=W, relu=1)
    wf = ops.arrange_weights(w, 9, cin, cout, cout, 0)
    ops.tap_gemm_pw(d, x, wf, bias, sc, sh, a, y)
    ref_a = torch.relu(conv_ref(x.interior(), w, bias))
    assert (a.interior().double() - ref_a).abs().max().item() < 1e-4
    assert (y.interior().double() - (ref_a * sc.double() + sh.double())).abs().max().item() < 1e-4
    assert a.border_abs_max() == 0.0 and y.border_abs_max() == 0.0                                     # borders and guards untouched
    ops.tap_gemm(d, x, w, bias, sc, sh, None, y1)
    assert (y.interior() - y1.interior()).abs().max().item() < 5e-5                                   # vs the LDS-staged kernel
    a2 = ops.Plane(B, H, W, cout)
    ops.tap_gemm_pw(d, x, wf, bias, sc, sh, a2, None)
    assert torch.equal(a.body, a2.body)                                                               # deterministic
    # data gradient = conv of dz with the mirrored, transposed weights; accumulate on top of an existing gradient
    dz = ops.Plane(B, H, W, cout); dz.set_interior(torch.randn(B, H, W, cout, device='cuda', generator=g))
    dx = ops.Plane(B, H, W, cin); dx.set_interior(torch.ones(B, H, W, cin, device='cuda'))
    bd = ops.gemm_desc(x.NP, cout, cin, cout, cout, 0, cin, ntaps=9, B=B, H=H, W=W, wmode=1, accumulate=1)
    ops.tap_gemm_pw(bd, dz, ops.arrange_weights(w, 9, cout, cin, cout, 1), None, None, None, None, dx)
    wt = w.flip(0, 1).permute(0, 1, 3, 2).contiguous()           # [3][3][cout][cin], taps mirrored
    ref = conv_ref(dz.interior(), wt) + 1.0
    assert (dx.interior().double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("M,K,N", [(1000, 72, 100), (300, 512, 36), (4096, 256, 512), (130, 6400, 128)])
def test_prearranged_dense_forward_and_data_gradient(M, K, N):
    from asr_dfcnn_transformer_amd import ops
    g = torch.Generator(device='cuda').manual_seed(1)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(K, N, device='cuda', generator=g) * (1.0 / K) ** 0.5
    bias = torch.randn(N, device='cuda', generator=g) * 0.1
    y = torch.zeros(M, N, device='cuda')
    d = ops.gemm_desc(M, K, N, K, N, N, 0, ntaps=1, relu=1)
    ops.tap_gemm_pw(d, a, ops.arrange_weights(w, 1, K, N, N, 0), bias, None, None, y, None)
    ref = torch.relu(a.double() @ w.double() + bias.double())
    assert (y.double() - ref).abs().max().item() < 1e-4
    dy = torch.randn(M, N, device='cuda', generator=g)
    da = torch.zeros(M, K, device='cuda')
    bd = ops.gemm_desc(M, N, K, N, N, 0, K, ntaps=1, wmode=1)
    ops.tap_gemm_pw(bd, dy, ops.arrange_weights(w, 1, N, K, N, 1), None, None, None, None, da)
    scale = float((dy.double() @ w.double().t()).abs().max())
    assert (da.double() - dy.double() @ w.double().t()).abs().max().item() < 1e-5 * max(1.0, scale) + 1e-4
