"""Winograd F(2x2,3x3) convolution (asr_tap_gemm_wino) against the tap-GEMM (asr_tap_gemm) and float64: forward with
bias + ReLU + BN affine on both outputs, the data-gradient view with accumulation, even / odd plane widths, tile counts that
do not fill a workgroup, borders left untouched; the forward with the 2x2 pool fused in (asr_tap_gemm_wino_pool) against
asr_tap_gemm_wino + asr_pool_fwd, bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from asr_dfcnn_transformer_amd import ops as o
    return o


def conv_ref(x, w, b=None):
    """float64 reference from the oracle (numpy; conv2d_same = tf.layers.conv2d 'same', NHWC x HWIO), not from a GPU library"""
    from oracle import nn as onn
    y = onn.conv2d_same(x.double().cpu().numpy(), w.double().cpu().numpy())
    if b is not None:
        y = y + b.double().cpu().numpy()
    return torch.from_numpy(y).to(x.device)


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 6, 5, 8, 64), (3, 4, 2, 16, 64), (2, 8, 12, 24, 128), (1, 16, 25, 32, 64), (5, 10, 7, 8, 192),
                                            # column-blocked tile order (wino11_kernel: tile columns split into blocks of
                                            # 11..15): two blocks 13 + 12 with partly filled last items, one block of 14 / 15 / 11
                                            # columns at odd and even widths, four blocks, many tile rows per block
                                            (3, 20, 50, 64, 128), (2, 14, 27, 48, 64), (2, 30, 29, 32, 64), (1, 200, 22, 32, 64),
                                            (2, 12, 100, 32, 64), (1, 6, 200, 32, 64),
                                            # channel counts that are odd multiples of 32 (wino11_kernel only)
                                            (2, 20, 50, 64, 32), (1, 16, 25, 32, 32), (3, 14, 27, 24, 96), (2, 200, 22, 8, 32),
                                            # odd plane heights (round 4; T_pad 1000 gives 125 x 25 planes): a half-filled last tile
                                            # row on both kernels -- wino8 (plain order), wino11 (column blocks)
                                            (2, 7, 5, 8, 64), (1, 125, 25, 32, 64), (2, 25, 50, 64, 128), (1, 125, 25, 32, 32), (3, 7, 27, 16, 96),
                                            (2, 1, 25, 32, 64), (4, 125, 25, 128, 128)])
def test_forward_matches_tap_gemm_and_float64(ops, B, H, W, cin, cout):
    g = torch.Generator(device='cuda').manual_seed(11)
    x = ops.Plane(B, H, W, cin); xi = torch.randn(B, H, W, cin, device='cuda', generator=g); x.set_interior(xi)
    w = torch.randn(3, 3, cin, cout, device='cuda', generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device='cuda', generator=g) * 0.1
    sc = 1 + 0.2 * torch.randn(cout, device='cuda', generator=g); sh = 0.1 * torch.randn(cout, device='cuda', generator=g)
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, cout, ntaps=9, B=B, H=H, W=W, relu=1)
    assert ops.winograd_supported(d)
    a0, y0, a1, y1 = (ops.Plane(B, H, W, cout) for _ in range(4))
    ops.tap_gemm(d, x, w, bias, sc, sh, a0, y0)
    ops.tap_gemm_wino(d, x, ops.winograd_weights(w, cin, cout, cout, 0), bias, sc, sh, a1, y1)
    ref = torch.relu(conv_ref(xi, w, bias))
    ea = (a1.interior().double() - ref).abs().max().item()
    assert ea < 2e-5 * max(1.0, ref.abs().max().item()), ea
    assert (a1.interior() - a0.interior()).abs().max().item() < 3e-5
    assert (y1.interior() - y0.interior()).abs().max().item() < 3e-5
    for p in (a1, y1):                                   # borders and guards stay EXACTLY zero (SAME padding of the next layer)
        assert p.border_abs_max() == 0.0


@pytest.mark.parametrize("pool", [1, 2])
@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 6, 5, 8, 64), (3, 4, 2, 16, 64), (2, 8, 12, 24, 128), (1, 16, 26, 32, 64), (2, 40, 100, 32, 64),
                                            (32, 30, 50, 16, 128), (2, 40, 100, 32, 32), (3, 12, 26, 16, 96)])
def test_fused_pool_equals_conv_then_pool_bitwise(ops, pool, B, H, W, cin, cout):
    """One launch writes the activation AND its pooled BN output; the pooled plane equals asr_pool_fwd of the stored activation
    bit for bit (average and maximum, odd widths drop the last column as asr_pool_fwd does, several items per workgroup)."""
    g = torch.Generator(device='cuda').manual_seed(21)
    x = ops.Plane(B, H, W, cin); x.set_interior(torch.randn(B, H, W, cin, device='cuda', generator=g))
    w = torch.randn(3, 3, cin, cout, device='cuda', generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device='cuda', generator=g) * 0.1
    sc = 1 + 0.2 * torch.randn(cout, device='cuda', generator=g); sh = 0.1 * torch.randn(cout, device='cuda', generator=g)
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, 0, ntaps=9, B=B, H=H, W=W, relu=1)
    wt = ops.winograd_weights(w, cin, cout, cout, 0)
    a0, a1 = ops.Plane(B, H, W, cout), ops.Plane(B, H, W, cout)
    y0, y1 = ops.Plane(B, H // 2, W // 2, cout), ops.Plane(B, H // 2, W // 2, cout)
    ops.tap_gemm_wino(d, x, wt, bias, sc, sh, a0, None)
    ops.pool_fwd(a0, sc, sh, pool, y0)
    ops.tap_gemm_wino_pool(d, x, wt, bias, sc, sh, a1, pool, y1)
    torch.cuda.synchronize()
    assert torch.equal(a0.buf, a1.buf)
    assert torch.equal(y0.buf, y1.buf)
    assert y1.interior().abs().sum().item() > 0


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 6, 5, 64, 8), (1, 12, 25, 128, 32), (3, 4, 4, 64, 16), (2, 20, 50, 64, 64), (3, 10, 23, 128, 48), (2, 20, 50, 32, 64), (1, 40, 25, 96, 256),
                                            (1, 125, 25, 64, 128), (2, 7, 5, 8, 64), (2, 25, 23, 32, 128), (3, 125, 25, 128, 128)])
def test_data_gradient_view_with_accumulation(ops, B, H, W, cin, cout):
    g = torch.Generator(device='cuda').manual_seed(12)
    w = torch.randn(3, 3, cin, cout, device='cuda', generator=g) * 0.1
    dz = ops.Plane(B, H, W, cout); dzi = torch.randn(B, H, W, cout, device='cuda', generator=g); dz.set_interior(dzi)
    prev = torch.randn(B, H, W, cin, device='cuda', generator=g)
    dx0, dx1 = ops.Plane(B, H, W, cin), ops.Plane(B, H, W, cin)
    dx0.set_interior(prev); dx1.set_interior(prev)
    bd = ops.gemm_desc(dz.NP, cout, cin, cout, cout, 0, cin, ntaps=9, B=B, H=H, W=W, wmode=1, accumulate=1)
    if not ops.winograd_supported(bd):
        pytest.skip('this view is not a Winograd shape (channel counts not multiples of 8 / 32, or no column blocks for N % 64 == 32)')
    ops.tap_gemm(bd, dz, w, None, None, None, None, dx0)
    ops.tap_gemm_wino(bd, dz, ops.winograd_weights(w, cout, cin, cout, 1), None, None, None, None, dx1)
    assert (dx1.interior() - dx0.interior()).abs().max().item() < 3e-5 * max(1.0, dx0.interior().abs().max().item() / 4)
    # float64 (oracle.nn.conv2d_same_bwd's rule: correlate dZ with the 180-degree-rotated, in/out-swapped kernel) + what was there
    ref = conv_ref(dzi, w.flip(0, 1).permute(0, 1, 3, 2).contiguous()) + prev.double()
    assert (dx1.interior().double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    assert dx1.border_abs_max() == 0.0


def test_unsupported_shapes_are_reported(ops):
    d = ops.gemm_desc(2 * 6 * 7, 8, 32, 8, 32, 32, 0, ntaps=9, B=2, H=5, W=6)      # N % 64 == 32 with too few tile columns for a column block
    assert not ops.winograd_supported(d)
    d = ops.gemm_desc(2 * 6 * 7, 12, 64, 12, 64, 64, 0, ntaps=9, B=2, H=5, W=6)     # K % 8 != 0
    assert not ops.winograd_supported(d)
    d = ops.gemm_desc(2 * 6 * 7, 8, 64, 8, 64, 64, 0, ntaps=9, B=2, H=5, W=6)       # odd height: supported since round 4
    assert ops.winograd_supported(d)


def test_engine_opt_in_gives_the_same_step(ops):
    """wino=True (the default) routes the supported 3x3 layers (forward, data-gradient and gated data-gradient) through the
    Winograd kernel; wino=False keeps all of them on the tap-GEMM: logits, loss and every gradient agree to rounding."""
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine
    rng = np.random.default_rng(3)
    B, T, F, V = 2, 32, 16, 20
    x = torch.tensor(rng.standard_normal((B, T, F)).astype(np.float32), device='cuda')
    target = np.zeros((B, 64), dtype=np.int32); target[:, :2] = rng.integers(1, V - 1, (B, 2))
    seq = [T // 8, T // 8]
    out = []
    for flag in (False, True):
        eng = DFCNNEngine(model='small', vocab=V, B=B, T=T, F=F, widths=(64, 64, 64, 64), seed=4, wino=flag)
        assert bool(eng.wt_f) == flag and bool(eng.wt_b) == flag
        logits = eng.forward(x).clone()
        eng.set_targets(seq, target); eng.loss_and_decode(); eng.backward()
        torch.cuda.synchronize()
        out.append((logits, eng.loss.clone(), eng.grad.clone()))
    (l0, s0, g0), (l1, s1, g1) = out
    assert (l0 - l1).abs().max().item() < 1e-4
    assert (s0 - s1).abs().max().item() < 1e-4 * max(1.0, s0.abs().max().item())
    assert (g0 - g1).abs().max().item() < 1e-4 * max(1e-6, g0.abs().max().item())


@pytest.mark.parametrize("B,H,W,cin,cout,hid", [(3, 20, 50, 32, 32, 32), (2, 40, 25, 64, 64, 32), (4, 25, 25, 128, 128, 64), (2, 16, 100, 32, 32, 32)])
def test_forward_with_squeeze_sums_feeds_the_se_block(ops, B, H, W, cin, cout, hid):
    """asr_tap_gemm_wino_sums (round 5): the activation and BN-output planes are those of asr_tap_gemm_wino bit for bit, and the partial
    rows fold to the per-image channel sums of y (float64, 1e-5 of their scale); asr_se_fwd_sums on them = asr_se_fwd with its own pass
    over the plane, to rounding (acoustic_model2.py:135-148: Global_Average_Pooling -> dense -> relu -> dense -> sigmoid -> scale)."""
    g = torch.Generator(device='cuda').manual_seed(23)
    x = ops.Plane(B, H, W, cin); x.set_interior(torch.randn(B, H, W, cin, device='cuda', generator=g))
    w = torch.randn(3, 3, cin, cout, device='cuda', generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device='cuda', generator=g) * 0.1
    sc = 1 + 0.2 * torch.randn(cout, device='cuda', generator=g); sh = 0.1 * torch.randn(cout, device='cuda', generator=g)
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, cout, ntaps=9, B=B, H=H, W=W, relu=1)
    rows = ops.winograd_sum_rows(d)
    assert rows > 0 and rows % B == 0
    wt = ops.winograd_weights(w, cin, cout, cout, 0)
    a0, y0, a1, y1 = (ops.Plane(B, H, W, cout) for _ in range(4))
    ops.tap_gemm_wino(d, x, wt, bias, sc, sh, a0, y0)
    sums = torch.full((rows * cout,), float('nan'), device='cuda')          # every partial row is written by the launch
    ops.tap_gemm_wino_sums(d, x, wt, bias, sc, sh, a1, y1, sums)
    torch.cuda.synchronize()
    assert torch.equal(a0.buf, a1.buf) and torch.equal(y0.buf, y1.buf)
    got = sums.view(B, rows // B, cout).double().sum(1)
    want = y1.interior().double().sum((1, 2))
    assert (got - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    # the SE block on top (main = another plane, branch = y1)
    main = ops.Plane(B, H, W, cout); main.set_interior(torch.randn(B, H, W, cout, device='cuda', generator=g))
    ssc = 1 + 0.1 * torch.randn(cout, device='cuda', generator=g); ssh = 0.1 * torch.randn(cout, device='cuda', generator=g)
    w1 = torch.randn(cout, hid, device='cuda', generator=g) * 0.2; b1 = torch.randn(hid, device='cuda', generator=g) * 0.1
    w2 = torch.randn(hid, cout, device='cuda', generator=g) * 0.2; b2 = torch.randn(cout, device='cuda', generator=g) * 0.1
    st0 = torch.zeros(ops.se_state_floats(B, cout, hid), device='cuda'); st1 = torch.zeros_like(st0)
    ws = torch.zeros(ops.se_fwd_workspace(B, H, W, cout) // 4 + 64, device='cuda')
    o0, o1 = ops.Plane(B, H, W, cout), ops.Plane(B, H, W, cout)
    ops.se_fwd(main, y1, hid, ssc, ssh, w1, b1, w2, b2, st0, ws, o0)
    ops.se_fwd_sums(main, y1, hid, ssc, ssh, w1, b1, w2, b2, st1, sums, rows // B, o1)
    torch.cuda.synchronize()
    assert (st0 - st1).abs().max().item() <= 2e-6 * max(1.0, st0.abs().max().item())
    assert (o0.interior() - o1.interior()).abs().max().item() <= 2e-6 * max(1.0, o0.interior().abs().max().item())
    assert o1.border_abs_max() == 0.0


@pytest.mark.parametrize("B,H,W,K,N", [(3, 20, 50, 64, 32), (2, 40, 25, 128, 64), (4, 25, 25, 256, 128), (2, 16, 100, 64, 32)])
def test_data_gradient_with_the_se_backward_reduction(ops, B, H, W, K, N):
    """asr_tap_gemm_wino_sesum (round 5): dy is asr_tap_gemm_wino's data-gradient bit for bit, and the partial rows fold to the per-image
    sums of dy * (sc * x + sh) over the pixels (float64, 1e-5 of their scale) -- se_reduce_kernel<1>'s result, the first step of the
    backward of squeeze_excitation_layer (acoustic_model2.py:135-148)."""
    g = torch.Generator(device='cuda').manual_seed(29)
    w = torch.randn(3, 3, N, K, device='cuda', generator=g) * 0.1
    dz = ops.Plane(B, H, W, K); dz.set_interior(torch.randn(B, H, W, K, device='cuda', generator=g))
    x = ops.Plane(B, H, W, N); x.set_interior(torch.randn(B, H, W, N, device='cuda', generator=g))
    sc = 1 + 0.2 * torch.randn(N, device='cuda', generator=g); sh = 0.1 * torch.randn(N, device='cuda', generator=g)
    bd = ops.gemm_desc(dz.NP, K, N, K, K, 0, N, ntaps=9, B=B, H=H, W=W, wmode=1)
    rows = ops.winograd_sum_rows(bd)
    assert rows > 0 and rows % B == 0
    wt = ops.winograd_weights(w, K, N, K, 1)
    dy0, dy1 = ops.Plane(B, H, W, N), ops.Plane(B, H, W, N)
    ops.tap_gemm_wino(bd, dz, wt, None, None, None, None, dy0)
    sums = torch.full((rows * N,), float('nan'), device='cuda')
    ops.tap_gemm_wino_sesum(bd, dz, wt, x, sc, sh, dy1, sums)
    torch.cuda.synchronize()
    assert torch.equal(dy0.buf, dy1.buf)
    got = sums.view(B, rows // B, N).double().sum(1)
    want = (dy1.interior().double() * (sc.double() * x.interior().double() + sh.double())).sum((1, 2))
    assert (got - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
