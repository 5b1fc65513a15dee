"""GPU parity tests: every C-ABI entry point against the float64 oracle on seeded inputs.
Tolerances are stated per test; integer outputs are bit-exact."""
import math

import numpy as np
import pytest
import torch

from oracle import nn as onn, ctc as octc, optim as oopt, fbank as ofb

pytestmark = pytest.mark.gpu


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device='cuda')


def report(name, got, want, tol):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    err = np.abs(got - want).max() if got.size else 0.0
    scale = max(1.0, np.abs(want).max()) if want.size else 1.0
    print('%-40s max|err| %.3e  (scale %.3e, tol %.1e)' % (name, err, scale, tol))
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert err <= tol * scale, '%s: err %.3e > %.1e*%.3e' % (name, err, tol, scale)


@pytest.fixture(scope='module')
def ops():
    from asr_dfcnn_transformer_amd import ops as _ops
    return _ops


def to_plane(ops, x):
    B, H, W, C = x.shape
    p = ops.Plane(B, H, W, C)
    p.set_interior(dev(x))
    return p


def plane_np(p):
    return p.interior().cpu().numpy()


def check_border_zero(p):
    v = p.view().clone()
    v[:, 1:, 1:, :] = 0
    assert float(v.abs().max()) == 0.0
    G, Cc = p.G, p.C
    assert float(p.buf[:G * Cc].abs().max()) == 0.0 and float(p.buf[-G * Cc:].abs().max()) == 0.0


# ------------------------------------------------------------------ tap-GEMM (conv fwd / dgrad / dense)
@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 12, 9, 32, 32), (2, 20, 25, 64, 128), (1, 37, 11, 32, 64),
                                            (3, 9, 50, 128, 256), (1, 8, 6, 8, 20)])
def test_conv3x3_cell_forward(ops, B, H, W, cin, cout):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((B, H, W, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * 0.1).astype(np.float32)
    bias = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    sc = (1 + 0.2 * rng.standard_normal(cout)).astype(np.float32)
    sh = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    xp = to_plane(ops, x)
    a = ops.Plane(B, H, W, cout)
    y = ops.Plane(B, H, W, cout)
    d = ops.gemm_desc(xp.NP, cin, cout, cin, cout, cout, cout, ntaps=9, B=B, H=H, W=W, relu=1)
    ops.tap_gemm(d, xp, dev(w), dev(bias), dev(sc), dev(sh), a, y)
    torch.cuda.synchronize()
    z = onn.conv2d_same(x.astype(np.float64), w.astype(np.float64)) + bias
    aref = np.maximum(z, 0)
    report('conv3x3 a', plane_np(a), aref, 2e-5)
    report('conv3x3 y', plane_np(y), sc * aref + sh, 2e-5)
    check_border_zero(a)
    check_border_zero(y)
    # unpadded y (feeds the dense head)
    yu = torch.zeros(B * H * W, cout, device='cuda')
    d2 = ops.gemm_desc(xp.NP, cin, cout, cin, cout, cout, cout, ntaps=9, B=B, H=H, W=W, relu=1, y_unpadded=1)
    ops.tap_gemm(d2, xp, dev(w), dev(bias), dev(sc), dev(sh), None, yu)
    report('conv3x3 y unpadded', yu.cpu().numpy().reshape(B, H, W, cout), sc * aref + sh, 2e-5)


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 12, 9, 32, 32), (2, 10, 25, 64, 128), (1, 13, 7, 128, 64)])
def test_conv3x3_dgrad_and_accumulate(ops, B, H, W, cin, cout):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, H, W, cin))
    w = (rng.standard_normal((3, 3, cin, cout)) * 0.1).astype(np.float32)
    dz = rng.standard_normal((B, H, W, cout)).astype(np.float32)
    dx_ref, _ = onn.conv2d_same_bwd(x, w.astype(np.float64), dz.astype(np.float64))
    dzp = to_plane(ops, dz)
    dx = ops.Plane(B, H, W, cin)
    d = ops.gemm_desc(dzp.NP, cout, cin, cout, cout, 0, cin, ntaps=9, B=B, H=H, W=W, wmode=1)
    ops.tap_gemm(d, dzp, dev(w), None, None, None, None, dx)
    report('dgrad', plane_np(dx), dx_ref, 2e-5)
    check_border_zero(dx)
    d.accumulate = 1
    ops.tap_gemm(d, dzp, dev(w), None, None, None, None, dx)
    report('dgrad accumulate', plane_np(dx), 2 * dx_ref, 2e-5)


@pytest.mark.parametrize("M,K,N", [(300, 64, 128), (517, 96, 1424), (64, 6400, 1536), (130, 128, 32), (700, 36, 64)])
def test_dense_fwd_and_dgrad(ops, M, K, N):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / math.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    out = torch.zeros(M, N, device='cuda')
    d = ops.gemm_desc(M, K, N, K, N, 0, N, ntaps=1)
    ops.tap_gemm(d, dev(x), dev(w), dev(b), None, None, None, out)
    report('dense fwd', out.cpu().numpy(), x.astype(np.float64) @ w.astype(np.float64) + b, 2e-5)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    dx = torch.zeros(M, K, device='cuda')
    d2 = ops.gemm_desc(M, N, K, N, N, 0, K, ntaps=1, wmode=1)
    ops.tap_gemm(d2, dev(dy), dev(w), None, None, None, None, dx)
    report('dense dgrad', dx.cpu().numpy(), dy.astype(np.float64) @ w.astype(np.float64).T, 2e-5)


def test_conv1x1_forward_and_dgrad(ops):
    rng = np.random.default_rng(3)
    B, H, W, cin, cout = 2, 9, 7, 64, 32
    x = rng.standard_normal((B, H, W, cin)).astype(np.float32)
    w = (rng.standard_normal((1, 1, cin, cout)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    xp = to_plane(ops, x)
    a = ops.Plane(B, H, W, cout)
    d = ops.gemm_desc(xp.NP, cin, cout, cin, cout, cout, 0, ntaps=1, B=B, H=H, W=W, relu=1)
    ops.tap_gemm(d, xp, dev(w), dev(bias), None, None, a, None)
    report('conv1x1', plane_np(a), np.maximum(x.astype(np.float64) @ w[0, 0].astype(np.float64) + bias, 0), 2e-5)
    check_border_zero(a)


# ------------------------------------------------------------------ wgrad
@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 12, 9, 32, 32), (2, 30, 25, 64, 128), (1, 40, 21, 32, 64),
                                            (2, 9, 12, 128, 256), (3, 50, 50, 32, 32),
                                            # Winograd F(3x3, 2x2) weight gradient (wino_wgrad.hip: 32 or 64 k input channels, 64 n
                                            # output channels, enough stages): two and four column blocks, odd tile-row counts, odd
                                            # widths, a one-column plane, several block pairs, 32 input channels (tile-pair halves)
                                            (4, 20, 50, 64, 64), (2, 16, 100, 32, 64), (3, 22, 27, 128, 64), (2, 36, 2, 64, 64),
                                            (2, 26, 25, 32, 128), (1, 64, 13, 128, 128),
                                            # odd heights on the Winograd route (round 4): the last tile row is half filled
                                            (2, 125, 25, 64, 64), (3, 7, 50, 32, 64), (2, 25, 25, 128, 128), (16, 1, 25, 64, 64),
                                            # 32 -> 32 channels (round 5, acoustic_model2.py:39; wino_wgrad4n32_kernel: stages of FOUR tile rows,
                                            # one per wave class): four column blocks, tile-row counts that are not multiples of four, odd
                                            # block widths (last tiles of two rows paired), odd heights, a single pixel row
                                            (2, 16, 100, 32, 32), (3, 22, 27, 32, 32), (2, 125, 25, 32, 32), (4, 30, 52, 32, 32),
                                            (16, 1, 25, 32, 32), (2, 38, 50, 32, 32)])
def test_conv3x3_wgrad(ops, B, H, W, cin, cout):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((B, H, W, cin)).astype(np.float32)
    dz = rng.standard_normal((B, H, W, cout)).astype(np.float32)
    _, dw_ref = onn.conv2d_same_bwd(x.astype(np.float64), np.zeros((3, 3, cin, cout)), dz.astype(np.float64))
    xp, dzp = to_plane(ops, x), to_plane(ops, dz)
    d = ops.gemm_desc(xp.NP, cin, cout, cin, cout, ntaps=9, B=B, H=H, W=W)
    ws = torch.zeros(max(4, ops.tap_wgrad_workspace(d) // 4), device='cuda')
    dw = torch.zeros(9 * cin * cout, device='cuda')
    ops.tap_wgrad(d, xp, dzp, cout, dw, ws)
    got = dw.cpu().numpy().reshape(3, 3, cin, cout)
    report('wgrad 3x3', got, dw_ref, 3e-5)
    dw2 = torch.zeros_like(dw)
    ops.tap_wgrad(d, xp, dzp, cout, dw2, ws)
    assert torch.equal(dw, dw2), 'wgrad must be bitwise reproducible'


@pytest.mark.parametrize("M,K,N,splits", [(640, 6400, 128, 4), (300, 512, 64, 2), (129, 1024, 192, 8), (6400, 6400, 128, 4)])
def test_dense_forward_split_k(ops, M, K, N, splits):
    """asr_tap_gemm_splitk = asr_tap_gemm up to the order of the K sum: bias + ReLU on out_a, affine on out_y; float64 check;
    bitwise reproducible."""
    rng = np.random.default_rng(15)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    sc = (1 + 0.2 * rng.standard_normal(N)).astype(np.float32); sh = (0.1 * rng.standard_normal(N)).astype(np.float32)
    d = ops.gemm_desc(M, K, N, K, N, N, N, ntaps=1, relu=1)
    a0, y0, a1, y1, a2 = (torch.zeros(M, N, device='cuda') for _ in range(5))
    ws = torch.zeros(ops.tap_gemm_splitk_workspace(d, splits) // 4 + 4, device='cuda')
    ops.tap_gemm(d, dev(x), dev(w), dev(b), dev(sc), dev(sh), a0, y0)
    ops.tap_gemm_splitk(d, dev(x), dev(w), dev(b), dev(sc), dev(sh), a1, y1, splits, ws)
    ops.tap_gemm_splitk(d, dev(x), dev(w), dev(b), None, None, a2, None, splits, ws)
    ref = np.maximum(x.astype(np.float64) @ w.astype(np.float64) + b, 0.0)
    report('split-K a', a1.cpu().numpy(), ref, 2e-5)
    report('split-K y', y1.cpu().numpy(), ref * sc + sh, 2e-5)
    assert (a1 - a0).abs().max().item() < 2e-5 * max(1.0, float(np.abs(ref).max()))
    assert torch.equal(a1, a2)


@pytest.mark.parametrize("M,K,N,splits", [(640, 6400, 128, 10), (300, 512, 64, 2), (129, 1024, 192, 8), (6400, 6400, 128, 10), (6400, 1536, 128, 8),
                                          (200, 6400, 128, 10)])
def test_dense_split_k_on_the_lds_dma_kernel(ops, M, K, N, splits):
    """asr_tap_gemm_nt_splitk (A [M][K], Bt [N][K]) against float64: bias + ReLU on out_a, affine on out_y; equal to asr_tap_gemm to
    rounding; bitwise reproducible; the rows of a one-utterance problem are bitwise the rows they are inside a batch."""
    rng = np.random.default_rng(16)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    sc = (1 + 0.2 * rng.standard_normal(N)).astype(np.float32); sh = (0.1 * rng.standard_normal(N)).astype(np.float32)
    d = ops.gemm_desc(M, K, N, K, N, N, N, ntaps=1, relu=1)
    a0, y0, a1, y1, a2 = (torch.zeros(M, N, device='cuda') for _ in range(5))
    ws = torch.zeros(ops.tap_gemm_nt_splitk_workspace(d, splits) // 4 + 4, device='cuda')
    wt = dev(np.ascontiguousarray(w.T))
    ops.tap_gemm(d, dev(x), dev(w), dev(b), dev(sc), dev(sh), a0, y0)
    ops.tap_gemm_nt_splitk(d, dev(x), wt, K, dev(b), dev(sc), dev(sh), a1, y1, splits, ws)
    ops.tap_gemm_nt_splitk(d, dev(x), wt, K, dev(b), None, None, a2, None, splits, ws)
    ref = np.maximum(x.astype(np.float64) @ w.astype(np.float64) + b, 0.0)
    report('split-K (LDS-DMA) a', a1.cpu().numpy(), ref, 2e-5)
    report('split-K (LDS-DMA) y', y1.cpu().numpy(), ref * sc + sh, 2e-5)
    assert (a1 - a0).abs().max().item() < 2e-5 * max(1.0, float(np.abs(ref).max()))
    assert torch.equal(a1, a2)
    if M > 256:
        m1 = 200 if M >= 400 else 100
        d1 = ops.gemm_desc(m1, K, N, K, N, N, N, ntaps=1, relu=1)
        a3 = torch.zeros(m1, N, device='cuda')
        ops.tap_gemm_nt_splitk(d1, dev(x[:m1]), wt, K, dev(b), None, None, a3, None, splits, ws)
        assert torch.equal(a3, a1[:m1]), 'rows must not depend on the batch around them'


def test_dense_split_k_argument_checks(ops):
    """asr_tap_gemm_nt_splitk refuses what it cannot run (split count, depth not a multiple of 32 x splits, narrow outputs, a pitch below
    the depth, a missing workspace) before anything is launched."""
    from asr_dfcnn_transformer_amd import _lib
    import ctypes as C_
    lib = _lib.load()
    M, K, N = 256, 1024, 128
    x = torch.zeros(M, K, device='cuda'); wt = torch.zeros(N, K, device='cuda'); y = torch.zeros(M, N, device='cuda')
    ws = torch.zeros(16 * M * N, device='cuda')
    p = lambda t: t.data_ptr()
    call = lambda d, splits, ldb=K, w=ws: lib.asr_tap_gemm_nt_splitk(C_.byref(d), p(x), p(wt), ldb, None, None, None, None, p(y), splits, p(w) if w is not None else None,
                                                                     4 * w.numel() if w is not None else 0, None)
    good = ops.gemm_desc(M, K, N, K, N, 0, N, ntaps=1)
    assert call(good, 4) == 0
    assert call(good, 1) == -1 and call(good, 17) == -1          # 2 <= splits <= 16
    assert call(good, 3) == -1                                   # 1024 % (32 * 3) != 0
    assert call(good, 4, ldb=K - 4) == -1
    assert call(good, 4, w=None) == -1
    assert call(good, 4, w=ws[:4 * M * N - 4]) == -1             # a slab smaller than splits x M x N floats
    assert call(good, 4, w=ws[:4 * M * N]) == 0
    assert call(ops.gemm_desc(M, K, 32, K, 32, 0, 32, ntaps=1), 4) == -1      # N < 64: the 64 x 64 tiles of asr_tap_gemm_splitk take those
    assert lib.asr_tap_gemm_nt_splitk_workspace(C_.byref(good), 4) == 4 * M * N * 4
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,K,N", [(300, 64, 128), (1000, 256, 32), (640, 6400, 1536)])
def test_dense_wgrad(ops, M, K, N):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((M, K)).astype(np.float32)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    d = ops.gemm_desc(M, K, N, K, N, ntaps=1)
    ws = torch.zeros(max(4, ops.tap_wgrad_workspace(d) // 4), device='cuda')
    dw = torch.zeros(K * N, device='cuda')
    ops.tap_wgrad(d, dev(x), dev(dy), N, dw, ws)
    report('dense wgrad', dw.cpu().numpy().reshape(K, N), x.astype(np.float64).T @ dy.astype(np.float64), 3e-5)


# ------------------------------------------------------------------ first cell
@pytest.mark.parametrize("pool", ['avg', 'max'])
@pytest.mark.parametrize("B,T,F,C", [(2, 16, 12, 32), (1, 33, 21, 8), (2, 24, 200, 32), (1, 35, 37, 32), (3, 18, 16, 32)])
def test_cell1_fwd_bwd(ops, pool, B, T, F, C):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((B, T, F, 1)).astype(np.float32)
    p = {'w': (rng.standard_normal((3, 3, 1, C)) * 0.3).astype(np.float32),
         'b': (rng.standard_normal(C) * 0.1).astype(np.float32),
         'gamma': (1 + 0.2 * rng.standard_normal(C)).astype(np.float32),
         'beta': (0.1 * rng.standard_normal(C)).astype(np.float32)}
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    out, cache = onn.cell_fwd(x.astype(np.float64), p64, pool)
    rs = 1.0 / math.sqrt(1.0 + onn.BN_EPS)
    sc, sh = dev(p['gamma'] * rs), dev(p['beta'])
    pm = 1 if pool == 'avg' else 2
    y = ops.Plane(B, T // 2, F // 2, C)
    xd = dev(x.reshape(B, T, F))
    ops.cell1_fwd(xd, dev(p['w']), dev(p['b']), sc, sh, pm, y)
    report('cell1 fwd ' + pool, plane_np(y), out, 2e-5)
    check_border_zero(y)
    dout = rng.standard_normal(out.shape).astype(np.float32)
    _, g = onn.cell_bwd(cache, p64, dout.astype(np.float64), pool)
    dyp = to_plane(ops, dout)
    dw, db, dsc, dsh = (torch.zeros(9 * C, device='cuda'), torch.zeros(C, device='cuda'),
                        torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda'))
    ws = torch.zeros(ops.cell1_bwd_workspace(B, T, F, C) // 4 + 4, device='cuda')
    ops.cell1_bwd(xd, dev(p['w']), dev(p['b']), sc, sh, pm, dyp, dw, db, dsc, dsh, ws)
    report('cell1 dw', dw.cpu().numpy().reshape(3, 3, 1, C), g['w'], 5e-5)
    report('cell1 db', db.cpu().numpy(), g['b'], 5e-5)
    report('cell1 dgamma', dsc.cpu().numpy() * rs, g['gamma'], 5e-5)
    report('cell1 dbeta', dsh.cpu().numpy(), g['beta'], 5e-5)


# ------------------------------------------------------------------ pool fwd + cell backward prologue
@pytest.mark.parametrize("pool", [None, 'avg', 'max'])
@pytest.mark.parametrize("B,H,W,C", [(2, 10, 8, 32), (1, 21, 25, 128), (3, 7, 5, 64)])
def test_pool_and_cell_bwd_pre(ops, pool, B, H, W, C):
    rng = np.random.default_rng(7)
    a = np.maximum(rng.standard_normal((B, H, W, C)), 0).astype(np.float32)     # post-ReLU
    gamma = (1 + 0.2 * rng.standard_normal(C)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(C)).astype(np.float32)
    rs = 1.0 / math.sqrt(1.0 + onn.BN_EPS)
    sc, sh = dev(gamma * rs), dev(beta)
    yfull = onn.bn_frozen(a.astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64))
    ap = to_plane(ops, a)
    pm = {None: 0, 'avg': 1, 'max': 2}[pool]
    if pool:
        yref = onn.avgpool2(yfull) if pool == 'avg' else onn.maxpool2(yfull)
        y = ops.Plane(B, H // 2, W // 2, C)
        ops.pool_fwd(ap, sc, sh, pm, y)
        report('pool fwd ' + pool, plane_np(y), yref, 1e-6)
        check_border_zero(y)
        dout = rng.standard_normal(yref.shape).astype(np.float32)
        dy_ref = (onn.avgpool2_bwd(yfull.shape, dout.astype(np.float64)) if pool == 'avg'
                  else onn.maxpool2_bwd(yfull, dout.astype(np.float64)))
        dyd, layout = to_plane(ops, dout), 1
    else:
        dout = rng.standard_normal(yfull.shape).astype(np.float32)
        dy_ref = dout.astype(np.float64)
        dyd, layout = to_plane(ops, dout), 0
    da, dg, dbt = onn.bn_frozen_bwd(a.astype(np.float64), gamma.astype(np.float64), dy_ref)
    dz_ref = da * (a > 0)
    dz = ops.Plane(B, H, W, C)
    dsc, dsh, dbias = (torch.zeros(C, device='cuda') for _ in range(3))
    ws = torch.zeros(ops.cell_bwd_pre_workspace(B, H, W, C) // 4 + 4, device='cuda')
    ops.cell_bwd_pre(dyd, layout, ap, sc, sh, pm, dz, dsc, dsh, dbias, ws)
    report('bwd_pre dz %s' % pool, plane_np(dz), dz_ref, 1e-6)
    check_border_zero(dz)
    report('bwd_pre dgamma', dsc.cpu().numpy() * rs, dg, 2e-5)
    report('bwd_pre dbeta', dsh.cpu().numpy(), dbt, 2e-5)
    report('bwd_pre dbias', dbias.cpu().numpy(), dz_ref.sum(axis=(0, 1, 2)), 2e-5)
    if not pool:     # unpadded dy (from the dense head)
        dz2 = ops.Plane(B, H, W, C)
        ops.cell_bwd_pre(dev(dout.reshape(-1, C)), 2, ap, sc, sh, 0, dz2, dsc, dsh, dbias, ws)
        report('bwd_pre dz unpadded-dy', plane_np(dz2), dz_ref, 1e-6)


# ------------------------------------------------------------------ SE
# (2, 200, 101, 16, 12): 40 split partials over sixteen thread groups, a hidden width that does not divide 256;
# (1, 9, 7, 512, 300): more channels / hidden units than threads (the one-thread-per-output paths of se_fold_partials / se_matvec)
@pytest.mark.parametrize("B,H,W,C,hid", [(2, 10, 8, 32, 32), (3, 50, 45, 64, 32), (2, 20, 25, 128, 64), (2, 200, 101, 16, 12),
                                         (1, 9, 7, 512, 300)])
def test_se_fwd_bwd(ops, B, H, W, C, hid):
    rng = np.random.default_rng(8)
    x = rng.standard_normal((B, H, W, C)).astype(np.float32)
    main = rng.standard_normal((B, H, W, C)).astype(np.float32)
    p = {'gamma': (1 + 0.2 * rng.standard_normal(C)), 'beta': 0.1 * rng.standard_normal(C),
         'w1': rng.standard_normal((C, hid)) * 0.3, 'b1': rng.standard_normal(hid) * 0.1,
         'w2': rng.standard_normal((hid, C)) * 0.3, 'b2': rng.standard_normal(C) * 0.1}
    p = {k: v.astype(np.float32) for k, v in p.items()}
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    s_ref, cache = onn.se_fwd(x.astype(np.float64), p64)
    rs = 1.0 / math.sqrt(1.0 + onn.BN_EPS)
    sc, sh = dev(p['gamma'] * rs), dev(p['beta'])
    xp, mp = to_plane(ops, x), to_plane(ops, main)
    out = ops.Plane(B, H, W, C)
    state = torch.zeros(ops.se_state_floats(B, C, hid), device='cuda')
    wsf = torch.zeros(ops.se_fwd_workspace(B, H, W, C) // 4 + 4, device='cuda')
    w1, b1, w2, b2 = dev(p['w1']), dev(p['b1']), dev(p['w2']), dev(p['b2'])
    ops.se_fwd(mp, xp, hid, sc, sh, w1, b1, w2, b2, state, wsf, out)
    report('se fwd', plane_np(out), main + s_ref, 2e-5)
    check_border_zero(out)
    dout = rng.standard_normal((B, H, W, C)).astype(np.float32)
    dx_ref, g = onn.se_bwd(cache, p64, dout.astype(np.float64))
    dop = to_plane(ops, dout)
    dx = ops.Plane(B, H, W, C)
    dsc, dsh = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    dw1, db1 = torch.zeros(C * hid, device='cuda'), torch.zeros(hid, device='cuda')
    dw2, db2 = torch.zeros(hid * C, device='cuda'), torch.zeros(C, device='cuda')
    wsb = torch.zeros(ops.se_bwd_workspace(B, H, W, C, hid) // 4 + 4, device='cuda')
    ops.se_bwd(dop, xp, hid, sc, sh, w1, w2, state, 0, dx, dsc, dsh, dw1, db1, dw2, db2, wsb)
    report('se dx', plane_np(dx), dx_ref, 2e-5)
    check_border_zero(dx)
    report('se dgamma', dsc.cpu().numpy() * rs, g['gamma'], 5e-5)
    report('se dbeta', dsh.cpu().numpy(), g['beta'], 5e-5)
    report('se dw1', dw1.cpu().numpy().reshape(C, hid), g['w1'], 5e-5)
    report('se db1', db1.cpu().numpy(), g['b1'], 5e-5)
    report('se dw2', dw2.cpu().numpy().reshape(hid, C), g['w2'], 5e-5)
    report('se db2', db2.cpu().numpy(), g['b2'], 5e-5)
    # asr_se_bwd_cell: the same with the backward prologue of the conv cell that produced x (a = its activations, x = BN(a)) fused in,
    # against asr_se_bwd + asr_cell_bwd_pre(pool 0) on the dx plane: dZ and every sum bit for bit, with and without add_dout
    a_np = np.maximum(rng.standard_normal((B, H, W, C)), 0).astype(np.float32)
    ap = to_plane(ops, a_np)
    csc, csh = dev(1 + 0.2 * rng.standard_normal(C)), dev(0.1 * rng.standard_normal(C))
    for add in (0, 1):
        ops.se_bwd(dop, xp, hid, sc, sh, w1, w2, state, add, dx, dsc, dsh, dw1, db1, dw2, db2, wsb)
        dz1 = ops.Plane(B, H, W, C)
        c1 = [torch.zeros(C, device='cuda') for _ in range(3)]
        wsp = torch.zeros(ops.cell_bwd_pre_workspace(B, H, W, C) // 4 + 64, device='cuda')
        ops.cell_bwd_pre(dx, 0, ap, csc, csh, 0, dz1, c1[0], c1[1], c1[2], wsp)
        want = [t.clone() for t in (dsc, dsh, dw1, db1, dw2, db2)]
        got = [torch.zeros_like(t) for t in want]
        dz2 = ops.Plane(B, H, W, C)
        c2 = [torch.zeros(C, device='cuda') for _ in range(3)]
        wsc = torch.zeros(ops.se_bwd_cell_workspace(B, H, W, C, hid) // 4 + 4, device='cuda')
        ops.se_bwd_cell(dop, xp, hid, sc, sh, w1, w2, state, add, got[0], got[1], got[2], got[3], got[4], got[5],
                        ap, csc, dz2, c2[0], c2[1], c2[2], wsc)
        assert torch.equal(dz1.buf, dz2.buf)
        check_border_zero(dz2)
        for u, v in zip(c1 + want, c2 + got):
            assert torch.equal(u, v)


# ------------------------------------------------------------------ head
@pytest.mark.parametrize("B,T,V", [(3, 7, 50), (4, 25, 1536), (2, 9, 1424)])
def test_softmax_log_fwd_bwd(ops, B, T, V):
    rng = np.random.default_rng(9)
    d = (rng.standard_normal((B, T, V)) * 3).astype(np.float32)
    out = torch.zeros(T, B, V, device='cuda')
    ops.softmax_log_fwd(dev(d), B, T, V, 1e-7, out)
    ref = onn.log_softmax_eps_tm(d.astype(np.float64))
    report('softmax_log fwd', out.cpu().numpy(), ref, 1e-5)
    g = rng.standard_normal((T, B, V)).astype(np.float32)
    dd = torch.zeros(B, T, V, device='cuda')
    ops.softmax_log_bwd(out, dev(g), B, T, V, 1e-7, 0.5, dd)
    report('softmax_log bwd', dd.cpu().numpy(), 0.5 * onn.log_softmax_eps_tm_bwd(d.astype(np.float64), g.astype(np.float64)), 2e-5)


def _ctc_inputs(rng, T, B, V, labels, seq):
    x = (rng.standard_normal((T, B, V)) * 2).astype(np.float32)
    max_label = 64
    lab = np.zeros((B, max_label), dtype=np.int32)
    for b, l in enumerate(labels):
        lab[b, :len(l)] = l
    return x, lab, np.array([len(l) for l in labels], dtype=np.int32), np.array(seq, dtype=np.int32), max_label


def test_ctc_loss_and_grad(ops):
    rng = np.random.default_rng(10)
    T, B, V = 40, 5, 97
    labels = [[1, 1, 2, 3], [5], [], list(rng.integers(0, V - 1, 18)), [7, 7, 7, 7, 7, 7]]
    seq = [40, 17, 9, 40, 11]
    x, lab, ll, sl, ml = _ctc_inputs(rng, T, B, V, labels, seq)
    loss_ref, g_ref = octc.ctc_loss_and_grad(x, labels, seq, V - 1)
    loss = torch.zeros(B, device='cuda')
    grad = torch.full((T, B, V), 7.0, device='cuda')
    status = torch.zeros(B, dtype=torch.int32, device='cuda')
    ws = torch.zeros(ops.ctc_workspace(T, B, ml) // 8 + 8, dtype=torch.float64, device='cuda')
    ops.ctc_loss(dev(x), T, B, V, dev(lab, torch.int32), ml, dev(ll, torch.int32), dev(sl, torch.int32), V - 1,
                 loss, grad, status, ws)
    assert status.cpu().tolist() == [0] * B
    report('ctc loss', loss.cpu().numpy(), loss_ref, 1e-6)
    report('ctc grad', grad.cpu().numpy(), g_ref, 2e-6)


def test_ctc_infeasible_and_full_width(ops):
    rng = np.random.default_rng(11)
    T, B, V = 200, 4, 1536
    labels = [list(rng.integers(1, V - 1, 64)), [3, 3], list(rng.integers(1, V - 1, 32)), [9]]
    seq = [200, 2, 125, 1]                     # utterance 1 needs 3 frames -> infeasible
    x, lab, ll, sl, ml = _ctc_inputs(rng, T, B, V, labels, seq)
    loss = torch.zeros(B, device='cuda')
    grad = torch.full((T, B, V), 7.0, device='cuda')
    status = torch.zeros(B, dtype=torch.int32, device='cuda')
    ws = torch.zeros(ops.ctc_workspace(T, B, ml) // 8 + 8, dtype=torch.float64, device='cuda')
    ops.ctc_loss(dev(x), T, B, V, dev(lab, torch.int32), ml, dev(ll, torch.int32), dev(sl, torch.int32), V - 1,
                 loss, grad, status, ws)
    assert status.cpu().tolist() == [0, 1, 0, 0]
    with pytest.raises(ValueError):
        octc.ctc_loss_and_grad(x[:, 1:2], [labels[1]], [2], V - 1)
    keep = [0, 2, 3]
    loss_ref, g_ref = octc.ctc_loss_and_grad(x[:, keep], [labels[i] for i in keep], [seq[i] for i in keep], V - 1)
    l = loss.cpu().numpy()
    assert np.isinf(l[1])
    report('ctc loss full', l[keep], loss_ref, 1e-6)
    g = grad.cpu().numpy()
    assert np.all(g[:, 1] == 0)
    report('ctc grad full', g[:, keep], g_ref, 2e-6)


def test_ctc_saturated_logits_fall_back_to_the_log_domain(ops):
    """A collapsed model: softmax saturated on the blank, every label at the eps floor -- log(p + 1e-7) = -16.1 per label
    emission (acoustic_model2.py:68).  With 64 (or 50) labels the states the likelihood needs sit e^-1000 below the frame's
    dominant state: the linear-domain lattice of the kernel flushes them to zero in float64; it must notice (tail == 0, or the
    occupancies of a frame not summing to 1) and redo those utterances in the log domain.  tf.nn.ctc_loss returns a finite
    loss here, and so does the oracle; before round 3 the kernel returned +inf and NaN gradients.  Utterance 2 (3 labels) and
    utterance 3 (ordinary random logits) stay on the fast path in the same launch."""
    rng = np.random.default_rng(21)
    T, B, V = 200, 4, 1536
    labels = [list(rng.permutation(np.arange(1, V - 1))[:64]), [int(v) for v in rng.integers(1, 40, 50)], [4, 9, 4],
              list(rng.integers(1, V - 1, 32))]
    seq = [200, 200, 200, 125]
    x, lab, ll, sl, ml = _ctc_inputs(rng, T, B, V, labels, seq)
    p = np.zeros((T, 3, V)); p[:, :, V - 1] = 1.0
    x[:, :3] = np.log(p + 1e-7).astype(np.float32)
    loss_ref, g_ref = octc.ctc_loss_and_grad(x, labels, seq, V - 1)
    assert np.all(np.isfinite(loss_ref)) and loss_ref[0] > 900 and loss_ref[1] > 650
    loss = torch.zeros(B, device='cuda')
    grad = torch.full((T, B, V), 7.0, device='cuda')
    status = torch.zeros(B, dtype=torch.int32, device='cuda')
    ws = torch.zeros(ops.ctc_workspace(T, B, ml) // 8 + 8, dtype=torch.float64, device='cuda')
    ops.ctc_loss(dev(x), T, B, V, dev(lab, torch.int32), ml, dev(ll, torch.int32), dev(sl, torch.int32), V - 1,
                 loss, grad, status, ws)
    assert status.cpu().tolist() == [0] * B
    l, g = loss.cpu().numpy(), grad.cpu().numpy()
    assert np.all(np.isfinite(l)) and np.all(np.isfinite(g))
    report('ctc loss saturated', l, loss_ref, 1e-6)
    report('ctc grad saturated', g, g_ref, 2e-6)
    # rows of the gradient w.r.t. the logits sum to 0 (softmax minus a distribution) wherever the utterance has frames
    assert np.abs(g[:125].sum(axis=2)).max() < 1e-4


def test_ctc_fallback_from_the_lds_resident_form(ops):
    """Round 5: an utterance whose lattice fits keeps its emission probabilities in LDS (Tb x (S rounded up to 4, + 4) doubles within
    the launch's budget: 125 frames x 50 labels here), the others read them from the workspace (200 frames x 64 labels).  Both forms must
    fall back to the log domain when the linear recursion loses states (saturated softmax), the LDS-resident one by gathering the float
    inputs into the space its probabilities held; repeated labels, an empty label row and a one-frame utterance ride along."""
    rng = np.random.default_rng(22)
    T, B, V = 200, 6, 1536
    labels = [[int(v) for v in rng.integers(1, 40, 50)], list(rng.permutation(np.arange(1, V - 1))[:64]), [4, 4, 9, 4, 4], [], [11],
              list(rng.integers(1, V - 1, 32))]
    seq = [125, 200, 125, 60, 1, 125]
    x, lab, ll, sl, ml = _ctc_inputs(rng, T, B, V, labels, seq)
    p = np.zeros((T, 2, V)); p[:, :, V - 1] = 1.0
    x[:, :2] = np.log(p + 1e-7).astype(np.float32)
    loss_ref, g_ref = octc.ctc_loss_and_grad(x, labels, seq, V - 1)
    assert np.all(np.isfinite(loss_ref)) and loss_ref[0] > 650
    loss = torch.zeros(B, device='cuda')
    grad = torch.full((T, B, V), 7.0, device='cuda')
    status = torch.zeros(B, dtype=torch.int32, device='cuda')
    ws = torch.zeros(ops.ctc_workspace(T, B, ml) // 8 + 8, dtype=torch.float64, device='cuda')
    for _ in range(2):                         # twice: the workspace of the first call is the second call's starting state
        ops.ctc_loss(dev(x), T, B, V, dev(lab, torch.int32), ml, dev(ll, torch.int32), dev(sl, torch.int32), V - 1,
                     loss, grad, status, ws)
    assert status.cpu().tolist() == [0] * B
    l, g = loss.cpu().numpy(), grad.cpu().numpy()
    assert np.all(np.isfinite(l)) and np.all(np.isfinite(g))
    report('ctc loss (LDS form + fallback)', l, loss_ref, 1e-6)
    report('ctc grad (LDS form + fallback)', g, g_ref, 2e-6)


@pytest.mark.parametrize("seed,T,V", [(101, 200, 1536), (102, 97, 50), (103, 200, 333)])
def test_ctc_random_batches(ops, seed, T, V):
    """Twenty-four utterances of random length, label count (0 .. 64) and label repetition per launch, against the float64 oracle: short and
    long lattices in one batch take different forms inside the kernel (two or four states per lane, probabilities in LDS or in the
    workspace), utterances at the edge of feasibility (frames = labels + repeats) and single-frame ones included."""
    rng = np.random.default_rng(seed)
    B = 24
    labels, seq = [], []
    for b in range(B):
        L = int(rng.integers(0, 65)) if b % 3 else int(rng.choice([0, 1, 2, 31, 32, 33, 63, 64]))
        L = min(L, T // 2)
        alphabet = int(rng.choice([2, 5, V - 1]))                  # few symbols: many repeats
        lab = [int(v) for v in rng.integers(0, min(alphabet, V - 1), L)]
        rep = sum(1 for i in range(1, L) if lab[i] == lab[i - 1])
        need = max(1, L + rep)
        s_ = need if b % 4 == 0 else int(rng.integers(need, T + 1))
        labels.append(lab); seq.append(min(s_, T))
    x, lab, ll, sl, ml = _ctc_inputs(rng, T, B, V, labels, seq)
    loss_ref, g_ref = octc.ctc_loss_and_grad(x, labels, seq, V - 1)
    loss = torch.zeros(B, device='cuda')
    grad = torch.full((T, B, V), 7.0, device='cuda')
    status = torch.zeros(B, dtype=torch.int32, device='cuda')
    ws = torch.zeros(ops.ctc_workspace(T, B, ml) // 8 + 8, dtype=torch.float64, device='cuda')
    ops.ctc_loss(dev(x), T, B, V, dev(lab, torch.int32), ml, dev(ll, torch.int32), dev(sl, torch.int32), V - 1,
                 loss, grad, status, ws)
    assert status.cpu().tolist() == [0] * B
    report('ctc loss, random batch', loss.cpu().numpy(), loss_ref, 1e-6)
    report('ctc grad, random batch', grad.cpu().numpy(), g_ref, 2e-6)
    g2 = torch.full((T, B, V), 3.0, device='cuda'); l2 = torch.zeros(B, device='cuda')
    ops.ctc_loss(dev(x), T, B, V, dev(lab, torch.int32), ml, dev(ll, torch.int32), dev(sl, torch.int32), V - 1, l2, g2, status, ws)
    assert torch.equal(g2, grad) and torch.equal(l2, loss), 'asr_ctc_loss must be bitwise reproducible'


def test_greedy_decode_bit_exact(ops):
    rng = np.random.default_rng(12)
    T, B, V = 60, 6, 1536
    x = rng.standard_normal((T, B, V)).astype(np.float32)
    # make runs, blanks and exact ties
    for b in range(B):
        for t in range(T):
            k = [5, 5, V - 1, 5, 9, 9, V - 1, V - 1, 0, 3][(t + b) % 10]
            x[t, b, k] = 20.0
    x[3, 0, 2] = 20.0          # tie between 2 and a later index -> lowest index wins
    seq = [60, 33, 1, 60, 17, 0]
    dec_ref, neg_ref = octc.ctc_greedy_decode(x, seq)
    ids = torch.zeros(B, T, dtype=torch.int32, device='cuda')
    n = torch.zeros(B, dtype=torch.int32, device='cuda')
    neg = torch.zeros(B, device='cuda')
    ops.ctc_greedy(dev(x), T, B, V, dev(np.array(seq), torch.int32), V - 1, ids, n, neg,
                   torch.zeros(ops.ctc_greedy_workspace(T, B) // 4 + 4, dtype=torch.int32, device='cuda'))
    ids, n = ids.cpu().numpy(), n.cpu().numpy()
    for b in range(B):
        assert n[b] == len(dec_ref[b])
        assert ids[b, :n[b]].tolist() == dec_ref[b]
        assert np.all(ids[b, n[b]:] == -1)
    report('greedy neg_sum_logits', neg.cpu().numpy(), neg_ref, 1e-5)


def test_edit_distance(ops):
    rng = np.random.default_rng(13)
    B, HP, TP = 9, 200, 64
    hyp = np.full((B, HP), -1, dtype=np.int32)
    tru = np.zeros((B, TP), dtype=np.int32)
    hl, tl = [], []
    pairs = []
    for b in range(B):
        n, m = [(0, 0), (5, 0), (0, 4), (64, 64), (1, 200), (64, 200), (30, 31), (17, 5), (40, 40)][b]
        t = rng.integers(1, 6, n)
        h = rng.integers(1, 6, m)
        if b == 8:
            h = t.copy()
        tru[b, :n], hyp[b, :m] = t, h
        hl.append(m), tl.append(n)
        pairs.append((list(h), list(t)))
    dist = torch.zeros(B, device='cuda')
    ops.edit_distance(dev(hyp, torch.int32), HP, dev(np.array(hl), torch.int32), dev(tru, torch.int32), TP,
                      dev(np.array(tl), torch.int32), B, dist)
    got = dist.cpu().numpy()
    for b, (h, t) in enumerate(pairs):
        want = octc.edit_distance_normalized(h, t)
        assert (np.isinf(want) and np.isinf(got[b])) or abs(got[b] - want) < 1e-6, (b, got[b], want)


def test_adam_tf(ops):
    rng = np.random.default_rng(14)
    n = 100003
    th, g = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    m, v = (rng.standard_normal(n) * 0.1).astype(np.float32), (rng.random(n) * 0.1).astype(np.float32)
    t, lr = 7, 7e-4
    lr_t = lr * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
    th_ref, m_ref, v_ref = oopt.adam_tf_step(th.astype(np.float64), 0.5 * g.astype(np.float64), m.astype(np.float64),
                                             v.astype(np.float64), lr, t)
    dth, dm, dv = dev(th), dev(m), dev(v)
    ops.adam_tf(dth, dev(g), dm, dv, lr_t, 0.9, 0.999, 1e-8, 0.5)
    report('adam theta', dth.cpu().numpy(), th_ref, 1e-6)
    report('adam m', dm.cpu().numpy(), m_ref, 1e-6)
    report('adam v', dv.cpu().numpy(), v_ref, 1e-6)


def test_colsum_and_relu_bwd(ops):
    rng = np.random.default_rng(15)
    x = rng.standard_normal((1000, 300)).astype(np.float32)
    out = torch.zeros(300, device='cuda')
    ws = torch.zeros(ops.colsum_workspace(1000, 300) // 4 + 4, device='cuda')
    ops.colsum(dev(x), 1000, 300, 300, out, ws)
    report('colsum', out.cpu().numpy(), x.astype(np.float64).sum(axis=0), 2e-5)
    h = np.maximum(rng.standard_normal(5000), 0).astype(np.float32)
    dy = rng.standard_normal(5000).astype(np.float32)
    dz = torch.zeros(5000, device='cuda')
    ops.relu_bwd(dev(dy), dev(h), dz)
    assert np.array_equal(dz.cpu().numpy(), dy * (h > 0))


@pytest.mark.parametrize("rows,cols,ld", [(6400, 1536, 1536), (4096, 6348, 6348), (1024, 64, 64), (1500, 68, 72), (20000, 2048, 2048),
                                          (32768, 512, 512), (70, 1000, 1000), (3000, 300, 300), (1030, 132, 136)])
def test_colsum_forms(ops, rows, cols, ld):
    """asr_colsum picks one of three forms by shape -- tall (float4 rows, >= 16384 rows), strips of 64 columns x row splits (round 5: wide
    matrices of a few thousand rows, a last strip may be partial), a column per thread -- each against float64 and bitwise reproducible;
    a row pitch larger than the width is honoured."""
    rng = np.random.default_rng(31)
    x = rng.standard_normal((rows, ld)).astype(np.float32)
    out = torch.full((cols,), 7.0, device='cuda'); out2 = torch.zeros(cols, device='cuda')
    ws = torch.zeros(ops.colsum_workspace(rows, cols) // 4 + 4, device='cuda')
    xd = dev(x)
    ops.colsum(xd, rows, cols, ld, out, ws)
    ops.colsum(xd, rows, cols, ld, out2, ws)
    report('colsum %d x %d' % (rows, cols), out.cpu().numpy(), x[:, :cols].astype(np.float64).sum(axis=0), 2e-5 * math.sqrt(rows / 1000.0 + 1))
    assert torch.equal(out, out2)


# ------------------------------------------------------------------ fbank
def test_fbank_matches_oracle():
    from asr_dfcnn_transformer_amd import wav_util
    rng = np.random.default_rng(1234)
    B = 3
    lens = [16000, 7321, 300]
    sig = np.zeros((B, 16000), dtype=np.float32)
    for b in range(B):
        sig[b, :lens[b]] = (0.1 * rng.standard_normal(lens[b])).astype(np.float32)
    sig[1, :lens[1]] += (0.3 * np.sin(2 * np.pi * 1000 * np.arange(lens[1]) / 16000)).astype(np.float32)
    ex = wav_util.FbankExtractor()
    feat, frames = ex.batch(dev(sig), dev(np.array(lens), torch.int32), 128)
    feat, frames = feat.cpu().numpy(), frames.cpu().numpy()
    for b in range(B):
        ref = ofb.compute_fbank_from_api(sig[b, :lens[b]].astype(np.float64), 16000, nfilt=200)
        T = ref.shape[0]
        assert frames[b] == T
        report('fbank utt %d' % b, feat[b, :T], ref.astype(np.float32), 2e-6)
        assert np.all(feat[b, T:] == 0)
        fb = ofb.get_filterbanks(200, 512, 16000)
        empty = np.where(fb.sum(axis=1) == 0)[0]
        assert np.all(feat[b, :T][:, empty] == 0)
    one = wav_util.compute_fbank_from_api(np.zeros(16000), 16000)
    assert one.shape == (99, 200) and np.all(one == 0)
