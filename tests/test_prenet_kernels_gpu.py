"""GPU parity of the end2end pre-net kernels (end2end/model.py:214-264) against oracle/prenet.py (torch float64 +
autograd), op by op, through the C ABI.  fp32 tolerances are written per check (relative to the tensor's scale)."""
import math

import numpy as np
import pytest
import torch

from oracle import prenet as opn

pytestmark = pytest.mark.gpu


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device='cuda')


def t64(a, grad=False):
    return torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=grad)


def report(name, got, want, tol):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = want.detach().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    got, want = got.astype(np.float64), want.astype(np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = np.abs(got - want).max()
    scale = max(1e-6, np.abs(want).max())
    print('%-40s max|err| %.3e (scale %.2e, tol %.0e)' % (name, err, scale, tol))
    assert err <= tol * max(1.0, scale), name


@pytest.fixture(scope='module')
def ops():
    from asr_dfcnn_transformer_amd import ops as _ops
    return _ops


def test_conv1_stride2_tanh(ops):
    rng = np.random.default_rng(0)
    B, T, F = 2, 12, 20
    x = rng.standard_normal((B, T, F)); w = rng.standard_normal((3, 3, 1, 64)) * 0.3; b = rng.standard_normal(64) * 0.1
    wt, bt = t64(w, True), t64(b, True)
    z = opn.conv(t64(x).unsqueeze(-1), wt, bt, 2)
    ref = torch.tanh(z)
    a1 = torch.zeros(B, T // 2, F // 2, 64, device='cuda')
    ops.prenet_conv1_fwd(dev(x), dev(w), dev(b), a1)
    report('conv1 fwd', a1, ref, 2e-6)
    dz = rng.standard_normal(z.shape)
    z.backward(t64(dz))
    dw, db = torch.zeros(3, 3, 1, 64, device='cuda'), torch.zeros(64, device='cuda')
    ws = torch.zeros(ops.prenet_conv1_bwd_workspace(B, T, F) // 4 + 16, device='cuda')
    ops.prenet_conv1_bwd(dev(x), dev(dz), dw, db, ws)
    report('conv1 dw', dw, wt.grad, 2e-6)
    report('conv1 db', db, bt.grad, 2e-6)


@pytest.mark.parametrize("B,H,W", [(2, 8, 12), (1, 20, 160)])
def test_conv_stride2_as_phase_split_taps(ops, B, H, W):
    """64 -> 64 stride-2 conv + tanh = 2x2-tap tap-GEMM over the phase-split plane; forward, dX, dW, db."""
    rng = np.random.default_rng(1)
    Cc = 64
    x = rng.standard_normal((B, H, W, Cc)); w = rng.standard_normal((3, 3, Cc, Cc)) * 0.05; b = rng.standard_normal(Cc) * 0.1
    xt, wt, bt = t64(x, True), t64(w, True), t64(b, True)
    z = opn.conv(xt, wt, bt, 2)
    ref = torch.tanh(z)
    H2, W2 = H // 2, W // 2
    xs = ops.Plane(B, H2, W2, 4 * Cc)
    v = xs.interior().view(B, H2, W2, 2, 2, Cc)                    # [b][i][j][ph][pw][c]
    v.copy_(dev(x).view(B, H2, 2, W2, 2, Cc).permute(0, 1, 3, 2, 4, 5))
    W4 = torch.zeros(4, 4 * Cc, Cc, device='cuda')
    ops.conv_s2_expand(dev(w), Cc, Cc, W4)
    out = ops.Plane(B, H2, W2, Cc)
    d = ops.gemm_desc(xs.NP, 4 * Cc, Cc, 4 * Cc, Cc, Cc, 0, ntaps=4, B=B, H=H2, W=W2, relu=2)
    ops.tap_gemm(d, xs, W4, dev(b), None, None, out, None)
    report('conv s2 fwd (tanh)', out.interior(), ref, 3e-6)
    assert float(out.view()[:, 0].abs().max()) == 0 and float(out.view()[:, :, 0].abs().max()) == 0
    dz = rng.standard_normal(z.shape)
    z.backward(t64(dz))
    dzp = ops.Plane(B, H2, W2, Cc)
    dzp.set_interior(dev(dz))
    # weight gradient (4 taps x 256 x 64) folded back to HWIO
    wd = ops.gemm_desc(xs.NP, 4 * Cc, Cc, 4 * Cc, Cc, ntaps=4, B=B, H=H2, W=W2)
    ws = torch.zeros(ops.tap_wgrad_workspace(wd) // 4 + 16, device='cuda')
    dW4 = torch.zeros_like(W4)
    ops.tap_wgrad(wd, xs, dzp, Cc, dW4, ws)
    dw = torch.zeros(3, 3, Cc, Cc, device='cuda')
    ops.conv_s2_gather(dW4, Cc, Cc, dw)
    report('conv s2 dW', dw, wt.grad, 1e-5)
    # data gradient into the phase-split plane
    dxs = ops.Plane(B, H2, W2, 4 * Cc)
    bd = ops.gemm_desc(xs.NP, Cc, 4 * Cc, Cc, Cc, 0, 4 * Cc, ntaps=4, B=B, H=H2, W=W2, wmode=1)
    ops.tap_gemm(bd, dzp, W4, None, None, None, None, dxs)
    got = dxs.interior().view(B, H2, W2, 2, 2, Cc).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, Cc)
    report('conv s2 dX', got, xt.grad, 1e-5)
    # round 6: the same data-gradient as one launch per phase of the input gradient on fragment-order weights (asr_conv_s2_dgrad)
    Wf9 = torch.zeros(ops.conv_s2_arrange_bytes(Cc) // 4, device='cuda')
    ops.conv_s2_arrange(W4, Cc, Wf9)
    dxs2 = ops.Plane(B, H2, W2, 4 * Cc)
    ops.conv_s2_dgrad(bd, dzp, Wf9, dxs2)
    got2 = dxs2.interior().view(B, H2, W2, 2, 2, Cc).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, Cc)
    report('conv s2 dX per phase', got2, xt.grad, 1e-5)
    assert float(dxs2.view()[:, 0].abs().max()) == 0 and float(dxs2.view()[:, :, 0].abs().max()) == 0       # borders stay zero
    from asr_dfcnn_transformer_amd import _lib
    lib = _lib.load()
    assert lib.asr_conv_s2_arrange(W4.data_ptr(), Cc, Wf9.data_ptr(), 4 * Wf9.numel() - 4, None) == -1      # a short buffer is refused before the launch
    assert lib.asr_conv_s2_arrange_bytes(48) == 0


@pytest.mark.parametrize("act", [0, 1, 2])
def test_batch_stat_bn(ops, act):
    rng = np.random.default_rng(2 + act)
    B, H, W, Cc = 2, 6, 10, 64
    z = rng.standard_normal((B, H, W, Cc)) * 1.5 + 0.3
    g = 1 + 0.2 * rng.standard_normal(Cc); be = 0.1 * rng.standard_normal(Cc)
    res = rng.standard_normal((B, H, W, Cc))
    zt, gt, bt = t64(z, True), t64(g, True), t64(be, True)
    a = zt if act == 0 else torch.relu(zt) if act == 1 else torch.tanh(zt)
    y = opn.batch_norm(a, gt, bt)
    ap = ops.Plane(B, H, W, Cc)
    ap.set_interior(dev(a.detach().numpy()))
    mean, rstd = torch.zeros(Cc, device='cuda'), torch.zeros(Cc, device='cuda')
    ws = torch.zeros(ops.bn_workspace(ap) // 4 + 16, device='cuda')
    ops.bn_stats(ap, opn.BN_EPS, mean, rstd, ws)
    an = a.detach().numpy()
    report('bn mean', mean, an.mean(axis=(0, 1, 2)), 1e-6)
    report('bn rstd', rstd, 1 / np.sqrt(an.var(axis=(0, 1, 2)) + opn.BN_EPS), 2e-6)
    yp = ops.Plane(B, H, W, Cc)
    ops.bn_apply(ap, mean, rstd, dev(g), dev(be), yp)
    report('bn apply', yp.interior(), y, 3e-6)
    # fused closing op: relu(bn(a) + res) into a plain tensor
    rp = ops.Plane(B, H, W, Cc); rp.set_interior(dev(res))
    out = torch.zeros(B, H, W, Cc, device='cuda')
    ops.bn_apply(ap, mean, rstd, dev(g), dev(be), out, res=rp, relu=True)
    report('relu(bn + res)', out, torch.relu(y + t64(res)), 3e-6)
    dy = rng.standard_normal(y.shape)
    y.backward(t64(dy))
    dyp = ops.Plane(B, H, W, Cc); dyp.set_interior(dev(dy))
    dzp = ops.Plane(B, H, W, Cc)
    dg, db = torch.zeros(Cc, device='cuda'), torch.zeros(Cc, device='cuda')
    ops.bn_bwd(dyp, ap, mean, rstd, dev(g), act, dzp, dg, db, ws)
    report('bn dz (act %d)' % act, dzp.interior(), zt.grad, 1e-5)
    report('bn dgamma', dg, gt.grad, 1e-5)
    report('bn dbeta', db, bt.grad, 1e-5)
    assert float(dzp.view()[:, 0].abs().max()) == 0


def test_bn_phase_split_roundtrip(ops):
    """BN(a1) written straight into the phase-split plane the stride-2 conv reads, and its backward reading dy from it."""
    rng = np.random.default_rng(5)
    B, H, W, Cc = 2, 4, 8, 64
    a = np.tanh(rng.standard_normal((B, H, W, Cc)))
    g = 1 + 0.2 * rng.standard_normal(Cc); be = 0.1 * rng.standard_normal(Cc)
    at, gt, bt = t64(a, True), t64(g, True), t64(be, True)
    y = opn.batch_norm(at, gt, bt)
    a_d = dev(a)
    mean, rstd = torch.zeros(Cc, device='cuda'), torch.zeros(Cc, device='cuda')
    ws = torch.zeros(ops.bn_workspace(a_d) // 4 + 16, device='cuda')
    ops.bn_stats(a_d, opn.BN_EPS, mean, rstd, ws)
    xs = ops.Plane(B, H // 2, W // 2, 4 * Cc)
    ops.bn_apply(a_d, mean, rstd, dev(g), dev(be), xs, dst_phase_split=True)
    got = xs.interior().view(B, H // 2, W // 2, 2, 2, Cc).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, Cc)
    report('bn -> phase split', got, y, 3e-6)
    dy = rng.standard_normal(y.shape)
    y.backward(t64(dy))
    dys = ops.Plane(B, H // 2, W // 2, 4 * Cc)
    dys.interior().view(B, H // 2, W // 2, 2, 2, Cc).copy_(dev(dy).view(B, H // 2, 2, W // 2, 2, Cc).permute(0, 1, 3, 2, 4, 5))
    dz = torch.zeros(B, H, W, Cc, device='cuda')
    dg, db = torch.zeros(Cc, device='cuda'), torch.zeros(Cc, device='cuda')
    ops.bn_bwd(dys, a_d, mean, rstd, dev(g), 0, dz, dg, db, ws, dy_phase_split=True)
    report('bn bwd from phase split', dz, at.grad, 1e-5)


@pytest.mark.parametrize("B,T", [(1, 8), (2, 200)])
def test_time_and_freq_attention(ops, B, T):
    rng = np.random.default_rng(6)
    Wd, Cc = 80, 64
    q, k, v = [rng.standard_normal((B, T, Wd, Cc)) for _ in range(3)]
    qt, kt, vt = t64(q, True), t64(k, True), t64(v, True)
    at = opn.attention(*[z.permute(0, 3, 1, 2) for z in (qt, kt, vt)]).permute(0, 2, 3, 1)
    af = opn.attention(*[z.permute(0, 3, 2, 1) for z in (qt, kt, vt)]).permute(0, 3, 2, 1)
    planes = []
    for z in (q, k, v):
        p = ops.Plane(B, T, Wd, Cc); p.set_interior(dev(z)); planes.append(p)
    QT, KT, VT = [torch.zeros(B, T, Cc, Wd, device='cuda') for _ in range(3)]
    for p, d in zip(planes, (QT, KT, VT)):
        ops.plane_to_T(p, 0, d)
    report('plane -> T', QT, np.transpose(q, (0, 1, 3, 2)).astype(np.float32), 0)
    OT, OF = torch.zeros_like(QT), torch.zeros_like(QT)
    lse = torch.zeros(B, Cc, T, device='cuda')
    ops.attention_nomask_fwd(QT, KT, VT, B, T, T, Cc * Wd, Cc, OT, lse)
    P = torch.zeros(B, Cc, Wd, Wd, device='cuda')
    ops.freq_attention_fwd(QT, KT, VT, B, T, P, OF)
    cat = ops.Plane(B, T, Wd, 2 * Cc)
    ops.T_to_plane(OT, None, cat, 0)
    ops.T_to_plane(OF, None, cat, Cc)
    report('time attention fwd', cat.interior()[..., :Cc], at, 1e-5)
    report('freq attention fwd', cat.interior()[..., Cc:], af, 1e-5)
    dat, daf = rng.standard_normal(at.shape), rng.standard_normal(af.shape)
    (at * t64(dat)).sum().backward(retain_graph=True)
    gt = [z.grad.clone() for z in (qt, kt, vt)]
    for z in (qt, kt, vt):
        z.grad = None
    (af * t64(daf)).sum().backward()
    gf = [z.grad.clone() for z in (qt, kt, vt)]
    dOT, dOF = dev(np.transpose(dat, (0, 1, 3, 2))), dev(np.transpose(daf, (0, 1, 3, 2)))
    dq1, dk1, dv1, dq2, dk2, dv2 = [torch.zeros_like(QT) for _ in range(6)]
    ws = torch.zeros(B * Cc * T + 64, device='cuda')
    ops.attention_nomask_bwd(QT, KT, VT, OT, dOT, lse, B, T, T, Cc * Wd, Cc, dq1, dk1, dv1, ws)
    dS = torch.zeros_like(P)
    ops.freq_attention_bwd(QT, KT, VT, P, dOF, B, T, dq2, dk2, dv2, dS)
    for name, g1, g2, r1, r2 in (('dq', dq1, dq2, gt[0], gf[0]), ('dk', dk1, dk2, gt[1], gf[1]), ('dv', dv1, dv2, gt[2], gf[2])):
        report('time attention ' + name, g1.permute(0, 1, 3, 2), r1, 2e-5)
        report('freq attention ' + name, g2.permute(0, 1, 3, 2), r2, 2e-5)
        back = ops.Plane(B, T, Wd, Cc)
        ops.T_to_plane(g1, g2, back, 0)
        report('T -> plane (sum) ' + name, back.interior(), r1 + r2, 3e-5)


def test_pix_add_layernorm(ops):
    rng = np.random.default_rng(7)
    B, H, W, Cc = 2, 5, 7, 64
    a, r = rng.standard_normal((B, H, W, Cc)), rng.standard_normal((B, H, W, Cc))
    g = 1 + 0.2 * rng.standard_normal(Cc); be = 0.1 * rng.standard_normal(Cc)
    at, rt, gt, bt = t64(a, True), t64(r, True), t64(g, True), t64(be, True)
    y = opn.layer_norm(at + rt, gt, bt)
    ap, rp, yp, xh = [ops.Plane(B, H, W, Cc) for _ in range(4)]
    ap.set_interior(dev(a)); rp.set_interior(dev(r))
    rstd = torch.zeros(B * H * W, device='cuda')
    ops.pix_add_ln_fwd(ap, rp, dev(g), dev(be), opn.LN_EPS, yp, xh, rstd)
    report('add + LN fwd', yp.interior(), y, 3e-6)
    assert float(yp.view()[:, 0].abs().max()) == 0 and float(yp.view()[:, :, 0].abs().max()) == 0
    dy = rng.standard_normal(y.shape)
    y.backward(t64(dy))
    dyp, dxp = ops.Plane(B, H, W, Cc), ops.Plane(B, H, W, Cc)
    dyp.set_interior(dev(dy))
    dg, db = torch.zeros(Cc, device='cuda'), torch.zeros(Cc, device='cuda')
    ws = torch.zeros(ops.pix_ln_bwd_workspace(dyp) // 4 + 16, device='cuda')
    ops.pix_ln_bwd(dyp, xh, rstd, dev(g), dxp, dg, db, ws)
    report('LN dx', dxp.interior(), at.grad, 1e-5)
    report('LN dgamma', dg, gt.grad, 1e-5)
    report('LN dbeta', db, bt.grad, 1e-5)


def test_relu_mask(ops):
    rng = np.random.default_rng(8)
    B, H, W, Cc = 2, 3, 5, 64
    dy, y = rng.standard_normal((B, H, W, Cc)), rng.standard_normal((B, H, W, Cc))
    dst = ops.Plane(B, H, W, Cc)
    ops.relu_mask(dev(dy), dev(y), dst)
    report('relu mask', dst.interior(), (dy * (y > 0)).astype(np.float32), 0)
