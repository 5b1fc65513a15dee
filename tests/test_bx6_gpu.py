"""GPU parity of the EXPERIMENTAL split-bf16 contraction kernels (asr_tap_gemm_bx6 / asr_tap_wgrad_bx6, DESIGN.md
section 9) against float64: the same bars as the fp32 kernels (outputs 1e-3 abs, here observed ~1e-5), plus a whole
DFCNN training step with ASR_BX6=1 against the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def conv_ref(x, w, b=None):
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None if b is None else b.double(), padding=1)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 9, 7, 20, 24), (1, 40, 25, 64, 64), (2, 16, 12, 32, 136), (1, 33, 50, 128, 32)])
def test_split_bf16_conv_forward_and_data_gradient(B, H, W, cin, cout):
    from asr_dfcnn_transformer_amd import ops
    g = torch.Generator(device='cuda').manual_seed(0)
    x = ops.Plane(B, H, W, cin); x.set_interior(torch.randn(B, H, W, cin, device='cuda', generator=g))
    w = torch.randn(3, 3, cin, cout, device='cuda', generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device='cuda', generator=g) * 0.1
    y = ops.Plane(B, H, W, cout)
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, 0, ntaps=9, B=B, H=H, W=W, relu=1)
    ops.tap_gemm_bx6(d, x, ops.split_weights(w, 9, cin, cout, cout, 0), bias, None, None, y, None)
    ref = torch.relu(conv_ref(x.interior(), w, bias))
    err = (y.interior().double() - ref).abs().max().item()
    print('bx6 conv fwd err %.2e' % err)
    assert err < 1e-4
    assert float(y.view()[:, 0].abs().max()) == 0 and float(y.view()[:, :, 0].abs().max()) == 0      # borders untouched
    # data gradient = conv of dz with the mirrored, transposed weights; accumulate on top of an existing gradient
    dz = ops.Plane(B, H, W, cout); dz.set_interior(torch.randn(B, H, W, cout, device='cuda', generator=g))
    dx = ops.Plane(B, H, W, cin); dx.set_interior(torch.ones(B, H, W, cin, device='cuda'))
    bd = ops.gemm_desc(x.NP, cout, cin, cout, cout, 0, cin, ntaps=9, B=B, H=H, W=W, wmode=1, accumulate=1)
    ops.tap_gemm_bx6(bd, dz, ops.split_weights(w, 9, cout, cin, cout, 1), None, None, None, None, dx, dgrad=True)
    wt = w.flip(0, 1).permute(0, 1, 3, 2).contiguous()           # [3][3][cout][cin], taps mirrored
    ref = conv_ref(dz.interior(), wt) + 1.0
    err = (dx.interior().double() - ref).abs().max().item()
    print('bx6 conv dgrad err %.2e' % err)
    assert err < 1e-4


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 20, 25, 32, 128), (1, 37, 11, 48, 136), (3, 64, 25, 128, 256)])
def test_split_bf16_weight_gradient(B, H, W, cin, cout):
    from asr_dfcnn_transformer_amd import ops
    g = torch.Generator(device='cuda').manual_seed(1)
    x = ops.Plane(B, H, W, cin); x.set_interior(torch.randn(B, H, W, cin, device='cuda', generator=g))
    dz = ops.Plane(B, H, W, cout); dz.set_interior(torch.randn(B, H, W, cout, device='cuda', generator=g))
    wd = ops.gemm_desc(x.NP, cin, cout, cin, cout, ntaps=9, B=B, H=H, W=W)
    ws = torch.zeros(ops.tap_wgrad_workspace(wd) // 4 + 16, device='cuda')
    got = torch.zeros(3, 3, cin, cout, device='cuda')
    ops.tap_wgrad_bx6(wd, x, dz, cout, got, ws)
    xp = F.pad(x.interior().double(), (0, 0, 1, 1, 1, 1))
    ref = torch.stack([torch.stack([torch.einsum('bhwk,bhwn->kn', xp[:, dh:dh + H, dw:dw + W], dz.interior().double())
                                    for dw in range(3)]) for dh in range(3)])
    rel = (got.double() - ref).abs().max().item() / ref.abs().max().item()
    print('bx6 wgrad rel err %.2e' % rel)
    assert rel < 1e-5
    again = torch.zeros_like(got)
    ops.tap_wgrad_bx6(wd, x, dz, cout, again, ws)
    assert torch.equal(got, again)                                # fixed-order chunk sums: bitwise reproducible


def test_dfcnn_step_in_split_bf16_mode_matches_oracle(monkeypatch):
    from oracle import dfcnn
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine
    monkeypatch.setenv('ASR_BX6', '1')
    rng = np.random.default_rng(3)
    B, T, Fd, V, widths = 2, 32, 16, 12, (8, 16, 32, 64)
    g = dfcnn.graph('m2', V, widths, feat=Fd)
    P = dfcnn.init_params(g, seed=3, perturb=True)
    P = {l: {k: v.astype(np.float32).astype(np.float64) for k, v in d.items()} for l, d in P.items()}
    x = rng.standard_normal((B, T, Fd, 1)).astype(np.float32)
    target = np.zeros((B, 64), dtype=np.int32); target[:, :2] = rng.integers(1, V - 1, (B, 2))
    seq = [4, 3]
    ref = dfcnn.train_step_oracle(g, P, x.astype(np.float64), seq, target)
    eng = DFCNNEngine(model='m2', vocab=V, B=B, T=T, F=Fd, widths=widths)
    assert eng.bx6 and len(eng.ws_f) > 0
    eng.load_params(P)
    logits = eng.forward(torch.tensor(x.reshape(B, T, Fd), device='cuda'))
    eng.set_targets(seq, target); eng.loss_and_decode(); eng.backward()
    torch.cuda.synchronize()
    assert np.abs(logits.cpu().numpy() - ref['logits']).max() < 1e-3
    assert np.abs(eng.loss.cpu().numpy() - ref['loss'][:, 0]).max() < 1e-3
    got = eng.grads_dict()
    for l in ref['grads']:
        for k, want in ref['grads'][l].items():
            assert np.abs(got[l][k] - want).max() <= 1e-3 * max(1e-6, np.abs(want).max()), (l, k)


def _bf16_round(x):
    """float64 -> nearest bfloat16 value (round to nearest even on the 16 dropped bits of the float32 pattern)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def test_split_bf16_worst_case_dropped_terms():
    """Adversarial inputs for the six-product scheme: every operand has mid and low pieces close to the largest the split
    allows and of ONE sign, so the three dropped products (mid x low, low x mid, low x low) of all K = 1152 terms of a dot
    product add up instead of cancelling.  Pieces of 8 significant bits give |mid| <= 2^-8 |x| and |low| <= 2^-17 |x|, so
    what is dropped is at most 2^-24 |x y| per product in the worst case (about 2^-28 for typical data) -- NOT 2^-32 as an
    earlier comment claimed.  Compared with float64 and with the fp32-MFMA kernel on the same data."""
    from asr_dfcnn_transformer_amd import ops
    rng = np.random.default_rng(0)
    K, M, N = 1152, 64, 32
    c = (1.0 + rng.random(400000)).astype(np.float32).astype(np.float64)           # candidates in [1, 2)
    h = _bf16_round(c); r1 = c - h; m = _bf16_round(r1); l = r1 - m
    assert (np.abs(m) <= 2.0 ** -8 * c).all() and (np.abs(l) <= 2.0 ** -17 * c).all()      # the piece bounds quoted above
    good = c[(m > 0.6 * 2.0 ** -8 * c) & (l > 0.6 * 2.0 ** -17 * c)]
    assert len(good) >= 256
    a = rng.choice(good, (M, K)); w = rng.choice(good, (K, N))
    split = lambda v: (lambda hh: (hh, _bf16_round(v - hh), v - hh - _bf16_round(v - hh)))(_bf16_round(v))
    ah, am, al = split(a); wh, wm, wl = split(w)
    dropped = am @ wl + al @ wm + al @ wl                            # all terms positive: the worst case
    exact = a @ w
    A = torch.tensor(a, dtype=torch.float32, device='cuda'); W = torch.tensor(w, dtype=torch.float32, device='cuda')
    d = ops.gemm_desc(M, K, N, K, N, 0, N, ntaps=1)
    y_bx = torch.zeros(M, N, device='cuda'); y_32 = torch.zeros(M, N, device='cuda')
    ops.tap_gemm_bx6(d, A, ops.split_weights(W, 1, K, N, N, 0), None, None, None, None, y_bx)
    ops.tap_gemm(d, A, W, None, None, None, None, y_32)
    s_bx = y_bx.double().cpu().numpy() - exact; s_32 = y_32.double().cpu().numpy() - exact
    rel_drop = (dropped / exact).max()
    print('dropped / exact %.2e (2^-24 = %.2e), split-bf16 max rel err %.2e, fp32 kernel %.2e'
          % (rel_drop, 2.0 ** -24, (np.abs(s_bx) / exact).max(), (np.abs(s_32) / exact).max()))
    assert 2.0 ** -27 < rel_drop < 2.0 ** -24                           # the construction is within 8x of the worst case
    # What this case shows (MI355X: dropped 3.2e-8, split-bf16 7.2e-6, fp32 kernel 2.0e-6 relative): the dropped products
    # are NOT what limits the scheme.  All six partial products of a term go into ONE fp32 accumulator; once it has grown
    # (same-sign data: ~2600 here, ulp 2.4e-4) the low-order products (~1e-5 each) fall below half an ulp and are rounded
    # away, 3.6x the fp32 chain's own stagnation error on this data.  Still 140x inside the 1e-3 bar of north_star, but the
    # claim "never worse than the fp32 kernels" only holds for data whose partial sums stay small (random signs) -- one of
    # the reasons the mode stays experimental.
    rel_bx, rel_32 = (np.abs(s_bx) / exact).max(), (np.abs(s_32) / exact).max()
    assert rel_bx < 2e-5 and rel_32 < 1e-5
    assert rel_bx < 6.0 * rel_32
    assert np.all(s_bx < 0)                                             # the lost low-order terms were all positive
