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
