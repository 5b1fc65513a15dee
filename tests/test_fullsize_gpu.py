"""Parity at BASELINE.json's full layer sizes (T_pad 1600, 200 mel bins, V 1536, reference channel widths).
One utterance is checked end to end against the independent float64 torch-CPU restatement
(oracle/torch_ref.py, itself pinned to the numpy oracle in tests/test_oracle_cpu.py); the batch-32 step is
checked through size-independent properties: per-utterance independence (bitwise), bitwise reproducibility,
CTC gradient rows summing to zero, zero rows beyond logits_length."""
import numpy as np
import pytest
import torch

from oracle import dfcnn, torch_ref, ctc as octc, fbank as ofb

pytestmark = pytest.mark.gpu
V, T, F = 1536, 1600, 200


def _inputs(B, seed=0, T=T):
    rng = np.random.default_rng(seed)
    x = np.zeros((B, T, F), dtype=np.float32)
    for b in range(B):
        nf = max(120, 999 - 27 * b)
        x[b, :nf] = rng.standard_normal((nf, F)).astype(np.float32)
    seq = np.array([min(200, max(120, 999 - 27 * b) // 8 + 1) for b in range(B)], dtype=np.int32)
    target = np.zeros((B, 64), dtype=np.int32)
    for b in range(B):
        L = min(32, int(seq[b]) // 3)            # always alignable, repeats included
        target[b, :L] = rng.integers(1, V - 1, L)
    return x, target, seq


# T_pad 1000 is north_star's "T≈1000" (the loader pads to feature_max_length, hparams.py:19): planes of 500 / 250 / 125 rows -- the odd
# height runs on the Winograd kernels since round 4 (a half-filled last tile row)
@pytest.mark.parametrize("model,T", [("m2", 1600), ("m1", 1600), ("small", 1600), ("m1", 1000), ("m2", 1000)])
def test_one_utterance_full_width_vs_float64_reference(model, T):
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine
    x, target, seq = _inputs(1, T=T)
    eng = DFCNNEngine(model=model, vocab=V, B=1, T=T, F=F, seed=5)
    P = eng.params_dict()
    logits = eng.forward(torch.tensor(x, device='cuda'))
    eng.set_targets(seq, target)
    eng.loss_and_decode()
    eng.backward()
    torch.cuda.synchronize()
    if T == 1000:
        assert set(eng.wt_f) and all(k in eng.wt_b for k in eng.wt_f), 'the 3x3 layers of a T_pad 1000 graph run on the Winograd kernels'
    g = dfcnn.graph(model, V)
    tP = torch_ref.to_torch_params({l: {k: v.astype(np.float64) for k, v in d.items()} for l, d in P.items()})
    labels = octc.dense_to_sparse(target)
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    ref_logits, ref_loss, _, _ = torch_ref.train_step(g, tP, torch.tensor(x[..., None], dtype=torch.float64), seq, labels)
    got = logits.cpu().numpy()
    err = np.abs(got - ref_logits.numpy()).max()
    print(model, 'full-width logits err', err, 'loss', float(eng.loss[0]), float(ref_loss[0]))
    assert err < 1e-3
    assert abs(float(eng.loss[0]) - float(ref_loss[0])) < 1e-3 * max(1.0, abs(float(ref_loss[0])))
    dec_ref, _ = octc.ctc_greedy_decode(ref_logits.numpy(), seq)
    assert eng.decoded_lists() == dec_ref
    G = eng.grads_dict()
    # max-pool graphs (m1): a near-tie inside a 2x2 window can pick a different arg-max in float32 than in the
    # float64 reference, which re-routes single gradient elements; the bar for its conv gradients is 5e-3
    tol = 5e-3 if model == 'm1' else 1e-3
    beyond, total = 0, 0
    for layer in P:                                       # EVERY layer of the graph
        for key in P[layer]:
            ref = tP[layer][key].grad.numpy()
            dev = np.abs(G[layer][key] - ref) / max(1e-12, np.abs(ref).max())
            rel = dev.max()
            beyond += int((dev >= 1e-3).sum()); total += dev.size
            print(model, layer, key, 'grad rel err %.2e' % rel)
            assert rel < tol, (layer, key, rel)
    # the 5e-3 ceiling of the max-pool graphs is for SINGLE re-routed elements: counted over every gradient element of the model, those
    # beyond north_star's 1e-3 (of their tensor's scale) must stay a vanishing fraction
    print(model, 'gradient elements beyond 1e-3 of their tensor scale: %d of %d' % (beyond, total))
    assert beyond <= 1e-5 * total, (beyond, total)


@pytest.mark.parametrize("model,B", [("m2", 32), ("m1", 32), ("small", 4)])
def test_batch32_properties(model, B):
    """BASELINE configs[2] / configs[1] (batch 32) and configs[0] (DFCNN-small, batch 4) at full size."""
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine
    x, target, seq = _inputs(B, seed=1)
    eng = DFCNNEngine(model=model, vocab=V, B=B, T=T, F=F, seed=5)
    xd = torch.tensor(x, device='cuda')
    logits = eng.forward(xd).clone()
    eng.set_targets(seq, target)
    eng.loss_and_decode()
    eng.backward()
    torch.cuda.synchronize()
    g1 = eng.grad.clone()
    # (1) CTC gradient rows sum to zero inside logits_length and are zero beyond it
    cg = eng.ctc_grad.cpu().numpy()
    rows = cg.sum(axis=2)
    assert np.abs(rows).max() < 2e-4
    for b in (0, B // 4, B - 1):
        assert np.all(cg[seq[b]:, b, :] == 0)
    loss = eng.loss.cpu().numpy()
    assert np.all(np.isfinite(loss)) and np.all(loss > 0)
    # (2) same step again: bitwise identical gradients (fixed-order reductions, no atomics)
    eng.forward(xd); eng.set_targets(seq, target); eng.loss_and_decode(); eng.backward()
    torch.cuda.synchronize()
    assert torch.equal(g1, eng.grad)
    # (3) utterances are independent (frozen BN, per-sample SE and CTC): utterance 5 alone gives bitwise the
    #     same logits as inside the batch of 32
    u = min(5, B - 1)
    e1 = DFCNNEngine(model=model, vocab=V, B=1, T=T, F=F, seed=5)
    l1 = e1.forward(xd[u:u + 1].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(l1[:, 0, :], logits[:, u, :])
    # (4) ... and so are their gradients: the batch gradient is the mean of the per-utterance gradients (reduce_mean,
    #     acoustic_model2.py:83).  Checked on utterance u's CTC gradient rows (bitwise) and on the whole step for the
    #     small model, where 4 single-utterance steps are cheap
    e1.set_targets(seq[u:u + 1], target[u:u + 1]); e1.loss_and_decode(); e1.backward()
    torch.cuda.synchronize()
    assert torch.equal(e1.ctc_grad[:, 0, :], eng.ctc_grad[:, u, :]) and torch.equal(e1.loss[0], eng.loss[u])
    if B <= 4:
        tot = torch.zeros_like(eng.grad, dtype=torch.float64)
        for b in range(B):
            e1.forward(xd[b:b + 1].contiguous()); e1.set_targets(seq[b:b + 1], target[b:b + 1]); e1.loss_and_decode(); e1.backward()
            tot += e1.grad.double()
        torch.cuda.synchronize()
        diff = (tot / B - g1.double()).abs().max().item()
        assert diff <= 1e-6 * g1.abs().max().item(), diff


def test_fbank_full_length_batch_matches_oracle_and_is_ragged_safe():
    from asr_dfcnn_transformer_amd import wav_util
    rng = np.random.default_rng(3)
    lens = [160000, 123457, 400, 16000]
    sig = np.zeros((4, 160000), dtype=np.float32)
    for b, n in enumerate(lens):
        sig[b, :n] = (0.1 * rng.standard_normal(n)).astype(np.float32)
    ex = wav_util.FbankExtractor()
    feat, frames = ex.batch(torch.tensor(sig, device='cuda'), torch.tensor(lens, dtype=torch.int32, device='cuda'), T)
    feat, frames = feat.cpu().numpy(), frames.cpu().numpy()
    assert frames.tolist() == [ofb.num_frames(n) for n in lens] and frames[0] == 999 and frames[2] == 1
    for b in (1, 3):
        ref = ofb.compute_fbank_from_api(sig[b, :lens[b]].astype(np.float64), 16000).astype(np.float32)
        assert np.abs(feat[b, :ref.shape[0]] - ref).max() <= 2e-6
        assert np.all(feat[b, ref.shape[0]:] == 0)
    assert np.all(feat[2] == 0)          # a single frame standardises to zero


@pytest.mark.parametrize("H,W,cin,cout", [(800, 100, 32, 64), (400, 50, 64, 128), (200, 25, 128, 128), (200, 25, 128, 256),
                                          (200, 25, 32, 256), (800, 100, 32, 32)])
def test_winograd_weight_gradient_equals_direct_at_layer_size(H, W, cin, cout):
    """asr_tap_wgrad (Winograd F(3x3,2x2), wino_wgrad.hip) against asr_tap_wgrad_direct (the direct eight-wave kernels: an independent
    implementation) on the plain DFCNN's 3x3 layer shapes at B = 8, plus the properties the training step relies on: bitwise
    reproducible, and blind to what lies in the partials workspace."""
    from asr_dfcnn_transformer_amd import ops
    B = 8
    g = torch.Generator(device='cuda').manual_seed(31)
    x = ops.Plane(B, H, W, cin); x.set_interior(torch.randn(B, H, W, cin, device='cuda', generator=g))
    dz = ops.Plane(B, H, W, cout); dz.set_interior(torch.randn(B, H, W, cout, device='cuda', generator=g))
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, ntaps=9, B=B, H=H, W=W)
    ws = torch.zeros(ops.tap_wgrad_workspace(d) // 4 + 64, device='cuda')
    dw0, dw1, dw2 = (torch.zeros(9 * cin * cout, device='cuda') for _ in range(3))
    ops.tap_wgrad(d, x, dz, cout, dw0, ws, direct=True)
    ops.tap_wgrad(d, x, dz, cout, dw1, ws)
    ws.fill_(float('nan'))                                   # every partial the reduce reads is written by this launch
    ops.tap_wgrad(d, x, dz, cout, dw2, ws)
    scale = dw0.abs().max().item()
    assert (dw1 - dw0).abs().max().item() < 1e-5 * scale
    assert torch.equal(dw1, dw2)
    # ... and against float64 (oracle.nn.conv2d_same_bwd) on ONE image of the same planes (B = 1 keeps the host side in seconds)
    from oracle import nn as onn
    x1 = ops.Plane(1, H, W, cin); x1.set_interior(x.interior()[:1].contiguous())
    dz1 = ops.Plane(1, H, W, cout); dz1.set_interior(dz.interior()[:1].contiguous())
    d1 = ops.gemm_desc(x1.NP, cin, cout, cin, cout, ntaps=9, B=1, H=H, W=W)
    ws1 = torch.zeros(ops.tap_wgrad_workspace(d1) // 4 + 64, device='cuda')
    dw3 = torch.zeros(9 * cin * cout, device='cuda')
    ops.tap_wgrad(d1, x1, dz1, cout, dw3, ws1)
    _, ref = onn.conv2d_same_bwd(x1.interior().double().cpu().numpy(), np.zeros((3, 3, cin, cout)), dz1.interior().double().cpu().numpy())
    err = np.abs(dw3.cpu().numpy().astype(np.float64).reshape(3, 3, cin, cout) - ref).max()
    assert err < 1e-5 * np.abs(ref).max(), err
