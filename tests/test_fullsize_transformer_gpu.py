"""The Transformer-family workloads at the sizes their numbers are quoted on (bench.py --workload transformer / lm /
am_lm): BASELINE configs[3] (pinyin ids -> hanzi encoder-decoder, d_model 512, 8 heads, 6 + 6 blocks, B = 64, T = 512,
V 1536 / 6347), the Language_Model of SURVEY 8 a12 (B = 64, T = 100, 12 blocks, V 1536 / 6345) and the joint AM + LM
graph of configs[4] (B = 32, T_pad 1600, 12 blocks).  One sequence of each Transformer is compared with the float64
oracle at full width; the full batches are checked through size-independent properties: bitwise reproducibility,
sequence independence (a sequence alone == inside the batch), cross-entropy / CTC gradient rows summing to zero, pad
columns of the vocabulary projection untouched."""
import numpy as np
import pytest
import torch

from oracle import transformer as otr

pytestmark = pytest.mark.gpu


def rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.abs(got - want).max() / max(1e-6, np.abs(want).max())


def rel2(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.linalg.norm(got - want) / max(1e-12, np.linalg.norm(want))


def check_grads_fullsize(G, R, tag):
    """Gradient bar at full width: element-wise 2e-2 of the tensor's scale, relative L2 error 5e-3.
    Why not the 1e-3 of the small configurations: at 512 positions x 512 channels x 12 blocks the float32 backward is
    ill-conditioned in places (LayerNorm backward subtracts two projections of the incoming gradient; ReLU gates sit on
    pre-activations that are zero to rounding), and ANY float32 evaluation order lands 1e-4 .. 7e-3 of scale away from
    float64 on a few tensors.  Measured with the independent torch-CPU restatement (tests/torch_transformer_ref.py) run in
    float32: on this container's CPU it differs from float64 by 7.2e-3 on enc5/wo and 4-5e-4 on the LayerNorm gains -- the
    same tensors and sizes as the HIP path -- and on the GPU box's CPU (another BLAS code path) by 2.6e-3 on enc5/wv and
    1.9e-3 on enc_emb instead (profiles/r02c_fp32_vs_f64_gradient_conditioning.txt).  Logits, loss and predictions keep the 1e-3 / exact bars."""
    worst, worst2 = 0.0, 0.0
    for k in R:
        r, r2 = rel(G[k], R[k]), rel2(G[k], R[k])
        worst, worst2 = max(worst, r), max(worst2, r2)
        assert r2 < 5e-3 and r < 2e-2, (tag, k, r, r2)
    print(tag, 'worst gradient error: element-wise %.2e of scale, relative L2 %.2e' % (worst, worst2))


def f32(P):
    return {k: (f32(v) if isinstance(v, dict) else np.asarray(v, np.float32).astype(np.float64)) for k, v in P.items()}


def flat_grads(G):
    out = {}
    for k, v in G.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                out['%s/%s' % (k, kk)] = vv
        else:
            out[k] = v
    return out


def _retie(P, blocks):
    for i in range(blocks):
        for k in ('wq', 'wk', 'wv', 'wo'):
            P['dec%d' % i][k] = P['enc%d' % i][k]
    for k in ('w1', 'b1', 'w2', 'b2'):
        P['dec_ffn'][k] = P['enc_ffn'][k]
    return P


C, H, BLK, VIN, VOUT, T3 = 512, 8, 6, 1536, 6347, 512


def _e2e_batch(N, seed):
    rng = np.random.default_rng(seed)
    x = rng.integers(1, VIN, (N, T3))
    y = rng.integers(3, VOUT, (N, T3))
    for n in range(N):                                  # ragged: trailing pad ids (0) on both sides
        k = (7 * n) % 97
        if k:
            x[n, T3 - k:] = 0
        k = (11 * n) % 61
        if k:
            y[n, T3 - k:] = 0
    y_in = np.concatenate([np.ones((N, 1), dtype=np.int64), y[:, :-1]], axis=1)
    return x, y_in, y


def test_configs3_one_sequence_full_width_vs_oracle():
    """E2EEngine(vin=1536, vout=6347, T = L = 512, C = 512, 8 heads, 6 + 6 blocks, tied) on ONE sequence against
    oracle.transformer.e2e_step with the id-input path: logits 1e-3, loss 1e-3, predictions exact, every gradient (bar:
    check_grads_fullsize)."""
    from asr_dfcnn_transformer_amd.transformer_engine import E2EEngine
    P = _retie(f32(otr.init_e2e(None, VOUT, C, H, BLK, 600, seed=4, perturb=True, tie=True, vin=VIN)), BLK)
    x, y_in, y = _e2e_batch(2, seed=9)
    x, y_in, y = x[1:2], y_in[1:2], y[1:2]              # sequence 1: has pads on both sides
    ref = otr.e2e_step(P, x, y_in, y, H, BLK, tie=True)
    eng = E2EEngine(vin=VIN, vout=VOUT, N=1, T=T3, L=T3, C=C, heads=H, blocks=BLK, pos_max=600, tie=True)
    eng.load_params(eng.flat_from_oracle(P))
    logits = eng.forward(x, y_in, y)
    eng.backward()
    torch.cuda.synchronize()
    got = logits.cpu().numpy().reshape(1, T3, -1)[:, :, :VOUT]
    err = np.abs(got - ref['logits']).max()
    print('configs[3] full-width logits err', err)
    assert err < 1e-3
    ml, acc = eng.fetch()
    assert abs(ml - ref['mean_loss']) < 1e-3 and abs(acc - ref['acc']) < 1e-6
    assert np.array_equal(eng.preds.cpu().numpy().reshape(1, T3), ref['preds'])
    check_grads_fullsize(eng.grads_dict(), flat_grads(ref['grads']), 'configs[3]')


def test_configs3_batch64_properties():
    from asr_dfcnn_transformer_amd.transformer_engine import E2EEngine
    N = 64
    x, y_in, y = _e2e_batch(N, seed=9)
    eng = E2EEngine(vin=VIN, vout=VOUT, N=N, T=T3, L=T3, C=C, heads=H, blocks=BLK, pos_max=600, tie=True, seed=3)
    logits = eng.forward(x, y_in, y).clone()
    eng.backward()
    torch.cuda.synchronize()
    g1 = eng.grad.clone()
    assert torch.isfinite(logits).all() and torch.isfinite(g1).all()
    # (1) label-smoothed CE: every row of dL/dlogits sums to zero over the real vocabulary (softmax and smoothed labels
    #     both sum to 1; rows with target 0 carry no loss at all), the 6348th pad column never receives gradient
    dl = eng.dlogits.view(N * T3, -1)
    rows = dl[:, :VOUT].double().sum(dim=1).abs().max().item()
    assert rows < 1e-7, rows
    assert not dl[:, VOUT:].any().item()
    pad_rows = torch.from_numpy((y.reshape(-1) == 0)).cuda()
    assert not dl[pad_rows].any().item()
    # (2) same step again: bitwise identical gradient (fixed-order reductions, no atomics)
    eng.forward(x, y_in, y); eng.backward(); torch.cuda.synchronize()
    assert torch.equal(g1, eng.grad)
    # (3) sequences are independent: sequence 5 alone gives the same logits as inside the batch of 64
    e1 = E2EEngine(vin=VIN, vout=VOUT, N=1, T=T3, L=T3, C=C, heads=H, blocks=BLK, pos_max=600, tie=True, seed=3)
    l1 = e1.forward(x[5:6], y_in[5:6], y[5:6])
    torch.cuda.synchronize()
    a, b = l1.view(T3, -1), logits.view(N, T3, -1)[5]
    print('sequence 5 alone vs in batch: max abs diff', (a - b).abs().max().item(), 'bitwise', torch.equal(a, b))
    assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
    assert torch.equal(e1.preds.view(-1), eng.preds.view(N, T3)[5])


def test_language_model_a12_size_vs_oracle_and_properties():
    """Language_Model at B = 64, T = 100, C = 512, 8 heads, 12 blocks, V 1536 -> 6345 (hparams.py:16,23; bench --workload lm)."""
    from asr_dfcnn_transformer_amd.transformer_engine import LMEngine
    N, T, blocks, Vin, Vout = 64, 100, 12, 1536, 6345
    rng = np.random.default_rng(5)
    x = rng.integers(1, Vin, (N, T)); y = rng.integers(1, Vout, (N, T))
    for n in range(N):
        k = (13 * n) % 41
        if k:
            x[n, T - k:] = 0; y[n, T - k:] = 0
    P = f32(otr.init_lm(Vin, Vout, C, H, blocks, 100, seed=6, perturb=True))
    ref = otr.lm_step(P, x[3:4], y[3:4], H, blocks)
    e1 = LMEngine(vin=Vin, vout=Vout, N=1, T=T, C=C, heads=H, blocks=blocks, pos_max=100)
    e1.load_params(e1.flat_from_oracle(P))
    l1 = e1.forward(x[3:4], y[3:4]).clone()
    e1.backward()
    torch.cuda.synchronize()
    got = l1.cpu().numpy().reshape(1, T, -1)[:, :, :Vout]
    assert np.abs(got - ref['logits']).max() < 1e-3
    ml, acc = e1.fetch()
    assert abs(ml - ref['mean_loss']) < 1e-3 and abs(acc - ref['acc']) < 1e-6
    assert np.array_equal(e1.preds.cpu().numpy().reshape(1, T), ref['preds'])
    check_grads_fullsize(e1.grads_dict(), flat_grads(ref['grads']), 'Language_Model a12')
    eng = LMEngine(vin=Vin, vout=Vout, N=N, T=T, C=C, heads=H, blocks=blocks, pos_max=100)
    eng.load_params(eng.flat_from_oracle(P))
    logits = eng.forward(x, y).clone()
    eng.backward(); torch.cuda.synchronize()
    g1 = eng.grad.clone()
    dl = eng.dlogits.view(N * T, -1)
    assert dl[:, :Vout].double().sum(dim=1).abs().max().item() < 1e-7 and not dl[:, Vout:].any().item()
    eng.forward(x, y); eng.backward(); torch.cuda.synchronize()
    assert torch.equal(g1, eng.grad)
    a, b = l1.view(T, -1), logits.view(N, T, -1)[3]
    print('lm sequence 3 alone vs in batch: max abs diff', (a - b).abs().max().item(), 'bitwise', torch.equal(a, b))
    assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()


def test_joint_am_lm_configs4_size_properties():
    """AMLMEngine at BASELINE configs[4]'s per-GPU size (B = 32, T_pad 1600, 200 mel bins, V 1536 / 6345, 12 blocks)."""
    from asr_dfcnn_transformer_amd.joint_engine import AMLMEngine
    B, T, F, VP, VH = 32, 1600, 200, 1536, 6345
    rng = np.random.default_rng(2)
    x = np.zeros((B, T, F), dtype=np.float32)
    wl = np.zeros(B, dtype=np.int32)
    for b in range(B):
        nf = 999 - 23 * b
        x[b, :nf] = rng.standard_normal((nf, F)).astype(np.float32)
        wl[b] = min(200, nf // 8 + 1)
    tp = np.zeros((B, 64), dtype=np.int32); tl = np.zeros(B, dtype=np.int32)
    hz = np.zeros((B, 64), dtype=np.int32)
    for b in range(B):
        L = min(30, int(wl[b]) // 3)
        tp[b, :L] = rng.integers(1, VP - 1, L); tl[b] = L
        hz[b, :L] = rng.integers(1, VH, L)
    xd = torch.tensor(x, device='cuda')

    def step(e, rows):
        e.forward(xd[rows].contiguous()); e.set_targets(wl[rows], tp[rows], tl[rows], hz[rows]); e.loss_and_decode(); e.backward()
        torch.cuda.synchronize()

    eng = AMLMEngine(v_pinyin=VP, v_hanzi=VH, B=B, T=T, F=F, blocks=12, pos_max=T // 8, seed=1)
    allrows = np.arange(B)
    step(eng, allrows)
    am_l, lm_l = eng.am.logits.clone(), eng.lm.logits.clone()
    g_am, g_lm = eng.am.grad.clone(), eng.lm.grad.clone()
    assert torch.isfinite(g_am).all() and torch.isfinite(g_lm).all()
    for cg in (eng.am.ctc_grad, eng.lm.ctc_grad):
        rows = cg.double().sum(dim=2).abs().max().item()
        assert rows < 2e-4, rows
    assert not eng.am.ctc_grad[int(wl[B - 1]):, B - 1].any().item()
    step(eng, allrows)
    assert torch.equal(g_am, eng.am.grad) and torch.equal(g_lm, eng.lm.grad)
    e1 = AMLMEngine(v_pinyin=VP, v_hanzi=VH, B=1, T=T, F=F, blocks=12, pos_max=T // 8, seed=1)
    step(e1, np.array([7]))
    assert torch.equal(e1.am.logits[:, 0], am_l[:, 7])
    d = (e1.lm.logits[:, 0] - lm_l[:, 7]).abs().max().item()
    print('joint model: utterance 7 alone vs in batch, language-half logits max abs diff', d)
    assert d <= 2e-5 * lm_l[:, 7].abs().max().item()
    am_dec, lm_dec = eng.decoded_lists()
    a1, l1 = e1.decoded_lists()
    assert am_dec[7] == a1[0] and lm_dec[7] == l1[0]
