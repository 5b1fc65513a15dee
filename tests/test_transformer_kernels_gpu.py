"""GPU parity of the Transformer kernels (asr_attention_*, asr_add_layernorm_*, asr_embed_*,
asr_smoothed_ce) against oracle/transformer.py.  fp32 tolerances are written per check."""
import numpy as np
import pytest
import torch

from oracle import transformer as otr

pytestmark = pytest.mark.gpu


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device='cuda')


def report(name, got, want, tol):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want).max()
    scale = max(1.0, np.abs(want).max())
    print('%-36s max|err| %.3e (scale %.2e, tol %.0e)' % (name, err, scale, tol))
    assert got.shape == want.shape
    assert err <= tol * scale, name


@pytest.fixture(scope='module')
def ops():
    from asr_dfcnn_transformer_amd import ops as _ops
    return _ops


# (Tq, Tk <= 128 run on the short-sequence kernels of attention_small.hip since round 6: rows of 16, the whole (sample, head) in LDS)
@pytest.mark.parametrize("N,Tq,Tk,H,causal", [(2, 7, 7, 2, True), (2, 100, 100, 8, True), (1, 130, 70, 2, False),
                                              (2, 33, 200, 4, True), (1, 512, 512, 8, False),
                                              (2, 300, 300, 2, True), (2, 520, 520, 2, True),
                                              (2, 100, 90, 2, False), (1, 128, 128, 2, True), (2, 17, 113, 2, False), (2, 128, 16, 2, True)])
def test_attention_fwd_bwd(ops, N, Tq, Tk, H, causal):
    rng = np.random.default_rng(0)
    C = H * 64
    relu = lambda x: np.maximum(x, 0)
    Q = relu(rng.standard_normal((N, Tq, C))).astype(np.float32)
    K = relu(rng.standard_normal((N, Tk, C))).astype(np.float32)
    V = relu(rng.standard_normal((N, Tk, C))).astype(np.float32)
    K[0, Tk - 3:, :] = 0                      # padded keys: zero rows in every head -> key mask
    K[N - 1, 1, :64] = 0                      # one key masked in head 0 only
    Q[0, Tq - 1, :] = 0                       # a zero query row -> query mask zeroes its output
    if Tk > 8:
        K[N - 1, :, 64:128] = 0               # head 1 of the last sample: ALL keys masked -> uniform softmax
        K[0, :2, :] = 0                       # causal: queries 0 and 1 of sample 0 see only masked keys, so their
        #                                       softmax is uniform over ALL Tk fill entries (no future tile may be skipped)
    Oref, cache = otr.attention_core(Q.astype(np.float64), K.astype(np.float64), V.astype(np.float64), H, causal)
    O = torch.zeros(N, Tq, C, device='cuda')
    lse = torch.zeros(2, N, H, Tq, device='cuda')
    dQ, dK, dV = dev(Q), dev(K), dev(V)
    ops.attention_fwd(dQ, dK, dV, N, Tq, Tk, C, H, causal, O, lse)
    report('attention fwd', O.cpu().numpy(), Oref, 2e-5)
    dO = rng.standard_normal((N, Tq, C)).astype(np.float32)
    gq_ref, gk_ref, gv_ref = otr.attention_core_bwd(cache, dO.astype(np.float64))
    gq, gk, gv = torch.zeros_like(dQ), torch.zeros_like(dK), torch.zeros_like(dV)
    ws = torch.zeros(N * H * Tq + 16, device='cuda')
    ops.attention_bwd(dQ, dK, dV, O, dev(dO), lse, N, Tq, Tk, C, H, causal, gq, gk, gv, ws)
    report('attention dQ', gq.cpu().numpy(), gq_ref, 3e-5)
    report('attention dK', gk.cpu().numpy(), gk_ref, 3e-5)
    report('attention dV', gv.cpu().numpy(), gv_ref, 3e-5)
    # relu_grad: the same gradients masked by (projection > 0), fused into the stores
    ops.attention_bwd(dQ, dK, dV, O, dev(dO), lse, N, Tq, Tk, C, H, causal, gq, gk, gv, ws, relu_grad=True)
    report('attention dQ (pre-relu)', gq.cpu().numpy(), gq_ref * (Q > 0), 3e-5)
    report('attention dK (pre-relu)', gk.cpu().numpy(), gk_ref * (K > 0), 3e-5)
    report('attention dV (pre-relu)', gv.cpu().numpy(), gv_ref * (V > 0), 3e-5)
    # round 5: the query masks / key biases taken once (asr_attention_stats) instead of per workgroup -- the same values, the same bits
    stats = torch.full((ops.attention_stats_floats(N, Tq, Tk, H),), 7.0, device='cuda')
    ops.attention_stats(dQ, dK, N, Tq, Tk, C, H, stats)
    qm = stats[:N * H * Tq].view(N, H, Tq).cpu().numpy()
    kb = stats[N * H * Tq:].view(N, H, Tk).cpu().numpy()
    assert np.array_equal(qm, (np.abs(Q).reshape(N, Tq, H, 64).sum(-1) != 0).transpose(0, 2, 1).astype(np.float32))
    assert np.array_equal(np.isinf(kb), (K.reshape(N, Tk, H, 64).sum(-1) != 0).transpose(0, 2, 1))
    O2, lse2 = torch.zeros_like(O), torch.zeros_like(lse)
    ops.attention_fwd(dQ, dK, dV, N, Tq, Tk, C, H, causal, O2, lse2, stats=stats)
    assert torch.equal(O2, O) and torch.equal(lse2, lse)
    g2 = [torch.zeros_like(dQ), torch.zeros_like(dK), torch.zeros_like(dV)]
    ops.attention_bwd(dQ, dK, dV, O, dev(dO), lse, N, Tq, Tk, C, H, causal, *g2, ws, relu_grad=True, stats=stats)
    assert torch.equal(g2[0], gq) and torch.equal(g2[1], gk) and torch.equal(g2[2], gv)


def test_attention_stats_on_column_blocks_and_its_argument_checks(ops):
    """asr_attention_stats on Q / K that are column blocks of one fused [rows][3C] projection buffer (row pitches 3C, as the engines call
    it) gives the statistics of the dense copies; with them the pitched attention gives the bits of the call without; bad arguments are
    refused before anything is launched."""
    from asr_dfcnn_transformer_amd import _lib
    rng = np.random.default_rng(3)
    N, T, H = 3, 70, 4
    C = H * 64
    buf = np.maximum(rng.standard_normal((N * T, 3 * C)), 0).astype(np.float32)
    buf[5, :C] = 0; buf[9, C:C + 64] = 0; buf[N * T - 1, C:2 * C] = 0
    bd = dev(buf)
    Q, K, V = bd[:, :C], bd[:, C:2 * C], bd[:, 2 * C:]
    stats = torch.zeros(ops.attention_stats_floats(N, T, T, H), device='cuda')
    ops.attention_stats(Q, K, N, T, T, C, H, stats, ldq=3 * C, ldk=3 * C)
    dense = torch.zeros_like(stats)
    ops.attention_stats(Q.contiguous(), K.contiguous(), N, T, T, C, H, dense)
    assert torch.equal(stats, dense)
    O1, O2 = torch.zeros(N * T, C, device='cuda'), torch.zeros(N * T, C, device='cuda')
    l1, l2 = torch.zeros(2, N, H, T, device='cuda'), torch.zeros(2, N, H, T, device='cuda')
    for causal in (False, True):
        ops.attention_fwd(Q, K, V, N, T, T, C, H, causal, O1, l1, ldq=3 * C, ldk=3 * C)
        ops.attention_fwd(Q, K, V, N, T, T, C, H, causal, O2, l2, ldq=3 * C, ldk=3 * C, stats=stats)
        assert torch.equal(O1, O2) and torch.equal(l1, l2)
    lib = _lib.load()
    p = lambda t: t.data_ptr()
    assert lib.asr_attention_stats(p(Q), p(K), N, T, T, C + 4, H, 3 * C, 3 * C, p(stats), None) == -1        # C != H * 64
    assert lib.asr_attention_stats(p(Q), p(K), N, T, T, C, H, C - 4, 3 * C, p(stats), None) == -1            # pitch below the width
    assert lib.asr_attention_stats(p(Q), None, N, T, T, C, H, 3 * C, 3 * C, p(stats), None) == -1


@pytest.mark.parametrize("rows,C,with_b", [(10, 512, True), (300, 64, False), (77, 2048, True), (5, 1030, True)])   # 1030: scalar kernels
def test_add_layernorm(ops, rows, C, with_b):
    rng = np.random.default_rng(1)
    a = rng.standard_normal((rows, C)).astype(np.float32)
    b = rng.standard_normal((rows, C)).astype(np.float32) if with_b else None
    g = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    be = (0.1 * rng.standard_normal(C)).astype(np.float32)
    x = a.astype(np.float64) + (b.astype(np.float64) if with_b else 0)
    yref, cache = otr.layer_norm(x, g.astype(np.float64), be.astype(np.float64))
    y, xh, rs = torch.zeros(rows, C, device='cuda'), torch.zeros(rows, C, device='cuda'), torch.zeros(rows, device='cuda')
    ops.add_layernorm_fwd(dev(a), dev(b) if with_b else None, dev(g), dev(be), rows, C, 1e-8, y, xh, rs)
    report('add_ln fwd', y.cpu().numpy(), yref, 1e-5)
    dy = rng.standard_normal((rows, C)).astype(np.float32)
    dxr, dgr, dbr = otr.layer_norm_bwd(cache, g.astype(np.float64), dy.astype(np.float64))
    dx = torch.zeros(rows, C, device='cuda')
    dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    ws = torch.zeros(ops.layernorm_bwd_workspace(rows, C) // 4 + 4, device='cuda')
    ops.layernorm_bwd(dev(dy), xh, rs, dev(g), rows, C, dx, False, dg, db, ws)
    report('ln dx', dx.cpu().numpy(), dxr, 2e-5)
    report('ln dgamma', dg.cpu().numpy(), dgr, 2e-5)
    report('ln dbeta', db.cpu().numpy(), dbr, 2e-5)
    ops.layernorm_bwd(dev(dy), xh, rs, dev(g), rows, C, dx, True, dg, db, ws)
    report('ln dx accumulate', dx.cpu().numpy(), 2 * dxr, 2e-5)
    if C % 4 == 0:
        # the fused form (residual fan-in + Dense(relu)/dropout backward in the same pass) against the separate calls: same bits
        z = np.maximum(rng.standard_normal((rows, C)), 0).astype(np.float32)
        res = rng.standard_normal((rows, C)).astype(np.float32)
        dx1 = torch.zeros(rows, C, device='cuda')
        ops.layernorm_bwd(dev(dy), xh, rs, dev(g), rows, C, dx1, False, dg, db, ws)
        want2 = dev(res); ops.axpy(want2, dx1, 1.0, True)
        wantz = torch.zeros(rows, C, device='cuda'); ops.relu_bwd_scaled(dx1, dev(z), 1.25, wantz)
        dg1, db1 = dg.clone(), db.clone()
        got, got2, gotz = torch.zeros(rows, C, device='cuda'), dev(res), torch.full((rows, C), 3.0, device='cuda')
        ops.layernorm_bwd_fused(dev(dy), xh, rs, dev(g), rows, C, got, got2, True, dev(z), 1.25, gotz, dg, db, ws)
        assert torch.equal(got, dx1) and torch.equal(got2, want2) and torch.equal(gotz, wantz)
        assert torch.equal(dg, dg1) and torch.equal(db, db1)
        # ... and with the generator's mask on top (the operand was dropped by add_layernorm_fwd_dropout)
        wantd = wantz.clone(); ops.dropout(wantd, 0.2, 77)                       # mask o relu mask, factor 1.25 * 1.25
        ops.layernorm_bwd_fused(dev(dy), xh, rs, dev(g), rows, C, None, None, False, dev(z), 1.25 * 1.25, gotz, dg, db, ws, 0.2, 77)
        keep = wantd != 0
        assert float((gotz - wantd).abs().max()) <= 1e-6 * float(wantd.abs().max())
        assert torch.equal(gotz == 0, wantd == 0)
        # forward: add_layernorm(dropout(a), b) in one pass = the two calls, a untouched
        a_d = dev(a); a_keep = a_d.clone()
        ops.dropout(a_d, 0.2, 77)
        y1, xh1, rs1 = torch.zeros(rows, C, device='cuda'), torch.zeros(rows, C, device='cuda'), torch.zeros(rows, device='cuda')
        ops.add_layernorm_fwd(a_d, dev(b) if with_b else None, dev(g), dev(be), rows, C, 1e-8, y1, xh1, rs1)
        y2, xh2, rs2 = torch.zeros(rows, C, device='cuda'), torch.zeros(rows, C, device='cuda'), torch.zeros(rows, device='cuda')
        ops.add_layernorm_fwd_dropout(a_keep, dev(b) if with_b else None, dev(g), dev(be), rows, C, 1e-8, 0.2, 77, y2, xh2, rs2)
        assert torch.equal(y1, y2) and torch.equal(xh1, xh2) and torch.equal(rs1, rs2)
        assert torch.equal(a_keep, dev(a))
        got2b = torch.full((rows, C), 9.0, device='cuda')
        ops.layernorm_bwd_fused(dev(dy), xh, rs, dev(g), rows, C, None, got2b, False, None, 0.0, None, dg, db, ws)
        assert torch.equal(got2b, dx1)


def test_deferred_layernorm_sums_in_one_batched_reduction(ops):
    """asr_layernorm_bwd_fused with NULL dgamma / dbeta leaves its block partials for asr_colsum_multi_batch: several LayerNorm
    sites of different row counts (one and two reduction levels, both rows-per-block rules) reduced in one call = the bits of the
    immediate form."""
    rng = np.random.default_rng(21)
    C = 512
    sites, items, wants = [], [], []
    for rows in (40, 900, 6400, 33000):
        dy = dev(rng.standard_normal((rows, C)).astype(np.float32))
        xh = dev(rng.standard_normal((rows, C)).astype(np.float32))
        rs = dev((0.5 + rng.random(rows)).astype(np.float32))
        g = dev((1 + 0.1 * rng.standard_normal(C)).astype(np.float32))
        ws = torch.zeros(ops.layernorm_bwd_workspace(rows, C) // 4 + 4, device='cuda')
        dx1, dg1, db1 = torch.zeros(rows, C, device='cuda'), torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
        ops.layernorm_bwd_fused(dy, xh, rs, g, rows, C, dx1, None, False, None, 1.0, None, dg1, db1, ws)
        nblk = ops.layernorm_bwd_blocks(rows)
        part = torch.zeros(nblk * 2 * C, device='cuda')
        dx2, dg2, db2 = torch.zeros(rows, C, device='cuda'), torch.full((C,), 5.0, device='cuda'), torch.full((C,), 5.0, device='cuda')
        ops.layernorm_bwd_fused(dy, xh, rs, g, rows, C, dx2, None, False, None, 1.0, None, None, None, part)
        assert torch.equal(dx1, dx2)
        items.append((part, nblk, 2 * C, [(C, dg2), (C, db2)]))
        wants.append((dg1, db1, dg2, db2))
    batch = ops.ReduceBatch(items)
    batch.run()
    for dg1, db1, dg2, db2 in wants:
        assert torch.equal(dg1, dg2) and torch.equal(db1, db2)


def test_embedding_fwd_bwd(ops):
    from asr_dfcnn_transformer_amd.transformer_engine import sorted_segments
    rng = np.random.default_rng(2)
    N, T, C, V = 3, 9, 64, 17
    table = rng.standard_normal((V, C)).astype(np.float32)
    pos = rng.standard_normal((T + 3, C)).astype(np.float32)
    ids = rng.integers(0, V, (N, T)).astype(np.int32)
    ids[0, :3] = 0
    ref = otr.embedding(table.astype(np.float64), ids, True, True) + pos[:T][None]
    out = torch.zeros(N, T, C, device='cuda')
    ops.embed_fwd(dev(table), dev(ids, torch.int32), dev(pos), N, T, C, True, C ** 0.5, out)
    report('embed fwd', out.cpu().numpy(), ref, 1e-6)
    dout = rng.standard_normal((N, T, C)).astype(np.float32)
    gref = otr.embedding_bwd((V, C), ids, dout.astype(np.float64), True, True)
    perm, uniq, seg = sorted_segments(ids)
    g = torch.zeros(V, C, device='cuda')
    ops.embed_bwd(dev(dout), dev(perm, torch.int32), dev(uniq, torch.int32), dev(seg, torch.int32), len(uniq), C, True,
                  C ** 0.5, g)
    report('embed bwd', g.cpu().numpy(), gref, 2e-6)
    # the same gradient straight from the ids (no sort): the same bits, absent ids untouched
    g2 = torch.full((V, C), 7.0, device='cuda')
    ops.embed_bwd_ids(dev(dout), dev(ids, torch.int32), N * T, V, C, True, C ** 0.5, g2)
    present = np.zeros(V, dtype=bool); present[np.unique(ids)] = True
    assert torch.equal(g2[torch.from_numpy(present).cuda()], g[torch.from_numpy(present).cuda()])
    assert float((g2[torch.from_numpy(~present).cuda()] - 7.0).abs().max()) == 0.0 if (~present).any() else True


def test_embed_bwd_from_ids_at_model_size(ops):
    """asr_embed_bwd_ids at the language model's size (6400 positions, 1536 ids, C 512) and with out-of-range ids: against the
    float64 oracle and bit-for-bit against the sorted-segment kernel."""
    from asr_dfcnn_transformer_amd.transformer_engine import sorted_segments
    rng = np.random.default_rng(12)
    rows, V, C = 6400, 1536, 512
    ids = rng.integers(0, V, rows).astype(np.int32)
    ids[rng.integers(0, rows, 200)] = rng.integers(0, 40, 200)          # a few heavily used ids
    dout = rng.standard_normal((rows, C)).astype(np.float32)
    gref = otr.embedding_bwd((V, C), ids.reshape(1, -1), dout.astype(np.float64).reshape(1, rows, C), True, True)
    perm, uniq, seg = sorted_segments(ids)
    g1, g2 = torch.zeros(V, C, device='cuda'), torch.zeros(V, C, device='cuda')
    ops.embed_bwd(dev(dout), dev(perm, torch.int32), dev(uniq, torch.int32), dev(seg, torch.int32), len(uniq), C, True, C ** 0.5, g1)
    ops.embed_bwd_ids(dev(dout), dev(ids, torch.int32), rows, V, C, True, C ** 0.5, g2)
    assert torch.equal(g1, g2)
    report('embed bwd from ids', g2.cpu().numpy(), gref, 2e-6)
    bad = ids.copy(); bad[:5] = [-1, V, V + 7, 2 ** 30, -(2 ** 31)]
    g3 = torch.zeros(V, C, device='cuda')
    ops.embed_bwd_ids(dev(dout), dev(bad, torch.int32), rows, V, C, True, C ** 0.5, g3)
    keep = np.ones(rows, dtype=bool); keep[:5] = False
    gref3 = otr.embedding_bwd((V, C), ids[keep].reshape(1, -1), dout[keep].astype(np.float64).reshape(1, -1, C), True, True)
    report('embed bwd, ids out of range ignored', g3.cpu().numpy(), gref3, 2e-6)


@pytest.mark.parametrize("rows,V", [(12, 13), (64, 6345), (7, 6347), (5, 7001)])     # 7001: wider than the register kernel
def test_smoothed_ce(ops, rows, V):
    rng = np.random.default_rng(3)
    ld = (V + 3) // 4 * 4
    logits = np.zeros((rows, ld), dtype=np.float32)
    logits[:, :V] = (rng.standard_normal((rows, V)) * 2).astype(np.float32)
    logits[:, V:] = 55.0                                   # pad columns must be ignored
    tgt = rng.integers(1, V, rows).astype(np.int32)
    tgt[0] = 0                                             # PAD: not counted
    tgt[1] = -1                                            # IGNORE: counted with an all-zero one-hot (Q9)
    tgt[2] = int(np.argmax(logits[2, :V]))                 # a correct prediction
    ml, acc, preds, loss, dl = otr.smoothed_ce(logits[None, :, :V].astype(np.float64), tgt[None].astype(np.int64))
    cnt = float((tgt != 0).sum())
    lr, pr = torch.zeros(rows, device='cuda'), torch.zeros(rows, dtype=torch.int32, device='cuda')
    st, dlg = torch.zeros(rows, 2, device='cuda'), torch.zeros(rows, ld, device='cuda')
    ops.smoothed_ce(dev(logits), ld, dev(tgt, torch.int32), rows, V, 0.1, 0, 1.0 / cnt, lr, pr, st, dlg)
    report('ce loss rows', lr.cpu().numpy(), loss[0], 1e-5)
    assert np.array_equal(pr.cpu().numpy(), preds[0])
    s = st.cpu().numpy().astype(np.float64).sum(axis=0)
    assert abs(s[0] / cnt - ml) < 1e-4 and abs(s[1] / cnt - acc) < 1e-6
    d = dlg.cpu().numpy()
    report('ce dlogits', d[:, :V], dl[0], 1e-6)
    assert np.all(d[:, V:] == 0)


def test_dropout_matches_the_oracle_generator(ops):
    """asr_dropout: the counter-based mask is restated in oracle/transformer.py; keep rate ~ 1 - rate; in place; the same
    call on a gradient is the backward."""
    rng = np.random.default_rng(5)
    n, rate, seed = 100003, 0.2, 123456789
    x = rng.standard_normal(n).astype(np.float32)
    m = otr.drop_scale_mask((n,), rate, seed)
    y = dev(x)
    ops.dropout(y, rate, seed)
    got = y.cpu().numpy()
    assert np.array_equal(got != 0, (m != 0) & (x != 0))
    report('dropout values', got, x * m, 1e-6)
    assert abs((m != 0).mean() - (1 - rate)) < 5e-3
    out = torch.zeros(n, device='cuda')
    ops.dropout(dev(x), rate, seed + 1, out)                       # another seed: another mask
    assert not np.array_equal(out.cpu().numpy() != 0, got != 0)


@pytest.mark.parametrize("N,Tq,Tk,H,causal", [(2, 70, 70, 2, True), (1, 130, 200, 2, False), (2, 300, 300, 2, True), (2, 100, 100, 2, False),
                                              (2, 128, 128, 2, True)])
def test_attention_with_weight_dropout(ops, N, Tq, Tk, H, causal):
    """Dropout of the attention weights (transformer.py:111) inside the fused kernels, forward and backward."""
    rng = np.random.default_rng(6)
    C, rate, seed = H * 64, 0.2, 987654321
    relu = lambda x: np.maximum(x, 0)
    Q = relu(rng.standard_normal((N, Tq, C))).astype(np.float32)
    K = relu(rng.standard_normal((N, Tk, C))).astype(np.float32)
    V = relu(rng.standard_normal((N, Tk, C))).astype(np.float32)
    K[0, Tk - 3:, :] = 0
    Q[0, Tq - 1, :] = 0
    K[N - 1, :, 64:128] = 0                                        # all keys of one head masked: uniform weights, then dropped
    M = otr.drop_scale_mask((N, H, Tq, Tk), rate, seed)
    Oref, cache = otr.attention_core(Q.astype(np.float64), K.astype(np.float64), V.astype(np.float64), H, causal, M)
    O = torch.zeros(N, Tq, C, device='cuda')
    lse = torch.zeros(2, N, H, Tq, device='cuda')
    dQ, dK, dV = dev(Q), dev(K), dev(V)
    ops.attention_fwd(dQ, dK, dV, N, Tq, Tk, C, H, causal, O, lse, rate, seed)
    report('attention fwd (dropout)', O.cpu().numpy(), Oref, 3e-5)
    dO = rng.standard_normal((N, Tq, C)).astype(np.float32)
    gq_ref, gk_ref, gv_ref = otr.attention_core_bwd(cache, dO.astype(np.float64))
    gq, gk, gv = torch.zeros_like(dQ), torch.zeros_like(dK), torch.zeros_like(dV)
    ws = torch.zeros(N * H * Tq + 16, device='cuda')
    ops.attention_bwd(dQ, dK, dV, O, dev(dO), lse, N, Tq, Tk, C, H, causal, gq, gk, gv, ws, dropout_rate=rate, seed=seed)
    report('attention dQ (dropout)', gq.cpu().numpy(), gq_ref, 5e-5)
    report('attention dK (dropout)', gk.cpu().numpy(), gk_ref, 5e-5)
    report('attention dV (dropout)', gv.cpu().numpy(), gv_ref, 5e-5)


@pytest.mark.parametrize("causal,rate", [(False, 0.0), (True, 0.25)])
def test_attention_with_row_pitches_equals_the_dense_form(causal, rate):
    """asr_attention_fwd_p / _bwd_p: Q, K, V (and dQ, dK, dV) as column blocks of one [rows][3C] buffer give bitwise the
    results of the dense call (the engines run the three projections of a block as one GEMM); asr_copy2d packs / scatters."""
    import torch
    from asr_dfcnn_transformer_amd import ops
    N, T, H = 3, 150, 2
    C = 64 * H
    g = torch.Generator(device='cuda').manual_seed(4)
    q, k, v = [torch.relu(torch.randn(N * T, C, device='cuda', generator=g)) for _ in range(3)]
    do = torch.randn(N * T, C, device='cuda', generator=g)
    o1, lse1 = torch.zeros(N * T, C, device='cuda'), torch.zeros(2 * N * H * T, device='cuda')
    ops.attention_fwd(q, k, v, N, T, T, C, H, causal, o1, lse1, rate, 7)
    dq1, dk1, dv1 = [torch.zeros(N * T, C, device='cuda') for _ in range(3)]
    ws = torch.zeros(N * H * T + 64, device='cuda')
    ops.attention_bwd(q, k, v, o1, do, lse1, N, T, T, C, H, causal, dq1, dk1, dv1, ws, relu_grad=True, dropout_rate=rate, seed=7)
    qkv = torch.zeros(N * T, 3 * C, device='cuda')
    for j, t in enumerate((q, k, v)):
        ops.copy2d(qkv.view(-1)[j * C:], 3 * C, t, C, N * T, C)
    assert torch.equal(qkv[:, C:2 * C], k)
    o2, lse2 = torch.zeros_like(o1), torch.zeros_like(lse1)
    f = qkv.view(-1)
    ops.attention_fwd(f, f[C:], f[2 * C:], N, T, T, C, H, causal, o2, lse2, rate, 7, ldq=3 * C, ldk=3 * C)
    assert torch.equal(o1, o2) and torch.equal(lse1, lse2)
    dqkv = torch.full((N * T, 3 * C), 9.0, device='cuda')
    d = dqkv.view(-1)
    ops.attention_bwd(f, f[C:], f[2 * C:], o2, do, lse2, N, T, T, C, H, causal, d, d[C:], d[2 * C:], ws, relu_grad=True,
                      dropout_rate=rate, seed=7, ldq=3 * C, ldk=3 * C)
    assert torch.equal(dqkv[:, :C], dq1) and torch.equal(dqkv[:, C:2 * C], dk1) and torch.equal(dqkv[:, 2 * C:], dv1)
    acc = torch.ones(N * T, C, device='cuda')
    ops.copy2d(acc, C, d[C:], 3 * C, N * T, C, True)                       # scatter-accumulate a column block
    assert torch.equal(acc, dk1 + 1.0)
