"""asr_tap_gemm_gated: the data-gradient GEMM of cell k with cell k-1's backward prologue (pool -> BN -> ReLU backward,
per-channel sums) fused into its epilogue, against the two-pass path it replaces (asr_tap_gemm[_pw] data-gradient, then
asr_cell_bwd_pre): dZ bitwise equal (same arithmetic per element), the three channel sums equal to 1e-5 of their scale
(different, fixed summation orders), for every pool mode, 3x3 (pre-arranged weights) and 1x1 (plain weights) GEMMs, with
and without an existing gradient contribution, and bitwise reproducible."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pool,ntaps,acc,B,Hq,Wq,N,K", [
    (0, 9, 0, 2, 9, 7, 32, 64), (0, 9, 1, 1, 20, 25, 128, 128), (0, 1, 0, 2, 9, 7, 256, 32), (0, 1, 1, 3, 8, 6, 64, 128),
    (1, 9, 0, 2, 6, 5, 32, 64), (1, 9, 1, 1, 10, 12, 64, 128), (2, 9, 0, 2, 6, 5, 32, 64), (2, 9, 1, 3, 8, 13, 64, 128),
    (2, 9, 0, 1, 50, 25, 128, 128)])
def test_gated_data_gradient_equals_gemm_then_prologue(pool, ntaps, acc, B, Hq, Wq, N, K):
    from asr_dfcnn_transformer_amd import ops
    from asr_dfcnn_transformer_amd.ops import Plane
    g = torch.Generator(device='cuda').manual_seed(1 + pool + ntaps + acc)
    rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)
    # cell k: input y(k-1) [Hq x Wq x N], output K channels; its dZ(k) is the GEMM's A operand
    dzk = Plane(B, Hq, Wq, K); dzk.set_interior(rnd(B, Hq, Wq, K))
    w = rnd(3 if ntaps == 9 else 1, 3 if ntaps == 9 else 1, N, K) * 0.1
    gh, gw = (Hq, Wq) if pool == 0 else (2 * Hq, 2 * Wq)
    a = Plane(B, gh, gw, N)
    av = torch.relu(rnd(B, gh, gw, N))
    if pool == 2:                                   # ties inside windows: the FIRST maximum takes the gradient
        av[:, 0:2, 0:2, :] = 0.7
        av[:, 2, 2, :] = av[:, 2, 3, :]
    a.set_interior(av)
    sc = 1.0 + 0.2 * rnd(N); sh = 0.1 * rnd(N)
    sc[0] = -0.5                                    # a negative BN scale turns the max-pool arg-max into an arg-min of a
    prev = Plane(B, Hq, Wq, N)
    if acc:
        prev.set_interior(rnd(B, Hq, Wq, N))
    d = ops.gemm_desc(dzk.NP, K, N, K, K, 0, N, ntaps=ntaps, B=B, H=Hq, W=Wq, wmode=1, accumulate=acc)
    wop = ops.arrange_weights(w, 9, K, N, K, 1) if ntaps == 9 else w.reshape(N, K).contiguous()
    if ntaps == 1:
        wop = w.reshape(N, K)                       # [cin = N][cout = K]: the data-gradient view of a 1x1 HWIO kernel
    # reference: two passes
    dy = Plane(B, Hq, Wq, N); dy.interior().copy_(prev.interior())
    (ops.tap_gemm_pw if ntaps == 9 else ops.tap_gemm)(d, dzk, wop, None, None, None, None, dy)
    dz_ref = Plane(B, gh, gw, N)
    dsc_r, dsh_r, db_r = torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda')
    ws = torch.zeros(ops.cell_bwd_pre_workspace(B, gh, gw, N) // 4 + 64, device='cuda')
    ops.cell_bwd_pre(dy, 1 if pool else 0, a, sc, sh, pool, dz_ref, dsc_r, dsh_r, db_r, ws)
    # fused
    dz = Plane(B, gh, gw, N)
    dz.buf.fill_(7.0)                               # every interior pixel must be overwritten ...
    dz.view()[:, 0].zero_(); dz.view()[:, :, 0].zero_(); dz.buf[:dz.G * N].zero_(); dz.buf[-dz.G * N:].zero_()   # ... borders stay 0
    dsc, dsh, db = torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda')
    ws2 = torch.zeros(ops.tap_gemm_gated_workspace(d) // 4 + 64, device='cuda')
    ops.tap_gemm_gated(d, dzk, wop, ntaps == 9, pool, a, sc, sh, prev if acc else None, dz, dsc, dsh, db, ws2)
    torch.cuda.synchronize()
    assert torch.equal(dz.interior(), dz_ref.interior())
    assert float(dz.view()[:, 0].abs().max()) == 0 and float(dz.view()[:, :, 0].abs().max()) == 0
    for got, want, name in ((dsc, dsc_r, 'dscale'), (dsh, dsh_r, 'dshift'), (db, db_r, 'dbias')):
        err = (got.double() - want.double()).abs().max().item()
        assert err <= 1e-5 * max(1.0, want.abs().max().item()), (name, err)
    again = Plane(B, gh, gw, N)
    dsc2, dsh2, db2 = torch.zeros_like(dsc), torch.zeros_like(dsh), torch.zeros_like(db)
    ops.tap_gemm_gated(d, dzk, wop, ntaps == 9, pool, a, sc, sh, prev if acc else None, again, dsc2, dsh2, db2, ws2)
    torch.cuda.synchronize()
    assert torch.equal(again.interior(), dz.interior()) and torch.equal(dsc, dsc2) and torch.equal(dsh, dsh2) and torch.equal(db, db2)


@pytest.mark.parametrize("pool,acc,B,Hq,Wq,N,K", [
    (0, 0, 2, 20, 50, 64, 128), (2, 1, 2, 10, 25, 64, 64),          # wino11_kernel, N % 64 == 0
    (2, 0, 2, 20, 50, 32, 64), (1, 1, 3, 10, 25, 32, 64), (0, 0, 1, 40, 25, 96, 32),      # wino11_kernel: N % 64 == 32
    (2, 0, 2, 125, 25, 128, 128), (0, 1, 1, 25, 25, 64, 64), (1, 0, 2, 7, 25, 32, 64)])    # odd heights (round 4)
def test_gated_winograd_data_gradient_matches_the_direct_one(pool, acc, B, Hq, Wq, N, K):
    """The same fused prologue behind the Winograd data-gradient kernels (prearranged == 2: wino11_kernel's gated epilogues) against
    the direct kernel (prearranged == 1): dZ to rounding (the gate decisions -- ReLU sign, max-pool winner -- come from the stored
    activations, not from the gradient, so they are identical), channel sums to 1e-4 of their scale, borders untouched,
    bitwise reproducible."""
    from asr_dfcnn_transformer_amd import ops
    from asr_dfcnn_transformer_amd.ops import Plane
    g = torch.Generator(device='cuda').manual_seed(40 + pool + acc)
    rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)
    dzk = Plane(B, Hq, Wq, K); dzk.set_interior(rnd(B, Hq, Wq, K))
    w = rnd(3, 3, N, K) * 0.1
    gh, gw = (Hq, Wq) if pool == 0 else (2 * Hq, 2 * Wq)
    a = Plane(B, gh, gw, N); a.set_interior(torch.relu(rnd(B, gh, gw, N)))
    sc = 1.0 + 0.2 * rnd(N); sh = 0.1 * rnd(N)
    prev = Plane(B, Hq, Wq, N)
    if acc:
        prev.set_interior(rnd(B, Hq, Wq, N))
    d = ops.gemm_desc(dzk.NP, K, N, K, K, 0, N, ntaps=9, B=B, H=Hq, W=Wq, wmode=1, accumulate=acc)
    assert ops.winograd_supported(d)
    ws = torch.zeros(ops.tap_gemm_gated_workspace(d) // 4 + 64, device='cuda')
    out = {}
    for pre, wop in ((1, ops.arrange_weights(w, 9, K, N, K, 1)), (2, ops.winograd_weights(w, K, N, K, 1)), (2, None)):
        if wop is None:
            wop = out[2][4]
        dz = Plane(B, gh, gw, N)
        sums = [torch.zeros(N, device='cuda') for _ in range(3)]
        ops.tap_gemm_gated(d, dzk, wop, pre, pool, a, sc, sh, prev if acc else None, dz, sums[0], sums[1], sums[2], ws)
        torch.cuda.synchronize()
        if pre in out:                                   # second Winograd run: bitwise reproducible
            assert torch.equal(dz.interior(), out[pre][0].interior())
            assert all(torch.equal(x, y) for x, y in zip(sums, out[pre][1:4]))
        else:
            out[pre] = (dz, sums[0], sums[1], sums[2], wop)
    ref, got = out[1], out[2]
    scale = max(1.0, ref[0].interior().abs().max().item())
    assert (got[0].interior() - ref[0].interior()).abs().max().item() < 3e-5 * scale
    assert got[0].border_abs_max() == 0.0
    for i, name in ((1, 'dscale'), (2, 'dshift'), (3, 'dbias')):
        err = (got[i].double() - ref[i].double()).abs().max().item()
        assert err <= 1e-4 * max(1.0, ref[i].abs().max().item()), (name, err)


@pytest.mark.parametrize("B,H,W,cin,N,K2,acc", [(2, 16, 50, 32, 64, 64, 0), (1, 40, 100, 32, 64, 128, 1), (2, 8, 52, 64, 128, 32, 0)])
def test_poolmax_compact_form_gives_the_bits_of_the_activation_plane_form(B, H, W, cin, N, K2, acc):
    """A max-pooled cell in the compact form (asr_tap_gemm_wino_poolmax / asr_tap_gemm_gated_poolmax: activation at each window's
    maximum + its 2-bit position instead of the pre-pool activation plane) against the form it replaces (asr_tap_gemm_wino_pool /
    asr_tap_gemm_gated with pool = 2): pooled output, dZ and the three channel sums bit for bit; negative BN scales (arg-max of the BN
    output = arg-min of the activation) and windows of equal values (ReLU zeros: the FIRST position takes the gradient) included."""
    from asr_dfcnn_transformer_amd import ops
    from asr_dfcnn_transformer_amd.ops import Plane
    g = torch.Generator(device='cuda').manual_seed(B + H + W + N)
    rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)
    x = Plane(B, H, W, cin); x.set_interior(rnd(B, H, W, cin))
    w = rnd(3, 3, cin, N) * (2.0 / (9 * cin)) ** 0.5
    bias = rnd(N) * 0.1 - 0.3                        # plenty of ReLU zeros
    sc = 1.0 + 0.2 * rnd(N); sh = 0.1 * rnd(N)
    sc[0] = -0.5; sc[N - 1] = -1.25
    fd = ops.gemm_desc(x.NP, cin, N, cin, N, N, 0, ntaps=9, B=B, H=H, W=W, relu=1)
    H2, W2 = H // 2, W // 2
    w2 = rnd(3, 3, N, K2) * 0.05
    dzk = Plane(B, H2, W2, K2); dzk.set_interior(rnd(B, H2, W2, K2))
    bd = ops.gemm_desc(dzk.NP, K2, N, K2, K2, 0, N, ntaps=9, B=B, H=H2, W=W2, wmode=1, accumulate=acc)
    assert ops.poolmax_supported(fd, bd)
    wt = ops.winograd_weights(w, cin, N, N, 0)
    wtb = ops.winograd_weights(w2, K2, N, K2, 1)
    # the form with the activation plane
    a, y1 = Plane(B, H, W, N), Plane(B, H2, W2, N)
    ops.tap_gemm_wino_pool(fd, x, wt, bias, sc, sh, a, 2, y1)
    assert ops.last_kernel() == 'wino11_kernel<0, 2>'
    prev = Plane(B, H2, W2, N)
    if acc:
        prev.set_interior(rnd(B, H2, W2, N))
    ws = torch.zeros(ops.tap_gemm_gated_workspace(bd) // 4 + 64, device='cuda')
    dz1 = Plane(B, H, W, N)
    s1 = [torch.zeros(N, device='cuda') for _ in range(3)]
    ops.tap_gemm_gated(bd, dzk, wtb, 2, 2, a, sc, sh, prev if acc else None, dz1, s1[0], s1[1], s1[2], ws)
    assert ops.last_kernel() == 'wino11_kernel<1, 8>'
    # the compact form
    y2, amax = Plane(B, H2, W2, N), Plane(B, H2, W2, N)
    idx = ops.poolmax_index(B, H2, W2, N)
    ops.tap_gemm_wino_poolmax(fd, x, wt, bias, sc, sh, y2, amax, idx)
    assert ops.last_kernel() == 'wino11_kernel<0, 9>'
    assert torch.equal(y1.buf, y2.buf)
    # a_max / index against the activation plane: first maximum of sc * a + sh in row-major window order
    av = a.interior().reshape(B, H2, 2, W2, 2, N).permute(0, 1, 3, 2, 4, 5).reshape(B, H2, W2, 4, N)
    bn = av * sc + sh                                  # (same fma? the comparison below is on positions: recompute exactly)
    bn = torch.addcmul(sh.expand_as(av), av, sc.expand_as(av))
    pos = torch.zeros(B, H2, W2, N, dtype=torch.long, device='cuda'); best = bn[..., 0, :].clone()
    for k in (1, 2, 3):
        better = bn[..., k, :] > best
        pos = torch.where(better, torch.full_like(pos, k), pos); best = torch.where(better, bn[..., k, :], best)
    want_amax = torch.gather(av, 3, pos.unsqueeze(3)).squeeze(3)
    # (torch's addcmul may round differently from the kernel's fma in a near-tie: compare where the margin is clear, the bits of dZ and
    #  of the sums below are the real check)
    srt = bn.sort(dim=3, descending=True).values
    clear = (srt[..., 0, :] - srt[..., 1, :]).abs() > 1e-5
    assert torch.equal(amax.interior()[clear], want_amax[clear])
    words = idx.view(B, H2 + 1, W2 + 1, N // 32, 2)[:, 1:, 1:]
    shifts = torch.arange(32, device='cuda', dtype=torch.int32)
    bits0 = ((words[..., 0].unsqueeze(-1) >> shifts) & 1).reshape(B, H2, W2, N)
    bits1 = ((words[..., 1].unsqueeze(-1) >> shifts) & 1).reshape(B, H2, W2, N)
    got_pos = (bits0 + 2 * bits1).long()
    assert torch.equal(got_pos[clear], pos[clear])
    dz2 = Plane(B, H, W, N)
    dz2.buf.fill_(7.0)
    dz2.view()[:, 0].zero_(); dz2.view()[:, :, 0].zero_(); dz2.buf[:dz2.G * N].zero_(); dz2.buf[-dz2.G * N:].zero_()
    s2 = [torch.zeros(N, device='cuda') for _ in range(3)]
    ops.tap_gemm_gated_poolmax(bd, dzk, wtb, H, W, amax, idx, sc, sh, prev if acc else None, dz2, s2[0], s2[1], s2[2], ws)
    assert ops.last_kernel() == 'wino11_kernel<1, 10>'
    torch.cuda.synchronize()
    assert torch.equal(dz1.interior(), dz2.interior())
    assert float(dz2.view()[:, 0].abs().max()) == 0 and float(dz2.view()[:, :, 0].abs().max()) == 0
    for u, v in zip(s1, s2):
        assert torch.equal(u, v)


@pytest.mark.parametrize("B,H,W,cin,N,K2,acc", [(2, 16, 50, 32, 64, 64, 0), (1, 40, 100, 32, 64, 128, 1), (2, 8, 52, 64, 128, 32, 0), (3, 20, 100, 32, 32, 64, 0)])
def test_poolavg_compact_form_against_the_activation_plane_form(B, H, W, cin, N, K2, acc):
    """An AVERAGE-pooled cell (the "maxpool" of acoustic_model2.py:116-124) in the compact form (round 5: asr_tap_gemm_wino_poolavg /
    asr_tap_gemm_gated_poolavg -- the window's activation sum + the ReLU sign of each position instead of the pre-pool activation plane)
    against the form it replaces (asr_tap_gemm_wino_pool / asr_tap_gemm_gated with pool = 1): pooled output, dZ, dshift and dbias bit for
    bit; dscale to 1e-5 of its scale (one multiply-add per window instead of four); the stored sums and signs against the plane."""
    from asr_dfcnn_transformer_amd import ops
    from asr_dfcnn_transformer_amd.ops import Plane
    g = torch.Generator(device='cuda').manual_seed(B + H + W + N + 1)
    rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)
    x = Plane(B, H, W, cin); x.set_interior(rnd(B, H, W, cin))
    w = rnd(3, 3, cin, N) * (2.0 / (9 * cin)) ** 0.5
    bias = rnd(N) * 0.1 - 0.3                        # plenty of ReLU zeros
    sc = 1.0 + 0.2 * rnd(N); sh = 0.1 * rnd(N)
    sc[0] = -0.5
    fd = ops.gemm_desc(x.NP, cin, N, cin, N, N, 0, ntaps=9, B=B, H=H, W=W, relu=1)
    H2, W2 = H // 2, W // 2
    w2 = rnd(3, 3, N, K2) * 0.05
    dzk = Plane(B, H2, W2, K2); dzk.set_interior(rnd(B, H2, W2, K2))
    bd = ops.gemm_desc(dzk.NP, K2, N, K2, K2, 0, N, ntaps=9, B=B, H=H2, W=W2, wmode=1, accumulate=acc)
    assert ops.poolmax_supported(fd, bd)
    wt = ops.winograd_weights(w, cin, N, N, 0)
    wtb = ops.winograd_weights(w2, K2, N, K2, 1)
    a, y1 = Plane(B, H, W, N), Plane(B, H2, W2, N)
    ops.tap_gemm_wino_pool(fd, x, wt, bias, sc, sh, a, 1, y1)
    assert ops.last_kernel() == 'wino11_kernel<0, 3>'
    prev = Plane(B, H2, W2, N)
    if acc:
        prev.set_interior(rnd(B, H2, W2, N))
    ws = torch.zeros(ops.tap_gemm_gated_workspace(bd) // 4 + 64, device='cuda')
    dz1 = Plane(B, H, W, N)
    s1 = [torch.zeros(N, device='cuda') for _ in range(3)]
    ops.tap_gemm_gated(bd, dzk, wtb, 2, 1, a, sc, sh, prev if acc else None, dz1, s1[0], s1[1], s1[2], ws)
    assert ops.last_kernel() == 'wino11_kernel<1, 7>'
    y2, asum = Plane(B, H2, W2, N), Plane(B, H2, W2, N)
    idx = ops.poolavg_index(B, H2, W2, N)
    ops.tap_gemm_wino_poolavg(fd, x, wt, bias, sc, sh, y2, asum, idx)
    assert ops.last_kernel() == 'wino11_kernel<0, 13>'
    assert torch.equal(y1.buf, y2.buf)
    av = a.interior().reshape(B, H2, 2, W2, 2, N).permute(0, 1, 3, 2, 4, 5).reshape(B, H2, W2, 4, N)
    want_sum = (av[..., 0, :] + av[..., 1, :]) + (av[..., 2, :] + av[..., 3, :])
    assert torch.equal(asum.interior(), want_sum)
    assert asum.border_abs_max() == 0.0
    words = idx.view(B, H2 + 1, W2 + 1, N // 32, 4)[:, 1:, 1:]
    shifts = torch.arange(32, device='cuda', dtype=torch.int32)
    for k in range(4):
        bits = ((words[..., k].unsqueeze(-1) >> shifts) & 1).reshape(B, H2, W2, N)
        assert torch.equal(bits.bool(), av[..., k, :] > 0)
    dz2 = Plane(B, H, W, N)
    dz2.buf.fill_(7.0)
    dz2.view()[:, 0].zero_(); dz2.view()[:, :, 0].zero_(); dz2.buf[:dz2.G * N].zero_(); dz2.buf[-dz2.G * N:].zero_()
    s2 = [torch.zeros(N, device='cuda') for _ in range(3)]
    ops.tap_gemm_gated_poolavg(bd, dzk, wtb, H, W, asum, idx, sc, sh, prev if acc else None, dz2, s2[0], s2[1], s2[2], ws)
    assert ops.last_kernel() == 'wino11_kernel<1, 14>'
    torch.cuda.synchronize()
    assert torch.equal(dz1.interior(), dz2.interior())
    assert float(dz2.view()[:, 0].abs().max()) == 0 and float(dz2.view()[:, :, 0].abs().max()) == 0
    assert torch.equal(s1[1], s2[1]) and torch.equal(s1[2], s2[2])                      # dshift, dbias: the same sums in the same order
    assert (s1[0].double() - s2[0].double()).abs().max().item() <= 1e-5 * max(1.0, s1[0].abs().max().item())


@pytest.mark.parametrize("B,H,W,Cc,K", [(4, 40, 25, 256, 128), (2, 200, 25, 256, 128), (8, 50, 25, 64, 1536), (1, 200, 25, 128, 128)])
def test_dense_data_gradient_with_the_cell_backward_in_its_epilogue(B, H, W, Cc, K):
    """asr_tap_gemm_gated_dense (round 5): the data-gradient of a dense layer fed by the flattened output of an un-pooled cell
    (reshape -> tf.layers.dense: acoustic_model.py:48-50, acoustic_model2.py:62-66) with that cell's BN / ReLU backward in the GEMM's
    epilogue, against the two passes it replaces (asr_tap_gemm data-gradient into the dense layout, then asr_cell_bwd_pre with layout 2):
    dZ bitwise equal, the three channel sums to 1e-5 of their scale, borders untouched, bitwise reproducible -- and against float64."""
    from asr_dfcnn_transformer_amd import ops
    from asr_dfcnn_transformer_amd.ops import Plane
    g = torch.Generator(device='cuda').manual_seed(3)
    rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)
    rows, cin = B * H, W * Cc
    dzd = rnd(rows, K)                                   # dL/d(dense output)
    w = rnd(cin, K) * 0.05                               # the dense kernel [cin][cout]
    a = Plane(B, H, W, Cc); av = torch.relu(rnd(B, H, W, Cc)); a.set_interior(av)
    sc = 1.0 + 0.2 * rnd(Cc); sh = 0.1 * rnd(Cc); sc[1] = -0.4
    d = ops.gemm_desc(rows, K, cin, K, K, 0, cin, ntaps=1, wmode=1)
    assert ops.tap_gemm_gated_dense_supported(d, H, W, Cc)
    # reference: two passes
    dflat = torch.zeros(rows, cin, device='cuda')
    ops.tap_gemm(d, dzd, w, None, None, None, None, dflat)
    dz_ref = Plane(B, H, W, Cc)
    r = [torch.zeros(Cc, device='cuda') for _ in range(3)]
    ws = torch.zeros(ops.cell_bwd_pre_workspace(B, H, W, Cc) // 4 + 64, device='cuda')
    ops.cell_bwd_pre(dflat, 2, a, sc, sh, 0, dz_ref, r[0], r[1], r[2], ws)
    # fused
    dz = Plane(B, H, W, Cc)
    dz.buf.fill_(7.0)
    dz.view()[:, 0].zero_(); dz.view()[:, :, 0].zero_(); dz.buf[:dz.G * Cc].zero_(); dz.buf[-dz.G * Cc:].zero_()
    f = [torch.zeros(Cc, device='cuda') for _ in range(3)]
    ws2 = torch.zeros(ops.tap_gemm_gated_dense_workspace(d, W, Cc) // 4 + 64, device='cuda')
    ws2.fill_(float('nan'))                              # every partial the fold reads is written by the launch
    ops.tap_gemm_gated_dense(d, dzd, w, a, sc, sh, dz, f[0], f[1], f[2], ws2)
    torch.cuda.synchronize()
    assert torch.equal(dz.interior(), dz_ref.interior())
    assert float(dz.view()[:, 0].abs().max()) == 0 and float(dz.view()[:, :, 0].abs().max()) == 0
    for got, want, name in zip(f, r, ('dscale', 'dshift', 'dbias')):
        err = (got.double() - want.double()).abs().max().item()
        assert err <= 1e-5 * max(1.0, want.abs().max().item()), (name, err)
    # float64: dy = dzd . w^T in the dense layout -> [B, H, W, C]; dZ = dy * scale where a > 0
    dy64 = (dzd.double() @ w.double().t()).reshape(B, H, W, Cc)
    dz64 = torch.where(av > 0, dy64 * sc.double(), torch.zeros_like(dy64))
    assert (dz.interior().double() - dz64).abs().max().item() <= 2e-5 * max(1.0, dz64.abs().max().item())
    assert (f[1].double() - dy64.sum((0, 1, 2))).abs().max().item() <= 1e-4 * max(1.0, dy64.sum((0, 1, 2)).abs().max().item())
    assert (f[0].double() - (dy64 * av.double()).sum((0, 1, 2))).abs().max().item() <= 1e-4 * max(1.0, (dy64 * av.double()).sum((0, 1, 2)).abs().max().item())
    again = Plane(B, H, W, Cc)
    f2 = [torch.zeros(Cc, device='cuda') for _ in range(3)]
    ops.tap_gemm_gated_dense(d, dzd, w, a, sc, sh, again, f2[0], f2[1], f2[2], ws2)
    torch.cuda.synchronize()
    assert torch.equal(again.interior(), dz.interior()) and all(torch.equal(x, y) for x, y in zip(f, f2))


def test_dense_gate_is_refused_where_the_kernel_does_not_apply():
    from asr_dfcnn_transformer_amd import ops
    d = ops.gemm_desc(8 * 4, 16, 3 * 24, 16, 16, 0, 3 * 24, ntaps=1, wmode=1)          # 24 channels: not a multiple of 32
    assert not ops.tap_gemm_gated_dense_supported(d, 4, 3, 24)
    d = ops.gemm_desc(2 * 4, 64, 2 * 64, 64, 64, 0, 2 * 64, ntaps=1, wmode=1)          # a handful of tiles: the small-GEMM kernel, two passes
    assert not ops.tap_gemm_gated_dense_supported(d, 4, 2, 64)
