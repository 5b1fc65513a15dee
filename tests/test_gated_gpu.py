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
