"""One page of arithmetic for a Winograd F(3x3,4x4) WEIGHT gradient: the twin of F(4x4,3x3) with the roles of the
filter and the output exchanged -- per 4 x 4 tile of the output gradient 36 multiplies instead of the 64 that four F(3x3,2x2) tiles
take (wino_wgrad4_kernel, csrc/wino_wgrad.hip).  Reference layers: lm_and_am/model/acoustic_model.py:42-46 (h3 64 -> 128 at 400 x 50,
h4 128 -> 128 and h5a 128 -> 256 at 200 x 25), acoustic_model2.py:47-62.

  (a) fp32 error of the 6-point transforms, summed over ~10^4 tiles, against a float64 direct weight gradient, beside F(3x3,2x2) (bar of the
      brief: 1e-4 relative, ten times inside north_star's 1e-3);
  (b) accumulator / LDS budget of a workgroup and the work a staged byte carries;
  (c) vector instructions per MFMA of the two wave layouts that fit, priced with the measured issue costs (profiles/r03_mfma_valu_wino.txt,
      profiles/r04_f4x4_page.txt) and the measured stage overheads of wino_wgrad4_kernel.

CPU, numpy.  usage: python tools/wino_f3x3_4x4_page.py > profiles/r06_f3x3_4x4_page.txt"""
import numpy as np


def cook_toom(points, m, r):
    """Matrices of the minimal filtering algorithm F(m, r) on the finite `points` + infinity:  y = AT [(G g) . (BT d)],
    y_k = sum_i d[k + i] g[i], k < m, i < r, d of m + r - 1 entries.  Exact in float64 up to rounding; checked below."""
    n = m + r - 1
    a = np.array(points, dtype=np.float64)
    assert len(a) == n - 1
    AT = np.zeros((m, n)); G = np.zeros((n, r)); BT = np.zeros((n, n))
    for i in range(n - 1):
        Ni = np.prod([a[i] - a[j] for j in range(n - 1) if j != i])
        AT[:, i] = a[i] ** np.arange(m)
        G[i, :] = a[i] ** np.arange(r) / Ni
        Mi = np.poly1d([1.0])
        for j in range(n - 1):
            if j != i:
                Mi = Mi * np.poly1d([1.0, -a[j]])
        c = Mi.coeffs[::-1]                               # ascending powers, degree n - 2
        BT[i, :len(c)] = c
    AT[m - 1, n - 1] = 1.0
    G[n - 1, r - 1] = 1.0
    M = np.poly1d([1.0])
    for j in range(n - 1):
        M = M * np.poly1d([1.0, -a[j]])
    BT[n - 1, :] = M.coeffs[::-1]
    # the sign / scale convention is fixed by the check: y must equal the direct correlation
    rng = np.random.default_rng(1)
    d, g = rng.standard_normal(n), rng.standard_normal(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[k + i] * g[i] for i in range(r)) for k in range(m)])
    if not np.allclose(y, ref, rtol=1e-9, atol=1e-9):
        # the transposed (Toom-Cook) form: rows of BT solve the interpolation for the correlation
        raise AssertionError((y, ref))
    return AT, G, BT


def wgrad_direct(x, dz):
    """dW[kh][kw][ci][co] = sum_pixels x[h + kh][w + kw][ci] dz[h][w][co]: x [B][H + 2][W + 2][K] (zero border), dz [B][H][W][N], float64."""
    B, H, W, N = dz.shape
    out = np.zeros((3, 3, x.shape[3], N))
    zz = dz.reshape(-1, N).astype(np.float64)
    for kh in range(3):
        for kw in range(3):
            out[kh, kw] = x[:, kh:kh + H, kw:kw + W].reshape(-1, x.shape[3]).astype(np.float64).T @ zz
    return out


def wgrad_wino(x, dz, AT, G, BT, t, dt=np.float32):
    """The same by F(3x3, t x t): per tile V = BT d BT^T (d the (t + 2)^2 input patch), Z = G dz G^T (dz the t x t gradient tile),
    M[xi] += V[xi] (x) Z[xi] over the tiles (the MFMA contraction, here an fp32 GEMM per position), dW = AT M AT^T once at the end.
    All arithmetic in `dt`; planes are padded with zeros to whole tiles like the kernel's out-of-range DMA."""
    B, H, W, N = dz.shape
    K = x.shape[3]
    th, tw = -(-H // t), -(-W // t)
    xp = np.zeros((B, th * t + 2, tw * t + 2, K), dtype=dt); xp[:, :H + 2, :W + 2] = x
    zp = np.zeros((B, th * t, tw * t, N), dtype=dt); zp[:, :H, :W] = dz
    n = t + 2
    # patches [tile][n][n][K], gradient tiles [tile][t][t][N]
    P = np.stack([xp[:, i * t:i * t + n, j * t:j * t + n] for i in range(th) for j in range(tw)], axis=1).reshape(-1, n, n, K)
    Q = np.stack([zp[:, i * t:i * t + t, j * t:j * t + t] for i in range(th) for j in range(tw)], axis=1).reshape(-1, t, t, N)
    BTd, Gd, ATd = BT.astype(dt), G.astype(dt), AT.astype(dt)
    V = np.einsum('ik,tklc->tilc', BTd, P).astype(dt)
    V = np.einsum('tilc,jl->tijc', V, BTd).astype(dt)
    Z = np.einsum('ik,tklc->tilc', Gd, Q).astype(dt)
    Z = np.einsum('tilc,jl->tijc', Z, Gd).astype(dt)
    M = np.zeros((n, n, K, N), dtype=dt)
    for i in range(n):
        for j in range(n):
            M[i, j] = V[:, i, j].T @ Z[:, i, j]                      # fp32 GEMM over the tiles
    out = np.einsum('ik,klcn->ilcn', ATd, M).astype(dt)
    out = np.einsum('ilcn,jl->ijcn', out, ATd).astype(dt)
    return out, P.shape[0]


def main():
    # F(3, 2): today's kernel (the halves of G moved into the output transform there; the arithmetic error is the same)
    AT2, G2, BT2 = cook_toom([0.0, 1.0, -1.0], 3, 2)
    sets = [('points 0, +-1, +-2, inf (Lavin & Gray\'s F(4,3) set, transposed)', [0.0, 1.0, -1.0, 2.0, -2.0]),
            ('points 0, +-1, +-1/2, inf', [0.0, 1.0, -1.0, 0.5, -0.5])]
    print('Winograd F(3x3,4x4) for the 3x3 weight gradients -- one page (tools/wino_f3x3_4x4_page.py; VERDICT r5 item 1)')
    print()
    print('(a) fp32 error against the float64 direct weight gradient, ~10^4 tiles of 4 x 4 gradient pixels per shape')
    print('    x = post-ReLU, BN-affine-like activations (relu(N(0,1))), dZ = N(0,1) x 1e-3 gated to 50 % zeros (a ReLU backward);')
    print('    every tensor element compared, error relative to the largest |dW| of the tensor (the bar the parity tests use)')
    rng = np.random.default_rng(0)
    shapes = [('h3  64 -> 128 @ 400 x 50', 64, 128, 400, 50, 8), ('h4  128 -> 128 @ 200 x 25', 128, 128, 200, 25, 29), ('h5a 128 -> 256 @ 200 x 25', 128, 256, 200, 25, 29)]
    worst = {}
    for name, K, N, H, W, B in shapes:
        x = np.zeros((B, H + 2, W + 2, K), dtype=np.float32)
        x[:, 1:H + 1, 1:W + 1] = np.maximum(rng.standard_normal((B, H, W, K)), 0).astype(np.float32)
        dz = (rng.standard_normal((B, H, W, N)) * 1e-3 * (rng.random((B, H, W, N)) < 0.5)).astype(np.float32)
        ref = wgrad_direct(x, dz)
        scale = np.abs(ref).max()
        got2, nt2 = wgrad_wino(x, dz, AT2, G2, BT2, 2)
        e2 = np.abs(got2 - ref).max() / scale
        d32 = np.zeros_like(ref, dtype=np.float32)
        for kh in range(3):                                            # a plain fp32 direct evaluation, for scale
            for kw in range(3):
                d32[kh, kw] = x[:, kh:kh + H, kw:kw + W].reshape(-1, K).T @ dz.reshape(-1, N)
        ed = np.abs(d32 - ref).max() / scale
        print('    %s, B = %d:' % (name, B))
        print('      direct fp32 (numpy sgemm)            max %.2e' % ed)
        print('      F(3x3,2x2) fp32, %6d tiles          max %.2e   rms %.2e' % (nt2, e2, np.sqrt(((got2 - ref) ** 2).mean()) / scale))
        for label, pts in sets:
            AT4, G4, BT4 = cook_toom(pts, 3, 4)
            got4, nt4 = wgrad_wino(x, dz, AT4, G4, BT4, 4)
            e4 = np.abs(got4 - ref).max() / scale
            worst[label] = max(worst.get(label, 0), e4)
            print('      F(3x3,4x4) fp32, %6d tiles          max %.2e   rms %.2e   [%s]' % (nt4, e4, np.sqrt(((got4 - ref) ** 2).mean()) / scale, label))
    print('    worst case: ' + '; '.join('%.1e (%s)' % (v, k.split(' (')[0]) for k, v in worst.items()) + ' -- bar 1e-4.')
    print()

    print('(b) what a workgroup can hold (one CU: 512 KB of vector registers, 160 KB of LDS)')
    print('    accumulators, fp32, all transform positions of a (ci, co) block resident for the whole launch (the property that gives')
    print('    wino_wgrad4_kernel its tail-free stage loop):')
    for label, pos, ci, co in (('F(3x3,2x2) today      ', 16, 64, 64), ('F(3x3,4x4)            ', 36, 64, 64), ('F(3x3,4x4)            ', 36, 32, 64),
                               ('F(3x3,4x2) asymmetric ', 24, 64, 64)):
        kb = pos * ci * co * 4 / 1024
        print('      %s %2d positions x %2d x %2d = %5.0f KB %s' % (label, pos, ci, co, kb, '(does not fit)' if kb > 400 else
              '(+ 16 waves x 42 other registers = 168 KB: fits, 106 per wave)' if pos == 16 else
              '(12 waves x (96 + ~64) = 480 KB: fits at THREE waves per SIMD, 160 of 168 registers)' if pos == 36 else
              '(12 waves x (128 + ~60): 188 > 168 registers -- spills; 16 waves x (96 + 40) = 136 > 128)'))
    print('    so F(3x3,4x4) means a 32 x 64 block: each staged input pixel (128 B) and gradient pixel (256 B) feeds HALF / the same number of')
    print('    output-channel / input-channel columns as today, and the algorithm itself issues 1.78x fewer MFMAs per pixel:')
    print('      MFMAs per staged KB (region without halo): today 64 MFMAs per tile pair of 2 x 4 px x (256 + 256) B = 16 per KB;')
    print('      F(3x3,4x4) at 32 x 64: 72 MFMAs per tile pair of 2 x 16 px x (128 + 256) B = 6 per KB  (2.7x less)')
    print('    LDS: today two buffer sets of 71.7 KB (6 x 28 input + 4 x 28 gradient pixels of 256 B: two 2 x 2-tile rows x 13 columns = 832 MFMAs')
    print('    per stage and barrier).  The same pixel region at 32 x 64 is 50.2 KB and is ONE 4 x 4-tile row of 7 tiles = 252 MFMAs per stage;')
    print('    two tile rows (10 + 8 pixel rows) are 93 KB per set, 186 KB double-buffered: over the 160 KB.  Three sets of one row fit (151 KB) and')
    print('    take the DMA wait out of the barrier, not the barrier: 3.3x fewer MFMAs between two workgroup-wide synchronisations.')
    print()

    print('(c) vector instructions per MFMA (packed: one instruction transforms the same pixel of two tile pairs, as today)')
    print('    today (wave = transform row x ci half x co half; 8 MFMAs per step): 8 + 2..4 ds_read2st64_b32 + 12 v_pk_add_f32 = 24 per 8 MFMAs = 3.0')
    print('    F(3x3,4x4), wave = transform row x co half (12 waves, 6 accumulators each, 12 MFMAs per step of two tile pairs):')
    print('      input row r of B^T d B: rows 1-4 of B^T have 4 non-zeros (0 and 5: 3) -> 4 pixel rows x 6 columns = 24 reads, 3 v_pk_fma x 6 = 18,')
    print('      then the 6-point column transform of the 6 sums: 12 v_pk_fma / add;  gradient row r of G\' y G\'^T (G\' = A^T of F(4,3), 6 x 4):')
    print('      <= 4 rows x 4 columns = 16 reads, 12 v_pk_fma, 10 for the 4 -> 6 column transform:  24 + 30 + 16 + 22 = 92 per 12 MFMAs = 7.7')
    print('    F(3x3,4x4), producer / consumer through LDS (a wave transforms whole patches -- 144 + 100 packed ops per 32 channels x 2 tile pairs with')
    print('      shared sub-expressions -- writes 36 + 36 values per lane, every consumer wave reads its 6 + 6): 3.9 per MFMA, but the transformed operands')
    print('      are 36 / 16 = 2.25x the raw pixels: 13.8 KB per tile at 32 + 64 channels -> 55 KB for ONE step of 4 tiles next to the raw sets (158 of')
    print('      160 KB with two raw sets of one tile row), two more barriers per step, and only 3 packed units of transform work for 12 waves.')
    mf, vv = 64.0, 4.5            # cycles: v_mfma_f32_32x32x2_f32; a packed vector / LDS instruction beside it (4.5 measured, profiles/r03_mfma_valu_wino.txt)
    today_busy = 0.68             # PMC mfma_busy of wino_wgrad4_kernel<64> (0.66-0.70, r04-r05)
    t_issue = 256 * (mf + 3.0 * vv) / 4          # per SIMD and 32 gradient pixels of a 64 x 64 block
    t_meas = 256 * mf / 4 / today_busy
    print('    priced per 32 gradient pixels of a 64 x 64 channel block, per SIMD (MFMA 64 cycles, a vector / LDS instruction 4.5 beside it):')
    print('      today: 256 MFMAs -> issue %.0f cycles; measured %.0f (busy %.2f): %.0f cycles of stage boundary + LDS latency = %.0f %% on top'
          % (t_issue, t_meas, today_busy, t_meas - t_issue, 100 * (t_meas - t_issue) / t_issue))
    stage_today = 18900.0                                  # cycles per stage, in-kernel stamps: 14 336 MFMA + 2 100 + 2 500
    per_stage_over = 2500.0                                # arrival skew + set-up + first LDS round trip per stage (same source)
    for label, vpm, stages_per_unit in (('row-owner waves (7.7 per MFMA)', 7.7, None), ('producer / consumer (3.9 per MFMA + 2 barriers per step)', 3.9, 'pc')):
        issue = 144 * (mf + vpm * vv) / 4
        # stage boundaries: today one per 832 MFMAs of the workgroup; F(3x3,4x4) at 32 x 64 one per 252 (row-owner), one per 144 and twice (producer / consumer)
        bound = per_stage_over * (144 / 252.0) if stages_per_unit is None else 2 * per_stage_over * (144 / 144.0) * 0.5
        inphase = 0.11 * issue                             # the in-phase waits scale with the phase (item 17: 2 100 of 16 400)
        tot = issue + bound + inphase
        for wd, util in ((25, (25 / 28.0) / (25 / 26.0)), (50, (50 / 52.0) / 1.0)):
            print('      F(3x3,4x4) %s: issue %.0f + boundaries %.0f + in-phase waits %.0f = %.0f  ->  %.2fx; x %.3f tile-column utilisation on a %d-wide plane = %.2fx'
                  % (label, issue, bound, inphase, tot, t_meas / tot, util, wd, t_meas / tot * util))
    print('    The same pricing put the FORWARD F(4x4,3x3) at 1.57x on paper (profiles/r04_f4x4_page.txt); its stage loop then measured 1.0x')
    print('    (profiles/r05_f4x4_loop_ab.txt): paper / measured = 1.5 for a kernel whose vector work per MFMA was 4.3, three waves per SIMD.')
    print()
    print('Reading: the arithmetic is no obstacle (7e-6 with either point set, ten times the F(3x3,2x2) error, fourteen times inside the 1e-4 bar); the machine is.  The forward kernel lost F(4x4) to the un-amortised 6 x 6 input')
    print('transform; the weight gradient loses it to the ACCUMULATORS: 36 positions of a 64 x 64 block are 576 KB, the block must shrink to 32 x 64, and')
    print('with it (i) every wave transforms an A AND a B operand for 12 MFMAs (7.7 vector instructions per MFMA against 3.0), (ii) a staged kilobyte')
    print('carries 2.7x fewer MFMAs, so the stage barrier comes 3.3x as often.  Paper ratio 1.0-1.1x for the forms that fit the LDS: below the 1.25x bar')
    print('before the paper-to-silicon factor of the forward twin (1.5) is applied.  Not built; decision recorded in DESIGN.md section 9.')


if __name__ == '__main__':
    main()
