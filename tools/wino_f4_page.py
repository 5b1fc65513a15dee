"""One page of arithmetic for Winograd F(4x4,3x3) on the 128-wide 200 x 25 layers: fp32 error against a float64
direct convolution beside F(2x2,3x3), and the instruction budget per MFMA.  CPU, numpy.  usage: python tools/wino_f4_page.py"""
import numpy as np

# Lavin & Gray's matrices
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def wino(x, w, BT, G, AT, m, dt):
    """x [H][W][C], w [3][3][C][N]; valid convolution of the interior tiles, all arithmetic in dtype dt."""
    H, W, C = x.shape
    N = w.shape[3]
    a = m + 2
    U = np.einsum('ik,klcn,jl->ijcn', G.astype(dt), w.astype(dt), G.astype(dt)).astype(dt)
    th, tw = (H - 2) // m, (W - 2) // m
    out = np.zeros((th * m, tw * m, N), dtype=dt)
    for i in range(th):
        for j in range(tw):
            d = x[i * m:i * m + a, j * m:j * m + a].astype(dt)
            V = np.einsum('ik,klc,jl->ijc', BT.astype(dt), d, BT.astype(dt)).astype(dt)
            M = np.einsum('ijc,ijcn->ijn', V, U).astype(dt)
            out[i * m:(i + 1) * m, j * m:(j + 1) * m] = np.einsum('ik,kln,jl->ijn', AT.astype(dt), M, AT.astype(dt)).astype(dt)
    return out


def direct(x, w):
    H, W, C = x.shape
    out = np.zeros((H - 2, W - 2, w.shape[3]))
    for kh in range(3):
        for kw in range(3):
            out += np.einsum('hwc,cn->hwn', x[kh:kh + H - 2, kw:kw + W - 2].astype(np.float64), w[kh, kw].astype(np.float64))
    return out


rng = np.random.default_rng(0)
C, N = 128, 128
x = np.maximum(rng.standard_normal((26, 26, C)), 0).astype(np.float32)          # post-ReLU activations
w = (rng.standard_normal((3, 3, C, N)) * (2.0 / (9 * C)) ** 0.5).astype(np.float32)
ref = direct(x, w)
for name, (BT, G, AT, m) in (('F(2x2,3x3)', (BT2, G2, AT2, 2)), ('F(4x4,3x3)', (BT4, G4, AT4, 4))):
    got = wino(x, w, BT, G, AT, m, np.float32)
    r = ref[:got.shape[0], :got.shape[1]]
    print('%s fp32: max |err| / max |ref| = %.2e   (rms %.2e)' % (name, np.abs(got - r).max() / np.abs(r).max(), np.sqrt(((got - r) ** 2).mean()) / np.abs(r).max()))
print()
print('Instruction budget, one wave = one row of the transform, 32 tiles x 32 output channels x 8 input channels per wave and chunk')
print('(v_mfma_f32_32x32x2_f32 = 64 cycles, a packed vector instruction = 4; vector issue ADDS to MFMA time on this pipe,')
print('profiles/r03_mfma_valu_wino.txt):')
rows = []
for name, a, m, mf, pk, rd in (('F(2x2,3x3)', 4, 2, 16, 16, 8 + 8), ('F(4x4,3x3)', 6, 4, 24, 72, 22 + 12)):
    px = 32 * m * m                              # output pixels the a waves of a tile group finish per chunk
    cyc = a * (mf * 64 + pk * 4)
    rows.append(cyc / px)
    print('  %s: %d waves x (%2d MFMAs + %2d packed vector ops + ~%2d LDS reads) per %3d output pixels = %5.1f issue cycles per pixel, MFMA share %.2f'
          % (name, a, mf, pk, rd, px, cyc / px, mf * 64 / (mf * 64 + pk * 4)))
print('  (F(4x4) transform: B^T has 3-4 non-zeros per row with constants 4, 5, 2 -> ~3 v_pk_fma_f32 per output column and stage, 6 columns,')
print('   2 stages, 4 channels per lane = 72 packed ops; F(2x2): 16 v_pk_add_f32)')
ideal = rows[0] / rows[1]
print('issue-bound ratio by this count: %.2fx.  MEASURED shares (tools/mfma_valu.hip r, MI355X, round 4): 16 v_pk_add_f32 + 16 MFMAs, four waves per' % ideal)
print('SIMD: time(0) / time(16) = 0.935;  72 v_pk_fma_f32 + 24 MFMAs, three waves per SIMD: 0.828  ->  1.778 x 0.828 / 0.935 = 1.57x.')
print()
print('What the 1.57x does not contain:')
print('  * registers: 6 accumulators = 96 + stage-1 results t[6] as float4 = 24 + up to 4 patch float4 in flight = 16 + one V[j] float4 = 4-8 +')
print('    ~20 of addressing = ~164 of the 168 a wave may have at three per SIMD (wino11_kernel: 128 at four per SIMD).  Round 4 item 1: the')
print('    first spill puts an s_waitcnt vmcnt(0) into the chunk loop that also waits for the LDS-DMA in flight;')
print('  * LDS: 36 weight positions per chunk = 36.9 KB per 32 channels and buffer set (16 KB today) + a 64-tile region of 38 KB: 150 KB double-')
print('    buffered -> ONE twelve-wave workgroup per CU, so the item tail (6 x 6 -> 4 x 4 inverse transform + epilogue of 16 pixels per tile: ~2.7x')
print('    today\'s tail per item) overlaps with nothing, as in round 3\'s wino9_kernel (tail 19-30 % of an item);')
print('  * the ragged last tile column of a 25-wide plane: 25 / 28 of the MFMAs used against 25 / 26.')
for busy in (0.64, 0.55, 0.50):
    r = 1.778 * busy / 0.64
    print('  if it kept the pipe as busy as %.2f (wino11_kernel: 0.58-0.64): %.2fx on whole tiles, %.2fx on a 25-wide plane'
          % (busy, r, r * (25 / 28) / (25 / 26)))
print('Reading: 1.3-1.6x on paper for the 200 x 25 layers (1.6 ms of the 6.9 ms plain-DFCNN step), i.e. at or above the 1.3x bar, fp32 error no obstacle')
print('(7.5e-6 against the 1e-3 bar) -- but only if a twelve-wave, 164-register, one-workgroup-per-CU kernel keeps the pipe as busy as today\'s.  Not built in')
print('round 4 (the round went into wino11_kernel, which had the same target); the next step is a prototype of the chunk loop alone, timed against')
print('wino11_kernel\'s with tools/trace_wino11.py stamps, before any epilogue is written.')
