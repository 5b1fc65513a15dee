"""The LDS-DMA "NT" GEMM of gemm1.hip (large 1-tap data-gradients through asr_tap_gemm, forward layers through
asr_tap_gemm_nt on transposed kernels) against float64 matmul: whole and ragged tiles in M, N and K (K % 32 != 0 reads zeros
through out-of-range buffer offsets), operands that are column blocks of wider matrices (row pitches), bias / ReLU /
accumulate epilogues, bitwise reproducibility, the fall-through to the register-staged kernels for small problems, and the
batched transpose that feeds it.  Bars: 1e-5 of the result's scale (fp32 chains of K <= 6348 terms)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from asr_dfcnn_transformer_amd import ops as o
    return o


def _rand(shape, gen, scale=1.0):
    return torch.randn(*shape, device='cuda', generator=gen) * scale


def _f64(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _host_ref(a, w, bias=None, relu=False):
    """float64 reference computed on the HOST with numpy (the GPU's own BLAS is not the oracle)."""
    r = _f64(a) @ _f64(w)
    if bias is not None:
        r = r + _f64(bias)
    return np.maximum(r, 0.0) if relu else r


# (the last three: grids of whole rounds of 512 slots, which keep the 128 x 128 tile; the others take the 128 x 64 tile)
@pytest.mark.parametrize("M,K,N", [(6144, 512, 1024), (6150, 512, 1028), (6144, 100, 1024), (8192, 6348, 768), (49 * 128, 32, 1024),
                                   (6144, 36, 1024), (8192, 512, 1024), (8187, 100, 1020), (16384, 36, 2048)])
def test_forward_on_transposed_kernel(ops, M, K, N):
    g = torch.Generator(device='cuda').manual_seed(M + K + N)
    a = _rand((M, K), g)
    w = _rand((K, N), g, (1.0 / K) ** 0.5)
    bias = _rand((N,), g, 0.1)
    wt = torch.zeros(N, K, device='cuda')
    ops.Copy2dBatch([(wt, K, w, N, K, N)]).run_transposed()
    assert torch.equal(wt, w.t().contiguous())
    y = torch.full((M, N), 7.0, device='cuda')
    d = ops.gemm_desc(M, K, N, K, N, 0, N, ntaps=1, relu=1)
    ops.tap_gemm_nt(d, a, w, wt, K, bias, None, None, None, y)
    assert ops.last_kernel() == ('gemm1_kernel<0, 2>' if M >= 8187 and N >= 1020 else 'gemm1_kernel<0, 1>'), ops.last_kernel()
    ref = _host_ref(a, w, bias, relu=True)                               # float64 on the host (numpy), not a GPU library
    err = np.abs(y.cpu().numpy().astype(np.float64) - ref).max()
    assert err < 1e-5 * max(1.0, np.abs(ref).max()), err
    y2 = torch.zeros(M, N, device='cuda')
    ops.tap_gemm_nt(d, a, w, wt, K, bias, None, None, None, y2)
    assert torch.equal(y, y2)                                            # deterministic
    # the same layer on the register-staged kernel: equal to fp32 rounding
    y3 = torch.zeros(M, N, device='cuda')
    ops.tap_gemm(d, a, w, bias, None, None, None, y3)
    assert not ops.last_kernel().startswith('gemm1')
    assert (y - y3).abs().max().item() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("M,Kout,N", [(6144, 1024, 512), (8192, 768, 6348), (6200, 1028, 516), (8192, 1024, 516)])
def test_data_gradient_routes_to_the_dma_kernel(ops, M, Kout, N):
    """dX (+)= dY . W^T with W [Kout][N] as the model stores it: no copy, the kernel reads W as its [N_gemm][K_gemm] operand."""
    g = torch.Generator(device='cuda').manual_seed(3 * M + N)
    dy = _rand((M, N), g)
    w = _rand((Kout, N), g, (1.0 / N) ** 0.5)
    dx = torch.ones(M, Kout, device='cuda')
    d = ops.gemm_desc(M, N, Kout, N, N, 0, Kout, ntaps=1, wmode=1, accumulate=1)
    ops.tap_gemm(d, dy, w, None, None, None, None, dx)
    assert ops.last_kernel().startswith('gemm1_kernel<1,'), ops.last_kernel()
    ref = _host_ref(dy, w.t(), None) + 1.0
    assert np.abs(dx.cpu().numpy().astype(np.float64) - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())


def test_the_tile_choice_changes_no_bit(ops):
    """gemm1_blocks() picks the 128 x 64 or the 128 x 128 tile from the grid, i.e. from the row count as well: the same rows must
    give the same bits in a 6144-row problem (128 x 64 tile) and in an 8192-row one (128 x 128) -- an utterance alone or inside
    a batch (the invariant engine.py states for its routing rules)."""
    g = torch.Generator(device='cuda').manual_seed(21)
    K, N = 512, 1024
    a = _rand((8192, K), g)
    w = _rand((K, N), g, (1.0 / K) ** 0.5)
    wt = w.t().contiguous()
    bias = _rand((N,), g, 0.1)
    ys = []
    for M, kern in ((8192, 'gemm1_kernel<0, 2>'), (6144, 'gemm1_kernel<0, 1>')):
        y = torch.zeros(M, N, device='cuda')
        ops.tap_gemm_nt(ops.gemm_desc(M, K, N, K, N, 0, N, ntaps=1, relu=1), a[:M], w, wt, K, bias, None, None, None, y)
        assert ops.last_kernel() == kern, ops.last_kernel()
        ys.append(y)
    assert torch.equal(ys[0][:6144], ys[1])
    # ... and in a problem small enough for the register-staged kernel (gemm1 needs >= 48 tiles of 128 x 128): both kernels sum
    # the K products of an output element in the same order
    y = torch.zeros(640, N, device='cuda')
    ops.tap_gemm_nt(ops.gemm_desc(640, K, N, K, N, 0, N, ntaps=1, relu=1), a[:640], w, wt, K, bias, None, None, None, y)
    assert ops.last_kernel().startswith('tap_gemm_kernel_v1'), ops.last_kernel()
    assert torch.equal(ys[0][:640], y)
    # data-gradient form
    dy = _rand((8192, N), g)
    dxs = []
    for M, kern in ((8192, 'gemm1_kernel<1, 2>'), (6144, 'gemm1_kernel<1, 1>')):
        dx = torch.zeros(M, 1024, device='cuda')
        wk = _rand((1024, N), torch.Generator(device='cuda').manual_seed(22), 0.03)
        ops.tap_gemm(ops.gemm_desc(M, N, 1024, N, N, 0, 1024, ntaps=1, wmode=1), dy[:M], wk, None, None, None, None, dx)
        assert ops.last_kernel() == kern, ops.last_kernel()
        dxs.append(dx)
    assert torch.equal(dxs[0][:6144], dxs[1])


def test_column_blocks_of_wider_matrices(ops):
    """[dK | dV] . [wk | wv]^T of an encoder-decoder attention block (transformer_engine._mha_bwd): the activations are a
    2C-wide column block of a [rows][3C] buffer... here: A = columns C.. of a [M][3C] matrix (pitch 3C), W = columns C.. of the
    packed [C][3C] kernel (pitch 3C, base offset C floats), output pitch wider than N."""
    C, M = 512, 16384
    g = torch.Generator(device='cuda').manual_seed(5)
    big = _rand((M, 3 * C), g)
    W3 = _rand((C, 3 * C), g, (1.0 / C) ** 0.5)
    out = torch.zeros(M, C + 64, device='cuda')
    d = ops.gemm_desc(M, 2 * C, C, 3 * C, 3 * C, 0, C + 64, ntaps=1, wmode=1)
    ops.tap_gemm(d, big.view(-1)[C:], W3.view(-1)[C:], None, None, None, None, out)
    assert ops.last_kernel().startswith('gemm1_kernel<1,')
    ref = _host_ref(big[:, C:], W3[:, C:].t())
    assert np.abs(_f64(out[:, :C]) - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())
    assert float(out[:, C:].abs().max()) == 0.0                          # nothing written past N


def test_small_problems_keep_the_register_staged_kernels(ops):
    g = torch.Generator(device='cuda').manual_seed(9)
    a, w = _rand((640, 512), g), _rand((512, 512), g, 0.05)                # 5 x 4 tiles
    wt = w.t().contiguous()
    y = torch.zeros(640, 512, device='cuda')
    d = ops.gemm_desc(640, 512, 512, 512, 512, 0, 512, ntaps=1)
    ops.tap_gemm_nt(d, a, w, wt, 512, None, None, None, None, y)
    assert ops.last_kernel().startswith('tap_gemm_kernel_v1')
    assert np.abs(_f64(y) - _host_ref(a, w)).max() < 1e-4


def test_transpose_batch_ragged_shapes(ops):
    g = torch.Generator(device='cuda').manual_seed(2)
    srcs = [_rand((37, 70), g), _rand((512, 6348), g), _rand((4, 4), g)]
    wide = _rand((100, 96), g)                                            # a column block: 40 columns at pitch 96
    dsts = [torch.zeros(s.shape[1], s.shape[0], device='cuda') for s in srcs] + [torch.full((40, 128), 3.0, device='cuda')]
    items = [(d_, s.shape[0], s, s.shape[1], s.shape[0], s.shape[1]) for d_, s in zip(dsts, srcs)]
    items.append((dsts[3], 128, wide.view(-1)[8:], 96, 100, 40))
    ops.Copy2dBatch(items).run_transposed()
    for d_, s in zip(dsts, srcs):
        assert torch.equal(d_, s.t().contiguous())
    assert torch.equal(dsts[3][:, :100], wide[:, 8:48].t().contiguous()) and float((dsts[3][:, 100:] - 3.0).abs().max()) == 0.0


@pytest.mark.parametrize("M,K,N", [(4096, 512, 512), (4100, 512, 640), (2048, 512, 6348), (5000, 256, 32), (3001, 200, 24), (70, 128, 128),
                                   (33000, 128, 1536)])
def test_dense_weight_gradient(ops, M, K, N):
    """dW = A^T dZ on wgrad1_kernel (128 x 128 tiles; 128 x 32 for outputs of up to 32 channels): whole and ragged runs, chunks
    and tiles, the slab sum; twice for bitwise reproducibility."""
    g = torch.Generator(device='cuda').manual_seed(M + 7 * K + N)
    a, dz = _rand((M, K), g), _rand((M, N), g)
    d = ops.gemm_desc(M, K, N, K, N, ntaps=1)
    ws = torch.zeros(ops.tap_wgrad_workspace(d) // 4 + 64, device='cuda')
    dw = torch.full((K, N), 5.0, device='cuda')
    ops.tap_wgrad(d, a, dz, N, dw, ws)
    assert ops.last_kernel().startswith('wgrad1_kernel<4, 1, 1, 1>' if N <= 32 else 'wgrad1_kernel<2, 2, 2, 2>'), ops.last_kernel()
    ref = _host_ref(a.t(), dz)
    err = np.abs(_f64(dw) - ref).max()
    assert err < 2e-6 * M ** 0.5 * max(1.0, np.abs(ref).max() / M ** 0.5), err
    dw2 = torch.zeros(K, N, device='cuda')
    ops.tap_wgrad(d, a, dz, N, dw2, ws)
    assert torch.equal(dw, dw2)


def test_dense_weight_gradient_of_a_column_block(ops):
    """[dwq | dwk | dwv] of a fused projection: dZ is the [M][3C] gradient, A the block input (transformer_engine._wgrad_packed);
    and a gradient taken from a column block of a wider matrix (row pitch ldz > N)."""
    M, C = 4096, 256
    g = torch.Generator(device='cuda').manual_seed(11)
    x, dqkv = _rand((M, C), g), _rand((M, 3 * C), g)
    d = ops.gemm_desc(M, C, C, C, C, ntaps=1)
    ws = torch.zeros(ops.tap_wgrad_workspace(d) // 4 + 64, device='cuda')
    dw = torch.zeros(C, C, device='cuda')
    ops.tap_wgrad(d, x, dqkv.view(-1)[C:], 3 * C, dw, ws)                  # the K block
    ref = _host_ref(x.t(), dqkv[:, C:2 * C])
    assert np.abs(_f64(dw) - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N", [(32768, 512, 2048), (6400, 512, 2048), (300, 64, 128), (129, 96, 40), (1000, 128, 1536)])
def test_dense_data_gradient_with_the_relu_backward_in_its_epilogue(M, K, N):
    """asr_tap_gemm_relu_bwd = asr_tap_gemm (wmode 1) followed by asr_relu_bwd, bit for bit -- on the LDS-DMA kernel (gemm1_relumask_kernel)
    where it takes the shape, as the two calls elsewhere; against float64 as well."""
    import numpy as np, torch
    from asr_dfcnn_transformer_amd import ops
    rng = np.random.default_rng(41)
    dy = torch.tensor(rng.standard_normal((M, K)).astype(np.float32), device='cuda')
    w = torch.tensor((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32), device='cuda')          # the layer's kernel [in N][out K]
    h = torch.tensor(np.maximum(rng.standard_normal((M, N)), 0).astype(np.float32), device='cuda')
    d = ops.gemm_desc(M, K, N, K, K, 0, N, ntaps=1, wmode=1)
    ref = torch.zeros(M, N, device='cuda'); out = torch.full((M, N), 7.0, device='cuda')
    ops.tap_gemm(d, dy, w, None, None, None, None, ref)
    ops.relu_bwd(ref, h, ref)
    ops.tap_gemm_relu_bwd(d, dy, w, h, out)
    assert torch.equal(out, ref)
    if M <= 6400:
        want = (dy.double().cpu().numpy() @ w.double().cpu().numpy().T) * (h.cpu().numpy() > 0)
        assert np.abs(out.cpu().numpy() - want).max() <= 3e-5 * max(1.0, np.abs(want).max())
