"""Thin Python layer over the C ABI: torch tensors are used as device allocations only
(``data_ptr()`` + the current HIP stream); every computation below runs in libasrhip.so."""
import ctypes as C

import torch

from . import _lib
from ._lib import GemmDesc, check


def _ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Plane:
    """A padded NHWC plane [B][H+1][W+1][C] (include/asr_hip.h): one zero row above every image and
    one zero column left of every row -- the right/bottom borders are the next row's / next image's
    left/top border -- plus (W+3)-pixel zero guards on both ends (the last image's bottom border lives
    in the tail guard).  Kernels only write the interior."""

    def __init__(self, B, H, W, C, device='cuda'):
        self.B, self.H, self.W, self.C = B, H, W, C
        self.HP, self.WP = H + 1, W + 1
        self.NP = B * self.HP * self.WP
        self.G = W + 3
        self.buf = torch.zeros((self.NP + 2 * self.G) * C, dtype=torch.float32, device=device)
        self.body = self.buf[self.G * C:(self.G + self.NP) * C]      # pixel 0 .. NP-1

    @property
    def ptr(self):
        return C.c_void_p(self.body.data_ptr())

    def view(self):
        return self.body.view(self.B, self.HP, self.WP, self.C)

    def interior(self):
        return self.view()[:, 1:, 1:, :]

    def set_interior(self, x):
        self.interior().copy_(x)

    def zero_(self):
        self.buf.zero_()

    def border_abs_max(self):
        """Largest |value| outside the interior (guards, top border rows, left border columns): must stay exactly 0 for the
        plane to be a valid SAME-padded conv input (tests)."""
        g = self.G * self.C
        v = self.view()
        m = torch.stack([self.buf[:g].abs().max(), self.buf[g + self.NP * self.C:].abs().max(),
                         v[:, 0].abs().max(), v[:, :, 0].abs().max()]).max()
        return float(m)


class KernelTimer:
    """Optional HIP-event timing of the contraction kernels, keyed by kernel instantiation
    (family, ntaps, wmode, N-class).  bench.py uses it to price the dominant kernel inside
    the timed region; events are recorded on the stream the kernels are launched on."""

    def __init__(self, only=None):
        self.only = only          # None: time every tap kernel; else a set of keys
        self.records = {}         # key -> list of (flops, ev0, ev1)

    def want(self, key):
        return self.only is None or key in self.only

    def add(self, key, flops, e0, e1):
        self.records.setdefault(key, []).append((flops, e0, e1))

    def summary(self):
        """key -> dict(launches, total_ms, total_flops, tflops)  (call after a device sync)."""
        out = {}
        for key, recs in self.records.items():
            ms = sum(e0.elapsed_time(e1) for _, e0, e1 in recs)
            fl = sum(f for f, _, _ in recs)
            out[key] = {'launches': len(recs), 'total_ms': ms, 'total_flops': fl,
                        'avg_us': 1e3 * ms / len(recs), 'tflops': fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0}
        return out


TIMER = None


def last_kernel():
    """Symbol of the contraction kernel the library launched in this thread's last tap_gemm / tap_wgrad call
    (asr_last_kernel, include/asr_hip.h) -- the launcher's own dispatch decision, spelled as in a rocprofv3 trace."""
    return _lib.load().asr_last_kernel().decode()


def _flops(d):
    rows = d.B * d.H * d.W if d.H > 0 else d.M
    return 2.0 * rows * d.K * d.N * d.ntaps


def _timed(d, fn, tag=''):
    """Runs fn(); with a KernelTimer installed, brackets it with HIP events on the launch stream and files the pair
    under the kernel symbol the library reports (+ ``tag``, for kernels that serve two directions under one symbol)."""
    t = TIMER
    if t is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    key = last_kernel() + tag
    if t.want(key):
        t.add(key, _flops(d), e0, e1)


def gemm_desc(M, K, N, lda, ldw, ldo_a=0, ldo_y=0, ntaps=1, B=0, H=0, W=0, wmode=0, relu=0,
              accumulate=0, y_unpadded=0):
    return GemmDesc(M, K, N, lda, ldw, ldo_a, ldo_y, ntaps, B, H, W, wmode, relu, accumulate, y_unpadded)


def tap_gemm(desc, A, W, bias=None, scale=None, shift=None, out_a=None, out_y=None):
    lib = _lib.load()
    pa = A.ptr if isinstance(A, Plane) else _ptr(A)
    po_a = out_a.ptr if isinstance(out_a, Plane) else _ptr(out_a)
    po_y = out_y.ptr if isinstance(out_y, Plane) else _ptr(out_y)
    _timed(desc, lambda: check(
        lib.asr_tap_gemm(C.byref(desc), pa, _ptr(W), _ptr(bias), _ptr(scale), _ptr(shift), po_a, po_y, _stream()),
        'asr_tap_gemm'))


def tap_gemm_nt(desc, A, W, Wt, ldwt, bias=None, scale=None, shift=None, out_a=None, out_y=None):
    """Dense forward with the kernel W [K][N] AND its transposed copy Wt [N][K] (asr_tap_gemm_nt): the large problems take the
    LDS-DMA kernel on Wt, the rest asr_tap_gemm on W."""
    lib = _lib.load()
    _timed(desc, lambda: check(
        lib.asr_tap_gemm_nt(C.byref(desc), _ptr(A), _ptr(W), _ptr(Wt), int(ldwt), _ptr(bias), _ptr(scale), _ptr(shift),
                            _ptr(out_a), _ptr(out_y), _stream()), 'asr_tap_gemm_nt'))


def tap_gemm_splitk_workspace(desc, splits):
    return _lib.load().asr_tap_gemm_splitk_workspace(C.byref(desc), int(splits))


def tap_gemm_splitk(desc, A, W, bias, scale, shift, out_a, out_y, splits, workspace):
    """Dense forward GEMM with the contraction split over the grid (asr_tap_gemm_splitk)."""
    lib = _lib.load()
    _timed(desc, lambda: check(lib.asr_tap_gemm_splitk(C.byref(desc), _ptr(A), _ptr(W), _ptr(bias), _ptr(scale), _ptr(shift), _ptr(out_a),
                                                       _ptr(out_y), int(splits), _ptr(workspace), _stream()), 'asr_tap_gemm_splitk'))


def tap_gemm_relu_bwd(desc, dY, W, H, dX):
    """dX = (dY . W^T) where H > 0 else 0: a dense data-gradient with the ReLU backward of the layer in front in its epilogue
    (asr_tap_gemm_relu_bwd)."""
    lib = _lib.load()
    _timed(desc, lambda: check(lib.asr_tap_gemm_relu_bwd(C.byref(desc), _ptr(dY), _ptr(W), _ptr(H), _ptr(dX), _stream()), 'asr_tap_gemm_relu_bwd'))


def tap_gemm_nt_splitk_workspace(desc, splits):
    return _lib.load().asr_tap_gemm_nt_splitk_workspace(C.byref(desc), int(splits))


def tap_gemm_nt_splitk(desc, A, Bt, ldb, bias, scale, shift, out_a, out_y, splits, workspace):
    """Split-K dense GEMM on the LDS-DMA kernel, both operands K-contiguous (asr_tap_gemm_nt_splitk)."""
    lib = _lib.load()
    _timed(desc, lambda: check(lib.asr_tap_gemm_nt_splitk(C.byref(desc), _ptr(A), _ptr(Bt), int(ldb), _ptr(bias), _ptr(scale), _ptr(shift),
                                                          _ptr(out_a), _ptr(out_y), int(splits), _ptr(workspace), workspace.numel() * workspace.element_size(),
                                                          _stream()), 'asr_tap_gemm_nt_splitk'))


def tap_gemm_gated_workspace(desc):
    return _lib.load().asr_tap_gemm_gated_workspace(C.byref(desc))


def tap_gemm_gated(desc, dZ, W, prearranged, pool, gate_a, bn_scale, bn_shift, dy_prev, dz_out, dscale, dshift, dbias, partials):
    """Data-gradient of a cell with the backward prologue of the cell in front fused into its epilogue
    (asr_tap_gemm_gated): gate_a / dz_out are Planes of that cell's pre-pool geometry."""
    lib = _lib.load()
    pz = dZ.ptr if isinstance(dZ, Plane) else _ptr(dZ)
    pp = dy_prev.ptr if isinstance(dy_prev, Plane) else _ptr(dy_prev)
    _timed(desc, lambda: check(
        lib.asr_tap_gemm_gated(C.byref(desc), pz, _ptr(W), int(prearranged), int(pool), gate_a.H, gate_a.W, gate_a.ptr,
                               _ptr(bn_scale), _ptr(bn_shift), pp, dz_out.ptr, _ptr(dscale), _ptr(dshift), _ptr(dbias),
                               _ptr(partials), _stream()), 'asr_tap_gemm_gated'))


def tap_gemm_gated_dense_supported(desc, H, W, Cc):
    return bool(_lib.load().asr_tap_gemm_gated_dense_supported(C.byref(desc), int(H), int(W), int(Cc)))


def tap_gemm_gated_dense_workspace(desc, W, Cc):
    return _lib.load().asr_tap_gemm_gated_dense_workspace(C.byref(desc), int(W), int(Cc))


def tap_gemm_gated_dense(desc, dZ, W, gate_a, bn_scale, bn_shift, dz_out, dscale, dshift, dbias, partials):
    """Data-gradient of a dense layer whose input is the (flattened) output of an un-pooled cell, with that cell's BN / ReLU backward
    in the epilogue (asr_tap_gemm_gated_dense): gate_a / dz_out are the cell's activation / dZ Planes."""
    lib = _lib.load()
    _timed(desc, lambda: check(
        lib.asr_tap_gemm_gated_dense(C.byref(desc), _ptr(dZ), _ptr(W), gate_a.H, gate_a.W, gate_a.C, gate_a.ptr, _ptr(bn_scale), _ptr(bn_shift),
                                     dz_out.ptr, _ptr(dscale), _ptr(dshift), _ptr(dbias), _ptr(partials), _stream()),
        'asr_tap_gemm_gated_dense'))


def poolmax_supported(fwd_desc, bwd_desc):
    return bool(_lib.load().asr_winograd_poolmax_supported(C.byref(fwd_desc), C.byref(bwd_desc)))


def poolmax_index(B, H2, W2, N, device='cuda'):
    """the position planes of asr_tap_gemm_wino_poolmax (uint32 words, held in an int32 tensor)"""
    n = _lib.load().asr_poolmax_index_bytes(B, H2, W2, N)
    assert n > 0
    return torch.zeros(n // 4, dtype=torch.int32, device=device)


def poolavg_index(B, H2, W2, N, device='cuda'):
    """the ReLU-sign planes of asr_tap_gemm_wino_poolavg (uint32 words, held in an int32 tensor)"""
    n = _lib.load().asr_poolavg_index_bytes(B, H2, W2, N)
    assert n > 0
    return torch.zeros(n // 4, dtype=torch.int32, device=device)


def tap_gemm_wino_poolavg(desc, A, Wt, bias, scale, shift, y_pooled, a_sum, index):
    """Forward conv of an AVERAGE-pooled cell in the compact form (asr_tap_gemm_wino_poolavg): no activation plane; y_pooled, the sum of
    each window's activations (a_sum, a Plane of the pooled geometry) and the ReLU signs of its positions (index)."""
    lib = _lib.load()
    pa = A.ptr if isinstance(A, Plane) else _ptr(A)
    _timed(desc, lambda: check(lib.asr_tap_gemm_wino_poolavg(C.byref(desc), pa, _ptr(Wt), _ptr(bias), _ptr(scale), _ptr(shift),
                                                             y_pooled.ptr, a_sum.ptr, _ptr(index), _stream()), 'asr_tap_gemm_wino_poolavg'))


def tap_gemm_gated_poolavg(desc, dZ, Wt, gate_H, gate_W, a_sum, index, bn_scale, bn_shift, dy_prev, dz_out, dscale, dshift, dbias, partials):
    """asr_tap_gemm_gated for an average-pooled cell in the compact form: (a_sum, index) in place of the activation plane."""
    lib = _lib.load()
    pz = dZ.ptr if isinstance(dZ, Plane) else _ptr(dZ)
    pp = dy_prev.ptr if isinstance(dy_prev, Plane) else _ptr(dy_prev)
    _timed(desc, lambda: check(
        lib.asr_tap_gemm_gated_poolavg(C.byref(desc), pz, _ptr(Wt), int(gate_H), int(gate_W), a_sum.ptr, _ptr(index), _ptr(bn_scale),
                                       _ptr(bn_shift), pp, dz_out.ptr, _ptr(dscale), _ptr(dshift), _ptr(dbias), _ptr(partials), _stream()),
        'asr_tap_gemm_gated_poolavg'))


def tap_gemm_wino_poolmax(desc, A, Wt, bias, scale, shift, y_pooled, a_max, index):
    """Forward conv of a MAX-pooled cell in the compact form (asr_tap_gemm_wino_poolmax): no activation plane; y_pooled, the
    activation at each window's maximum (a_max, a Plane of the pooled geometry) and its position (index)."""
    lib = _lib.load()
    pa = A.ptr if isinstance(A, Plane) else _ptr(A)
    _timed(desc, lambda: check(lib.asr_tap_gemm_wino_poolmax(C.byref(desc), pa, _ptr(Wt), _ptr(bias), _ptr(scale), _ptr(shift),
                                                             y_pooled.ptr, a_max.ptr, _ptr(index), _stream()), 'asr_tap_gemm_wino_poolmax'))


def tap_gemm_gated_poolmax(desc, dZ, Wt, gate_H, gate_W, a_max, index, bn_scale, bn_shift, dy_prev, dz_out, dscale, dshift, dbias, partials):
    """asr_tap_gemm_gated for a max-pooled cell in the compact form: (a_max, index) in place of the activation plane."""
    lib = _lib.load()
    pz = dZ.ptr if isinstance(dZ, Plane) else _ptr(dZ)
    pp = dy_prev.ptr if isinstance(dy_prev, Plane) else _ptr(dy_prev)
    _timed(desc, lambda: check(
        lib.asr_tap_gemm_gated_poolmax(C.byref(desc), pz, _ptr(Wt), int(gate_H), int(gate_W), a_max.ptr, _ptr(index), _ptr(bn_scale),
                                       _ptr(bn_shift), pp, dz_out.ptr, _ptr(dscale), _ptr(dshift), _ptr(dbias), _ptr(partials), _stream()),
        'asr_tap_gemm_gated_poolmax'))


def tap_wgrad_workspace(desc):
    return _lib.load().asr_tap_wgrad_workspace(C.byref(desc))


def tap_wgrad(desc, A, dZ, ldz, dW, partials, direct=False):
    """direct=True: asr_tap_wgrad_direct (never the Winograd kernel)."""
    lib = _lib.load()
    pa = A.ptr if isinstance(A, Plane) else _ptr(A)
    pz = dZ.ptr if isinstance(dZ, Plane) else _ptr(dZ)
    fn = lib.asr_tap_wgrad_direct if direct else lib.asr_tap_wgrad
    _timed(desc, lambda: check(fn(C.byref(desc), pa, pz, ldz, _ptr(dW), _ptr(partials), _stream()), 'asr_tap_wgrad'))


def cell1_fwd(x, w, bias, sc, sh, pool, y):
    B, T, F = x.shape[0], x.shape[1], x.shape[2]
    check(_lib.load().asr_cell1_fwd(_ptr(x), B, T, F, y.C, _ptr(w), _ptr(bias), _ptr(sc), _ptr(sh), pool, y.ptr,
                                    _stream()), 'asr_cell1_fwd')


def cell1_bwd_workspace(B, T, F, Cc):
    return _lib.load().asr_cell1_bwd_workspace(B, T, F, Cc)


def cell1_bwd(x, w, bias, sc, sh, pool, dy, dw, db, dscale, dshift, partials):
    B, T, F = x.shape[0], x.shape[1], x.shape[2]
    check(_lib.load().asr_cell1_bwd(_ptr(x), B, T, F, dy.C, _ptr(w), _ptr(bias), _ptr(sc), _ptr(sh), pool, dy.ptr,
                                    _ptr(dw), _ptr(db), _ptr(dscale), _ptr(dshift), _ptr(partials), _stream()),
          'asr_cell1_bwd')


def pool_fwd(a, sc, sh, pool, y):
    check(_lib.load().asr_pool_fwd(a.ptr, a.B, a.H, a.W, a.C, _ptr(sc), _ptr(sh), pool, y.ptr, _stream()),
          'asr_pool_fwd')


def cell_bwd_pre_workspace(B, H, W, Cc):
    return _lib.load().asr_cell_bwd_pre_workspace(B, H, W, Cc)


def cell_bwd_pre(dy, dy_layout, a, sc, sh, pool, dz, dscale, dshift, dbias, partials):
    pdy = dy.ptr if isinstance(dy, Plane) else _ptr(dy)
    check(_lib.load().asr_cell_bwd_pre(pdy, dy_layout, a.ptr, a.B, a.H, a.W, a.C, _ptr(sc), _ptr(sh), pool, dz.ptr,
                                       _ptr(dscale), _ptr(dshift), _ptr(dbias), _ptr(partials), _stream()),
          'asr_cell_bwd_pre')


def se_state_floats(B, Cc, hid):
    return _lib.load().asr_se_state_floats(B, Cc, hid)


def se_fwd_workspace(B, H, W, Cc):
    return _lib.load().asr_se_fwd_workspace(B, H, W, Cc)


def se_bwd_workspace(B, H, W, Cc, hid):
    return _lib.load().asr_se_bwd_workspace(B, H, W, Cc, hid)


def se_fwd(main_in, x, hid, sc, sh, w1, b1, w2, b2, state, partials, out):
    check(_lib.load().asr_se_fwd(main_in.ptr, x.ptr, x.B, x.H, x.W, x.C, hid, _ptr(sc), _ptr(sh), _ptr(w1), _ptr(b1),
                                 _ptr(w2), _ptr(b2), _ptr(state), _ptr(partials), out.ptr, _stream()), 'asr_se_fwd')


def se_fwd_sums(main_in, x, hid, sc, sh, w1, b1, w2, b2, state, sums, nsplit, out):
    check(_lib.load().asr_se_fwd_sums(main_in.ptr, x.ptr, x.B, x.H, x.W, x.C, hid, _ptr(sc), _ptr(sh), _ptr(w1), _ptr(b1),
                                      _ptr(w2), _ptr(b2), _ptr(state), _ptr(sums), int(nsplit), out.ptr, _stream()), 'asr_se_fwd_sums')


def se_bwd(dout, x, hid, sc, sh, w1, w2, state, add_dout, dx, dscale, dshift, dw1, db1, dw2, db2, partials):
    check(_lib.load().asr_se_bwd(dout.ptr, x.ptr, x.B, x.H, x.W, x.C, hid, _ptr(sc), _ptr(sh), _ptr(w1), _ptr(w2),
                                 _ptr(state), add_dout, dx.ptr, _ptr(dscale), _ptr(dshift), _ptr(dw1), _ptr(db1),
                                 _ptr(dw2), _ptr(db2), _ptr(partials), _stream()), 'asr_se_bwd')


def se_bwd_cell_workspace(B, H, W, Cc, hid):
    return _lib.load().asr_se_bwd_cell_workspace(B, H, W, Cc, hid)


def se_bwd_cell(dout, x, hid, sc, sh, w1, w2, state, add_dout, dscale, dshift, dw1, db1, dw2, db2, cell_a, cell_scale, cell_dz,
                cell_dscale, cell_dshift, cell_dbias, partials):
    """asr_se_bwd with the backward prologue of the conv cell that produced x fused in (asr_se_bwd_cell): no dx plane."""
    check(_lib.load().asr_se_bwd_cell(dout.ptr, x.ptr, x.B, x.H, x.W, x.C, hid, _ptr(sc), _ptr(sh), _ptr(w1), _ptr(w2),
                                      _ptr(state), add_dout, _ptr(dscale), _ptr(dshift), _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2),
                                      cell_a.ptr, _ptr(cell_scale), cell_dz.ptr, _ptr(cell_dscale), _ptr(cell_dshift), _ptr(cell_dbias),
                                      _ptr(partials), _stream()), 'asr_se_bwd_cell')


def se_bwd_cell_sums(dout, x, hid, sc, sh, w1, w2, state, add_dout, dscale, dshift, dw1, db1, dw2, db2, cell_a, cell_scale, cell_dz,
                     cell_dscale, cell_dshift, cell_dbias, xsums, nsplit, partials):
    """se_bwd_cell with the block's first reduction handed in (asr_se_bwd_cell_sums; xsums from tap_gemm_wino_sesum)."""
    check(_lib.load().asr_se_bwd_cell_sums(dout.ptr, x.ptr, x.B, x.H, x.W, x.C, hid, _ptr(sc), _ptr(sh), _ptr(w1), _ptr(w2),
                                           _ptr(state), add_dout, _ptr(dscale), _ptr(dshift), _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2),
                                           cell_a.ptr, _ptr(cell_scale), cell_dz.ptr, _ptr(cell_dscale), _ptr(cell_dshift), _ptr(cell_dbias),
                                           _ptr(xsums), int(nsplit), _ptr(partials), _stream()), 'asr_se_bwd_cell_sums')


def axpy(dst, src, alpha=1.0, accumulate=False):
    check(_lib.load().asr_axpy(_ptr(dst), _ptr(src), dst.numel(), alpha, int(accumulate), _stream()), 'asr_axpy')


def softmax_log_fwd(d, B, T, V, eps, logits_tm):
    check(_lib.load().asr_softmax_log_fwd(_ptr(d), B, T, V, eps, _ptr(logits_tm), _stream()), 'asr_softmax_log_fwd')


def softmax_log_bwd(logits_tm, g_tm, B, T, V, eps, gscale, dd):
    check(_lib.load().asr_softmax_log_bwd(_ptr(logits_tm), _ptr(g_tm), B, T, V, eps, gscale, _ptr(dd), _stream()),
          'asr_softmax_log_bwd')


def relu_bwd(dy, h, dz):
    check(_lib.load().asr_relu_bwd(_ptr(dy), _ptr(h), dy.numel(), _ptr(dz), _stream()), 'asr_relu_bwd')


def relu_bwd_scaled(dy, h, scale, dz):
    """dz = (h > 0) ? dy * scale : 0 (see asr_relu_bwd_scaled)."""
    check(_lib.load().asr_relu_bwd_scaled(_ptr(dy), _ptr(h), dy.numel(), float(scale), _ptr(dz), _stream()), 'asr_relu_bwd_scaled')


def colsum_workspace(rows, cols):
    return _lib.load().asr_colsum_workspace(rows, cols)


def colsum(x, rows, cols, ldx, out, partials):
    check(_lib.load().asr_colsum(_ptr(x), rows, cols, ldx, _ptr(out), _ptr(partials), _stream()), 'asr_colsum')


def ctc_workspace(T, B, max_label):
    return _lib.load().asr_ctc_workspace(T, B, max_label)


def ctc_loss(logits_tm, T, B, V, labels, max_label, label_len, seq_len, blank, loss, grad, status, workspace):
    check(_lib.load().asr_ctc_loss(_ptr(logits_tm), T, B, V, _ptr(labels), max_label, _ptr(label_len), _ptr(seq_len),
                                   blank, _ptr(loss), _ptr(grad), _ptr(status), _ptr(workspace), _stream()),
          'asr_ctc_loss')


def ctc_greedy_workspace(T, B):
    return _lib.load().asr_ctc_greedy_workspace(T, B)


def ctc_greedy(logits_tm, T, B, V, seq_len, blank, out_ids, out_len, neg_sum, workspace):
    check(_lib.load().asr_ctc_greedy(_ptr(logits_tm), T, B, V, _ptr(seq_len), blank, _ptr(out_ids), _ptr(out_len),
                                     _ptr(neg_sum), _ptr(workspace), _stream()), 'asr_ctc_greedy')


def edit_distance(hyp, hyp_pitch, hyp_len, truth, truth_pitch, truth_len, B, dist):
    check(_lib.load().asr_edit_distance(_ptr(hyp), hyp_pitch, _ptr(hyp_len), _ptr(truth), truth_pitch,
                                        _ptr(truth_len), B, _ptr(dist), _stream()), 'asr_edit_distance')


def adam_tf(theta, grad, m, v, lr_t, beta1, beta2, eps, gscale=1.0):
    check(_lib.load().asr_adam_tf(_ptr(theta), _ptr(grad), _ptr(m), _ptr(v), theta.numel(), lr_t, beta1, beta2, eps,
                                  gscale, _stream()), 'asr_adam_tf')


def fbank(signal, nsamples, frame_len, frame_step, nfft, preemph, nfilt, fb_start, fb_count, fb_weight, fb_width,
          twiddle, logfb, max_frames, out, t_pad, frames):
    B, max_samples = signal.shape
    check(_lib.load().asr_fbank(_ptr(signal), _ptr(nsamples), B, max_samples, frame_len, frame_step, nfft, preemph,
                                nfilt, _ptr(fb_start), _ptr(fb_count), _ptr(fb_weight), fb_width, _ptr(twiddle),
                                _ptr(logfb), max_frames, _ptr(out), t_pad, _ptr(frames), _stream()), 'asr_fbank')


def lfr(feat, frames, m, n, t_out, out=None):
    """feat [B, t_pad, D] f32, frames [B] i32 (device) -> [B, t_out, m*D] stacked / skipped frames (asr_lfr)."""
    B, t_pad, D = feat.shape
    if out is None:
        out = torch.empty(B, t_out, m * D, dtype=torch.float32, device=feat.device)
    check(_lib.load().asr_lfr(_ptr(feat), _ptr(frames), B, t_pad, D, m, n, t_out, _ptr(out), _stream()), 'asr_lfr')
    return out


# ------------------------------------------------------------------ Transformer path
def attention_stats_floats(N, Tq, Tk, H):
    return _lib.load().asr_attention_stats_floats(N, Tq, Tk, H)


def attention_stats(Q, K, N, Tq, Tk, Cc, H, stats, ldq=None, ldk=None):
    """The query masks and key biases of every (sample, head) of an attention call, once (asr_attention_stats): handed to attention_fwd /
    attention_bwd as ``stats`` they replace what every workgroup of those kernels otherwise recomputes from Q / K (the same values)."""
    check(_lib.load().asr_attention_stats(_ptr(Q), _ptr(K), N, Tq, Tk, Cc, H, ldq or Cc, ldk or Cc, _ptr(stats), _stream()), 'asr_attention_stats')


def attention_fwd(Q, K, V, N, Tq, Tk, Cc, H, causal, O, lse, dropout_rate=0.0, seed=0, ldq=None, ldk=None, stats=None):
    """ldq / ldk: row pitches of Q and of K, V when they are column blocks of a fused projection buffer (default Cc)."""
    check(_lib.load().asr_attention_fwd_s(_ptr(Q), _ptr(K), _ptr(V), N, Tq, Tk, Cc, H, ldq or Cc, ldk or Cc, int(causal),
                                          float(dropout_rate), int(seed) & 0xffffffff, _ptr(O), _ptr(lse), _ptr(stats), _stream()),
          'asr_attention_fwd')


def attention_bwd(Q, K, V, O, dO, lse, N, Tq, Tk, Cc, H, causal, dQ, dK, dV, delta_ws, relu_grad=False, dropout_rate=0.0, seed=0,
                  ldq=None, ldk=None, stats=None):
    check(_lib.load().asr_attention_bwd_s(_ptr(Q), _ptr(K), _ptr(V), _ptr(O), _ptr(dO), _ptr(lse), N, Tq, Tk, Cc, H,
                                          ldq or Cc, ldk or Cc, int(causal), int(relu_grad), float(dropout_rate),
                                          int(seed) & 0xffffffff, _ptr(dQ), _ptr(dK), _ptr(dV), _ptr(delta_ws), _ptr(stats), _stream()),
          'asr_attention_bwd')


def copy2d(dst, ldd, src, lds, rows, cols, accumulate=False):
    check(_lib.load().asr_copy2d(_ptr(dst), ldd, _ptr(src), lds, rows, cols, int(accumulate), _stream()), 'asr_copy2d')


class Copy2dBatch:
    """A table of strided copies run as one launch (asr_copy2d_batch).  ``items``: (dst, ldd, src, lds, rows, cols) with dst /
    src device tensors (views allowed); the tensors must outlive the batch and keep their storage."""

    def __init__(self, items):
        import struct
        assert items
        raw = b''.join(struct.pack('<QQiiii', _ptr(d).value or 0, _ptr(s_).value or 0, ldd, lds, rows, cols)
                       for d, ldd, s_, lds, rows, cols in items)
        self._keep = [(d, s_) for d, _, s_, _, _, _ in items]
        self.n = len(items)
        self.max_elems = max(rows * cols for _, _, _, _, rows, cols in items)
        dev = items[0][0].device
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)

    def run(self, accumulate=False):
        check(_lib.load().asr_copy2d_batch(_ptr(self.table), self.n, self.max_elems, int(accumulate), _stream()), 'asr_copy2d_batch')

    def run_transposed(self):
        """dst[c][r] = src[r][c] for every item (rows / cols / lds describe src): asr_transpose_batch."""
        check(_lib.load().asr_transpose_batch(_ptr(self.table), self.n, self.max_elems, _stream()), 'asr_transpose_batch')


class ReduceBatch:
    """Deferred column reductions run as one asr_colsum_multi_batch (two launches).  ``items``: (partials tensor, rows, ld,
    [(width, out tensor), ...] with up to four segments); scratch is owned here; the tensors must keep their storage."""

    def __init__(self, items):
        import struct
        assert items
        dev = items[0][0].device
        self._keep, raw, self.max_cols = [], b'', 0
        for part, rows, ld, segs in items:
            assert 1 <= len(segs) <= 4
            cols = sum(w for w, _ in segs)
            tmp = torch.zeros(64 * cols, dtype=torch.float32, device=dev)
            self._keep.append((part, tmp, [o for _, o in segs]))
            widths = [w for w, _ in segs] + [0] * (4 - len(segs))
            outs = [(_ptr(o).value or 0) for _, o in segs] + [0] * (4 - len(segs))
            raw += struct.pack('<QQiii4i4xQQQQ', _ptr(part).value or 0, _ptr(tmp).value or 0, rows, ld, len(segs), *widths, *outs)
            self.max_cols = max(self.max_cols, cols)
        assert len(raw) == 80 * len(items)
        self.n = len(items)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)

    def run(self):
        check(_lib.load().asr_colsum_multi_batch(_ptr(self.table), self.n, self.max_cols, _stream()), 'asr_colsum_multi_batch')


def layernorm_bwd_blocks(rows):
    return _lib.load().asr_layernorm_bwd_blocks(rows)


def dropout(x, rate, seed, y=None):
    """y = keep(i, seed) ? x / (1 - rate) : 0 (in place when y is None); the same call on a gradient is the backward."""
    y = x if y is None else y
    check(_lib.load().asr_dropout(_ptr(x), x.numel(), float(rate), int(seed) & 0xffffffff, _ptr(y), _stream()), 'asr_dropout')


def add_layernorm_fwd(a, b, gamma, beta, rows, Cc, eps, y, xhat, rstd):
    check(_lib.load().asr_add_layernorm_fwd(_ptr(a), _ptr(b), _ptr(gamma), _ptr(beta), rows, Cc, eps, _ptr(y),
                                            _ptr(xhat), _ptr(rstd), _stream()), 'asr_add_layernorm_fwd')


def layernorm_bwd_workspace(rows, Cc):
    return _lib.load().asr_layernorm_bwd_workspace(rows, Cc)


def layernorm_bwd(dy, xhat, rstd, gamma, rows, Cc, dx, accumulate, dgamma, dbeta, partials):
    check(_lib.load().asr_layernorm_bwd(_ptr(dy), _ptr(xhat), _ptr(rstd), _ptr(gamma), rows, Cc, _ptr(dx),
                                        int(accumulate), _ptr(dgamma), _ptr(dbeta), _ptr(partials), _stream()),
          'asr_layernorm_bwd')


def embed_fwd(table, ids, pos, N, T, Cc, zero_pad, scale, out):
    check(_lib.load().asr_embed_fwd(_ptr(table), _ptr(ids), _ptr(pos), N, T, Cc, int(zero_pad), scale, _ptr(out),
                                    _stream()), 'asr_embed_fwd')


def layernorm_bwd_fused(dy, xhat, rstd, gamma, rows, Cc, dx, dx2, accumulate2, z, zscale, dz, dgamma, dbeta, partials,
                        drop_rate=0.0, drop_seed=0):
    """asr_layernorm_bwd with its consumers fused: dx (or None), dx2 (+)= dx (or None), dz = mask ? dx * zscale : 0 (or None),
    mask = (z > 0 if z is given) and (asr_dropout's keep(element, drop_seed) if drop_rate > 0)."""
    check(_lib.load().asr_layernorm_bwd_fused(_ptr(dy), _ptr(xhat), _ptr(rstd), _ptr(gamma), rows, Cc, _ptr(dx), _ptr(dx2),
                                              int(accumulate2), _ptr(z), float(drop_rate), int(drop_seed) & 0xffffffff, float(zscale),
                                              _ptr(dz), _ptr(dgamma), _ptr(dbeta), _ptr(partials), _stream()), 'asr_layernorm_bwd_fused')


def add_layernorm_fwd_dropout(a, b, gamma, beta, rows, Cc, eps, rate, seed, y, xhat, rstd):
    """add_layernorm_fwd(dropout(a, rate, seed), b) in one pass; a is left as it is."""
    check(_lib.load().asr_add_layernorm_fwd_dropout(_ptr(a), _ptr(b), _ptr(gamma), _ptr(beta), rows, Cc, eps, float(rate),
                                                    int(seed) & 0xffffffff, _ptr(y), _ptr(xhat), _ptr(rstd), _stream()),
          'asr_add_layernorm_fwd_dropout')


def embed_bwd_ids(dout, ids, rows, V, Cc, zero_pad, scale, dtable):
    check(_lib.load().asr_embed_bwd_ids(_ptr(dout), _ptr(ids), rows, V, Cc, int(zero_pad), scale, _ptr(dtable), _stream()),
          'asr_embed_bwd_ids')


def embed_bwd(dout, perm, uniq, seg, n_uniq, Cc, zero_pad, scale, dtable):
    check(_lib.load().asr_embed_bwd(_ptr(dout), _ptr(perm), _ptr(uniq), _ptr(seg), n_uniq, Cc, int(zero_pad), scale,
                                    _ptr(dtable), _stream()), 'asr_embed_bwd')


def smoothed_ce(logits, ld, target, rows, V, eps, pad_id, inv_count, loss_rows, preds, stats, dlogits):
    check(_lib.load().asr_smoothed_ce(_ptr(logits), ld, _ptr(target), rows, V, eps, pad_id, inv_count, _ptr(loss_rows),
                                      _ptr(preds), _ptr(stats), _ptr(dlogits), _stream()), 'asr_smoothed_ce')


# ---------------------------------------------------------------------------- end2end pre-net (include/asr_hip.h)
from ._lib import PixMap  # noqa: E402


def pixmap(t, C_=None, phase_split=False):
    """(pointer, asr_pixmap) of a Plane (kind 0, or kind 2 when it holds a phase-split tensor of 2H x 2W pixels and
    C/4 channels) or of a plain [B][H][W][C] tensor (kind 1)."""
    if isinstance(t, Plane):
        if phase_split:
            return t.ptr, PixMap(2, t.B, 2 * t.H, 2 * t.W, t.C // 4, t.C)
        return t.ptr, PixMap(0, t.B, t.H, t.W, C_ or t.C, t.C)
    B, H, W, Cc = t.shape
    return _ptr(t), PixMap(1, B, H, W, Cc, Cc)


def prenet_conv1_fwd(x, w, b, a1):
    B, T, F = x.shape
    check(_lib.load().asr_prenet_conv1_fwd(_ptr(x), _ptr(w), _ptr(b), B, T, F, _ptr(a1), _stream()), 'asr_prenet_conv1_fwd')


def prenet_conv1_bwd_workspace(B, T, F):
    return _lib.load().asr_prenet_conv1_bwd_workspace(B, T, F)


def prenet_conv1_bwd(x, dz, dw, db, ws):
    B, T, F = x.shape
    check(_lib.load().asr_prenet_conv1_bwd(_ptr(x), _ptr(dz), B, T, F, _ptr(dw), _ptr(db), _ptr(ws), _stream()),
          'asr_prenet_conv1_bwd')


def bn_workspace(t, **kw):
    _, m = pixmap(t, **kw)
    return _lib.load().asr_bn_workspace(C.byref(m))


def bn_stats(src, eps, mean, rstd, ws, **kw):
    p, m = pixmap(src, **kw)
    check(_lib.load().asr_bn_stats(p, C.byref(m), eps, _ptr(mean), _ptr(rstd), _ptr(ws), _stream()), 'asr_bn_stats')


def bn_moving(mean, rstd, C_, eps, count, momentum, mov_mean, mov_var, inf_rstd=None):
    """One momentum step of a Keras BatchNormalization's moving statistics and / or its inference-mode 1/sqrt(var + eps)."""
    check(_lib.load().asr_bn_moving(_ptr(mean), _ptr(rstd), C_, eps, float(count), momentum, _ptr(mov_mean), _ptr(mov_var),
                                    _ptr(inf_rstd), _stream()), 'asr_bn_moving')


def bn_apply(src, mean, rstd, gamma, beta, dst, res=None, relu=False, dst_phase_split=False):
    ps, sm = pixmap(src)
    pd, dm = pixmap(dst, phase_split=dst_phase_split)
    pr, rm = pixmap(res) if res is not None else (C.c_void_p(0), sm)
    check(_lib.load().asr_bn_apply(ps, C.byref(sm), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), pr, C.byref(rm),
                                   int(relu), pd, C.byref(dm), _stream()), 'asr_bn_apply')


def bn_bwd(dy, a, mean, rstd, gamma, act, dz, dgamma, dbeta, ws, dy_phase_split=False):
    py, ym = pixmap(dy, phase_split=dy_phase_split)
    pa, am = pixmap(a)
    pz, zm = pixmap(dz)
    check(_lib.load().asr_bn_bwd(py, C.byref(ym), pa, C.byref(am), _ptr(mean), _ptr(rstd), _ptr(gamma), act, pz, C.byref(zm),
                                 _ptr(dgamma), _ptr(dbeta), _ptr(ws), _stream()), 'asr_bn_bwd')


def relu_mask(dy, y, dst):
    py, ym = pixmap(dy)
    po, om = pixmap(y)
    pd, dm = pixmap(dst)
    check(_lib.load().asr_relu_mask(py, C.byref(ym), po, C.byref(om), pd, C.byref(dm), _stream()), 'asr_relu_mask')


def conv_s2_expand(w, Cin, Cout, W4):
    check(_lib.load().asr_conv_s2_expand(_ptr(w), Cin, Cout, _ptr(W4), _stream()), 'asr_conv_s2_expand')


def conv_s2_gather(dW4, Cin, Cout, dw):
    check(_lib.load().asr_conv_s2_gather(_ptr(dW4), Cin, Cout, _ptr(dw), _stream()), 'asr_conv_s2_gather')


def conv_s2_arrange_bytes(Cc):
    return _lib.load().asr_conv_s2_arrange_bytes(Cc)


def conv_s2_arrange(W4, Cc, out):
    """The nine non-zero (phase, tap) blocks of the expanded stride-2 weights in data-gradient fragment order (asr_conv_s2_arrange)."""
    check(_lib.load().asr_conv_s2_arrange(_ptr(W4), Cc, _ptr(out), out.numel() * out.element_size(), _stream()), 'asr_conv_s2_arrange')


def conv_s2_dgrad(desc, dz, Wf9, dx):
    """Data-gradient of the phase-split stride-2 conv, one launch per phase of the input gradient (asr_conv_s2_dgrad)."""
    lib = _lib.load()
    pz = dz.ptr if isinstance(dz, Plane) else _ptr(dz)
    px = dx.ptr if isinstance(dx, Plane) else _ptr(dx)
    _timed(desc, lambda: check(lib.asr_conv_s2_dgrad(C.byref(desc), pz, _ptr(Wf9), px, _stream()), 'asr_conv_s2_dgrad'))


def plane_to_T(plane, choff, dst):
    check(_lib.load().asr_plane_to_T(plane.ptr, plane.B, plane.H, plane.W, plane.C, choff, _ptr(dst), _stream()), 'asr_plane_to_T')


def T_to_plane(srcA, srcB, plane, choff):
    check(_lib.load().asr_T_to_plane(_ptr(srcA), _ptr(srcB), plane.B, plane.H, plane.W, plane.C, choff, plane.ptr, _stream()),
          'asr_T_to_plane')


def attention_nomask_fwd(Q, K, V, N, Tq, Tk, Cc, H, O, lse):
    check(_lib.load().asr_attention_nomask_fwd(_ptr(Q), _ptr(K), _ptr(V), N, Tq, Tk, Cc, H, _ptr(O), _ptr(lse), _stream()),
          'asr_attention_nomask_fwd')


def attention_nomask_bwd(Q, K, V, O, dO, lse, N, Tq, Tk, Cc, H, dQ, dK, dV, delta_ws):
    check(_lib.load().asr_attention_nomask_bwd(_ptr(Q), _ptr(K), _ptr(V), _ptr(O), _ptr(dO), _ptr(lse), N, Tq, Tk, Cc, H,
                                               _ptr(dQ), _ptr(dK), _ptr(dV), _ptr(delta_ws), _stream()),
          'asr_attention_nomask_bwd')


def freq_attention_fwd(Q, K, V, B, T, P, O):
    check(_lib.load().asr_freq_attention_fwd(_ptr(Q), _ptr(K), _ptr(V), B, T, _ptr(P), _ptr(O), _stream()), 'asr_freq_attention_fwd')


def freq_attention_bwd(Q, K, V, P, dO, B, T, dQ, dK, dV, dS_ws):
    check(_lib.load().asr_freq_attention_bwd(_ptr(Q), _ptr(K), _ptr(V), _ptr(P), _ptr(dO), B, T, _ptr(dQ), _ptr(dK), _ptr(dV),
                                             _ptr(dS_ws), _stream()), 'asr_freq_attention_bwd')


def pix_add_ln_fwd(a, r, gamma, beta, eps, y, xhat, rstd):
    pa, m = pixmap(a)
    check(_lib.load().asr_pix_add_ln_fwd(pa, pixmap(r)[0], C.byref(m), _ptr(gamma), _ptr(beta), eps, pixmap(y)[0],
                                         pixmap(xhat)[0], _ptr(rstd), _stream()), 'asr_pix_add_ln_fwd')


def pix_ln_bwd_workspace(t):
    _, m = pixmap(t)
    return _lib.load().asr_pix_ln_bwd_workspace(C.byref(m))


def pix_ln_bwd(dy, xhat, rstd, gamma, dx, dgamma, dbeta, ws):
    pd, m = pixmap(dy)
    check(_lib.load().asr_pix_ln_bwd(pd, pixmap(xhat)[0], _ptr(rstd), C.byref(m), _ptr(gamma), pixmap(dx)[0], _ptr(dgamma),
                                     _ptr(dbeta), _ptr(ws), _stream()), 'asr_pix_ln_bwd')


def maxpool_bwd(dy, y, dx):
    """dy: pooled-gradient Plane, y: the pooled tensor's input Plane, dx: Plane like y."""
    check(_lib.load().asr_maxpool_bwd(dy.ptr, y.ptr, y.B, y.H, y.W, y.C, dx.ptr, _stream()), 'asr_maxpool_bwd')


# ---------------------------------------------------------------------------- fp32 contraction on pre-arranged weights
def arrange_weights(W, ntaps, K, N, ldw, wmode=0, out=None):
    """fp32 weights in MFMA fragment order (asr_arrange_weights); returns the float32 buffer."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(lib.asr_arrange_weights_bytes(ntaps, K, N) // 4, dtype=torch.float32, device=W.device)
    check(lib.asr_arrange_weights(_ptr(W), ntaps, K, N, ldw, wmode, _ptr(out), _stream()), 'asr_arrange_weights')
    return out


def winograd_weights_floats(K, N):
    """floats of the buffer asr_winograd_weights2 fills (two layouts side by side, asr_winograd_weights_bytes)"""
    return _lib.load().asr_winograd_weights_bytes(K, N) // 4


def winograd_weights(W, K, N, ldw, wmode=0, out=None):
    """U = G g G^T of a 3x3 layer: [16][K][N] and, 16 K N floats further, wino11_kernel's chunk-major layout (asr_winograd_weights2)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(lib.asr_winograd_weights_bytes(K, N) // 4, dtype=torch.float32, device=W.device)
    check(lib.asr_winograd_weights2(_ptr(W), K, N, ldw, wmode, _ptr(out), out.numel() * 4, _stream()), 'asr_winograd_weights2')
    return out


def winograd_supported(desc):
    return bool(_lib.load().asr_winograd_supported(C.byref(desc)))


def tap_gemm_wino(desc, A, Wt, bias=None, scale=None, shift=None, out_a=None, out_y=None):
    """asr_tap_gemm on Winograd-transformed weights (3x3 convs that asr_winograd_supported accepts)."""
    lib = _lib.load()
    pa = A.ptr if isinstance(A, Plane) else _ptr(A)
    po_a = out_a.ptr if isinstance(out_a, Plane) else _ptr(out_a)
    po_y = out_y.ptr if isinstance(out_y, Plane) else _ptr(out_y)
    _timed(desc, lambda: check(lib.asr_tap_gemm_wino(C.byref(desc), pa, _ptr(Wt), _ptr(bias), _ptr(scale), _ptr(shift), po_a, po_y,
                                                     _stream()), 'asr_tap_gemm_wino'))


def winograd_sum_rows(desc):
    """partial rows asr_tap_gemm_wino_sums writes (0: not a shape of that launch)"""
    return _lib.load().asr_winograd_sum_rows(C.byref(desc))


def tap_gemm_wino_sums(desc, A, Wt, bias, scale, shift, out_a, out_y, y_sums):
    """tap_gemm_wino + the per-image channel sums of out_y as partial rows (asr_tap_gemm_wino_sums): the squeeze of the SE block whose
    branch this cell is."""
    lib = _lib.load()
    _timed(desc, lambda: check(lib.asr_tap_gemm_wino_sums(C.byref(desc), A.ptr, _ptr(Wt), _ptr(bias), _ptr(scale), _ptr(shift), out_a.ptr, out_y.ptr,
                                                          _ptr(y_sums), _stream()), 'asr_tap_gemm_wino_sums'))


def tap_gemm_wino_sesum(desc, dZ, Wt, x, se_scale, se_shift, dy, xsums):
    """Winograd data-gradient into dy (no accumulate) + partial rows of sum dy * (se_scale * x + se_shift) (asr_tap_gemm_wino_sesum)."""
    lib = _lib.load()
    _timed(desc, lambda: check(lib.asr_tap_gemm_wino_sesum(C.byref(desc), dZ.ptr, _ptr(Wt), x.ptr, _ptr(se_scale), _ptr(se_shift), dy.ptr,
                                                           _ptr(xsums), _stream()), 'asr_tap_gemm_wino_sesum'))


def tap_gemm_wino_pool(desc, A, Wt, bias, scale, shift, out_a, pool, y_pooled):
    """Forward conv of a pooled cell in one launch (asr_tap_gemm_wino_pool): out_a as tap_gemm_wino writes it, y_pooled =
    pool_fwd(out_a) bit for bit."""
    lib = _lib.load()
    pa = A.ptr if isinstance(A, Plane) else _ptr(A)
    _timed(desc, lambda: check(lib.asr_tap_gemm_wino_pool(C.byref(desc), pa, _ptr(Wt), _ptr(bias), _ptr(scale), _ptr(shift), out_a.ptr,
                                                          int(pool), y_pooled.ptr, _stream()), 'asr_tap_gemm_wino_pool'))


def tap_gemm_pw(desc, A, Wf, bias=None, scale=None, shift=None, out_a=None, out_y=None):
    """asr_tap_gemm on pre-arranged weights; desc.wmode only labels the launch (forward / data-gradient symbol)."""
    lib = _lib.load()
    pa = A.ptr if isinstance(A, Plane) else _ptr(A)
    po_a = out_a.ptr if isinstance(out_a, Plane) else _ptr(out_a)
    po_y = out_y.ptr if isinstance(out_y, Plane) else _ptr(out_y)
    _timed(desc, lambda: check(lib.asr_tap_gemm_pw(C.byref(desc), pa, _ptr(Wf), _ptr(bias), _ptr(scale), _ptr(shift), po_a, po_y,
                                                   _stream()), 'asr_tap_gemm_pw'))
