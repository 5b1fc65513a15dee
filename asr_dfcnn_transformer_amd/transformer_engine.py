"""Host-side engines of the Transformer path: ``LMEngine`` (the pinyin->hanzi
``Language_Model`` of lm_and_am/model/language_model.py:22-78) and ``E2EEngine`` (the
encoder-decoder of end2end/model.py:267-370).  They own flat parameter / gradient / Adam
buffers and activation tensors and enqueue libasrhip kernels; no arithmetic happens here.

Live graph as written in the reference: in every block loop the FFN result is
stored in a different attribute than the one the next block reads, so N stacked MHA
sub-layers and ONE live FFN (the last block's) remain; the decoder block is a single
causal cross-attention.  Projections are Dense(relu, no bias).  ``tie`` (default on)
re-uses encoder block i's dense kernels in decoder block i (variable-scope reuse, Q8).
Dropout: identity (parity mode).
"""
import math

import numpy as np
import torch

from . import ops

LN_EPS = 1e-8
SMOOTH_EPS = 0.1


def sorted_segments(ids):
    """Positions sorted by id (stable) + segment table: the inputs of asr_embed_bwd (_Base._embed_bwd says when the engines use it
    and when asr_embed_bwd_ids, which needs neither)."""
    flat = np.asarray(ids).reshape(-1)
    perm = np.argsort(flat, kind='stable').astype(np.int32)
    sv = flat[perm]
    uniq, start = np.unique(sv, return_index=True)
    seg = np.concatenate([start, [len(flat)]]).astype(np.int32)
    return perm, uniq.astype(np.int32), seg


def _r4(n):
    return (n + 3) // 4 * 4


class _Base:
    """Flat parameter storage + block helpers shared by the two graphs."""

    def __init__(self, C, heads, device, lr, beta2, decay_steps, min_lr, dropout_rate=0.0, drop_seed=0):
        self.C, self.H, self.device = C, heads, device
        # tf.layers.dropout(training=True) at the reference's sites (transformer.py:111,154,226; model.py:290;
        # language_model.py:34) with the counter-based mask of asr_dropout; 0 = off (all parity runs against TF-free
        # arithmetic use 0, the dropout parity runs compare with the oracle's restatement of the same generator)
        self.dropout_rate, self.drop_seed = float(dropout_rate), int(drop_seed)
        assert C == heads * 64, 'the attention kernels are built for 64-wide heads (512 / 8)'
        self.entries = {}       # name -> (offset, shape)   shape = physical (padded) shape
        self.logical = {}       # name -> logical shape
        self._off = 0
        self.lr0, self.beta1, self.beta2, self.adam_eps = lr, 0.9, beta2, 1e-8
        self.decay_steps, self.min_lr = decay_steps, min_lr
        self.global_step = 0

    # ---- parameters
    def _add(self, name, shape, phys=None):
        phys = tuple(phys or shape)
        self.entries[name] = (self._off, phys)
        self.logical[name] = tuple(shape)
        self._off += _r4(int(np.prod(phys)))

    def _add_mha(self, name, share=None):
        C = self.C
        for k in ('wq', 'wk', 'wv', 'wo'):
            if share is not None:
                self.entries['%s/%s' % (name, k)] = self.entries['%s/%s' % (share, k)]
                self.logical['%s/%s' % (name, k)] = (C, C)
            else:
                self._add('%s/%s' % (name, k), (C, C))
        self._add(name + '/ln_g', (C,)); self._add(name + '/ln_b', (C,))

    def _add_ffn(self, name, share=None):
        C = self.C
        for k, shp in (('w1', (C, 4 * C)), ('b1', (4 * C,)), ('w2', (4 * C, C)), ('b2', (C,))):
            if share is not None:
                self.entries['%s/%s' % (name, k)] = self.entries['%s/%s' % (share, k)]
                self.logical['%s/%s' % (name, k)] = shp
            else:
                self._add('%s/%s' % (name, k), shp)
        self._add(name + '/ln_g', (C,)); self._add(name + '/ln_b', (C,))

    def _finish_params(self):
        z = lambda: torch.zeros(self._off, dtype=torch.float32, device=self.device)
        self.theta, self.grad, self.adam_m, self.adam_v = z(), z(), z(), z()
        self.n_params = len({v[0] for v in self.entries.values()})

    def p(self, name, buf=None):
        off, shape = self.entries[name]
        return (self.theta if buf is None else buf)[off:off + int(np.prod(shape))]

    def g(self, name):
        return self.p(name, self.grad)

    def load_params(self, flat):
        """flat: {name: ndarray} with logical shapes (vocabulary matrices are padded here)."""
        host = self.theta.cpu().numpy()
        for name, (off, phys) in self.entries.items():
            v = np.asarray(flat[name], dtype=np.float32)
            assert tuple(v.shape) == self.logical[name], (name, v.shape, self.logical[name])
            buf = np.zeros(phys, dtype=np.float32)
            buf[tuple(slice(0, s) for s in v.shape)] = v
            host[off:off + buf.size] = buf.ravel()
        self.theta.copy_(torch.from_numpy(host))

    def grads_dict(self, buf=None):
        host = (self.grad if buf is None else buf).cpu().numpy()
        out = {}
        for name, (off, phys) in self.entries.items():
            a = host[off:off + int(np.prod(phys))].reshape(phys)
            out[name] = a[tuple(slice(0, s) for s in self.logical[name])].copy()
        return out

    def params_dict(self):
        return self.grads_dict(self.theta)

    def init_params(self, seed=0):
        """xavier/glorot-uniform matrices, zero biases, LayerNorm gamma 1 / beta 0."""
        rng = np.random.default_rng(seed)
        flat, seen = {}, {}
        for name, (off, phys) in self.entries.items():
            shp = self.logical[name]
            if off in seen:
                flat[name] = flat[seen[off]]
                continue
            seen[off] = name
            if len(shp) == 2:
                lim = math.sqrt(6.0 / (shp[0] + shp[1]))
                flat[name] = rng.uniform(-lim, lim, shp)
            elif name.endswith('ln_g'):
                flat[name] = np.ones(shp)
            else:
                flat[name] = np.zeros(shp)
        self.load_params(flat)

    # ---- dropout sites (ids shared with oracle/transformer.py DROP_SITES / drop_site_seed)
    def _drop_seed(self, site):
        if isinstance(site, tuple):
            kind, i, which = site
            sid = {'enc': 10, 'mha': 10, 'dec': 40}[kind] + 2 * i + (0 if which == 'att' else 1)
        else:
            sid = {'emb_enc': 0, 'emb': 0, 'emb_dec': 1, 'enc_ffn': 70, 'ffn': 70, 'dec_ffn': 71}[site]
        return (self.drop_seed + 1009 * self.global_step + 7919 * sid) & 0xFFFFFFFF

    @staticmethod
    def _block_site(name):
        kind = name.rstrip('0123456789')
        return kind, int(name[len(kind):])

    # ---- low-level helpers
    def _t(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def _ln_part(self, name, rows):
        """The block-partial buffer of LayerNorm site ``name``: its backward leaves the partial sums of dgamma / dbeta there
        (asr_layernorm_bwd_fused with NULL outputs) and _ln_reduce folds the partials of ALL sites of a step in one batched reduction
        (two launches instead of two per LayerNorm; the optimiser is the only reader of those gradients)."""
        reg = self.__dict__.setdefault('_ln_sites', {})
        nblk = ops.layernorm_bwd_blocks(rows)
        if name not in reg or reg[name][1] != nblk:
            # (a site called with another row count: a new buffer and a new batch table -- the kernel writes cdiv(rows, rows per
            # block) partial rows, an old buffer would be overrun or leave stale rows in the fold)
            reg[name] = (self._t(nblk * 2 * self.C), nblk)
            self._ln_batch = None
        return reg[name][0]

    def _ln_reduce(self):
        reg = self.__dict__.get('_ln_sites')
        if not reg:
            return
        if self.__dict__.get('_ln_batch') is None:
            C = self.C
            self._ln_batch = ops.ReduceBatch([(part, nblk, 2 * C, [(C, self.g(n + '/ln_g')), (C, self.g(n + '/ln_b'))])
                                              for n, (part, nblk) in reg.items()])
        self._ln_batch.run()

    def _embed_bwd(self, dout, ids_dev, ids_host, rows, V, zero_pad, scale, dtable):
        """Gradient of an embedding table.  Small tables x few positions (the language model: 1536 x 6400) straight from the ids
        on the device (asr_embed_bwd_ids: a wave per table row scans the ids -- V * rows / 64 wave iterations, ~10 us there, no
        host work); the large ones (6348 x 32768: 200 us of scanning) keep the host-side stable sort + asr_embed_bwd, whose
        uploads ride a GPU-bound step.  Both give the same bits."""
        C = self.C
        if V * rows <= (1 << 25):
            ops.embed_bwd_ids(dout, ids_dev, rows, V, C, zero_pad, scale, dtable)
            return
        perm, uniq, seg = sorted_segments(ids_host)
        if not hasattr(self, '_seg') or self._seg[0].numel() < rows:
            self._seg = [self._t(rows, dtype=torch.int32), self._t(rows, dtype=torch.int32), self._t(rows + 1, dtype=torch.int32)]
        self._upload('seg0', self._seg[0], perm); self._upload('seg1', self._seg[1], uniq); self._upload('seg2', self._seg[2], seg)
        ops.embed_bwd(dout, self._seg[0], self._seg[1], self._seg[2], len(uniq), C, zero_pad, scale, dtable)

    def _upload(self, key, dst, arr):
        """dst[:len] <- int32 host array through a PINNED staging buffer owned by the engine (two per site, alternating, each
        guarded by the event of its last copy).  A copy_ from pageable memory runs as several blit KERNELS on the compute stream
        (rocprofv3: 22 __amd_rocclr_copyBuffer launches = 0.57 ms per language-model step for five small id arrays); from pinned
        memory it is one DMA transfer."""
        arr = np.ascontiguousarray(arr, dtype=np.int32).reshape(-1)
        n = arr.size
        if not hasattr(self, '_stage'):
            self._stage = {}
        slot = self._stage.setdefault(key, {'i': 0, 'buf': [None, None], 'ev': [None, None]})
        i = slot['i']; slot['i'] = i ^ 1
        if slot['buf'][i] is None or slot['buf'][i].numel() < n:
            slot['buf'][i] = torch.empty(max(n, 16), dtype=torch.int32, pin_memory=True)
            slot['ev'][i] = None
        if slot['ev'][i] is not None:
            slot['ev'][i].synchronize()                      # the copy that last read this buffer (two uploads ago) is done
        slot['buf'][i].numpy()[:n] = arr
        dst.view(-1)[:n].copy_(slot['buf'][i][:n], non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
        slot['ev'][i] = ev

    def _dense(self, x, rows, K, N, w, b, out, relu):
        """tf.layers.dense forward.  The kernel also exists transposed ([N][K], refreshed once per forward by ONE batched
        launch, _wt_for): the large GEMMs read both operands K-contiguous through LDS-DMA (asr_tap_gemm_nt, gemm1.hip)."""
        d = ops.gemm_desc(rows, K, N, K, N, 0, N, ntaps=1, relu=1 if relu else 0)
        ops.tap_gemm_nt(d, x, w, self._wt_for([(w, K, N, N)]), K, b, None, None, None, out)

    def _wt_for(self, parts):
        """The transposed copy of a dense kernel given as column blocks ``parts`` = [(w [K][n_j] tensor, K, n_j, pitch)]:
        a [sum n_j][K] tensor whose rows are the kernel's columns.  Registered on first use (and transposed at once);
        afterwards the first request of a forward pass re-transposes EVERY registered kernel in one launch
        (asr_transpose_batch) -- the parameters may have been stepped, loaded or perturbed since the last forward."""
        reg = self.__dict__.setdefault('_wt_reg', {})
        key = tuple((w.data_ptr(), K, n, ld) for w, K, n, ld in parts)
        if key not in reg:
            K = parts[0][1]
            wt = self._t(sum(n for _, _, n, _ in parts) * K)
            items, row = [], 0
            for w, _, n, ld in parts:
                items.append((wt[row * K:], K, w, ld, K, n))
                row += n
            ops.Copy2dBatch(items).run_transposed()
            reg[key] = (wt, items)
            self._wt_batch = None
            return wt
        if self.__dict__.get('_wt_dirty', True):
            if self.__dict__.get('_wt_batch') is None:
                self._wt_batch = ops.Copy2dBatch([it for _, items in reg.values() for it in items])
            self._wt_batch.run_transposed()
            self._wt_dirty = False
        return reg[key][0]

    def _dense_dgrad(self, dy, rows, K, N, w, dx, accumulate):
        # dx[rows,K] (+)= dy[rows,N] . w[K,N]^T
        d = ops.gemm_desc(rows, N, K, N, N, 0, K, ntaps=1, wmode=1, accumulate=1 if accumulate else 0)
        ops.tap_gemm(d, dy, w, None, None, None, None, dx)

    def _wgrad(self, x, dy, rows, K, N, name):
        """grad[name] (+)= x^T dy; a tensor shared by two layers receives the sum."""
        d = ops.gemm_desc(rows, K, N, K, N, ntaps=1)
        shared = name_off(self, name) in self._written
        self._written.add(name_off(self, name))

        if shared:
            ops.tap_wgrad(d, x, dy, N, self._wtmp[:K * N], self.ws)
            ops.axpy(self.g(name), self._wtmp[:K * N], 1.0, True)
        else:
            ops.tap_wgrad(d, x, dy, N, self.g(name), self.ws)

    def _bgrad(self, dy, rows, N, name):
        if name_off(self, name) in self._written:
            ops.colsum(dy, rows, N, N, self._btmp[:N], self.ws)
            ops.axpy(self.g(name), self._btmp[:N], 1.0, True)
        else:
            ops.colsum(dy, rows, N, N, self.g(name), self.ws)
            self._written.add(name_off(self, name))

    # ---- blocks
    def _mha_alloc(self, N, Tq, Tk):
        C, H = self.C, self.H
        # Q | K | V live in ONE buffer so that the projections can run as one GEMM (see _mha_fwd): [rows][3C] for
        # self-attention, [rq][C] followed by [rk][2C] when queries and keys come from different tensors
        return {'QKV': self._t(N * Tq * C + 2 * N * Tk * C), 'W3': self._t(C, 3 * C), 'A': self._t(N * Tq, C),
                'Z': self._t(N * Tq, C), 'xhat': self._t(N * Tq, C), 'rstd': self._t(N * Tq), 'lse': self._t(2 * N * H * Tq),
                'out': self._t(N * Tq, C), 'N': N, 'Tq': Tq, 'Tk': Tk, 'stats': self._t(ops.attention_stats_floats(N, Tq, Tk, self.H))}

    def _qkv_views(self, st, fused3):
        C, rq, rk = self.C, st['N'] * st['Tq'], st['N'] * st['Tk']
        buf = st['QKV']
        if fused3:                                   # Q, K, V = column blocks of one [rows][3C] matrix
            return buf[0:], buf[C:], buf[2 * C:], 3 * C, 3 * C
        kv = buf[rq * C:]                            # Q dense [rq][C]; K, V = column blocks of [rk][2C]
        return buf[0:], kv[0:], kv[C:], C, 2 * C

    def _mha_fwd(self, name, st, q_in, k_in, causal):
        """multihead_attention (transformer.py:117-158).  The three projections Dense(relu) of a block read at most two
        inputs, so they run as ONE GEMM against [wq | wk | wv] (self-attention, N = 3C) or as wq plus ONE GEMM against
        [wk | wv] (keys from another tensor): the weight matrices are packed side by side once per step (asr_copy2d) and
        the attention kernels take Q, K, V as column blocks of the result (row pitches, asr_attention_fwd_p).  Same
        arithmetic per output element; a 32768 x 512 x 1536 GEMM runs at ~120 TFLOP/s where three x512 ones reach ~106."""
        C, N, Tq, Tk = self.C, st['N'], st['Tq'], st['Tk']
        rq, rk = N * Tq, N * Tk
        st['q_in'], st['k_in'], st['causal'] = q_in, k_in, causal
        fused3 = (q_in is k_in) and rq == rk
        st['fused3'] = fused3
        W3 = st['W3']
        self._pack_qkv(name, st)
        Q, K, V, ldq, ldk = self._qkv_views(st, fused3)
        wq, wk, wv = (self.p('%s/%s' % (name, k)) for k in ('wq', 'wk', 'wv'))
        if fused3:
            d = ops.gemm_desc(rq, C, 3 * C, C, 3 * C, 0, 3 * C, ntaps=1, relu=1)
            ops.tap_gemm_nt(d, q_in, W3, self._wt_for([(wq, C, C, C), (wk, C, C, C), (wv, C, C, C)]), C, None, None, None, None, Q)
        else:
            self._dense(q_in, rq, C, C, wq, None, Q, True)
            d = ops.gemm_desc(rk, C, 2 * C, C, 3 * C, 0, 2 * C, ntaps=1, relu=1)
            ops.tap_gemm_nt(d, k_in, W3.view(-1)[C:], self._wt_for([(wk, C, C, C), (wv, C, C, C)]), C, None, None, None, None, K)
        rate = self._rate
        kind, blk = self._block_site(name)
        st['seed_att'], st['seed_out'] = self._drop_seed((kind, blk, 'att')), self._drop_seed((kind, blk, 'out'))
        # query masks / key biases of the block once (the attention kernels' workgroups each recomputed them from Q / K: 6-12 % of their time)
        ops.attention_stats(Q, K, N, Tq, Tk, C, self.H, st['stats'], ldq=ldq, ldk=ldk)
        ops.attention_fwd(Q, K, V, N, Tq, Tk, C, self.H, causal, st['A'], st['lse'], rate, st['seed_att'], ldq=ldq, ldk=ldk, stats=st['stats'])
        self._dense(st['A'], N * Tq, C, C, self.p(name + '/wo'), None, st['Z'], True)
        # dropout(Z) + residual + LayerNorm in one pass; Z itself stays undropped, the backward regenerates the mask
        ops.add_layernorm_fwd_dropout(st['Z'], q_in, self.p(name + '/ln_g'), self.p(name + '/ln_b'), N * Tq, C, LN_EPS,
                                      rate, st['seed_out'], st['out'], st['xhat'], st['rstd'])
        return st['out']

    def _pack_qkv(self, name, st):
        """[wq | wk | wv] of every attention block, packed side by side for the fused projection GEMMs.  The first forward
        copies block by block and records the copies; afterwards the first block of a forward runs them all as ONE launch
        (asr_copy2d_batch; 3 x blocks launches per step before)."""
        C = self.C
        reg = self.__dict__.setdefault('_pack_reg', {})
        key = (name, id(st))
        if key not in reg:
            items = [(st['W3'].view(-1)[j * C:], 3 * C, self.p('%s/%s' % (name, k)), C, C, C) for j, k in enumerate(('wq', 'wk', 'wv'))]
            for it in items:
                ops.copy2d(*it)
            reg[key] = items
            self._pack_batch = None
            return
        if self.__dict__.get('_pack_dirty', True):
            if self.__dict__.get('_pack_batch') is None:
                self._pack_batch = ops.Copy2dBatch([it for items in reg.values() for it in items])
            self._pack_batch.run(False)
            self._pack_dirty = False

    def _begin_forward(self):
        """Start of a forward pass: the packed projection weights are rebuilt from the parameters once (they may have
        been stepped, loaded or perturbed since the last forward), and so are the transposed kernels (_wt_for)."""
        self._pack_dirty = True
        self._wt_dirty = True

    def _wgrad_packed(self, x, dy, rows, K, names, ldz):
        """[g(names[0]) | g(names[1]) | ...] += x^T dy for weight matrices packed side by side (dy [rows][len(names) * C],
        row pitch ldz): one weight-gradient GEMM into scratch, scattered back into the separate gradient tensors."""
        C = self.C
        Nf = len(names) * C
        tmp = self._wtmp[:K * Nf]
        d = ops.gemm_desc(rows, K, Nf, K, Nf, ntaps=1)
        ops.tap_wgrad(d, x, dy, ldz, tmp, self.ws)
        cache = self.__dict__.setdefault('_unpack_batches', {})
        key = (tuple(names), K, tmp.data_ptr())
        if key not in cache:
            cache[key] = ops.Copy2dBatch([(self.g(nm), C, tmp[j * C:], Nf, K, C) for j, nm in enumerate(names)])
        cache[key].run(True)                                            # the gradient buffer was zeroed at the start of backward
        for nm in names:
            self._written.add(name_off(self, nm))

    def _mha_bwd(self, name, st, dout, dq_in, dq_acc, dk_in, dk_acc):
        """dq_in (+)= dL/d(queries), dk_in (+)= dL/d(keys); dk_in may be dq_in (self-attention)."""
        C, N, Tq, Tk = self.C, st['N'], st['Tq'], st['Tk']
        rq, rk = N * Tq, N * Tk
        dZ, dA = self.sc['b'][:rq * C], self.sc['c'][:rq * C]
        # one pass: LayerNorm backward, dq_in (+)= its result (the residual branch) and dZ = the dropout + Dense(relu) backward
        # of it (mask = kept by the generator AND Z > 0, factor 1 / (1 - rate))
        ops.layernorm_bwd_fused(dout, st['xhat'], st['rstd'], self.p(name + '/ln_g'), rq, C, None, dq_in, dq_acc, st['Z'],
                                1.0 / (1.0 - self._rate) if self._rate > 0 else 1.0, dZ,
                                None, None, self._ln_part(name, rq), self._rate, st['seed_out'])
        self._wgrad(st['A'], dZ, rq, C, C, name + '/wo')
        self._dense_dgrad(dZ, rq, C, C, self.p(name + '/wo'), dA, False)
        fused3 = st['fused3']
        Q, K, V, ldq, ldk = self._qkv_views(st, fused3)
        dbuf = self.sc['qkv']
        if fused3:
            dQ, dK, dV = dbuf[0:], dbuf[C:], dbuf[2 * C:]
        else:
            dkv = dbuf[rq * C:]
            dQ, dK, dV = dbuf[0:], dkv[0:], dkv[C:]
        ops.attention_bwd(Q, K, V, st['A'], dA, st['lse'], N, Tq, Tk, C, self.H, st['causal'],
                          dQ, dK, dV, self.ws, relu_grad=True,      # gradients of the pre-ReLU projections
                          dropout_rate=self._rate, seed=st['seed_att'], ldq=ldq, ldk=ldk, stats=st['stats'])
        W3 = st['W3']
        if fused3:
            self._wgrad_packed(st['q_in'], dQ, rq, C, [name + '/wq', name + '/wk', name + '/wv'], 3 * C)
            self._dense_dgrad(dQ, rq, C, 3 * C, W3, dq_in, True)              # dx += [dQ | dK | dV] . [wq | wk | wv]^T
        else:
            self._wgrad(st['q_in'], dQ, rq, C, C, name + '/wq')
            self._dense_dgrad(dQ, rq, C, C, self.p(name + '/wq'), dq_in, True)
            self._wgrad_packed(st['k_in'], dK, rk, C, [name + '/wk', name + '/wv'], 2 * C)
            d = ops.gemm_desc(rk, 2 * C, C, 2 * C, 3 * C, 0, C, ntaps=1, wmode=1, accumulate=1 if dk_acc else 0)
            ops.tap_gemm(d, dK, W3.view(-1)[C:], None, None, None, None, dk_in)   # dk_in (+)= [dK | dV] . [wk | wv]^T

    def _ffn_alloc(self, rows):
        C = self.C
        return {'H': self._t(rows, 4 * C), 'Y': self._t(rows, C), 'xhat': self._t(rows, C), 'rstd': self._t(rows),
                'out': self._t(rows, C), 'rows': rows}

    def _ffn_fwd(self, name, st, x):
        C, rows = self.C, st['rows']
        st['x'] = x
        self._dense(x, rows, C, 4 * C, self.p(name + '/w1'), self.p(name + '/b1'), st['H'], True)
        self._dense(st['H'], rows, 4 * C, C, self.p(name + '/w2'), self.p(name + '/b2'), st['Y'], False)
        st['seed'] = self._drop_seed(name)
        ops.add_layernorm_fwd_dropout(st['Y'], x, self.p(name + '/ln_g'), self.p(name + '/ln_b'), rows, C, LN_EPS,
                                      self._rate, st['seed'], st['out'], st['xhat'], st['rstd'])
        return st['out']

    def _ffn_bwd(self, name, st, dout, dx, dx_acc):
        C, rows = self.C, st['rows']
        dY = self.sc['b'][:rows * C]
        dH = self.sc['h'][:rows * 4 * C]
        # one pass: dx (+)= the LayerNorm input gradient (the residual branch), dY = the same through the dropout of the FFN output
        ops.layernorm_bwd_fused(dout, st['xhat'], st['rstd'], self.p(name + '/ln_g'), rows, C, None, dx, dx_acc, None,
                                1.0 / (1.0 - self._rate) if self._rate > 0 else 1.0, dY,
                                None, None, self._ln_part(name, rows), self._rate, st['seed'])
        self._bgrad(dY, rows, C, name + '/b2')
        self._wgrad(st['H'], dY, rows, 4 * C, C, name + '/w2')
        # dH = (dY . w2^T) where H > 0: the ReLU backward of the first Dense rides in the data-gradient's epilogue (no pass over rows x 4C)
        ops.tap_gemm_relu_bwd(ops.gemm_desc(rows, C, 4 * C, C, C, 0, 4 * C, ntaps=1, wmode=1), dY, self.p(name + '/w2'), st['H'], dH)
        self._bgrad(dH, rows, 4 * C, name + '/b1')
        self._wgrad(st['x'], dH, rows, C, 4 * C, name + '/w1')
        self._dense_dgrad(dH, rows, C, 4 * C, self.p(name + '/w1'), dx, True)

    def _alloc_scratch(self, max_rows, max_w, gemms):
        """gemms: (rows, K, N) of every weight-gradient GEMM, to size the split-K slab workspace."""
        C = self.C
        # One stream: running the weight-gradient GEMMs on a second stream beside the data-gradient GEMMs was measured 2 % SLOWER
        # on configs[3] in round 1 (both are MFMA-bound one-round grids, there is no HBM-bound prologue to hide) and removed.
        self.sc = {k: self._t(max_rows * C) for k in 'abcdef'}
        self.sc['h'] = self._t(max_rows * 4 * C)
        self.sc['qkv'] = self._t(max_rows * 3 * C)            # d[Q | K | V] of a block (see _mha_bwd)
        self._wtmp = self._t(max_w)
        self._btmp = self._t(max(4 * C, self.Vp, 1024))      # bias-gradient scratch (_bgrad)
        ws = max(ops.layernorm_bwd_workspace(max_rows, C), ops.colsum_workspace(max_rows, 4 * C),
                 ops.colsum_workspace(max_rows, self.Vp), 4 * (max_rows * self.H + 64), 1 << 20)
        for rows, K, N in list(gemms) + [(max_rows, C, 3 * C), (max_rows, C, 2 * C)]:
            ws = max(ws, ops.tap_wgrad_workspace(ops.gemm_desc(rows, K, N, K, N, ntaps=1)))
        self.ws = self._t(ws // 4 + 64)

    # ---- loss head
    def _head_alloc(self, rows):
        self.logits = self._t(rows, self.Vp)
        self.dlogits = self._t(rows, self.Vp)
        self.loss_rows = self._t(rows)
        self.preds = self._t(rows, dtype=torch.int32)
        self.stats = self._t(rows, 2)
        self.stat_sum = self._t(4)
        self.target = self._t(rows, dtype=torch.int32)

    def _head_fwd(self, x, rows, target_host, train):
        C = self.C
        self._dense(x, rows, C, self.Vp, self.p('out_w'), self.p('out_b'), self.logits, False)
        tg = np.ascontiguousarray(np.asarray(target_host, dtype=np.int32).reshape(-1))
        self._upload('target', self.target, tg)
        cnt = float((tg != 0).sum())
        self._count = cnt
        ops.smoothed_ce(self.logits, self.Vp, self.target, rows, self.V, SMOOTH_EPS, 0, 1.0 / max(cnt, 1.0),
                        self.loss_rows, self.preds, self.stats, self.dlogits if train else None)
        ops.colsum(self.stats, rows, 2, 2, self.stat_sum[:2], self.ws)

    def _head_bwd(self, x, rows, dx):
        C = self.C
        self._wgrad(x, self.dlogits, rows, C, self.Vp, 'out_w')
        self._bgrad(self.dlogits, rows, self.Vp, 'out_b')
        self._dense_dgrad(self.dlogits, rows, C, self.Vp, self.p('out_w'), dx, False)

    def fetch(self):
        s = self.stat_sum.cpu().numpy()
        c = max(self._count, 1.0)
        return float(s[0]) / c, float(s[1]) / c            # mean_loss, acc

    # ---- optimiser (tf.train.polynomial_decay(cycle, power 0.5) + AdamOptimizer)
    def current_learning_rate(self, step=None):
        step = float(self.global_step if step is None else step)
        ds = float(self.decay_steps)
        mult = 1.0 if step == 0 else math.ceil(step / ds)
        return (self.lr0 - self.min_lr) * math.sqrt(max(0.0, 1.0 - step / (ds * mult))) + self.min_lr

    def apply_adam(self, gscale=1.0):
        lr = self.current_learning_rate()
        t = self.global_step + 1
        lr_t = lr * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        ops.adam_tf(self.theta, self.grad, self.adam_m, self.adam_v, lr_t, self.beta1, self.beta2, self.adam_eps, gscale)
        self.global_step += 1
        return lr


def name_off(eng, name):
    return eng.entries[name][0]


class LMEngine(_Base):
    """Language_Model (language_model.py:22-78): embedding*sqrt(d) (zero_pad) + learned positions ->
    ``num_blocks`` causal self-attention MHA sub-layers -> one FFN -> dense(V_hanzi) ->
    label-smoothed CE masked by y != 0; Adam(beta2 0.999) on the polynomial-decay lr_lm."""

    def __init__(self, vin=1536, vout=6345, N=64, T=100, C=512, heads=8, blocks=12, pos_max=100, lr=5e-5,
                 decay_steps=5000, min_lr=1e-6, seed=0, device='cuda', dropout_rate=0.0, drop_seed=0):
        super().__init__(C, heads, device, lr, 0.999, decay_steps, min_lr, dropout_rate, drop_seed)
        assert T <= pos_max, 'positions >= position_max_length index past the table (language_model.py:29-30)'
        self.vin, self.V, self.Vp, self.N, self.T, self.blocks, self.pos_max = vin, vout, _r4(vout), N, T, blocks, pos_max
        self._add('emb', (vin, C)); self._add('pos', (pos_max, C))
        for i in range(blocks):
            self._add_mha('mha%d' % i)
        self._add_ffn('ffn')
        self._add('out_w', (C, vout), (C, self.Vp)); self._add('out_b', (vout,), (self.Vp,))
        self._finish_params()
        self.init_params(seed)
        rows = N * T
        self.ids = self._t(N, T, dtype=torch.int32)
        self.x0 = self._t(rows, C)
        self.mha = [self._mha_alloc(N, T, T) for _ in range(blocks)]
        self.ffn = self._ffn_alloc(rows)
        self._head_alloc(rows)
        self.dstream = [self._t(rows * C), self._t(rows * C)]
        self._alloc_scratch(rows, max(C * self.Vp, 4 * C * C),
                            [(rows, C, C), (rows, C, 4 * C), (rows, 4 * C, C), (rows, C, self.Vp)])

    def flat_from_oracle(self, P):
        flat = {'emb': P['emb'], 'pos': P['pos'], 'out_w': P['out_w'], 'out_b': P['out_b']}
        for i in range(self.blocks):
            for k, v in P['mha%d' % i].items():
                flat['mha%d/%s' % (i, k)] = v
        for k, v in P['ffn'].items():
            flat['ffn/%s' % k] = v
        return flat

    def forward(self, x_ids, y=None, train=True):
        self._begin_forward()
        N, T, C = self.N, self.T, self.C
        xi = np.ascontiguousarray(np.asarray(x_ids, dtype=np.int32))
        assert xi.shape == (N, T)
        self._x_host = xi
        self._upload('ids', self.ids, xi)
        ops.embed_fwd(self.p('emb'), self.ids, self.p('pos'), N, T, C, True, float(C) ** 0.5, self.x0)
        self._rate = self.dropout_rate if train else 0.0
        self._seed_emb = self._drop_seed('emb')
        if self._rate > 0:
            ops.dropout(self.x0, self._rate, self._seed_emb)          # language_model.py:34
        enc = self.x0
        for i in range(self.blocks):
            enc = self._mha_fwd('mha%d' % i, self.mha[i], enc, enc, True)
        self.enc = enc
        out = self._ffn_fwd('ffn', self.ffn, enc)
        if y is not None:
            self._head_fwd(out, N * T, y, train)
        else:
            self._dense(out, N * T, C, self.Vp, self.p('out_w'), self.p('out_b'), self.logits, False)
        return self.logits

    def backward(self):
        N, T, C = self.N, self.T, self.C
        rows = N * T
        self.grad.zero_()
        self._written = set()
        d0, d1 = self.dstream
        self._head_bwd(self.ffn['out'], rows, d0)
        self._ffn_bwd('ffn', self.ffn, d0, d1, False)
        cur, nxt = d1, d0
        for i in reversed(range(self.blocks)):
            self._mha_bwd('mha%d' % i, self.mha[i], cur, nxt, False, nxt, True)
            cur, nxt = nxt, cur
        if self._rate > 0:
            ops.dropout(cur, self._rate, self._seed_emb)
        self._embed_bwd(cur, self.ids, self._x_host, rows, self.vin, True, float(C) ** 0.5, self.g('emb'))
        ops.colsum(cur, N, T * C, T * C, self.g('pos')[:T * C], self.ws)
        self._ln_reduce()


class E2EEngine(_Base):
    """Transformer_Model.{embedding_input, encoder, decoder, loss} (end2end/model.py:267-370).
    Encoder input: features [N,T,Din] -> dense(relu)+LayerNorm+enc_pe; or, for the pinyin->hanzi
    configuration of BASELINE.json configs[3], pinyin ids through an embedding (``vin``)."""

    def __init__(self, din=5120, vout=6347, N=8, T=150, L=50, C=512, heads=8, blocks=6, pos_max=600, tie=True, vin=None,
                 lr=5e-4, decay_steps=5000, min_lr=1e-6, seed=0, device='cuda', need_dx=False, dropout_rate=0.0, drop_seed=0):
        super().__init__(C, heads, device, lr, 0.98, decay_steps, min_lr, dropout_rate, drop_seed)
        self.need_dx = need_dx and vin is None       # dL/d(features) for the pre-net in front (prenet_engine.py)
        assert T <= pos_max and L <= pos_max
        self.din, self.V, self.Vp, self.N, self.T, self.L = din, vout, _r4(vout), N, T, L
        self.blocks, self.tie, self.vin, self.pos_max = blocks, tie, vin, pos_max
        if vin is None:
            self._add('in_w', (din, C)); self._add('in_b', (C,)); self._add('in_ln_g', (C,)); self._add('in_ln_b', (C,))
        else:
            self._add('enc_emb', (vin, C))
        self._add('enc_pe', (pos_max, C)); self._add('dec_pe', (pos_max, C)); self._add('dec_input', (vout, C))
        for i in range(blocks):
            self._add_mha('enc%d' % i)
        for i in range(blocks):
            self._add_mha('dec%d' % i, share=('enc%d' % i) if tie else None)
        self._add_ffn('enc_ffn')
        self._add_ffn('dec_ffn', share='enc_ffn' if tie else None)
        self._add('out_w', (C, vout), (C, self.Vp)); self._add('out_b', (vout,), (self.Vp,))
        self._finish_params()
        self.init_params(seed)
        re, rd = N * T, N * L
        self.u = self._t(re, C); self.u_xhat = self._t(re, C); self.u_rstd = self._t(re)
        self.enc0 = self._t(re, C); self.dec0 = self._t(rd, C); self.pe_tmp = self._t(re, C)
        self.x_ids = self._t(N, T, dtype=torch.int32)
        self.y_ids = self._t(N, L, dtype=torch.int32)
        self.enc = [self._mha_alloc(N, T, T) for _ in range(blocks)]
        self.dec = [self._mha_alloc(N, L, T) for _ in range(blocks)]
        self.enc_ffn, self.dec_ffn = self._ffn_alloc(re), self._ffn_alloc(rd)
        self._head_alloc(rd)
        mr = max(re, rd)
        self.dstream = [self._t(mr * C), self._t(mr * C)]
        self.dmem = self._t(re * C)
        self.dx_feat = self._t(re, din) if self.need_dx else None
        self._alloc_scratch(mr, max(C * self.Vp, 4 * C * C, din * C),
                            [(r, C, C) for r in (re, rd)] + [(r, C, 4 * C) for r in (re, rd)] +
                            [(r, 4 * C, C) for r in (re, rd)] + [(rd, C, self.Vp), (re, din, C)])

    def flat_from_oracle(self, P):
        flat = {}
        for k, v in P.items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    flat['%s/%s' % (k, kk)] = vv
            else:
                flat[k] = v
        return flat

    def forward(self, x, y_in, y_tgt=None, train=True):
        self._begin_forward()
        N, T, L, C = self.N, self.T, self.L, self.C
        re, rd = N * T, N * L
        self._rate = self.dropout_rate if train else 0.0
        if self.vin is None:
            assert tuple(x.shape) == (N, T, self.din) and x.is_contiguous()
            self.x_feat = x
            self._dense(x, re, self.din, C, self.p('in_w'), self.p('in_b'), self.u, True)
            ops.add_layernorm_fwd(self.u, None, self.p('in_ln_g'), self.p('in_ln_b'), re, C, LN_EPS, self.enc0,
                                  self.u_xhat, self.u_rstd)
            ops.embed_fwd(None, None, self.p('enc_pe'), N, T, C, False, 1.0, self.pe_tmp)
            ops.axpy(self.enc0, self.pe_tmp, 1.0, True)
        else:
            xi = np.ascontiguousarray(np.asarray(x, dtype=np.int32))
            self._x_host = xi
            self._upload('x_ids', self.x_ids, xi)
            ops.embed_fwd(self.p('enc_emb'), self.x_ids, self.p('enc_pe'), N, T, C, True, float(C) ** 0.5, self.enc0)
        self._seed_emb = self._drop_seed('emb_enc')
        if self._rate > 0:
            ops.dropout(self.enc0, self._rate, self._seed_emb)        # model.py:290 (the decoder input is not dropped)
        yi = np.ascontiguousarray(np.asarray(y_in, dtype=np.int32))
        self._y_host = yi
        self._upload('y_ids', self.y_ids, yi)
        ops.embed_fwd(self.p('dec_input'), self.y_ids, self.p('dec_pe'), N, L, C, False, 1.0, self.dec0)
        e = self.enc0
        for i in range(self.blocks):
            e = self._mha_fwd('enc%d' % i, self.enc[i], e, e, False)
        memory = self._ffn_fwd('enc_ffn', self.enc_ffn, e)
        self.memory = memory
        d = self.dec0
        for i in range(self.blocks):
            d = self._mha_fwd('dec%d' % i, self.dec[i], d, memory, True)
        out = self._ffn_fwd('dec_ffn', self.dec_ffn, d)
        if y_tgt is not None:
            self._head_fwd(out, rd, y_tgt, train)
        else:
            self._dense(out, rd, C, self.Vp, self.p('out_w'), self.p('out_b'), self.logits, False)
        return self.logits

    def backward(self):
        N, T, L, C = self.N, self.T, self.L, self.C
        re, rd = N * T, N * L
        self.grad.zero_()
        self._written = set()
        d0, d1 = self.dstream[0][:rd * C], self.dstream[1][:rd * C]
        self._head_bwd(self.dec_ffn['out'], rd, d0)
        self._ffn_bwd('dec_ffn', self.dec_ffn, d0, d1, False)
        cur, nxt = d1, d0
        first = True
        for i in reversed(range(self.blocks)):
            self._mha_bwd('dec%d' % i, self.dec[i], cur, nxt, False, self.dmem, not first)
            first = False
            cur, nxt = nxt, cur
        self._embed_bwd(cur, self.y_ids, self._y_host, rd, self.logical['dec_input'][0], False, 1.0, self.g('dec_input'))
        ops.colsum(cur, N, L * C, L * C, self.g('dec_pe')[:L * C], self.ws)
        e0, e1 = self.dstream[0][:re * C], self.dstream[1][:re * C]
        self._ffn_bwd('enc_ffn', self.enc_ffn, self.dmem, e0, False)
        cur, nxt = e0, e1
        for i in reversed(range(self.blocks)):
            self._mha_bwd('enc%d' % i, self.enc[i], cur, nxt, False, nxt, True)
            cur, nxt = nxt, cur
        if self._rate > 0:
            ops.dropout(cur, self._rate, self._seed_emb)
        ops.colsum(cur, N, T * C, T * C, self.g('enc_pe')[:T * C], self.ws)
        if self.vin is None:
            du = self.sc['b'][:re * C]
            ops.layernorm_bwd(cur, self.u_xhat, self.u_rstd, self.p('in_ln_g'), re, C, du, False, self.g('in_ln_g'),
                              self.g('in_ln_b'), self.ws)
            ops.relu_bwd(du, self.u, du)
            self._wgrad(self.x_feat, du, re, self.din, C, 'in_w')
            self._bgrad(du, re, C, 'in_b')
            if self.need_dx:
                self._dense_dgrad(du, re, self.din, C, self.p('in_w'), self.dx_feat, False)
        else:
            self._embed_bwd(cur, self.x_ids, self._x_host, re, self.vin, True, float(C) ** 0.5, self.g('enc_emb'))
        self._ln_reduce()
