"""``CNNCTCModel``: the acoustic-model class of the reference (lm_and_am/model/
acoustic_model2.py:10-93 and its siblings acoustic_model.py / acoustic_model3.py) with the
same constructor, placeholder names and fetch names, driven session-style:

    model = CNNCTCModel(am_hp, acoustic_vocab_size, label_vocab_size)
    loss, mean_loss, lr, summary, label_err, _ = model.run(
        [model.loss, model.mean_loss, model.current_learning, model.summary, model.label_err, model.train_op],
        feed_dict={model.wav_input: x, model.logits_length: n, model.target_py: y,
                   model.target_length: l, model.drop_rate: 0.5})        # lm_and_am/train.py:59-69

Placeholders / fetches are plain string handles; ``run`` enqueues the HIP kernels for what
the fetch list needs (forward; + CTC/decode; + backward/Adam when ``train_op`` is fetched).
``drop_rate`` is accepted and ignored: tf.layers.dropout without training= is the identity
in the reference; ``target_length`` is ignored like on the reference's sparse
label path (Q6)."""
import numpy as np
import torch

from .engine import DFCNNEngine
from .parallel import BucketedAllReduce


class SparseTensorValue:
    """(indices [n,2] int64, values [n] int64, dense_shape [2]) as tf.nn.ctc_greedy_decoder returns."""

    def __init__(self, decoded):
        idx = [(b, j) for b, ids in enumerate(decoded) for j in range(len(ids))]
        self.indices = np.asarray(idx, dtype=np.int64).reshape(-1, 2)
        self.values = np.asarray([v for ids in decoded for v in ids], dtype=np.int64)
        self.dense_shape = np.asarray([len(decoded), max([len(d) for d in decoded] + [0])], dtype=np.int64)


class CNNCTCModel:
    variant = 'm2'          # acoustic_model2.py (the class lm_and_am/train.py:8 imports)

    # handles (placeholders and fetches)
    wav_input, logits_length, target_py, target_length, drop_rate = (
        'wav_input', 'logits_length', 'target_py', 'target_length', 'dropout_rate')
    logits, loss, mean_loss, current_learning, summary, label_err, train_op, log_prob = (
        'logits', 'loss', 'mean_loss', 'current_learning', 'summary', 'label_err', 'train_op', 'log_prob')
    decoded = ('decoded',)

    def __init__(self, args, acoustic_vocab_size, label_vocab_size, batch_size=None, variant=None, widths=None,
                 seed=0, device='cuda', data_parallel=True):
        self.acoustic_vocab_size = acoustic_vocab_size
        self.label_vocab_size = label_vocab_size
        self.gpu_nums = args.gpu_nums
        self.lm_lr = args.am_lr
        self.dacay_step = args.dacay_step
        self.min_learning_rate = args.min_learning_rate
        self.feature_dim = args.feature_dim
        self.feature_max_length = args.feature_max_length
        self.is_training = args.is_training
        self.variant = variant or type(self).variant
        self.engine = DFCNNEngine(model=self.variant, vocab=acoustic_vocab_size,
                                  B=batch_size or args.am_batch_size, T=self.feature_max_length, F=self.feature_dim,
                                  widths=widths, seed=seed, device=device, lr=self.lm_lr,
                                  decay_steps=self.dacay_step, min_lr=self.min_learning_rate)
        e = self.engine
        # bucket 0: the dense head (final before the conv-stack backward starts); buckets 1, 2:
        # BN gammas and the conv/SE tensors (final at the end of backward)
        self.reducer = (BucketedAllReduce(e.grad, [(e.n_gamma, e.dense_end), (0, e.n_gamma),
                                                   (e.dense_end, e.grad.numel())]) if data_parallel else None)

    @property
    def global_step(self):
        return self.engine.global_step

    def run(self, fetches, feed_dict):
        single = not isinstance(fetches, (list, tuple)) or fetches is self.decoded
        flist = [fetches] if single else list(fetches)
        e = self.engine
        x = feed_dict[self.wav_input]
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x, dtype=np.float32))
        x = x.to(e.device, dtype=torch.float32).reshape(x.shape[0], e.T, e.F).contiguous()
        n = x.shape[0]                       # B' <= B rows survived the loader (data_loader.py:149-156); the rest is padding
        if n > e.B:
            raise ValueError('batch %d > engine batch %d (build the model with batch_size=...)' % (n, e.B))
        r = self.reducer
        dp = r is not None and r.world > 1
        need_loss = any(f in (self.loss, self.mean_loss, self.label_err, self.train_op, self.log_prob) or f is self.decoded
                        or f == self.decoded[0] for f in flist)
        e.forward(x)
        lr = None
        denom = n
        if need_loss:
            tp = feed_dict.get(self.target_py)
            if tp is None:      # decode only (lm_and_am/test.py:48-50)
                self._decode_only(feed_dict)
            else:
                # reduce_mean over the rows of the (global) batch, acoustic_model2.py:83: under data parallelism every rank
                # divides by the GLOBAL number of surviving rows, so the summed all-reduce is that mean whatever each
                # rank's B' is (a rank may even hold 0 rows: it still enters every collective)
                # The host-side target checks (label length, "not enough time for target transition sequence": where TF raises
                # InvalidArgumentError) run BEFORE the first collective of the step and their verdict rides on it: a bad batch
                # on one rank stops every rank here, none is left waiting inside a gradient all-reduce.
                prepared, err = None, None
                try:
                    prepared = e.prepare_targets(np.asarray(feed_dict[self.logits_length]), np.asarray(tp), n_valid=n)
                except ValueError as ex:
                    err = ex
                if dp and self.train_op in flist:
                    try:
                        denom = r.sum_count(n, ok=err is None)
                    except RuntimeError:
                        if err is not None:
                            raise err
                        raise
                elif err is not None:
                    raise err
                e.commit_targets(prepared, loss_denom=max(denom, 1))
                e.loss_and_decode(defer_decode_join=self.train_op in flist)
        if self.train_op in flist:
            if not self.is_training:
                raise RuntimeError('train_op needs is_training=True')
            if dp:
                e.backward(on_dense_grads_ready=lambda: r.launch(0))
                r.launch(1)
                r.launch(2)
                r.wait()
            else:
                e.backward()
            if denom > 0:                   # a step whose every row was dropped (on every rank) updates nothing
                lr = e.apply_adam(1.0)      # the 1/rows factor is already in the gradient (loss_denom)
        out = []
        scal = None
        for f in flist:
            if f == self.logits:
                out.append(e.logits)
            elif f == self.loss:
                out.append(e.loss.cpu().numpy()[:n].reshape(-1, 1))
            elif f in (self.mean_loss, self.label_err):
                scal = scal or e.fetch_scalars()
                out.append(scal[0] if f == self.mean_loss else scal[1])
            elif f == self.current_learning:
                out.append(lr if lr is not None else e.current_learning_rate())
            elif f == self.summary:
                scal = scal or e.fetch_scalars()
                out.append({'mean_loss': scal[0], 'accuracy': scal[1]})     # the two tf.summary.scalar tags
            elif f == self.train_op:
                out.append(None)
            elif f is self.decoded or f == self.decoded[0]:
                out.append(SparseTensorValue(e.decoded_lists()))
            elif f == self.log_prob:
                out.append(e.neg_sum.cpu().numpy()[:n].reshape(-1, 1))
            else:
                raise KeyError(f)
        return out[0] if single else out

    def _decode_only(self, feed_dict):
        from . import ops
        e = self.engine
        sl = np.zeros(e.B, dtype=np.int32)
        ln = np.asarray(feed_dict[self.logits_length], dtype=np.int32).reshape(-1)
        sl[:len(ln)] = ln
        e.n_valid = len(ln)
        e.seq_len.copy_(torch.from_numpy(sl).to(e.device))
        ops.ctc_greedy(e.logits, e.T8, e.B, e.V, e.seq_len, e.V - 1, e.dec_ids, e.dec_len, e.neg_sum, e.dec_ws)


class CNNCTCModel1(CNNCTCModel):
    """lm_and_am/model/acoustic_model.py (plain DFCNN, max-pool, NiN cell, 128-unit hidden dense)."""
    variant = 'm1'


class CNNCTCModel3(CNNCTCModel):
    """lm_and_am/model/acoustic_model3.py."""
    variant = 'm3'
