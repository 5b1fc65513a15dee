"""``CNNCTCModel`` of lm_and_am/model/am_lm_model.py:17-155 -- the joint acoustic + language model graph that
lm_and_am/am_lm_train.py:44-77 trains -- with the reference's constructor, placeholder and fetch names, driven
session-style like acoustic_model.CNNCTCModel:

    model = CNNCTCModel(am_hp, acoustic_vocab_size, language_vocab_size, batch_size=16)
    mean_loss, label_err, han_wer, summary, _ = model.run(
        [model.lm_mean_loss, model.label_err, model.han_wer, model.summary, model.train_op],
        feed_dict={model.wav_input: x, model.wav_length: n, model.target_py: py, model.target_py_length: pl,
                   model.target_hanzi: hz, model.target_hanzi_length: hl})            # am_lm_train.py:64-75

The reference file does not run as written: `self.am_out` is undefined, so the language half reads
h7 and hidden_units / num_heads become 128 / 2 whatever ``args`` says; position_max_length is raised to the 200 time steps
it is indexed with.  ``target_hanzi_length`` is accepted and unused (the reference only uses dense_to_sparse(target_hanzi)).
"""
import numpy as np
import torch

from .acoustic_model import SparseTensorValue
from .joint_engine import AMLMEngine


class CNNCTCModel:
    wav_input, wav_length, target_py, target_py_length, target_hanzi, target_hanzi_length = (
        'wav_input', 'logits_length', 'target_py', 'target_py_length', 'target_hanzi', 'target_hanzi_length')
    (am_logits, lm_logits, am_loss, lm_loss, am_mean_loss, lm_mean_loss, mean_loss, label_err, han_wer, summary, train_op,
     lm_out) = ('am_logits', 'lm_logits', 'am_loss', 'lm_loss', 'am_mean_loss', 'lm_mean_loss', 'mean_loss',
                'label_error_rate', 'han_wer', 'summary', 'train_op', 'lm_out')
    decoded = ('decoded',)

    def __init__(self, args, acoustic_vocab_size, language_vocab_size, batch_size=None, widths=None, seed=0, device='cuda'):
        self.acoustic_vocab_size = acoustic_vocab_size
        self.language_vocab_size = language_vocab_size
        self.gpu_nums = args.gpu_nums
        self.lr = args.am_lr
        self.feature_dim = args.feature_dim
        self.feature_max_length = args.feature_max_length
        self.is_training = args.is_training
        self.num_blocks = args.num_blocks
        self.dropout_rate = args.dropout_rate
        t8 = self.feature_max_length // 8
        self.position_max_length = max(args.position_max_length, t8)       # indexed with range(200) (:85-87)
        self.engine = AMLMEngine(v_pinyin=acoustic_vocab_size, v_hanzi=language_vocab_size,
                                 B=batch_size or args.am_batch_size, T=self.feature_max_length, F=self.feature_dim,
                                 widths=widths, heads=None, blocks=self.num_blocks, pos_max=self.position_max_length,
                                 lr=self.lr, seed=seed, device=device,
                                 dropout_rate=self.dropout_rate if self.is_training else 0.0)
        self.hidden_units = self.engine.lm.C            # = width of h7 (D1), not args.hidden_units
        self.num_heads = self.engine.lm.H
        import torch.distributed as dist
        # one process per GPU (am_lm_train.train_model): gradients summed over ranks, Adam averages
        self.reducers = self.engine.make_reducers() if (dist.is_initialized() and dist.get_world_size() > 1) else None

    @property
    def global_step(self):
        return self.engine.am.global_step

    def run(self, fetches, feed_dict):
        single = not isinstance(fetches, (list, tuple)) or fetches is self.decoded
        flist = [fetches] if single else list(fetches)
        e = self.engine
        x = feed_dict[self.wav_input]
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x, dtype=np.float32))
        x = x.to(e.am.device, dtype=torch.float32).reshape(x.shape[0], e.am.T, e.am.F).contiguous()
        if x.shape[0] != e.B:
            raise ValueError('batch %d != engine batch %d (build the model with batch_size=...)' % (x.shape[0], e.B))
        train = self.train_op in flist
        if train and not self.is_training:
            raise RuntimeError('train_op needs is_training=True')
        e.forward(x, train=self.is_training)
        only_logits = all(f in (self.am_logits, self.lm_logits) for f in flist)
        if not only_logits:
            e.set_targets(np.asarray(feed_dict[self.wav_length]), np.asarray(feed_dict[self.target_py]),
                          np.asarray(feed_dict[self.target_py_length]), feed_dict.get(self.target_hanzi))
            e.loss_and_decode()
        if train:
            e.backward(self.reducers)
            e.apply_adam(self.reducers[0].grad_scale if self.reducers else 1.0)
        out, scal = [], None
        for f in flist:
            if f == self.am_logits:
                out.append(e.am.logits)
            elif f == self.lm_logits:
                out.append(e.lm.logits[:, :, :self.language_vocab_size])
            elif f == self.am_loss:
                out.append(e.am.loss.cpu().numpy().reshape(-1, 1))
            elif f == self.lm_loss:
                out.append(e.lm.loss.cpu().numpy().reshape(-1, 1))
            elif f in (self.am_mean_loss, self.lm_mean_loss, self.mean_loss, self.label_err):
                scal = scal or e.fetch()
                out.append(scal[{self.am_mean_loss: 0, self.lm_mean_loss: 1, self.mean_loss: 2, self.label_err: 3}[f]])
            elif f == self.han_wer:
                out.append(e.han_wer())
            elif f == self.summary:
                scal = scal or e.fetch()
                out.append({'acc': e.han_wer(), 'mean_loss': scal[2]})      # the two tf.summary.scalar tags (:123,138)
            elif f == self.train_op:
                out.append(None)
            elif f is self.decoded or f == self.decoded[0]:
                out.append(SparseTensorValue(e.decoded_lists()[1]))          # self.decoded ends up as the lm decode (:120)
            elif f == self.lm_out:
                ids = e.am.decoded_lists()                                   # sparse_to_dense(am decode, 0) (:76)
                w = max([len(d) for d in ids] + [0])
                out.append(np.asarray([d + [0] * (w - len(d)) for d in ids], dtype=np.int64).reshape(len(ids), w))
            else:
                raise KeyError(f)
        return out[0] if single else out
