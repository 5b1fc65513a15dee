"""``Language_Model``: the pinyin->hanzi Transformer of lm_and_am/model/language_model.py:5-78
with the reference's constructor and attribute names, driven session-style like
``CNNCTCModel``:

    cost, cur_lr, _ = lm_model.run([lm_model.mean_loss, lm_model.current_learning, lm_model.train_op],
                                   feed_dict={lm_model.x: input_batch, lm_model.y: label_batch})   # train.py:138-141

The reference calls ``feedforward(self.enc, num_units=...)`` without its two required
arguments (language_model.py:52), so the class cannot be constructed at HEAD;
this implements the evidently intended call form of end2end/model.py:307-310.  Batches are
zero-padded to ``position_max_length`` internally: with the causal mask, trailing pad
positions cannot influence real ones, and the loss masks y == 0."""
import numpy as np

from .transformer_engine import LMEngine


class Language_Model:
    x, y, preds, acc, mean_loss, current_learning, train_op, merged, logits = (
        'x', 'y', 'preds', 'acc', 'mean_loss', 'current_learning', 'train_op', 'merged', 'logits')

    def __init__(self, arg, acoustic_vocab_size, language_vocab_size, batch_size=None, seed=0, device='cuda'):
        self.is_training = arg.is_training
        self.hidden_units = arg.hidden_units
        self.input_vocab_size = acoustic_vocab_size
        self.label_vocab_size = language_vocab_size
        self.num_heads, self.num_blocks = arg.num_heads, arg.num_blocks
        self.position_max_length = arg.position_max_length
        self.lm_lr, self.dacay_step, self.min_learning_rate = arg.lm_lr, arg.dacay_step, arg.min_learning_rate
        self.dropout_rate = arg.dropout_rate            # language_model.py:19,34,46: active while is_training
        self.engine = LMEngine(vin=acoustic_vocab_size, vout=language_vocab_size, N=batch_size or arg.lm_batch_size,
                               T=self.position_max_length, C=self.hidden_units, heads=self.num_heads,
                               blocks=self.num_blocks, pos_max=self.position_max_length, lr=self.lm_lr,
                               decay_steps=self.dacay_step, min_lr=self.min_learning_rate, seed=seed, device=device,
                               dropout_rate=self.dropout_rate if self.is_training else 0.0, drop_seed=seed)

    @property
    def global_step(self):
        return self.engine.global_step

    def _reducer(self):
        from .parallel import BucketedAllReduce
        if getattr(self, '_red', None) is None or self._red.flat is not self.engine.grad:
            self._red = BucketedAllReduce(self.engine.grad, [(0, self.engine.grad.numel())])
        return self._red

    def _pad(self, a):
        e = self.engine
        a = np.asarray(a)
        if a.shape[1] > e.T:
            raise IndexError('sequence longer than position_max_length (embedding_lookup would fail in the reference)')
        out = np.zeros((e.N, e.T), dtype=np.int32)
        out[:a.shape[0], :a.shape[1]] = a
        return out

    def run(self, fetches, feed_dict):
        single = not isinstance(fetches, (list, tuple))
        flist = [fetches] if single else list(fetches)
        e = self.engine
        x = np.asarray(feed_dict[self.x])
        n, t = x.shape
        y = feed_dict.get(self.y)
        train = self.train_op in flist
        # inference (lm_and_am/test.py:60) feeds no labels: an all-PAD target still yields preds from the CE kernel
        e.forward(self._pad(x), np.zeros((e.N, e.T), dtype=np.int32) if y is None else self._pad(y), train=train)
        lr = None
        if train:
            e.backward()
            # one process per GPU: sum the flat gradient over the ranks (RCCL), Adam averages (parallel.py); every rank
            # trains on its own batch of the same step, so the update is the mean of the per-rank mean-token losses
            red = self._reducer()
            red.launch(0); red.wait()
            lr = e.apply_adam(red.grad_scale)
        out, sc = [], None
        for f in flist:
            if f in (self.mean_loss, self.acc):
                sc = sc or e.fetch()
                out.append(sc[0] if f == self.mean_loss else sc[1])
            elif f == self.merged:
                sc = sc or e.fetch()
                out.append({'mean_loss': sc[0], 'acc': sc[1]})
            elif f == self.current_learning:
                out.append(lr if lr is not None else e.current_learning_rate())
            elif f == self.preds:
                out.append(e.preds.view(e.N, e.T)[:n, :t].cpu().numpy())
            elif f == self.logits:
                out.append(e.logits.view(e.N, e.T, -1)[:n, :t, :e.V])
            elif f == self.train_op:
                out.append(None)
            else:
                raise KeyError(f)
        return out[0] if single else out
