"""``Transformer_Model``: the encoder-decoder of end2end/model.py:176-370 with the reference's
placeholder / fetch names (x_input, y_input, y_target, learning_rate; mean_loss, merged,
current_learning, train_op), driven session-style (end2end/model.py:104-109).

``x_input`` is the reference's placeholder ``[batch, T, 4 * dimension]`` (model.py:203): stacked frames that go
through ``pre_net`` (model.py:214-264; prenet_engine.PreNetEngine) and ``embedding_input``.  A feed whose last
dimension equals ``input_dim`` instead is taken as an already-computed, flattened pre-net output (the entry point
of ``embedding_input``, model.py:267-279).  The encoder is not causal, so padding would change results: engines
are (re)built per (T, L) shape and the parameter / Adam state carried over."""
import numpy as np
import torch

from .prenet_engine import PreNetEngine
from .transformer_engine import E2EEngine


class Transformer_Model:
    x_input, y_input, y_target, learning_rate = 'x_input', 'y_input', 'y_target', 'learning_rate'
    mean_loss, merged, current_learning, train_op, preds, acc, logits = (
        'mean_loss', 'merged', 'current_learning', 'train_op', 'preds', 'acc', 'logits')

    def __init__(self, arg, label_vocab_size=6347, input_dim=None, tie=True, seed=0, device='cuda'):
        self.arg = arg
        self.batch_size, self.num_blocks, self.num_heads = arg.batch_size, arg.num_blocks, arg.num_heads
        self.hidden_units, self.position_max_length = arg.hidden_units, arg.position_max_length
        self.dacay_step, self.min_learning_rate = arg.dacay_step, arg.min_learning_rate
        self.label_vocab_size = label_vocab_size
        self.input_dim = input_dim or (arg.feature_dim * 4 // 4) * 64          # [T/4, F/4, 64] flattened (model.py:270)
        self.raw_dim = 4 * getattr(arg, 'dimension', arg.feature_dim)           # x_input's last dimension (model.py:203)
        self.tie, self.seed, self.device = tie, seed, device
        self.engine = None
        self.prenet = None
        self._restore = None          # parts of a checkpoint waiting for their engine (restore_checkpoint)

    def build_transformer(self):
        return self

    def restore_checkpoint(self, path):
        """saver.restore(sess, tf.train.latest_checkpoint(model_path)) of transformerTrain.train (end2end/model.py:81-88).  The
        engines of this shim are built on the first run() (their shapes come from the first batch), so the checkpoint's parts
        are kept here and copied into each engine the moment it is created: variables, Adam slots and global_step."""
        ck = torch.load(path, map_location='cpu', weights_only=True)
        if ck.get('format') != 2 or '' not in ck['parts']:
            raise ValueError('%s is not a checkpoint of an encoder-decoder model' % path)
        self._restore = dict(ck['parts'])
        for name, e in (('', self.engine), ('prenet', self.prenet)):
            if e is not None:
                self._restore_into(name, e)

    def _restore_into(self, name, e):
        st = (self._restore or {}).pop(name, None)
        if st is None:
            return
        ent = {str(k): (int(v[0]), tuple(int(d) for d in v[1])) for k, v in e.entries.items()}
        if st['variant'] != str(getattr(e, 'model', type(e).__name__)) or st['entries'] != ent:
            raise ValueError('checkpoint part %r was written by a different model (%s)' % (name, st['variant']))
        e.theta.copy_(st['theta']); e.adam_m.copy_(st['adam_m']); e.adam_v.copy_(st['adam_v'])
        e.global_step = int(st['global_step'])

    def _prenet_for(self, T):
        p = self.prenet
        if p is not None and p.T == T:
            return p
        new = PreNetEngine(self.batch_size, T, self.raw_dim, seed=self.seed + 1, device=self.device)
        if p is not None:
            new.theta.copy_(p.theta); new.adam_m.copy_(p.adam_m); new.adam_v.copy_(p.adam_v)
            new.global_step = p.global_step
        else:
            self._restore_into('prenet', new)
        self.prenet = new
        return new

    def _engine_for(self, T, L, lr, need_dx=False):
        e = self.engine
        if e is not None and (e.T, e.L) == (T, L) and (e.need_dx or not need_dx):
            e.lr0 = lr
            return e
        new = E2EEngine(din=self.input_dim, vout=self.label_vocab_size, N=self.batch_size, T=T, L=L, C=self.hidden_units,
                        heads=self.num_heads, blocks=self.num_blocks, pos_max=self.position_max_length, tie=self.tie,
                        lr=lr, decay_steps=self.dacay_step, min_lr=self.min_learning_rate, seed=self.seed, device=self.device,
                        need_dx=need_dx,
                        dropout_rate=self.arg.dropout_rate if getattr(self.arg, 'is_training', True) else 0.0, drop_seed=self.seed)
        if e is not None:
            new.theta.copy_(e.theta); new.adam_m.copy_(e.adam_m); new.adam_v.copy_(e.adam_v)
            new.global_step = e.global_step
        else:
            self._restore_into('', new)
        self.engine = new
        return new

    def run(self, fetches, feed_dict):
        single = not isinstance(fetches, (list, tuple))
        flist = [fetches] if single else list(fetches)
        x = feed_dict[self.x_input]
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x, dtype=np.float32))
        x = x.to(self.device, dtype=torch.float32).contiguous()
        y_in, y_tgt = np.asarray(feed_dict[self.y_input]), feed_dict.get(self.y_target)
        raw = x.shape[2] == self.raw_dim and x.shape[2] != self.input_dim
        pre = None
        if raw:
            if x.shape[1] % 4:
                # the pre-net kernels take whole 4-frame groups (two stride-2 convs): zero frames are appended, as
                # wav_padding (end2end/data_loader.py:84-99) appends them behind every shorter utterance of a batch.
                # (For a T that is not a multiple of 4 TensorFlow's 'same' padding would put one of its pad rows in
                # FRONT of an odd-length axis: results on such batches differ from the reference by that shift.)
                pad = 4 - x.shape[1] % 4
                x = torch.cat([x, torch.zeros(x.shape[0], pad, x.shape[2], dtype=x.dtype, device=x.device)], dim=1).contiguous()
            pre = self._prenet_for(x.shape[1])
            x = pre.forward(x)                          # [B, T/4, 80*64]
        elif x.shape[2] != self.input_dim:
            raise ValueError('x_input has %d features: expected %d (stacked frames) or %d (flattened pre-net output)' %
                             (x.shape[2], self.raw_dim, self.input_dim))
        e = self._engine_for(x.shape[1], y_in.shape[1], float(feed_dict.get(self.learning_rate, self.arg.learning_rate)),
                             need_dx=raw)
        train = self.train_op in flist
        e.forward(x, y_in, np.zeros_like(y_in) if y_tgt is None else np.asarray(y_tgt), train=train)
        lr = None
        if train:
            from .parallel import BucketedAllReduce
            e.backward()
            red = BucketedAllReduce(e.grad, [(0, e.grad.numel())])          # data parallel: summed over ranks, Adam averages
            red.launch(0)
            if pre is not None:
                pre.backward(e.dx_feat)
                red_pre = BucketedAllReduce(pre.grad, [(0, pre.grad.numel())])
                red_pre.launch(0); red_pre.wait()
            red.wait()
            lr = e.apply_adam(red.grad_scale)
            if pre is not None:
                pre.apply_adam(lr, red.grad_scale)      # one AdamOptimizer over all variables (model.py:366-370)
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
                out.append(e.preds.view(e.N, e.L).cpu().numpy())
            elif f == self.logits:
                out.append(e.logits.view(e.N, e.L, -1)[:, :, :e.V])
            elif f == self.train_op:
                out.append(None)
            else:
                raise KeyError(f)
        return out[0] if single else out


class E2EHparams:
    """The module-level argparse defaults of end2end/model.py:15-55 (the ones the path reads)."""
    gpu_nums, mode, is_training, batch_size, epochs = 1, 'train', True, 8, 100
    feature_max_length, dimension, shuffle, data_length, save_nums = 1600, 80, True, None, 3
    num_blocks, hidden_units, num_heads = 6, 512, 8
    position_max_length, dropout_rate, feature_dim = 600, 0.2, 80
    summary_step, save_every_n, log_every_n = 200, 1000, 2
    learning_rate, dacay_step, min_learning_rate = 5e-4, 5000, 1e-6
