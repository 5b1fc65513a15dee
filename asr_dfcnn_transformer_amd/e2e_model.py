"""``Transformer_Model``: the encoder-decoder of end2end/model.py:176-370 with the reference's
placeholder / fetch names (x_input, y_input, y_target, learning_rate; mean_loss, merged,
current_learning, train_op), driven session-style (end2end/model.py:104-109).

``x_input`` here is the flattened pre-net output fed to ``embedding_input`` (model.py:267-279);
the stride-2 conv + 2-D attention pre-net (model.py:214-264) is the next widening step
(SURVEY 8f.1).  The encoder is not causal, so padding would change results: an engine is
(re)built per (T, L) shape and the parameter / Adam state carried over."""
import numpy as np
import torch

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
        self.tie, self.seed, self.device = tie, seed, device
        self.engine = None

    def build_transformer(self):
        return self

    def _engine_for(self, T, L, lr):
        e = self.engine
        if e is not None and (e.T, e.L) == (T, L):
            e.lr0 = lr
            return e
        new = E2EEngine(din=self.input_dim, vout=self.label_vocab_size, N=self.batch_size, T=T, L=L, C=self.hidden_units,
                        heads=self.num_heads, blocks=self.num_blocks, pos_max=self.position_max_length, tie=self.tie,
                        lr=lr, decay_steps=self.dacay_step, min_lr=self.min_learning_rate, seed=self.seed, device=self.device)
        if e is not None:
            new.theta.copy_(e.theta); new.adam_m.copy_(e.adam_m); new.adam_v.copy_(e.adam_v)
            new.global_step = e.global_step
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
        e = self._engine_for(x.shape[1], y_in.shape[1], float(feed_dict.get(self.learning_rate, self.arg.learning_rate)))
        train = self.train_op in flist
        e.forward(x, y_in, np.zeros_like(y_in) if y_tgt is None else np.asarray(y_tgt), train=train)
        lr = None
        if train:
            e.backward()
            lr = e.apply_adam()
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
    """The module-level argparse defaults of end2end/model.py:15-55."""
    batch_size, num_blocks, hidden_units, num_heads = 8, 6, 512, 8
    position_max_length, dropout_rate, feature_dim = 600, 0.2, 80
    learning_rate, dacay_step, min_learning_rate = 5e-4, 5000, 1e-6
    is_training = True
