"""``CNNCTCModel`` of lm_and_am/model/cnn_ctc.py:15-93 (the original Keras DFCNN) with the reference's constructor and
methods, on keras_engine.KerasDFCNNEngine:

    model = CNNCTCModel(args, acoustic_vocab_size)                 # args.am_lr, args.is_training      (:16-26)
    loss = model.train_on_batch(inputs, outputs)                   # ctc_model.fit_generator's step    (:51-65)
    ids = model.predict(fbank_features, length)                    # greedy pinyin ids of one utterance (:67-83)
    model.save_model(name) / model.load_model(name)                # weights file                       (:85-89)

``predict`` is the inference path of lm_and_am/read_wav.py:46-50 / test.py (pred_pinyin): the features of ONE utterance
are zero-padded to 1600 frames and replicated ``batch_size`` times, ``self.model.predict`` (Keras learning phase 0:
BatchNormalization on its moving statistics, Dropout off) gives the softmax output [batch, 200, vocab], and
``util.utils.decode_ctc`` -- K.ctc_decode(greedy=True) over ``length`` frames -- returns the ids of the first row.
"""
import os

import numpy as np
import torch

from .keras_engine import KerasDFCNNEngine, CELLS
from .utils import decode_ctc


class CNNCTCModel:
    def __init__(self, args, acoustic_vocab_size, batch_size=None, feature_max_length=1600, cells=CELLS, hidden=128, seed=0,
                 device='cuda', model_dir='.'):
        self.vocab_size = acoustic_vocab_size
        self.lr = args.am_lr
        self.feature_length = getattr(args, 'feature_dim', 200)           # 200 in the reference (:20)
        self.is_training = args.is_training
        self.feature_max_length = feature_max_length
        self.batch_size = batch_size or getattr(args, 'am_batch_size', 1)
        self.cells, self.hidden, self.seed, self.device, self.model_dir = list(cells), hidden, seed, device, model_dir
        self._engines = {}
        self.engine = self._engine_for(self.batch_size)

    def _engine_for(self, B):
        """Engines are built per batch size (the Keras graph has a None batch dimension) and share ONE set of variables:
        the parameter / Adam / moving-statistics buffers of the first engine."""
        if B not in self._engines:
            e = KerasDFCNNEngine(vocab=self.vocab_size, B=B, T=self.feature_max_length, F=self.feature_length, cells=self.cells,
                                 hidden=self.hidden, lr=self.lr, seed=self.seed, device=self.device,
                                 dropout_rate=0.3 if self.is_training else 0.0, drop_seed=self.seed)
            first = next(iter(self._engines.values()), None)
            if first is not None:
                e.theta, e.adam_m, e.adam_v = first.theta, first.adam_m, first.adam_v
            self._engines[B] = e
        return self._engines[B]

    @property
    def global_step(self):
        return self.engine.global_step

    def train_on_batch(self, inputs, outputs=None):
        """One optimiser step of ``ctc_model`` (:51-65): inputs = {'the_inputs' [B, 1600, 200, 1], 'the_labels' [B, L],
        'input_length' [B(,1)], 'label_length' [B(,1)]} as DataLoader's Keras generator yields them; returns the mean CTC loss."""
        if not self.is_training:
            raise RuntimeError('the ctc model only exists with is_training (cnn_ctc.py:23-25)')
        x = inputs['the_inputs']
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x, dtype=np.float32))
        x = x.to(self.device, dtype=torch.float32).reshape(x.shape[0], self.feature_max_length, self.feature_length).contiguous()
        e = self._engine_for(x.shape[0])
        e.global_step = self.engine.global_step
        e.forward(x, train=True)
        e.set_targets(np.asarray(inputs['input_length']).reshape(-1), np.asarray(inputs['the_labels']),
                      np.asarray(inputs['label_length']).reshape(-1))
        e.loss_and_decode()
        e.backward()
        e.apply_adam()
        self.engine.global_step = e.global_step
        return e.fetch_loss()

    def predict(self, data_input, length, batch_size=1):
        """cnn_ctc.py:67-83.  data_input: [T, feature_length(, 1)] features of one utterance; returns the decoded ids."""
        d = np.asarray(data_input, dtype=np.float32).reshape(len(data_input), self.feature_length)
        x_in = np.zeros((batch_size, self.feature_max_length, self.feature_length), dtype=np.float32)
        n = min(len(d), self.feature_max_length)
        x_in[:, :n] = d[:n]
        e = self._engine_for(batch_size)
        e.forward(torch.from_numpy(x_in).to(self.device), train=False)
        # the softmax output of the first row, [1, 200, vocab]: the engine keeps log(softmax + 1e-7) time-major -- exactly what
        # K.ctc_decode would compute from it (utils.decode_ctc takes either form)
        return decode_ctc(e.logits[:, :1, :], length, log_time_major=True)

    def _path(self, model):
        return os.path.join(self.model_dir, model + '.pt')

    def save_model(self, model):
        from .train import save_checkpoint
        save_checkpoint(self.engine, self._path(model))

    def load_model(self, model):
        from .train import load_checkpoint
        load_checkpoint(self.engine, self._path(model))
