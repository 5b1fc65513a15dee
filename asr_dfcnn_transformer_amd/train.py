"""Training driver of the acoustic model: the counterpart of ``train_acoustic_model``
(lm_and_am/train.py:21-97) with the same feed / fetch contract per step (train.py:59-69),
one process per GPU, gradients averaged with RCCL (parallel.py).  Checkpoints are plain
``torch.save`` dictionaries of the flat parameter and Adam buffers plus ``global_step``
(the reference's tf.train.Saver format is out of scope: SURVEY.md section 5).

    python -m asr_dfcnn_transformer_amd.train --synthetic 64 --epochs 1
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m asr_dfcnn_transformer_amd.train --synthetic 512
"""
import argparse
import os

import torch

from .acoustic_model import CNNCTCModel
from .data_loader import DataLoader, SyntheticSource
from .hparams import AmLmHparams, AmDataHparams
from .parallel import init_from_env


def _engine_of(obj):
    return getattr(obj, 'engine', obj)


def save_checkpoint(model, path):
    """Everything tf.train.Saver keeps for the reference (train.py:38,91-96; model.py:81-88): the variables, the Adam
    slots and global_step -- of any engine here (DFCNN variants, Keras DFCNN, LM, encoder-decoder, pre-net) or of a
    shim that owns one.  Own format (torch.save); TF checkpoint compatibility is out of scope."""
    e = _engine_of(model)
    torch.save({'theta': e.theta.cpu(), 'adam_m': e.adam_m.cpu(), 'adam_v': e.adam_v.cpu(),
                'global_step': e.global_step, 'variant': getattr(e, 'model', type(e).__name__),
                'entries': {str(k): v for k, v in e.entries.items()}}, path)


def load_checkpoint(model, path):
    e = _engine_of(model)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    if ck['variant'] != getattr(e, 'model', type(e).__name__) or ck['entries'] != {str(k): v for k, v in e.entries.items()}:
        raise ValueError('checkpoint %s was written by a different model (%s)' % (path, ck['variant']))
    e.theta.copy_(ck['theta']); e.adam_m.copy_(ck['adam_m']); e.adam_v.copy_(ck['adam_v'])
    e.global_step = int(ck['global_step'])


def train_acoustic_model(data_args, am_hp, train_source, dev_source=None, ckpt_dir=None, log_every=2, model_cls=CNNCTCModel):
    rank, world, local = init_from_env()
    torch.cuda.set_device(local)
    train_loader = DataLoader(train_source, data_args, am_hp)
    model = model_cls(am_hp, train_loader.acoustic_vocab_size, train_loader.language_vocab_size)
    if ckpt_dir and os.path.exists(os.path.join(ckpt_dir, 'final_model.pt')):
        load_checkpoint(model, os.path.join(ckpt_dir, 'final_model.pt'))
    batch_nums = len(train_loader)
    history = []
    for epoch in range(am_hp.epochs):
        total_loss = 0.0
        steps = 0
        for train_step in range(rank, batch_nums, world):           # utterance shards: independent batches per rank
            item = train_loader[train_step]
            if item is None or item[0].shape[0] != model.engine.B:
                continue                                           # rows were dropped (data_loader.py:149-156)
            x, in_len, y, y_len, _, _ = item
            feed = {model.wav_input: x, model.logits_length: in_len, model.target_py: y,
                    model.target_length: y_len, model.drop_rate: am_hp.dropout_rate}
            loss, mean_loss, lr, summary, label_err, _ = model.run(
                [model.loss, model.mean_loss, model.current_learning, model.summary, model.label_err, model.train_op],
                feed_dict=feed)
            total_loss += mean_loss
            steps += 1
            if rank == 0 and steps % log_every == 0:
                print('epoch: %d    step: %d/%d    mean_loss: %.4f    total_loss: %.4f  lr: %.6f   label_err: %.4f'
                      % (epoch + 1, train_step + 1, batch_nums, mean_loss, total_loss / steps, lr, label_err), flush=True)
            history.append((mean_loss, lr, label_err))
        if dev_source is not None:
            dev_loader = DataLoader(dev_source, data_args, am_hp)
            tot_l, tot_e, n = 0.0, 0.0, 0
            for item in dev_loader.am_generator():
                if item[0].shape[0] != model.engine.B:
                    continue
                x, in_len, y, y_len, _, _ = item
                ml, le = model.run([model.mean_loss, model.label_err],
                                   {model.wav_input: x, model.logits_length: in_len, model.target_py: y,
                                    model.target_length: y_len, model.drop_rate: 0})
                tot_l += ml; tot_e += le; n += 1
            if rank == 0 and n:
                print('epoch: ', epoch + 1, ': average loss = ', tot_l / n, ' wer = ', tot_e / n, flush=True)
        if ckpt_dir and rank == 0:
            os.makedirs(ckpt_dir, exist_ok=True)
            save_checkpoint(model, os.path.join(ckpt_dir, 'final_model.pt'))
    return model, history


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--synthetic', type=int, default=64, help='number of synthetic 10 s utterances')
    ap.add_argument('--epochs', type=int, default=1)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--ckpt', default=None)
    a = ap.parse_args()
    am_hp = AmLmHparams().args
    am_hp.epochs, am_hp.am_batch_size = a.epochs, a.batch
    data_args = AmDataHparams().args
    train_acoustic_model(data_args, am_hp, SyntheticSource(a.synthetic), ckpt_dir=a.ckpt)


if __name__ == '__main__':
    main()
