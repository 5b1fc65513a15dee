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


def _parts(model):
    """name -> sub-engine that owns flat theta / Adam buffers.  A plain engine (or a shim around one) is one part '';
    the joint model is {'am', 'lm'} (joint_engine.AMLMEngine), the speech Transformer is {'engine', 'prenet'}
    (e2e_model.Transformer_Model: pre_net + encoder-decoder under ONE AdamOptimizer, end2end/model.py:366-370)."""
    parts = {}
    e = _engine_of(model)
    if hasattr(e, 'theta'):
        parts[''] = e
    for holder in (model, e):
        for name in ('am', 'lm', 'prenet'):
            sub = getattr(holder, name, None)
            if sub is not None and hasattr(sub, 'theta') and sub not in parts.values():
                parts[name] = sub
    if not parts:
        raise ValueError('%s owns no engine with parameters (build it / run one step first)' % type(model).__name__)
    return parts


def _state(e):
    return {'theta': e.theta.cpu(), 'adam_m': e.adam_m.cpu(), 'adam_v': e.adam_v.cpu(), 'global_step': int(e.global_step),
            'variant': str(getattr(e, 'model', type(e).__name__)),
            'entries': {str(k): (int(v[0]), tuple(int(d) for d in v[1])) for k, v in e.entries.items()}}


def save_checkpoint(model, path):
    """Everything tf.train.Saver keeps for the reference (train.py:38,91-96; model.py:81-88): the variables, the Adam
    slots and global_step -- of any engine here (DFCNN variants, Keras DFCNN, LM, encoder-decoder, pre-net), of a shim
    that owns one, or of a composite (joint AM+LM, pre-net + encoder-decoder): one state per part.  Own format
    (torch.save of tensors / ints / strings / tuples only); TF checkpoint compatibility is out of scope."""
    torch.save({'format': 2, 'parts': {name: _state(e) for name, e in _parts(model).items()}}, path)


def load_checkpoint(model, path):
    ck = torch.load(path, map_location='cpu', weights_only=True)      # plain data only: nothing is unpickled into code
    parts = _parts(model)
    if ck.get('format') != 2 or set(ck['parts']) != set(parts):
        raise ValueError('checkpoint %s holds parts %s, the model has %s' % (path, sorted(ck.get('parts', {})), sorted(parts)))
    for name, e in parts.items():
        st, mine = ck['parts'][name], _state(e)
        if st['variant'] != mine['variant'] or st['entries'] != mine['entries']:
            raise ValueError('checkpoint %s was written by a different model (%s)' % (path, st['variant']))
    for name, e in parts.items():
        st = ck['parts'][name]
        e.theta.copy_(st['theta']); e.adam_m.copy_(st['adam_m']); e.adam_v.copy_(st['adam_v'])
        e.global_step = int(st['global_step'])


def rank_batches(batch_nums, world, rank):
    """Batch indices rank ``rank`` of ``world`` trains on in one epoch: r, r + world, ... -- the SAME count
    (batch_nums // world) on every rank, so that all ranks issue the same collectives."""
    return [it * world + rank for it in range(batch_nums // world)]


def train_acoustic_model(data_args, am_hp, train_source, dev_source=None, ckpt_dir=None, log_every=2, model_cls=CNNCTCModel,
                         loader_cls=DataLoader):
    rank, world, local = init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    train_loader = loader_cls(train_source, data_args, am_hp)
    model = model_cls(am_hp, train_loader.acoustic_vocab_size, train_loader.language_vocab_size)
    if ckpt_dir and os.path.exists(os.path.join(ckpt_dir, 'final_model.pt')):
        load_checkpoint(model, os.path.join(ckpt_dir, 'final_model.pt'))
    batch_nums = len(train_loader)
    history = []
    for epoch in range(am_hp.epochs):
        total_loss = 0.0
        steps = 0
        # utterance shards: rank r takes batches r, r + world, ...  EVERY rank takes batch_nums // world steps (the
        # batch_nums % world left-over batches are dropped, as the loader itself drops the utterances beyond a whole
        # batch, data_loader.py:279-280), and a batch that lost rows -- even all of them -- is still fed: the model
        # trains on the B' surviving rows (train.py:59-69 feeds whatever data_generation returned) and every rank
        # enters the same collectives in the same order.
        for train_step in rank_batches(batch_nums, world, rank):
            x, in_len, y, y_len, _, _ = train_loader[train_step]
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
            dev_loader = loader_cls(dev_source, data_args, am_hp)
            tot_l, tot_e, n = 0.0, 0.0, 0
            for item in dev_loader.am_generator():
                x, in_len, y, y_len, _, _ = item
                if x.shape[0] == 0:
                    continue
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
