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
from .parallel import init_from_env, all_agree


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
    plans = {}
    for name, e in parts.items():
        st, mine = ck['parts'][name], _state(e)
        if st['variant'] != mine['variant']:
            raise ValueError('checkpoint %s was written by a different model (%s)' % (path, st['variant']))
        plans[name] = None if st['entries'] == mine['entries'] else _migration(st['entries'], mine['entries'], path)
    for name, e in parts.items():
        st = ck['parts'][name]
        if plans[name] is None:
            e.theta.copy_(st['theta']); e.adam_m.copy_(st['adam_m']); e.adam_v.copy_(st['adam_v'])
        else:
            for buf, key in ((e.theta, 'theta'), (e.adam_m, 'adam_m'), (e.adam_v, 'adam_v')):
                host = torch.zeros(buf.numel(), dtype=torch.float32)
                for dst, src, n, fill in plans[name]:
                    host[dst:dst + n] = st[key][src:src + n] if src >= 0 else (fill if key == 'theta' else 0.0)
                buf.copy_(host)
        e.global_step = int(st['global_step'])


def _migration(old, new, path):
    """Copy plan [(dst offset, src offset or -1, floats, fill)] from a checkpoint whose parameter table lacks entries this build
    added to an existing variant: the Keras DFCNN's BatchNormalization moving statistics ('<layer>/mm', '<layer>/mv', round 3),
    which start at Keras' initial values (moving_mean 0, moving_variance 1; zero Adam slots -- the optimiser never moves them).
    Anything else that differs is a different model."""
    import numpy as np
    missing = [k for k in new if k not in old]
    if any(k not in new for k in old) or not all(k.endswith(('/mm', '/mv')) for k in missing) or \
            any(tuple(old[k][1]) != tuple(new[k][1]) for k in old):
        raise ValueError('checkpoint %s was written by a different model (parameter tables differ)' % path)
    plan = []
    for k, (off, shape) in new.items():
        n = int(np.prod(shape))
        plan.append((int(off), int(old[k][0]) if k in old else -1, n, 1.0 if k.endswith('/mv') else 0.0))
    return plan


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


def train_language_model(data_args, am_hp, train_source, dev_source=None, ckpt_dir=None, log_every=10, model_cls=None,
                         loader_cls=DataLoader, seed=0):
    """Counterpart of ``train_language_model`` (lm_and_am/train.py:100-165): per epoch every lm_batch_size-sized batch of
    ``DataLoader.get_lm_batch()`` is fed as ``{x: input_batch, y: label_batch}`` and ``[mean_loss, current_learning,
    train_op]`` fetched (:138-141); then the dev set's mean accuracy (:152-160) decides whether ``final_model`` is kept.
    One process per GPU: all ranks draw the same (seeded) batch order and rank r trains on batches r, r + world, ...
    (rank_batches), gradients summed over RCCL inside ``Language_Model.run``.  (The reference's dev loop adds the last
    TRAINING cost to its loss total, :157 ``total_loss += cost``; the dev loss is accumulated here.)"""
    import random
    from .language_model import Language_Model
    model_cls = model_cls or Language_Model
    rank, world, local = init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    loader = loader_cls(train_source, data_args, am_hp)
    model = model_cls(am_hp, loader.acoustic_vocab_size, loader.language_vocab_size)
    if ckpt_dir and os.path.exists(os.path.join(ckpt_dir, 'final_model.pt')):
        load_checkpoint(model, os.path.join(ckpt_dir, 'final_model.pt'))
    batch_num = len(loader.pny_lst) // am_hp.lm_batch_size
    mine = set(rank_batches(batch_num, world, rank))
    history, old_acc = [], 0.0
    for epoch in range(am_hp.epochs):
        total_loss, steps = 0.0, 0
        for i, (input_batch, _, label_batch) in zip(sorted(mine), loader.get_lm_batch(rng=random.Random(seed + epoch), select=mine)):
            cost, cur_lr, _ = model.run([model.mean_loss, model.current_learning, model.train_op],
                                        feed_dict={model.x: input_batch, model.y: label_batch})
            total_loss += cost
            steps += 1
            if rank == 0 and i % log_every == 0:
                print("epoch: %d    step: %d/%d lr:%.6f train loss=%.6f" % (epoch + 1, i, batch_num, cur_lr, cost), flush=True)
            history.append((cost, cur_lr))
        if rank == 0 and steps:
            print('epochs', epoch + 1, ': average loss = ', total_loss / steps, flush=True)
        if dev_source is not None:
            dev_loader = loader_cls(dev_source, data_args, am_hp)
            tot_acc, tot_loss, n = 0.0, 0.0, 0
            for input_batch, _, label_batch in dev_loader.get_lm_batch(rng=random.Random(seed)):
                if input_batch.shape[0] == 0:
                    continue
                loss, acc = model.run([model.mean_loss, model.acc], feed_dict={model.x: input_batch, model.y: label_batch})
                tot_loss += loss; tot_acc += acc; n += 1
            if n:
                acc = tot_acc / n
                if rank == 0:
                    print("epoch: %d test acc:%.4f  test loss=%.6f" % (epoch + 1, acc, tot_loss / n), flush=True)
                if acc > old_acc:
                    old_acc = acc
                    if ckpt_dir and rank == 0:
                        os.makedirs(ckpt_dir, exist_ok=True)
                        save_checkpoint(model, os.path.join(ckpt_dir, 'final_model_%d.pt' % (epoch + 1)))
        if ckpt_dir and rank == 0:
            os.makedirs(ckpt_dir, exist_ok=True)
            save_checkpoint(model, os.path.join(ckpt_dir, 'final_model.pt'))
    return model, history


def train_transformer(args, loader, model=None, ckpt_dir=None, seed=0):
    """Counterpart of ``transformerTrain.train`` (end2end/model.py:74-126): every batch of
    ``dataloader.get_transformer_batch()`` is fed as ``{x_input, y_input, y_target, learning_rate}`` and ``[mean_loss, merged,
    current_learning, train_op]`` fetched (:104-109); a checkpoint every ``save_every_n`` steps and as ``final_model``
    (:114-120), a log line every ``log_every_n`` (:122-126).  ``args``: the module-level argparse namespace of
    end2end/model.py:15-55 (e2e_model.E2EHparams).  One process per GPU as in train_language_model."""
    import random
    from datetime import datetime
    from .e2e_model import Transformer_Model
    rank, world, local = init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    model = model or Transformer_Model(args, label_vocab_size=loader.language_vocab_size)
    model.build_transformer()
    if ckpt_dir and os.path.exists(os.path.join(ckpt_dir, 'final_model.pt')):
        # model.py:81-88: restore the latest checkpoint before the loop.  The shim builds its engines on the first batch, so
        # the restore is deferred to that moment (Transformer_Model.restore_checkpoint); train_steps restarts at 0 as in the
        # reference, global_step (the learning-rate schedule) continues
        model.restore_checkpoint(os.path.join(ckpt_dir, 'final_model.pt'))
    batch_nums = len(loader)
    mine = set(rank_batches(batch_nums, world, rank))
    train_steps, history = 0, []
    save_every_n, log_every_n = getattr(args, 'save_every_n', 1000), getattr(args, 'log_every_n', 10)
    for epoch in range(args.epochs):
        total_loss, done = 0.0, 0
        for the_inputs, the_labels, ground_truth in loader.get_transformer_batch(rng=random.Random(seed + epoch), select=mine):
            # x_input has a STATIC batch dimension (model.py:195: [batch_size, None, 320]): a batch that lost rows cannot
            # be fed.  The decision to skip a step is taken by ALL ranks together (one MIN all-reduce), so that no rank
            # enters a gradient all-reduce the others skip.
            if not all_agree(the_inputs.shape[0] == model.batch_size):
                continue
            train_steps += 1
            feed = {model.x_input: the_inputs, model.y_input: the_labels, model.y_target: ground_truth,
                    model.learning_rate: args.learning_rate}
            train_loss, summary, lr, _ = model.run([model.mean_loss, model.merged, model.current_learning, model.train_op],
                                                   feed_dict=feed)
            total_loss += train_loss
            done += 1
            history.append((train_loss, lr))
            if ckpt_dir and rank == 0 and train_steps % save_every_n == 0:
                os.makedirs(ckpt_dir, exist_ok=True)
                save_checkpoint(model, os.path.join(ckpt_dir, 'model_%d.pt' % train_steps))
                save_checkpoint(model, os.path.join(ckpt_dir, 'final_model.pt'))
            if rank == 0 and train_steps % log_every_n == 0:
                print('Epoch: {0:>3}, Iter: {1:>6}, LR:{2:>10.6f} Average Loss: {3:>6.6f}, Time: {4}'.format(
                    epoch + 1, train_steps, lr, total_loss / done, datetime.now()), flush=True)
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
