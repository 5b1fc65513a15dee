"""Training driver of the joint acoustic + language model: the counterpart of ``train_model``
(lm_and_am/am_lm_train.py:27-116).  Per step the six arrays of ``DataLoader.end2end_generator`` are fed as
``{wav_input, wav_length, target_py, target_py_length, target_hanzi, target_hanzi_length}`` and
``[lm_mean_loss, label_err, han_wer, summary, train_op]`` fetched (:61-75); after every epoch the dev set is evaluated and a
checkpoint ``model_<epoch>-<loss>-<acc>`` written, ``final_model`` when the label error rate improved (:80-112).

Read as evidently intended where the source cannot run (DESIGN.md section 10 does the same for the model file): the dev
loop unpacks the six-tuple of the loader into four names and feeds the LAST TRAINING batch's hanzi targets (:87-93) -- here
every dev batch feeds its own six arrays --, ``am_loss`` is undefined (:102, dropped), and ``latest = None`` right after
``tf.train.latest_checkpoint`` (:54-55) disables resuming: ``resume=False`` keeps that, ``resume=True`` loads
``final_model`` first.  The joint graph has a static batch dimension in this build (engine planes): a batch that lost rows
is skipped -- by all ranks together under data parallelism (parallel.all_agree), as train.train_transformer does.
"""
import os

import torch

from .am_lm_model import CNNCTCModel
from .data_loader import DataLoader
from .parallel import init_from_env, all_agree
from .train import rank_batches, save_checkpoint, load_checkpoint


def train_model(data_args, am_hp, train_source=None, dev_source=None, ckpt_dir=None, resume=False, log_every=2, model_kw=None,
                loader_cls=DataLoader, data_dir='data', audio_root=''):
    """``train_source`` / ``dev_source``: DataUtil-like sources (path_lst / pny_lst / han_lst / read_audio); when omitted they
    are built as the reference builds them (am_lm_train.py:36-37) from the index files under ``data_dir``."""
    rank, world, local = init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if train_source is None:
        from .data_util import DataUtil
        train_source = DataUtil(data_args, batch_size=am_hp.am_batch_size, mode='train', data_length=None, shuffle=True,
                                data_dir=data_dir, audio_root=audio_root)
        dev_source = DataUtil(data_args, batch_size=am_hp.am_batch_size, mode='dev', data_length=None, shuffle=True,
                              data_dir=data_dir, audio_root=audio_root)
    train_loader = loader_cls(train_source, data_args, am_hp)
    dev_loader = loader_cls(dev_source, data_args, am_hp) if dev_source is not None else None
    model = CNNCTCModel(am_hp, train_loader.acoustic_vocab_size, train_loader.language_vocab_size, **(model_kw or {}))
    if resume and ckpt_dir and os.path.exists(os.path.join(ckpt_dir, 'final_model.pt')):
        load_checkpoint(model, os.path.join(ckpt_dir, 'final_model.pt'))
    B = model.engine.B
    batch_nums = len(train_loader)
    mine = rank_batches(batch_nums, world, rank)
    history, old_wer = [], 1.0

    def feed_of(item):
        x, in_len, py, py_len, hz, hz_len = item
        return {model.wav_input: x, model.wav_length: in_len, model.target_py: py, model.target_py_length: py_len,
                model.target_hanzi: hz, model.target_hanzi_length: hz_len}

    for epoch in range(am_hp.epochs):
        total_loss, done = 0.0, 0
        for train_step in mine:
            item = train_loader[train_step]
            if not all_agree(item[0].shape[0] == B):
                continue
            mean_loss, label_err, han_wer, summary, _ = model.run(
                [model.lm_mean_loss, model.label_err, model.han_wer, model.summary, model.train_op], feed_dict=feed_of(item))
            total_loss += mean_loss
            done += 1
            history.append((mean_loss, label_err, han_wer))
            if rank == 0 and done % log_every == 0:
                print('epoch: {0:d}   step:{1:d}/{2:d}   average loss:{3:.4f}   label_err:{4:.4f}   acc:{5:.4f}'.format(
                    epoch + 1, train_step + 1, batch_nums, total_loss / done, label_err, han_wer), flush=True)
        if dev_loader is None:
            continue
        total_wer = total_acc = total_dev = 0.0
        eval_steps = 0
        was_training, model.is_training = model.is_training, False        # dev: no dropout, no update
        for item in dev_loader.end2end_generator():
            if item[0].shape[0] != B:
                continue
            mean_loss, label_err, acc = model.run([model.lm_mean_loss, model.label_err, model.han_wer], feed_dict=feed_of(item))
            total_wer += label_err; total_dev += mean_loss; total_acc += acc
            eval_steps += 1
        model.is_training = was_training
        if not eval_steps:
            continue
        wer, acc, mean_loss = total_wer / eval_steps, total_acc / eval_steps, total_dev / eval_steps
        if rank == 0:
            print('epoch:%d   loss:%.4f   wer:%.4f   acc:%.4f' % (epoch + 1, mean_loss, wer, acc), flush=True)
            if ckpt_dir:
                os.makedirs(ckpt_dir, exist_ok=True)
                save_checkpoint(model, os.path.join(ckpt_dir, 'model_%d-%.2f-%.2f.pt' % (epoch, mean_loss, acc)))
                if wer < old_wer:
                    save_checkpoint(model, os.path.join(ckpt_dir, 'final_model.pt'))
        if wer < old_wer:
            old_wer = wer
    return model, history
