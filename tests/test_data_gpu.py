"""Device side of the data seams and the drivers around them: asr_lfr against the LFR oracle, the end-to-end loader
(fbank 80 -> LFR(4, 3) -> padding, SOS / EOS / IGNORE labels; end2end/data_loader.py:263-302) on synthesised WAV files
named by the hand-written index fixtures, the language-model and end-to-end training loops (lm_and_am/train.py:100-165,
end2end/model.py:74-126) and speech_test with its pred_log (lm_and_am/test.py:25-101)."""
import math
import os
import wave

import numpy as np
import pytest
import torch

from oracle import fbank as ofb

pytestmark = pytest.mark.gpu
INDEX = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'index')


def _write_wavs(root, paths, seed=0, seconds=(0.9, 1.7)):
    """16-bit PCM 16 kHz files of different lengths for every index entry; returns {path: float64 samples as decoded}."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, p in enumerate(paths):
        n = int(16000 * (seconds[0] + (seconds[1] - seconds[0]) * ((k * 37) % 11) / 10.0))
        t = np.arange(n) / 16000.0
        x = 0.2 * np.sin(2 * np.pi * (200 + 90 * k) * t * (1 + 0.5 * t)) + 0.05 * rng.standard_normal(n)
        pcm = np.clip(np.round(x * 32768), -32768, 32767).astype('<i2')
        f = os.path.join(root, p)
        os.makedirs(os.path.dirname(f), exist_ok=True)
        with wave.open(f, 'wb') as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
        out[p] = pcm.astype(np.float64) / 32768.0
    return out


def test_lfr_kernel_matches_oracle_on_a_ragged_batch():
    from asr_dfcnn_transformer_amd import ops
    rng = np.random.default_rng(0)
    B, t_pad, D, m, n = 5, 41, 80, 4, 3
    frames = [41, 40, 39, 10, 1]
    feat = np.zeros((B, t_pad, D), dtype=np.float32)
    for b, f in enumerate(frames):
        feat[b, :f] = rng.standard_normal((f, D)).astype(np.float32)
    t_out = max(math.ceil(f / n) for f in frames)
    got = ops.lfr(torch.tensor(feat, device='cuda'), torch.tensor(frames, dtype=torch.int32, device='cuda'), m, n, t_out).cpu().numpy()
    assert got.shape == (B, t_out, m * D)
    for b, f in enumerate(frames):
        ref = ofb.build_LFR_features(feat[b, :f], m, n)
        assert np.array_equal(got[b, :ref.shape[0]], ref) and not got[b, ref.shape[0]:].any()
    # (m, n) = (1, 1) is the identity, (1, 3) skips
    g1 = ops.lfr(torch.tensor(feat, device='cuda'), torch.tensor(frames, dtype=torch.int32, device='cuda'), 1, 1, 41).cpu().numpy()
    assert np.array_equal(g1, feat)


def test_get_transformer_batch_from_wav_files(tmp_path):
    from asr_dfcnn_transformer_amd.const import Const
    from asr_dfcnn_transformer_amd.data_util import DataUtil
    from asr_dfcnn_transformer_amd.e2e_data_loader import dataloader
    from asr_dfcnn_transformer_amd.e2e_model import E2EHparams
    from asr_dfcnn_transformer_amd.hparams import TransDataHparams
    hp = TransDataHparams().args
    args = E2EHparams(); args.batch_size, args.shuffle = 3, False
    du = DataUtil(hp, 3, 'train', data_dir=INDEX, audio_root=str(tmp_path))
    sigs = _write_wavs(str(tmp_path), du.path_lst)
    dl = dataloader(args, hp, data_util=du)
    batches = list(dl.get_transformer_batch())
    assert len(batches) == len(dl) == 2
    for bi, (wav, lab, tgt) in enumerate(batches):
        rows = range(3 * bi, 3 * bi + 3)
        refs = [ofb.build_LFR_features(ofb.compute_fbank_from_api(sigs[du.path_lst[r]], 16000, nfilt=80), 4, 3) for r in rows]
        tmax = max(r.shape[0] for r in refs)
        assert tuple(wav.shape) == (3, tmax, 320) and wav.dtype == torch.float32 and wav.is_cuda
        got = wav.cpu().numpy()
        for k, ref in enumerate(refs):
            assert np.abs(got[k, :ref.shape[0]] - ref).max() <= 2e-6       # bit-exact after the float32 cast in practice
            assert not got[k, ref.shape[0]:].any()
        ids = [dl.han2id(du.han_lst[r]) for r in rows]
        L = max(len(i) for i in ids) + 1
        assert lab.shape == (3, L) and tgt.shape == (3, L) and lab.dtype == np.int32
        for k, i in enumerate(ids):
            assert lab[k].tolist() == [Const.SOS] + i + [Const.EOS] * (L - 1 - len(i))
            assert tgt[k].tolist() == i + [Const.EOS] + [Const.IGNORE] * (L - 1 - len(i))
    # an utterance with a character outside hanzi.txt is dropped from all three arrays
    du.han_lst[1] = '今天Q好'
    wav, lab, tgt = next(dl.get_transformer_batch())
    assert wav.shape[0] == 2 and lab.shape[0] == 2 and tgt.shape[0] == 2


def _small_am_lm(src):
    from asr_dfcnn_transformer_amd.acoustic_model import CNNCTCModel
    from asr_dfcnn_transformer_amd.data_loader import DataLoader
    from asr_dfcnn_transformer_amd.hparams import AmLmHparams, TransDataHparams
    from asr_dfcnn_transformer_amd.language_model import Language_Model
    hp = AmLmHparams().args
    hp.am_batch_size, hp.lm_batch_size, hp.hidden_units, hp.num_heads, hp.num_blocks, hp.position_max_length = 2, 2, 128, 2, 1, 32
    hp.lm_lr, hp.dropout_rate = 2e-3, 0.0
    dl = DataLoader(src, TransDataHparams().args, hp)
    return hp, dl


def test_language_model_training_loop_on_the_index_fixture(tmp_path):
    from asr_dfcnn_transformer_amd import train as tr
    from asr_dfcnn_transformer_amd.data_util import DataUtil
    from asr_dfcnn_transformer_amd.hparams import TransDataHparams
    dhp = TransDataHparams().args
    src = DataUtil(dhp, 2, 'train', shuffle=True, data_dir=INDEX, seed=2)
    dev = DataUtil(dhp, 2, 'dev', data_dir=INDEX)
    hp, _ = _small_am_lm(src)
    hp.epochs = 12
    model, hist = tr.train_language_model(dhp, hp, src, dev_source=dev, ckpt_dir=str(tmp_path), log_every=1000)
    assert len(hist) == 12 * 4 and model.global_step == 48
    first, last = np.mean([h[0] for h in hist[:4]]), np.mean([h[0] for h in hist[-4:]])
    assert last < 0.7 * first, (first, last)
    assert os.path.exists(tmp_path / 'final_model.pt')
    # resume: a fresh model picks the checkpoint up (train.py:121-127)
    hp.epochs = 1
    m2, h2 = tr.train_language_model(dhp, hp, src, ckpt_dir=str(tmp_path), log_every=1000)
    assert m2.global_step == 48 + 4 and h2[0][0] < first


def test_end2end_training_loop_from_wav_files(tmp_path):
    from asr_dfcnn_transformer_amd import train as tr
    from asr_dfcnn_transformer_amd.data_util import DataUtil
    from asr_dfcnn_transformer_amd.e2e_data_loader import dataloader
    from asr_dfcnn_transformer_amd.e2e_model import E2EHparams
    from asr_dfcnn_transformer_amd.hparams import TransDataHparams
    hp = TransDataHparams().args
    args = E2EHparams()
    args.batch_size, args.hidden_units, args.num_heads, args.num_blocks, args.epochs, args.dropout_rate = 2, 128, 2, 1, 6, 0.0
    args.learning_rate, args.log_every_n, args.save_every_n = 2e-3, 1000, 5
    du = DataUtil(hp, 2, 'train', data_dir=INDEX, audio_root=str(tmp_path / 'audio'))
    _write_wavs(str(tmp_path / 'audio'), du.path_lst, seconds=(0.5, 0.8))
    dl = dataloader(args, hp, data_util=du)
    model, hist = tr.train_transformer(args, dl, ckpt_dir=str(tmp_path / 'ck'))
    assert len(hist) == 6 * 4 and model.engine.global_step == 24 and model.prenet.global_step == 24
    assert np.mean([h[0] for h in hist[-4:]]) < 0.8 * np.mean([h[0] for h in hist[:4]])
    assert os.path.exists(tmp_path / 'ck' / 'final_model.pt') and os.path.exists(tmp_path / 'ck' / 'model_20.pt')
    from asr_dfcnn_transformer_amd.train import load_checkpoint
    load_checkpoint(model, str(tmp_path / 'ck' / 'model_20.pt'))         # composite checkpoint: encoder-decoder + pre-net
    assert model.engine.global_step == 20 and model.prenet.global_step == 20
    # resume (end2end/model.py:81-88 restores the latest checkpoint before the loop): a second call with the same ckpt_dir and a
    # FRESH model -- whose engines do not exist before its first batch -- continues from final_model.pt (written at step 20):
    # global_step and the learning-rate schedule go on, the first losses are those of a trained model, not of a random one
    args.epochs = 1
    m2, h2 = tr.train_transformer(args, dl, ckpt_dir=str(tmp_path / 'ck'))
    assert m2 is not model and len(h2) == 4
    assert m2.engine.global_step == 20 + 4 and m2.prenet.global_step == 20 + 4
    assert h2[0][1] == pytest.approx(model.engine.current_learning_rate(20), rel=1e-6)
    assert np.mean([h[0] for h in h2]) < 0.8 * np.mean([h[0] for h in hist[:4]])


def test_speech_test_loop_writes_pred_log(tmp_path):
    from asr_dfcnn_transformer_amd.acoustic_model import CNNCTCModel
    from asr_dfcnn_transformer_amd.data_util import DataUtil
    from asr_dfcnn_transformer_amd.hparams import TransDataHparams
    from asr_dfcnn_transformer_amd.language_model import Language_Model
    from asr_dfcnn_transformer_amd.test_pipeline import speech_test
    dhp = TransDataHparams().args
    dhp.aishell = False
    du = DataUtil(dhp, 1, 'test', data_dir=INDEX, audio_root=str(tmp_path / 'audio'))
    _write_wavs(str(tmp_path / 'audio'), du.path_lst)
    hp, dl = _small_am_lm(du)
    hp.is_training = False
    am = CNNCTCModel(hp, dl.acoustic_vocab_size, dl.language_vocab_size, widths=(8, 16, 16, 32), batch_size=1)
    lm = Language_Model(hp, dl.acoustic_vocab_size, dl.language_vocab_size, batch_size=1)
    py_acc, han_acc = speech_test(am, lm, dl, num=3, pred_dir=str(tmp_path / 'pred'), start=1, verbose=False)
    assert 0.0 <= py_acc <= 1.0 and 0.0 <= han_acc <= 1.0
    log = open(tmp_path / 'pred' / 'pred_log', encoding='utf-8').read()
    assert log.count('原文汉字结果:') == 3 and log.count('预测拼音结果:') == 3
    assert '原文汉字结果:北京天气' in log and '原文拼音结果:bei3 jing1 tian1 qi4' in log           # index 1 first (start=1), wraps to 0
    assert log.index('北京天气') < log.index('我是大学人') < log.index('你好中国人')
    assert log.rstrip().endswith('%') and '拼音 word accuracy ratio: ' in log and '汉字 word accuracy ratio: ' in log
    # bookkeeping: errors are capped at the sentence length (test.py:78-81)
    from asr_dfcnn_transformer_amd.test_pipeline import AccuracyMeter
    m = AccuracyMeter(); m.update([1, 2, 3], [9, 9, 9, 9, 9, 9, 9]); assert (m.words, m.errors) == (3, 3)
    m.update([1, 2, 3, 4], [1, 2, 4]); assert (m.words, m.errors) == (7, 4)


def test_compute_fbank_from_file_both_readers(tmp_path):
    """util/wav_util.py:13-19,34-45.  sf_flag=True: the soundfile view of the file (float in [-1, 1), 1-D) through
    compute_fbank_from_api.  sf_flag=False: read_wav_data returns the int16 frames as [1, n], and python_speech_features'
    pre-emphasis ``append(signal[0], signal[1:] - 0.97 * signal[:-1])`` of a [1, n] array is the row itself -- that path is
    NOT pre-emphasised (a reference quirk the oracle's logfbank reproduces by the same numpy expression)."""
    from asr_dfcnn_transformer_amd.wav_util import compute_fbank_from_file, read_wav_data
    sigs = _write_wavs(str(tmp_path), ['a/one.wav'], seed=3, seconds=(0.7, 0.7))
    f = str(tmp_path / 'a' / 'one.wav')
    wd, sr = read_wav_data(f)
    assert sr == 16000 and wd.dtype == np.int16 and wd.shape == (1, len(sigs['a/one.wav']))
    assert np.array_equal(wd[0].astype(np.float64) / 32768.0, sigs['a/one.wav'])
    got_sf = compute_fbank_from_file(f, feature_dim=200, sf_flag=True)
    ref_sf = ofb.compute_fbank_from_api(sigs['a/one.wav'], 16000, nfilt=200)
    assert got_sf.shape == ref_sf.shape and np.abs(got_sf - ref_sf).max() <= 2e-6
    got = compute_fbank_from_file(f, feature_dim=200)
    ref = ofb.compute_fbank_from_api(wd, 16000, nfilt=200)                  # 2-D in: no pre-emphasis, int16 scale
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-6
    assert np.abs(got - got_sf).max() > 1e-2                                # the two readers do give different features
