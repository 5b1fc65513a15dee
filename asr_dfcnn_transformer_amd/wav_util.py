"""Device fbank features: the drop-in for ``util/wav_util.py`` of the reference.

``compute_fbank_from_api(signal, sample_rate, nfilt=200)`` keeps the reference signature
(util/wav_util.py:22-31) and returns a float64 numpy array of shape [T, nfilt]; the batched
entry ``FbankExtractor.batch`` keeps everything on the GPU and writes the zero-padded
``[B, feature_max_length, nfilt]`` float32 tensor the acoustic model consumes
(lm_and_am/data_loader.py:107,146).  All arithmetic runs in libasrhip.so (asr_fbank);
the host only builds the (constant) mel filterbank and FFT twiddle tables.
"""
import math

import numpy as np
import torch

from . import ops


def _hz2mel(hz):
    return 2595.0 * np.log10(1.0 + hz / 700.0)


def _mel2hz(mel):
    return 700.0 * (10.0 ** (mel / 2595.0) - 1.0)


def mel_filterbank_banded(nfilt, nfft, samplerate, lowfreq=0.0, highfreq=None):
    """Triangular mel filters on FFT-bin edges floor((nfft+1)*hz/sr) as python_speech_features
    0.6 defines them, returned in banded form: (start[nfilt], count[nfilt], weight[nfilt][width])."""
    highfreq = highfreq or samplerate / 2
    mel = np.linspace(_hz2mel(lowfreq), _hz2mel(highfreq), nfilt + 2)
    edges = np.floor((nfft + 1) * _mel2hz(mel) / samplerate).astype(np.int64)
    start = edges[:-2].copy()
    count = edges[2:] - edges[:-2]
    width = int(max(1, count.max()))
    weight = np.zeros((nfilt, width), dtype=np.float64)
    for j in range(nfilt):
        lo, mid, hi = edges[j], edges[j + 1], edges[j + 2]
        for i in range(lo, mid):
            weight[j, i - lo] = (i - lo) / float(mid - lo)
        for i in range(mid, hi):
            weight[j, i - lo] = (hi - i) / float(hi - mid)
    return start.astype(np.int32), count.astype(np.int32), weight, width


def num_frames(nsamples, frame_len=400, frame_step=160):
    if nsamples <= frame_len:
        return 1
    return 1 + int(math.ceil((nsamples - frame_len) / float(frame_step)))


class FbankExtractor:
    def __init__(self, sample_rate=16000, nfilt=200, nfft=512, winlen=0.025, winstep=0.01, preemph=0.97,
                 device='cuda'):
        self.sample_rate, self.nfilt, self.nfft, self.preemph = sample_rate, nfilt, nfft, preemph
        self.frame_len = int(math.floor(winlen * sample_rate + 0.5))
        self.frame_step = int(math.floor(winstep * sample_rate + 0.5))
        st, cnt, w, width = mel_filterbank_banded(nfilt, nfft, sample_rate)
        self.fb_width = width
        self.fb_start = torch.tensor(st, dtype=torch.int32, device=device)
        self.fb_count = torch.tensor(cnt, dtype=torch.int32, device=device)
        self.fb_weight = torch.tensor(w, dtype=torch.float64, device=device)
        k = np.arange(nfft // 2, dtype=np.float64)
        tw = np.stack([np.cos(2 * np.pi * k / nfft), -np.sin(2 * np.pi * k / nfft)], axis=1)
        self.twiddle = torch.tensor(tw, dtype=torch.float64, device=device)
        self.device = device
        self._ws = None

    def batch(self, signal, nsamples, t_pad, out=None):
        """signal [B, max_samples] float32 cuda, nsamples [B] int32 cuda -> (feat [B,t_pad,nfilt] f32, frames [B] i32)."""
        B, max_samples = signal.shape
        max_frames = num_frames(max_samples, self.frame_len, self.frame_step)
        need = B * max_frames * self.nfilt
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.float64, device=self.device)
        if out is None:
            out = torch.empty(B, t_pad, self.nfilt, dtype=torch.float32, device=self.device)
        frames = torch.empty(B, dtype=torch.int32, device=self.device)
        ops.fbank(signal, nsamples, self.frame_len, self.frame_step, self.nfft, self.preemph, self.nfilt,
                  self.fb_start, self.fb_count, self.fb_weight, self.fb_width, self.twiddle, self._ws,
                  max_frames, out, t_pad, frames)
        return out, frames


_extractors = {}


def compute_fbank_from_api(signal, sample_rate, nfilt=200):
    """Reference signature (util/wav_util.py:22-31).  ``signal`` is a 1-D array (soundfile: float in [-1, 1)).
    A 2-D ``[channels, n]`` array -- what ``read_wav_data`` returns -- goes through python_speech_features' pre-emphasis as
    the reference's does: ``numpy.append(signal[0], signal[1:] - 0.97 * signal[:-1])`` on a [1, n] array is the row ITSELF
    (``signal[1:]`` is empty), i.e. a mono file read that way is NOT pre-emphasised (and keeps its int16 scale, which the
    per-column standardisation removes again); that append is done here on the host and the device pass runs with
    pre-emphasis 0."""
    sig_np = np.asarray(signal)
    preemph = 0.97
    if sig_np.ndim == 2:
        sig_np = np.append(sig_np[0], sig_np[1:] - 0.97 * sig_np[:-1])
        preemph = 0.0
    key = (int(sample_rate), int(nfilt), preemph)
    if key not in _extractors:
        _extractors[key] = FbankExtractor(sample_rate=int(sample_rate), nfilt=int(nfilt), preemph=preemph)
    ex = _extractors[key]
    sig = torch.as_tensor(np.asarray(sig_np, dtype=np.float32)).reshape(1, -1).cuda()
    n = torch.tensor([sig.shape[1]], dtype=torch.int32, device='cuda')
    T = num_frames(sig.shape[1], ex.frame_len, ex.frame_step)
    feat, _ = ex.batch(sig, n, T)
    return feat[0].double().cpu().numpy()


def read_wav_data(filename):
    """util/wav_util.py:34-45: the frames of a 16-bit PCM WAV file as an int16 array ``[channels, n]`` and the frame rate
    (``np.fromstring`` there; the removed alias of ``np.frombuffer``)."""
    import wave
    with wave.open(filename, 'rb') as w:
        num_frame, num_channel, framerate = w.getnframes(), w.getnchannels(), w.getframerate()
        str_data = w.readframes(num_frame)
    wave_data = np.frombuffer(str_data, dtype=np.short).reshape(-1, num_channel).T
    return wave_data, framerate


def compute_fbank_from_file(file, feature_dim=200, sf_flag=False):
    """util/wav_util.py:13-19: ``sf_flag`` reads the file the way soundfile does (float64 in [-1, 1), 1-D for a mono file;
    16-bit PCM is decoded with the standard library: data_util.read_wav_pcm16), else through ``read_wav_data``."""
    if sf_flag:
        from .data_util import read_wav_pcm16
        signal, sample_rate = read_wav_pcm16(file)
    else:
        signal, sample_rate = read_wav_data(file)
    return compute_fbank_from_api(signal, sample_rate, nfilt=feature_dim)
