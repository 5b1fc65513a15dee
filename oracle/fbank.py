"""Oracle for the fbank feature path (float64 numpy).  Test infrastructure only.

Follows ``util/wav_util.py:22-31`` (``compute_fbank_from_api``):
``logfbank(signal, sample_rate, nfilt=nfilt)`` from python_speech_features
0.6 (requirements.txt:42; not installed, algorithm restated from the
published source: SURVEY.md Appendix A1) followed by
``sklearn.preprocessing.scale`` (Appendix A2).  Parity unpinned: the
reference holds no fixture for this function.
"""
import math

import numpy as np


def hz2mel(hz):
    return 2595.0 * np.log10(1.0 + hz / 700.0)


def mel2hz(mel):
    return 700.0 * (10.0 ** (mel / 2595.0) - 1.0)


def round_half_up(x):
    return int(math.floor(x + 0.5))


def get_filterbanks(nfilt=20, nfft=512, samplerate=16000, lowfreq=0, highfreq=None):
    """python_speech_features.base.get_filterbanks (v0.6)."""
    highfreq = highfreq or samplerate / 2
    lowmel = hz2mel(lowfreq)
    highmel = hz2mel(highfreq)
    melpoints = np.linspace(lowmel, highmel, nfilt + 2)
    bins = np.floor((nfft + 1) * mel2hz(melpoints) / samplerate)
    fb = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(nfilt):
        for i in range(int(bins[j]), int(bins[j + 1])):
            fb[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
        for i in range(int(bins[j + 1]), int(bins[j + 2])):
            fb[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
    return fb


def preemphasis(signal, coeff=0.97):
    return np.append(signal[0], signal[1:] - coeff * signal[:-1])


def framesig(sig, frame_len, frame_step):
    """python_speech_features.sigproc.framesig with the default all-ones window."""
    slen = len(sig)
    frame_len = round_half_up(frame_len)
    frame_step = round_half_up(frame_step)
    if slen <= frame_len:
        numframes = 1
    else:
        numframes = 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))
    padlen = int((numframes - 1) * frame_step + frame_len)
    padsignal = np.concatenate((sig, np.zeros((padlen - slen,))))
    idx = (np.arange(frame_len)[None, :] + frame_step * np.arange(numframes)[:, None])
    return padsignal[idx]


def powspec(frames, nfft):
    return 1.0 / nfft * np.square(np.absolute(np.fft.rfft(frames, nfft)))


def logfbank(signal, samplerate=16000, winlen=0.025, winstep=0.01, nfilt=26,
             nfft=512, lowfreq=0, highfreq=None, preemph=0.97):
    """python_speech_features.logfbank (v0.6 defaults; called at wav_util.py:29)."""
    signal = np.asarray(signal, dtype=np.float64)
    highfreq = highfreq or samplerate / 2
    signal = preemphasis(signal, preemph)
    frames = framesig(signal, winlen * samplerate, winstep * samplerate)
    pspec = powspec(frames, nfft)
    fb = get_filterbanks(nfilt, nfft, samplerate, lowfreq, highfreq)
    feat = np.dot(pspec, fb.T)
    feat = np.where(feat == 0, np.finfo(float).eps, feat)
    return np.log(feat)


def scale(x):
    """sklearn.preprocessing.scale(X) on a 2-D array (wav_util.py:30), restated
    from sklearn/preprocessing/_data.py (version unpinned by the reference;
    >= 0.24 behaviour): centre, re-centre if the centred mean is not ~0, divide
    by the population std (std < 10*eps -> 1), re-centre again if needed.  The
    re-centring matters: a constant column c has mean c+d (rounding), becomes
    d/|d| = +-1 after the division and is brought to exactly 0 by the last
    step -- which is how the 43 empty mel filters end up exactly 0."""
    x = np.asarray(x, dtype=np.float64)
    mean = x.mean(axis=0)
    std = np.sqrt(np.mean(np.abs(x - mean) ** 2, axis=0))      # np.nanstd
    xr = x - mean
    mean_1 = xr.mean(axis=0)
    if not np.allclose(mean_1, 0):
        xr = xr - mean_1
    std = np.where(std < 10 * np.finfo(np.float64).eps, 1.0, std)
    xr = xr / std
    mean_2 = xr.mean(axis=0)
    if not np.allclose(mean_2, 0):
        xr = xr - mean_2
    return xr


def compute_fbank_from_api(signal, sample_rate, nfilt=200):
    """util/wav_util.py:22-31."""
    return scale(logfbank(signal, sample_rate, nfilt=nfilt))


def num_frames(nsamples, frame_len=400, frame_step=160):
    if nsamples <= frame_len:
        return 1
    return 1 + int(math.ceil((1.0 * nsamples - frame_len) / frame_step))


def build_LFR_features(inputs, m, n):
    """util/utils.py:7-31, restated rule by rule (docstring :9-13): output frame i stacks the m input frames that start
    at i*n; T_lfr = ceil(T / n) frames; a frame that runs past the end is completed with copies of the LAST input frame
    (:25-29).  m = n = 1 returns the input; m = 1 skips; n = 1 stacks.  Written as explicit loops on purpose: the product
    (asr_dfcnn_transformer_amd.utils.build_LFR_features and the device kernel asr_lfr) gathers by a clamped index, and the
    two are compared in tests/test_oracle_cpu.py next to a hand-written fixture."""
    inputs = np.asarray(inputs)
    T, D = inputs.shape
    T_lfr = (T + n - 1) // n
    out = np.empty((T_lfr, m * D), dtype=inputs.dtype)
    for i in range(T_lfr):
        for j in range(m):
            t = i * n + j
            src = inputs[t] if t < T else inputs[T - 1]
            out[i, j * D:(j + 1) * D] = src
    return out
