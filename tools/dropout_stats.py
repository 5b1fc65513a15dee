"""Statistics of the counter-based dropout generators (CPU, numpy): the shipped one (murmur3 finaliser per element) against cheaper
candidates for the attention kernels.  Keep fraction, serial correlation at the lags the attention layout makes adjacent, row /
column spread of a [rows][512] mask against the binomial expectation, agreement of masks of neighbouring seeds, chi-square of
8-element keep patterns.  usage: python tools/dropout_stats.py"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def fmix32(h):
    h = h & M32
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    return h


def gen_murmur(idx, seed):
    return fmix32((idx * np.uint64(0x9E3779B1) + np.uint64(seed)) & M32)


def gen_light(idx, seed):
    s2 = fmix32(np.array([seed], dtype=np.uint64))[0]
    x = (idx * np.uint64(0x9E3779B1)) & M32
    x ^= s2
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & M32
    return x


def gen_light2(idx, seed):
    x = gen_light(idx, seed)
    x ^= x >> np.uint64(13)
    return x


def report(name, gen, rate=0.2):
    thr = np.uint64(int(rate * 4294967296.0 + 0.5))
    n = 1 << 24
    idx = np.arange(n, dtype=np.uint64)
    out = []
    keep = (gen(idx, 5) >= thr)
    out.append('keep %.5f' % keep.mean())
    k = keep.astype(np.float64) - keep.mean()
    var = (k * k).mean()
    cors = []
    for lag in (1, 2, 3, 4, 8, 32, 64, 100, 512, 1024, 512 * 512):
        cors.append(abs((k[:-lag] * k[lag:]).mean() / var))
    out.append('max |corr| over lags %.2e (3 sigma = %.2e)' % (max(cors), 3 / np.sqrt(n)))
    m = keep.reshape(-1, 512)
    p = keep.mean()
    out.append('row-mean var / binomial %.3f, col-mean var / binomial %.3f' % (m.mean(1).var() / (p * (1 - p) / 512), m.mean(0).var() / (p * (1 - p) / m.shape[0])))
    agree = []
    for ds in (1, 1009, 7919, 1009 + 7919):
        k2 = gen(idx, 5 + ds) >= thr
        agree.append((keep == k2).mean())
    out.append('agreement with neighbouring seeds %s (independent: %.4f)' % (' '.join('%.4f' % a for a in agree), p * p + (1 - p) ** 2))
    bits = keep.reshape(-1, 8)
    code = (bits * (1 << np.arange(8))).sum(1)
    cnt = np.bincount(code, minlength=256).astype(np.float64)
    pop = np.array([bin(c).count('1') for c in range(256)])
    exp = len(code) * p ** pop * (1 - p) ** (8 - pop)
    out.append('chi2 of 8-element patterns %.0f (255 dof)' % (((cnt - exp) ** 2 / exp).sum()))
    # a 2-D view: stride-512 neighbours (same key, consecutive queries)
    col = keep.reshape(-1, 512)[:, 7].astype(np.float64) - p
    out.append('|corr| of consecutive rows at one column %.2e (3 sigma %.2e)' % (abs((col[:-1] * col[1:]).mean() / var), 3 / np.sqrt(len(col))))
    print('%-8s %s' % (name, '; '.join(out)))


if __name__ == '__main__':
    for name, g in (('murmur', gen_murmur), ('light', gen_light), ('light2', gen_light2)):
        report(name, g)
