"""Oracle layer arithmetic (float64 numpy, NHWC), forward and hand-derived
backward.  Test infrastructure only; parity unpinned (see oracle/__init__.py).

Restates the TensorFlow-1.x layer semantics the reference relies on
 at these call sites:
  conv2d 3x3/1x1 SAME + bias + ReLU   lm_and_am/model/acoustic_model2.py:102-110
  batch_normalization (frozen affine)  acoustic_model2.py:112-113  
  average / max pooling 2x2 VALID      acoustic_model2.py:115-117, acoustic_model.py:111-112
  squeeze-excitation                   acoustic_model2.py:135-148
  dense (+relu / softmax)              acoustic_model2.py:120-121
  log(softmax + 1e-7), time-major      acoustic_model2.py:67-68
"""
import numpy as np

BN_EPS = 1e-3          # tf.layers.batch_normalization default epsilon
K_EPSILON = 1e-7       # keras.backend.epsilon()


# ----------------------------------------------------------------- conv
def _patches(x, kh, kw):
    """x [B,H,W,C] -> SAME-padded patches [B,H,W,kh,kw,C] (stride 1)."""
    B, H, W, C = x.shape
    ph, pw = (kh - 1) // 2, (kw - 1) // 2
    xp = np.zeros((B, H + kh - 1, W + kw - 1, C), dtype=x.dtype)
    xp[:, ph:ph + H, pw:pw + W, :] = x
    s = xp.strides
    return np.lib.stride_tricks.as_strided(
        xp, shape=(B, H, W, kh, kw, C),
        strides=(s[0], s[1], s[2], s[1], s[2], s[3]), writeable=False)


def conv2d_same(x, w):
    """Cross-correlation, NHWC x HWIO, stride 1, SAME (tf.layers.conv2d)."""
    kh, kw, cin, cout = w.shape
    p = _patches(x, kh, kw)
    return np.tensordot(p, w, axes=([3, 4, 5], [0, 1, 2]))


def conv2d_same_bwd(x, w, dz):
    """Returns (dx, dw) for z = conv2d_same(x, w)."""
    kh, kw, cin, cout = w.shape
    p = _patches(x, kh, kw)
    dw = np.tensordot(p, dz, axes=([0, 1, 2], [0, 1, 2]))
    # dx = correlate dz with the 180-degree-rotated, in/out-swapped kernel
    wr = w[::-1, ::-1].transpose(0, 1, 3, 2)
    dx = conv2d_same(dz, wr)
    return dx, dw


# ----------------------------------------------------------------- BN
def bn_frozen(a, gamma, beta, mov_mean=None, mov_var=None):
    """tf.layers.batch_normalization(x) with training=False."""
    C = a.shape[-1]
    mm = np.zeros(C) if mov_mean is None else mov_mean
    mv = np.ones(C) if mov_var is None else mov_var
    return gamma * (a - mm) / np.sqrt(mv + BN_EPS) + beta


def bn_frozen_bwd(a, gamma, dy, mov_mean=None, mov_var=None):
    C = a.shape[-1]
    mm = np.zeros(C) if mov_mean is None else mov_mean
    mv = np.ones(C) if mov_var is None else mov_var
    rs = 1.0 / np.sqrt(mv + BN_EPS)
    axes = tuple(range(a.ndim - 1))
    dgamma = (dy * (a - mm) * rs).sum(axis=axes)
    dbeta = dy.sum(axis=axes)
    da = dy * gamma * rs
    return da, dgamma, dbeta


def bn_batch(a, gamma, beta, eps=BN_EPS):
    """Batch-statistics form (training=True; biased variance)."""
    axes = tuple(range(a.ndim - 1))
    mu = a.mean(axis=axes)
    var = a.var(axis=axes)
    xhat = (a - mu) / np.sqrt(var + eps)
    return gamma * xhat + beta, (xhat, var)


def bn_batch_bwd(cache, gamma, dy, eps=BN_EPS):
    xhat, var = cache
    axes = tuple(range(dy.ndim - 1))
    n = np.prod([dy.shape[i] for i in axes])
    dgamma = (dy * xhat).sum(axis=axes)
    dbeta = dy.sum(axis=axes)
    dxhat = dy * gamma
    da = (dxhat - dxhat.mean(axis=axes) - xhat * (dxhat * xhat).sum(axis=axes) / n) / np.sqrt(var + eps)
    return da, dgamma, dbeta


# ----------------------------------------------------------------- pooling
def avgpool2(y):
    B, H, W, C = y.shape
    h2, w2 = H // 2, W // 2
    v = y[:, :h2 * 2, :w2 * 2, :].reshape(B, h2, 2, w2, 2, C)
    return v.mean(axis=(2, 4))


def avgpool2_bwd(yshape, dp):
    B, H, W, C = yshape
    h2, w2 = H // 2, W // 2
    dy = np.zeros(yshape, dtype=dp.dtype)
    dy[:, :h2 * 2, :w2 * 2, :] = np.repeat(np.repeat(dp, 2, axis=1), 2, axis=2) * 0.25
    return dy


def maxpool2(y):
    B, H, W, C = y.shape
    h2, w2 = H // 2, W // 2
    v = y[:, :h2 * 2, :w2 * 2, :].reshape(B, h2, 2, w2, 2, C)
    return v.max(axis=(2, 4))


def maxpool2_bwd(y, dp):
    """Gradient goes to the FIRST maximum in row-major window order (Appendix A5)."""
    B, H, W, C = y.shape
    h2, w2 = H // 2, W // 2
    v = y[:, :h2 * 2, :w2 * 2, :].reshape(B, h2, 2, w2, 2, C).transpose(0, 1, 3, 5, 2, 4)
    v = v.reshape(B, h2, w2, C, 4)
    arg = v.argmax(axis=-1)             # first occurrence
    onehot = (np.arange(4)[None, None, None, None, :] == arg[..., None]).astype(dp.dtype)
    d = onehot * dp[..., None]
    d = d.reshape(B, h2, w2, C, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(B, h2 * 2, w2 * 2, C)
    dy = np.zeros(y.shape, dtype=dp.dtype)
    dy[:, :h2 * 2, :w2 * 2, :] = d
    return dy


# ----------------------------------------------------------------- cell
def cell_fwd(x, p, pool=None):
    """cnn_cell without NiN: conv -> +bias -> ReLU -> BN(frozen) -> [pool]
    (acoustic_model2.py:126-133; order per SURVEY Q2)."""
    z = conv2d_same(x, p['w']) + p['b']
    a = np.maximum(z, 0.0)
    y = bn_frozen(a, p['gamma'], p['beta'])
    if pool == 'avg':
        out = avgpool2(y)
    elif pool == 'max':
        out = maxpool2(y)
    else:
        out = y
    return out, (x, a, y)


def cell_bwd(cache, p, dout, pool=None):
    x, a, y = cache
    if pool == 'avg':
        dy = avgpool2_bwd(y.shape, dout)
    elif pool == 'max':
        dy = maxpool2_bwd(y, dout)
    else:
        dy = dout
    da, dgamma, dbeta = bn_frozen_bwd(a, p['gamma'], dy)
    dz = da * (a > 0)
    db = dz.sum(axis=(0, 1, 2))
    dx, dw = conv2d_same_bwd(x, p['w'], dz)
    return dx, {'w': dw, 'b': db, 'gamma': dgamma, 'beta': dbeta}


# ----------------------------------------------------------------- SE
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def se_fwd(x, p):
    """squeeze_excitation_layer (acoustic_model2.py:141-148): returns BN(x) * e."""
    xt = bn_frozen(x, p['gamma'], p['beta'])
    s = xt.mean(axis=(1, 2))                       # [B,C]
    u = s @ p['w1'] + p['b1']
    r = np.maximum(u, 0.0)
    v = r @ p['w2'] + p['b2']
    e = sigmoid(v)
    return xt * e[:, None, None, :], (x, xt, s, r, e)


def se_bwd(cache, p, dout):
    x, xt, s, r, e = cache
    B, H, W, C = x.shape
    de = (dout * xt).sum(axis=(1, 2))
    dxt = dout * e[:, None, None, :]
    dv = de * e * (1.0 - e)
    dw2 = r.T @ dv
    db2 = dv.sum(axis=0)
    dr = dv @ p['w2'].T
    du = dr * (r > 0)
    dw1 = s.T @ du
    db1 = du.sum(axis=0)
    ds = du @ p['w1'].T
    dxt = dxt + ds[:, None, None, :] / (H * W)
    dx, dgamma, dbeta = bn_frozen_bwd(x, p['gamma'], dxt)
    return dx, {'gamma': dgamma, 'beta': dbeta, 'w1': dw1, 'b1': db1, 'w2': dw2, 'b2': db2}


# ----------------------------------------------------------------- head
def dense_fwd(x, w, b):
    return x @ w + b


def dense_bwd(x, w, dy):
    x2 = x.reshape(-1, x.shape[-1])
    dy2 = dy.reshape(-1, dy.shape[-1])
    return (dy @ w.T), x2.T @ dy2, dy2.sum(axis=0)


def softmax(d):
    m = d.max(axis=-1, keepdims=True)
    ex = np.exp(d - m)
    return ex / ex.sum(axis=-1, keepdims=True)


def log_softmax_eps_tm(d):
    """d [B,T,V] dense pre-activations -> time-major log(softmax(d)+1e-7) [T,B,V]
    (acoustic_model2.py:67-68)."""
    p = softmax(d)
    return np.log(np.transpose(p, (1, 0, 2)) + K_EPSILON)


def log_softmax_eps_tm_bwd(d, g_tm):
    """g_tm = dL/d(logits_tm) [T,B,V] -> dL/dd [B,T,V]."""
    p = softmax(d)
    g = np.transpose(g_tm, (1, 0, 2))
    dp = g / (p + K_EPSILON)
    return p * (dp - (dp * p).sum(axis=-1, keepdims=True))
