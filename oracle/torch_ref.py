"""Second, independent CPU restatement of the DFCNN(+SE)+CTC step on torch-CPU
ops (oneDNN conv/GEMM, torch ctc_loss, autograd).  Test infrastructure only.

Two uses: (1) cross-checks oracle/dfcnn.py (numpy, hand-derived backward) in
tests/; (2) the ``cpu_baseline`` leg of bench.py ("port": a stand-in for the
reference's TF-CPU path, which cannot run offline -- SURVEY.md §8d).
Cites the same reference lines as oracle/dfcnn.py.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3
K_EPSILON = 1e-7


def to_torch_params(P, dtype=torch.float64, requires_grad=True):
    out = {}
    for name, d in P.items():
        out[name] = {k: torch.tensor(v, dtype=dtype, requires_grad=requires_grad) for k, v in d.items()}
    return out


def cell(x, p, pool):
    # x NCHW; w stored HWIO
    w = p['w'].permute(3, 2, 0, 1)
    k = w.shape[-1]
    z = F.conv2d(x, w, p['b'], padding=k // 2)
    a = F.relu(z)
    y = a * (p['gamma'] / math.sqrt(1.0 + BN_EPS)).view(1, -1, 1, 1) + p['beta'].view(1, -1, 1, 1)
    if pool == 'avg':
        y = F.avg_pool2d(y, 2)
    elif pool == 'max':
        y = F.max_pool2d(y, 2)
    return y


def se(x, p, use_bn):
    if use_bn:
        x = x * (p['gamma'] / math.sqrt(1.0 + BN_EPS)).view(1, -1, 1, 1) + p['beta'].view(1, -1, 1, 1)
    s = x.mean(dim=(2, 3))
    e = torch.sigmoid(F.relu(s @ p['w1'] + p['b1']) @ p['w2'] + p['b2'])
    return x * e[:, :, None, None]


def forward(ops, P, x_nhwc):
    acts = {'x': x_nhwc.permute(0, 3, 1, 2)}
    for op in ops:
        if op[0] == 'cell':
            _, src, dst, cin, cout, k, pool = op
            acts[dst] = cell(acts[src], P[dst], pool)
        elif op[0] == 'se':
            _, main, br, dst, C, hid, use_bn = op
            acts[dst] = acts[main] + se(acts[br], P[dst], use_bn)
        elif op[0] == 'dense':
            _, src, dst, cin, cout, act = op
            h = acts[src]
            if h.dim() == 4:                       # NCHW -> [B,T,W*C]
                h = h.permute(0, 2, 3, 1).reshape(h.shape[0], h.shape[2], -1)
            z = h @ P[dst]['w'] + P[dst]['b']
            acts[dst] = F.relu(z) if act == 'relu' else z
    return acts[ops[-1][2]], acts


def loss_head(d, logits_length, labels, blank):
    """d [B,T,V] -> (logits_tm, per-utterance CTC loss [B])."""
    logits_tm = torch.log(torch.softmax(d, dim=-1).transpose(0, 1) + K_EPSILON)
    lp = torch.log_softmax(logits_tm, dim=-1)          # tf.nn.ctc_loss applies its own softmax
    flat = torch.tensor([v for l in labels for v in l], dtype=torch.long)
    tl = torch.tensor([len(l) for l in labels], dtype=torch.long)
    il = torch.as_tensor(logits_length, dtype=torch.long)
    loss = F.ctc_loss(lp, flat, il, tl, blank=blank, reduction='none', zero_infinity=False)
    return logits_tm, loss


def train_step(ops, P, x_nhwc, logits_length, labels):
    d, acts = forward(ops, P, x_nhwc)
    logits_tm, loss = loss_head(d, logits_length, labels, d.shape[-1] - 1)
    mean_loss = loss.mean()
    mean_loss.backward()
    return logits_tm.detach(), loss.detach(), mean_loss.detach(), acts


def adam_tf_(params, grads, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    with torch.no_grad():
        for p, g, mm, vv in zip(params, grads, m, v):
            mm.mul_(beta1).add_(g, alpha=1.0 - beta1)
            vv.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
            p.addcdiv_(mm, vv.sqrt().add_(eps), value=-lr_t)
