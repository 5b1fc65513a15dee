"""Oracle for the optimiser path (float64).  Test infrastructure only; parity unpinned.

Restates tf.train.polynomial_decay(cycle=True, power=0.5) and
tf.train.AdamOptimizer as used at lm_and_am/model/acoustic_model2.py:85-91
(SURVEY.md Appendix A10, A11).
"""
import math

import numpy as np


def polynomial_decay(lr0, step, decay_steps=5000, end_lr=1e-6, power=0.5, cycle=True):
    step = float(step)
    ds = float(decay_steps)
    if cycle:
        mult = 1.0 if step == 0 else math.ceil(step / ds)
        ds = ds * mult
    else:
        step = min(step, ds)
    p = step / ds
    return (lr0 - end_lr) * (1.0 - p) ** power + end_lr


def adam_tf_step(theta, g, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    """One TF-Adam update; ``t`` is the 1-based update count.  Returns (theta, m, v)."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    theta = theta - lr_t * m / (np.sqrt(v) + eps)
    return theta, m, v
