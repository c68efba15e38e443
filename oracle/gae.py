"""ORACLE (test infrastructure): GAE(lambda) over a flat concatenated batch.

Restates ``estimate_advantages`` (/root/reference/core/common.py:5-25): a single reverse
sweep over the WHOLE flat batch (previous value/advantage start at 0 and are gated by the
mask of the current sample), returns = values + advantages, then standardisation with the
unbiased std (torch.std default). Pinned against tests/golden/gae.npz.
"""
import numpy as np


def estimate_advantages(rewards, masks, values, gamma, tau):
    r = np.asarray(rewards, float).ravel()
    m = np.asarray(masks, float).ravel()
    v = np.asarray(values, float).ravel()
    n = r.shape[0]
    adv = np.empty(n)
    nxt_v, nxt_a = 0.0, 0.0
    for i in range(n - 1, -1, -1):
        delta = r[i] + gamma * nxt_v * m[i] - v[i]
        adv[i] = delta + gamma * tau * nxt_a * m[i]
        nxt_v, nxt_a = v[i], adv[i]
    ret = v + adv
    adv_n = (adv - adv.mean()) / adv.std(ddof=1)
    return adv_n.reshape(-1, 1), ret.reshape(-1, 1), adv.reshape(-1, 1)
