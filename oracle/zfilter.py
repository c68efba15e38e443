"""ORACLE (test infrastructure): running observation normaliser.

Restates ``RunningStat`` / ``ZFilter`` (/root/reference/utils/zfilter.py:7-67): Welford
update per sample, var = S/(n-1) for n>1 else mean^2, y = clip((x-mean)/(std+1e-8), +-clip).
Also provides the batched Chan merge the product uses, so tests can show that merging a
block equals pushing its rows one by one. Pinned against tests/golden/zfilter.npz.
"""
import numpy as np


class RunningStatOracle:
    def __init__(self, dim):
        self.n = 0
        self.mean = np.zeros(dim)
        self.S = np.zeros(dim)

    def push(self, x):
        x = np.asarray(x, float)
        self.n += 1
        if self.n == 1:
            self.mean = x.copy()
        else:
            old = self.mean
            self.mean = old + (x - old) / self.n
            self.S = self.S + (x - old) * (x - self.mean)

    def merge_block(self, X):
        """Chan/Golub/LeVeque pairwise merge of a (B,dim) block (what the batched kernel does)."""
        X = np.atleast_2d(np.asarray(X, float))
        nb = X.shape[0]
        if nb == 0:
            return
        mb = X.mean(axis=0)
        Sb = ((X - mb) ** 2).sum(axis=0)
        if self.n == 0:
            self.n, self.mean, self.S = nb, mb, Sb
            return
        d = mb - self.mean
        tot = self.n + nb
        self.S = self.S + Sb + d * d * (self.n * nb / tot)
        self.mean = self.mean + d * (nb / tot)
        self.n = tot

    @property
    def var(self):
        return self.S / (self.n - 1) if self.n > 1 else np.square(self.mean)

    @property
    def std(self):
        return np.sqrt(self.var)


def zfilter_apply(x, mean, std, clip):
    y = (np.asarray(x, float) - mean) / (std + 1e-8)
    return np.clip(y, -clip, clip) if clip else y


class ZFilterOracle:
    def __init__(self, dim, clip=10.0):
        self.rs = RunningStatOracle(dim)
        self.clip = clip

    def __call__(self, x, update=True):
        if update:
            self.rs.push(x)
        return zfilter_apply(x, self.rs.mean, self.rs.std, self.clip)
