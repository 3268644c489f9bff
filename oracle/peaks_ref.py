"""CPU restatement of ``inference.find_N_peaks`` (reference ``inference.py:21-29``).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  ``maximum_filter(size=r,
mode='wrap')`` is restated explicitly: truly periodic, window for output i covers
inputs [i - r//2, i - r//2 + r - 1] (SURVEY.md section 4 KAT 3).  Pinned against the
reference function + SciPy by ``oracle/gen_golden.py``.
"""
import numpy as np


def maximum_filter_wrap(signal, r):
    n = signal.shape[0]
    lo = -(r // 2)
    idx = (np.arange(n)[:, None] + np.arange(lo, lo + r)[None, :]) % n
    return signal[idx].max(axis=1)


def find_N_peaks(signal, r=29, min_v=0.05, N=None):
    max_v = maximum_filter_wrap(signal, r)
    pk_loc = np.where(max_v == signal)[0]
    pk_loc = pk_loc[signal[pk_loc] > min_v]
    if N is not None:
        order = np.argsort(-signal[pk_loc])
        pk_loc = pk_loc[order[:N]]
        pk_loc = pk_loc[np.argsort(pk_loc)]
    return pk_loc, signal[pk_loc]


def sigmoid_f32(x):
    """torch.sigmoid on float32 as used at inference.py:80 (restated: 1/(1+exp(-x)) in f32)."""
    x = np.asarray(x, dtype=np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-x))).astype(np.float32)
