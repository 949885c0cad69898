"""Shared helpers for the parity tests."""
import glob
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def golden_names():
    """forward/backward goldens (prepare_* are goldens of the input producer, metrics_* of the evaluation metrics, augment_* of the synthetic generator)"""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, '*.npz')))
    return [n for n in names if not n.startswith(('prepare_', 'metrics_', 'augment_'))]


def golden_depth(name):
    import re
    return int(re.search(r'_K(\d+)_', name).group(1))


def shallow_golden_names():
    """goldens with K <= 10: the 1e-5 bar against the reference's fp32 output is meaningful there"""
    return [n for n in golden_names() if golden_depth(n) <= 10]


def deep_golden_names():
    """the reference's own run configurations (K=15 main.py:209-213, K=30 main.py:108): fp32 rounding is amplified"""
    return [n for n in golden_names() if golden_depth(n) > 10]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return {k: z[k] for k in z.files}


def cfg_of(g):
    return dict(latent_dim=int(g['latent_dim']), hidden_dim=int(g['hidden_dim']), K=int(g['K']),
                gamma=float(g['gamma']), multiple_phi=bool(int(g['multiple_phi'])))


def rel_err(a, b):
    """max|a-b| / max|b| - the per-tensor measure of SURVEY section 8c (theta straddles 0)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def assert_close(a, b, rel, abs_floor=1e-6, what=''):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f'{what}: shape {a.shape} vs {b.shape}'
    assert np.all(np.isfinite(a)), f'{what}: non-finite values'
    tol = abs_floor + rel * np.max(np.abs(b))
    worst = float(np.max(np.abs(a - b)))
    assert worst <= tol, f'{what}: max|diff|={worst:.3e} > tol={tol:.3e} (rel_err={rel_err(a, b):.3e})'


def t(x):
    return torch.as_tensor(np.asarray(x))
