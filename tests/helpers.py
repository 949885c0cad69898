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


def odd_topologies():
    """Hand-made small topologies with the shapes the case files do not have: parallel lines in both directions, a hub with a
    high in-degree, buses without any line (they take the highest ids: the reference gathers per-line arrays with bus ids,
    GNS/main.py:41, so every CONNECTED bus id must be <= E), one-directional chains (buses with no incoming / no outgoing line),
    a generator on every bus, one generator only, two generators on one bus.  name -> (N, f_bus, t_bus, gen_bus), 1-based."""
    rng = np.random.default_rng(77)
    out = {}
    out['pair'] = (2, [1, 2], [2, 1], [1])
    out['hub_all_gens'] = (5, [2, 3, 4, 5, 2, 3, 4, 5], [1, 1, 1, 1, 1, 1, 1, 1], [1, 2, 3, 4, 5])
    out['ring_isolated_dupgen'] = (12, [1, 2, 3, 4, 5, 6, 7, 8, 9, 1, 3, 9], [2, 3, 4, 5, 6, 7, 8, 9, 1, 5, 7, 2], [2, 2, 11, 6])
    out['chain_one_way'] = (17, list(range(1, 17)) + [1], list(range(2, 18)) + [17], [17, 4])
    f = list(rng.integers(2, 22, size=40)); out['hub_indegree_40'] = (21, [int(a) for a in f], [1] * 40, [1, 9, 15])
    n, e = 40, 70
    f, t_ = [], []
    while len(f) < e:
        a, b = (int(z) for z in rng.integers(1, n + 1, size=2))
        if a != b:
            f.append(a); t_.append(b)
    out['random_40_one_gen'] = (n, f, t_, [23])
    n, e = 33, 64
    f, t_ = [], []
    while len(f) < e:
        a, b = (int(z) for z in rng.integers(1, n + 1, size=2))
        if a != b:
            f.append(a); t_.append(b)
    out['random_33_many_gens'] = (n, f, t_, sorted(int(z) for z in rng.choice(n, size=20, replace=False) + 1))
    return out


def grids_on_topology(n, f_bus, t_bus, gen_bus, batch, seed):
    """(buses[B,N,6], lines[B,E,7], generators[B,Gn,7]) float32 CPU tensors in the layout of GNS/utils.py:17-41 on a given topology,
    continuous columns drawn in the ranges of opf_graph_neural_solver_amd.synth (per-unit values of a 100 MVA base)."""
    g = torch.Generator().manual_seed(int(seed))
    e, gn = len(f_bus), len(gen_bus)
    u = lambda lo, hi, *s: torch.rand(s, generator=g) * (hi - lo) + lo
    buses = torch.zeros(batch, n, 6)
    buses[:, :, 0] = torch.arange(1, n + 1, dtype=torch.float32)
    buses[:, :, 1] = 1.0
    buses[:, :, 2] = u(0.0, 0.9, batch, n) * (torch.rand(batch, n, generator=g) < 0.8)
    buses[:, :, 3] = u(-0.05, 0.25, batch, n)
    buses[:, :, 4], buses[:, :, 5] = 0.01, -0.01
    lines = torch.zeros(batch, e, 7)
    lines[:, :, 0] = torch.tensor(f_bus, dtype=torch.float32)
    lines[:, :, 1] = torch.tensor(t_bus, dtype=torch.float32)
    lines[:, :, 2] = u(0.0, 0.25, batch, e) * (torch.rand(batch, e, generator=g) < 0.85)
    lines[:, :, 3] = u(0.04, 0.6, batch, e)
    lines[:, :, 4] = u(0.0, 0.06, batch, e)
    lines[:, :, 5] = u(0.8, 1.2, batch, e)
    lines[:, :, 6] = u(-0.2, 0.2, batch, e) * (np.pi / 180.0)
    gens = torch.zeros(batch, gn, 7)
    gens[:, :, 0] = torch.tensor(gen_bus, dtype=torch.float32)
    gens[:, :, 1] = u(1.0, 3.5, batch, gn)
    gens[:, :, 2] = 0.0
    gens[:, :, 3] = gens[:, :, 1] * u(0.25, 0.75, batch, gn)
    gens[:, :, 4] = u(0.95, 1.1, batch, gn)
    gens[:, :, 5] = u(-0.2, 0.5, batch, gn)
    gens[:, :, 6] = gens[:, :, 3]
    return buses, lines, gens
