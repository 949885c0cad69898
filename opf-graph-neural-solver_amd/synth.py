"""Synthetic power grids in the layout the GNS hot path consumes.

No case30/118/300 data exists offline and the reference's shipped case14 pickles cannot be opened
with a non-executing loader, so every test and benchmark input is generated here.  The generator
mirrors, vectorised over the batch and on any torch device,
  * the perturbation ranges of the reference's ``GNS/augment_grids.py:12-53`` (r, x, b x U[0.9,1.1];
    tau ~ U[0.8,1.2]; shift ~ U[-0.2,0.2] degrees; vg x U[0.95,1.05]; Pg ~ U[.25,.75] of the unit's
    range; Pd x U[.5,1.5] then rescaled so that sum(Pd) == sum(Pg); Qd x U[.5,1.5]), and
  * the column selection / per-unit normalisation of ``GNS/utils.py:17-41`` (``prepare_grid``):
    buses [N,6] = (bus_i, type, Pd, Qd, Gs=1/baseMVA, Bs=-1/baseMVA), lines [E,7] =
    (f_bus, t_bus, r, x, b, tau, shift[rad]), generators [Gn,7] = (bus_i, Pmax, Pmin, Pg_set, vg, qg, Pg).
Topology is one fixed graph per case and is shared by the whole batch, exactly like the reference's
data (``augment_grids.py`` perturbs continuous columns only).  Sizes come from ``GNS/utils.py:45-56``.
case14 uses the public IEEE 14-bus test system as its base; the larger cases are case-SHAPED random
connected graphs with contiguous bus ids (the real case300 has non-contiguous ids, which the reference's
``bus_id - 1`` indexing cannot address).
"""
from __future__ import annotations

import math

import numpy as np
import torch

# 200 is not a reference case: a case-shaped synthetic grid between case118 and case300 (exercises the 64-160 KB LDS plane of the
# lane-per-grid forward, which case118 (59 KB) and case300 (HBM path) both miss)
CASE_SHAPES = {14: (14, 20, 5), 30: (30, 41, 6), 118: (118, 186, 54), 200: (200, 290, 45), 300: (300, 411, 69)}
BASE_MVA = 100.0

# IEEE 14-bus test system (public data): from, to, r, x, b
_IEEE14_BRANCH = [
    (1, 2, 0.01938, 0.05917, 0.0528), (1, 5, 0.05403, 0.22304, 0.0492), (2, 3, 0.04699, 0.19797, 0.0438),
    (2, 4, 0.05811, 0.17632, 0.0340), (2, 5, 0.05695, 0.17388, 0.0346), (3, 4, 0.06701, 0.17103, 0.0128),
    (4, 5, 0.01335, 0.04211, 0.0), (4, 7, 0.0, 0.20912, 0.0), (4, 9, 0.0, 0.55618, 0.0),
    (5, 6, 0.0, 0.25202, 0.0), (6, 11, 0.09498, 0.19890, 0.0), (6, 12, 0.12291, 0.25581, 0.0),
    (6, 13, 0.06615, 0.13027, 0.0), (7, 8, 0.0, 0.17615, 0.0), (7, 9, 0.0, 0.11001, 0.0),
    (9, 10, 0.03181, 0.08450, 0.0), (9, 14, 0.12711, 0.27038, 0.0), (10, 11, 0.08205, 0.19207, 0.0),
    (12, 13, 0.22092, 0.19988, 0.0), (13, 14, 0.17093, 0.34802, 0.0)]
_IEEE14_PD = [0.0, 21.7, 94.2, 47.8, 7.6, 11.2, 0.0, 0.0, 29.5, 9.0, 3.5, 6.1, 13.5, 14.9]
_IEEE14_QD = [0.0, 12.7, 19.0, -3.9, 1.6, 7.5, 0.0, 0.0, 16.6, 5.8, 1.8, 1.6, 5.8, 5.0]
# bus, Pg, Qg, Vg, Pmax, Pmin
_IEEE14_GEN = [(1, 232.4, -16.9, 1.06, 332.4, 0.0), (2, 40.0, 42.4, 1.045, 140.0, 0.0), (3, 0.0, 23.4, 1.01, 100.0, 0.0),
               (6, 0.0, 12.2, 1.07, 100.0, 0.0), (8, 0.0, 17.4, 1.09, 100.0, 0.0)]


def base_case(case_nr: int) -> dict:
    """Un-augmented base quantities (float64 numpy, MW / MVAr / p.u. like a PYPOWER case)."""
    if case_nr not in CASE_SHAPES:
        raise ValueError(f'unknown case {case_nr}; known: {sorted(CASE_SHAPES)}')
    n, e, gn = CASE_SHAPES[case_nr]
    if case_nr == 14:
        br = np.array(_IEEE14_BRANCH, dtype=np.float64)
        gen = np.array(_IEEE14_GEN, dtype=np.float64)
        return dict(f_bus=br[:, 0].astype(np.int64), t_bus=br[:, 1].astype(np.int64), r=br[:, 2], x=br[:, 3], b=br[:, 4],
                    Pd=np.array(_IEEE14_PD), Qd=np.array(_IEEE14_QD), gen_bus=gen[:, 0].astype(np.int64),
                    Pg=gen[:, 1], Qg=gen[:, 2], Vg=gen[:, 3], Pmax=gen[:, 4], Pmin=gen[:, 5])
    rng = np.random.default_rng(1000 + case_nr)
    # random spanning tree over a random bus order, then extra lines (parallel lines allowed)
    order = rng.permutation(n) + 1
    f, t = [], []
    for i in range(1, n):
        a, b_ = int(order[rng.integers(0, i)]), int(order[i])
        if rng.random() < 0.5:
            a, b_ = b_, a
        f.append(a); t.append(b_)
    while len(f) < e:
        a, b_ = (int(z) for z in rng.integers(1, n + 1, size=2))
        if a != b_:
            f.append(a); t.append(b_)
    perm = rng.permutation(e)
    f_bus, t_bus = np.array(f, dtype=np.int64)[perm], np.array(t, dtype=np.int64)[perm]
    gen_bus = np.sort(rng.choice(n, size=gn, replace=False) + 1).astype(np.int64)
    pmax = rng.uniform(100.0, 350.0, size=gn)
    load = rng.uniform(0.0, 1.0, size=n) * (rng.random(n) < 0.8)
    return dict(f_bus=f_bus, t_bus=t_bus, r=rng.uniform(0.0, 0.25, size=e) * (rng.random(e) < 0.85),
                x=rng.uniform(0.04, 0.6, size=e), b=rng.uniform(0.0, 0.06, size=e) * (rng.random(e) < 0.7),
                Pd=load * pmax.sum() * 0.5 / max(load.sum(), 1e-9), Qd=rng.uniform(-5.0, 25.0, size=n),
                gen_bus=gen_bus, Pg=pmax * 0.5, Qg=rng.uniform(-20.0, 50.0, size=gn), Vg=rng.uniform(1.0, 1.09, size=gn),
                Pmax=pmax, Pmin=np.zeros(gn))


def case_topology(case_nr: int):
    """(f_bus[E], t_bus[E], gen_bus[Gn]) as 1-based int64 numpy arrays."""
    c = base_case(case_nr)
    return c['f_bus'], c['t_bus'], c['gen_bus']


def _hash32(x):
    """lowbias32 integer hash on int64 tensors holding 32-bit values (same bits on CPU and GPU)."""
    m = 0xFFFFFFFF
    x = (x ^ (x >> 16)) & m
    x = (x * 0x7FEB352D) & m
    x = (x ^ (x >> 15)) & m
    x = (x * 0x846CA68B) & m
    return (x ^ (x >> 16)) & m


def counter_uniform(seed: int, stream: int, first_index: int, batch: int, width: int, device) -> torch.Tensor:
    """Counter-based U[0,1) of shape [batch, width]: the value for (grid g, element j) is a hash of
    (seed, stream, first_index + g, j) and of nothing else - no generator state - so any shard of a batch, on any
    device, reproduces the rows of the unsharded batch bit for bit (every GPU count sees the same grids)."""
    g = torch.arange(first_index, first_index + batch, dtype=torch.int64, device=device).unsqueeze(1)
    j = torch.arange(width, dtype=torch.int64, device=device).unsqueeze(0)
    key = _hash32(torch.full((1, 1), (int(seed) * 0x9E3779B1 + int(stream) * 0x85EBCA6B + 0x1234567) & 0xFFFFFFFF,
                             dtype=torch.int64, device=device))
    h = _hash32((key + (g & 0xFFFFFFFF)) & 0xFFFFFFFF)
    h = _hash32((h ^ ((g >> 32) & 0xFFFFFFFF)) & 0xFFFFFFFF)
    h = _hash32((h + j * 0x27D4EB2F) & 0xFFFFFFFF)
    return (h >> 8).to(torch.float32) * (1.0 / 16777216.0)          # 24 random bits: exact in fp32


def synth_grids(case_nr: int, batch: int, seed: int = 0, device='cpu', load_scale: float = 1.0, augment: bool = True,
                first_index: int = 0):
    """Return (buses[B,N,6], lines[B,E,7], generators[B,Gn,7]) float32 on ``device`` for the grids
    ``first_index .. first_index + batch - 1`` of the (seed, case) data set.

    Every random draw is a counter-based hash of (seed, grid index, column, element): ``synth_grids(c, 8, s)[..][2:5]``
    equals ``synth_grids(c, 3, s, first_index=2)`` bit for bit, on the CPU and on the device.
    ``load_scale`` multiplies Pd after balancing; < ~0.6 drives the lambda < 0.5 branch of
    ``global_active_compensation`` (GNS/main.py:48,54), which balanced random-weight grids never reach.
    """
    c = base_case(case_nr)
    n, e, gn = CASE_SHAPES[case_nr]
    dev = torch.device(device)
    f32 = dict(dtype=torch.float32, device=dev)
    stream = [0]

    def tens(a):
        return torch.as_tensor(np.asarray(a, dtype=np.float32), device=dev)

    def uni(lo, hi, *shape):
        stream[0] += 1
        assert shape[0] == batch and len(shape) == 2
        return counter_uniform(seed, stream[0], first_index, batch, shape[1], dev) * (hi - lo) + lo

    def rowsum(t):
        # fixed left-to-right order in float64: the same bits whatever the batch size or the device's reduction strategy
        acc = torch.zeros(t.shape[0], dtype=torch.float64, device=dev)
        for j in range(t.shape[1]):
            acc = acc + t[:, j].double()
        return acc.unsqueeze(1)

    one = lambda *shape: torch.ones(shape, **f32)
    amp = (lambda lo, hi, *s: uni(lo, hi, *s)) if augment else (lambda lo, hi, *s: one(*s))
    r = tens(c['r']) * amp(0.9, 1.1, batch, e)
    x = tens(c['x']) * amp(0.9, 1.1, batch, e)
    b = tens(c['b']) * amp(0.9, 1.1, batch, e)
    tau = uni(0.8, 1.2, batch, e) if augment else one(batch, e)
    shift_deg = uni(-0.2, 0.2, batch, e) if augment else torch.zeros(batch, e, **f32)
    vg = tens(c['Vg']) * amp(0.95, 1.05, batch, gn)
    pmax, pmin = tens(c['Pmax']).expand(batch, gn), tens(c['Pmin']).expand(batch, gn)
    span = pmax - pmin
    pg = (pmin + span * 0.25) + uni(0.0, 1.0, batch, gn) * (span * 0.75 - (pmin + span * 0.25)) if augment \
        else tens(c['Pg']).expand(batch, gn)
    pd = tens(c['Pd']) * amp(0.5, 1.5, batch, n)
    ratio = (rowsum(pg) / rowsum(pd).clamp_min(1e-9)).to(torch.float32)        # sum(Pd) == sum(Pg) (augment_grids.py:51)
    pd = pd * ratio * load_scale
    qd = tens(c['Qd']) * amp(0.5, 1.5, batch, n)
    qg = tens(c['Qg']).expand(batch, gn)

    buses = torch.zeros(batch, n, 6, **f32)
    buses[:, :, 0] = torch.arange(1, n + 1, **f32)
    buses[:, :, 1] = 1.0
    buses[:, :, 2] = pd / BASE_MVA
    buses[:, :, 3] = qd / BASE_MVA
    buses[:, :, 4] = 1.0 / BASE_MVA
    buses[:, :, 5] = -1.0 / BASE_MVA
    lines = torch.zeros(batch, e, 7, **f32)
    lines[:, :, 0] = tens(c['f_bus'])
    lines[:, :, 1] = tens(c['t_bus'])
    lines[:, :, 2], lines[:, :, 3], lines[:, :, 4] = r, x, b
    lines[:, :, 5] = torch.where(tau == 0, torch.ones_like(tau), tau)
    lines[:, :, 6] = shift_deg * (math.pi / 180.0)
    gens = torch.zeros(batch, gn, 7, **f32)
    gens[:, :, 0] = tens(c['gen_bus'])
    gens[:, :, 1] = pmax / BASE_MVA
    gens[:, :, 2] = pmin / BASE_MVA
    gens[:, :, 3] = pg / BASE_MVA
    gens[:, :, 4] = vg
    gens[:, :, 5] = qg / BASE_MVA
    gens[:, :, 6] = pg / BASE_MVA
    return buses, lines, gens


def raw_case_arrays(case_nr: int, batch: int, seed: int = 0, zero_tau_fraction: float = 0.3):
    """Synthetic PYPOWER case-format arrays ``bus[B,N,13]``, ``branch[B,E,13]``, ``gen[B,Gn,21]`` (float64, MW / degrees)
    for the input producer ``prepare.prepare_grids``.  A fraction of the lines carries ratio 0 (= "no transformer" in
    MATPOWER), which ``GNS/utils.py:33`` maps to 1."""
    c = base_case(case_nr)
    n, e, gn = CASE_SHAPES[case_nr]
    g = torch.Generator().manual_seed(int(seed))
    f64 = torch.float64
    u = lambda lo, hi, *s: torch.rand(s, generator=g, dtype=f64) * (hi - lo) + lo
    T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64))
    bus = torch.zeros(batch, n, 13, dtype=f64)
    bus[:, :, 0] = torch.arange(1, n + 1, dtype=f64)
    bus[:, :, 1] = 1
    bus[:, :, 2] = T(c['Pd']) * u(0.5, 1.5, batch, n)
    bus[:, :, 3] = T(c['Qd']) * u(0.5, 1.5, batch, n)
    bus[:, :, 4] = u(0.0, 5.0, batch, n)             # Gs, Bs of the raw case are overwritten by prepare_grid
    bus[:, :, 5] = u(0.0, 20.0, batch, n)
    bus[:, :, 7], bus[:, :, 9], bus[:, :, 11], bus[:, :, 12] = 1.0, 135.0, 1.06, 0.94
    br = torch.zeros(batch, e, 13, dtype=f64)
    br[:, :, 0], br[:, :, 1] = T(c['f_bus']), T(c['t_bus'])
    br[:, :, 2] = T(c['r']) * u(0.9, 1.1, batch, e)
    br[:, :, 3] = T(c['x']) * u(0.9, 1.1, batch, e)
    br[:, :, 4] = T(c['b']) * u(0.9, 1.1, batch, e)
    tau = u(0.8, 1.2, batch, e)
    br[:, :, 8] = torch.where(torch.rand(batch, e, generator=g) < zero_tau_fraction, torch.zeros_like(tau), tau)
    br[:, :, 9] = u(-0.2, 0.2, batch, e)
    br[:, :, 10], br[:, :, 11], br[:, :, 12] = 1, -360, 360
    ge = torch.zeros(batch, gn, 21, dtype=f64)
    ge[:, :, 0] = T(c['gen_bus'])
    ge[:, :, 8], ge[:, :, 9] = T(c['Pmax']), T(c['Pmin'])
    ge[:, :, 1] = T(c['Pmin']) + (T(c['Pmax']) - T(c['Pmin'])) * u(0.25, 0.75, batch, gn)
    ge[:, :, 2] = T(c['Qg'])
    ge[:, :, 3], ge[:, :, 4] = 100.0, -50.0
    ge[:, :, 5] = T(c['Vg']) * u(0.95, 1.05, batch, gn)
    ge[:, :, 6], ge[:, :, 7] = BASE_MVA, 1
    return bus, br, ge
