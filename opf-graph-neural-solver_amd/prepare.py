"""Batched, device-side input producer: raw PYPOWER arrays -> the tensors the GNS hot path consumes.

Replaces the reference's per-file ``prepare_grid`` + ``load_all_grids`` loop (``GNS/utils.py:17-41,44-68``): the same
column picks and normalisation, vectorised over the batch on whatever device the inputs live on (no Python loop,
no pickle).  Input layout is PYPOWER's case format v2: ``bus[B,N,13]``, ``branch[B,E,13]``, ``gen[B,Gn,21]``.
"""
from __future__ import annotations

import math

import torch


def prepare_grids(bus: torch.Tensor, branch: torch.Tensor, gen: torch.Tensor, base_mva=100.0):
    """Return ``(buses[B,N,6], lines[B,E,7], generators[B,Gn,7])`` as float32 (2-D inputs give 2-D outputs).

    * buses: columns 0-5 of ``bus``; Gs := 1, Bs := -1 (``utils.py:25-26``); Pd, Qd, Gs, Bs divided by baseMVA (``:30``)
    * lines: columns (0,1,2,3,4,8,9) of ``branch``; tau == 0 -> 1 (``:33``); shift degrees -> radians (``:35``)
    * generators: columns (0,8,9,1,5,2) of ``gen`` + a copy of Pg (``:37-38``); Pmax, Pmin, Pg_set, qg, Pg divided by baseMVA (``:40``)
    ``base_mva`` may be a number or a ``[B]`` tensor.
    """
    single = bus.dim() == 2
    if single:
        bus, branch, gen = bus.unsqueeze(0), branch.unsqueeze(0), gen.unsqueeze(0)
    if bus.shape[-1] < 6 or branch.shape[-1] < 10 or gen.shape[-1] < 10:
        raise ValueError('expected PYPOWER case arrays: bus[...,>=6], branch[...,>=10], gen[...,>=10]')
    f32 = torch.float32
    base = torch.as_tensor(base_mva, dtype=f32, device=bus.device).reshape(-1, 1)          # [1,1] or [B,1]
    b = bus[..., 0:6].to(f32).clone()
    b[..., 4] = 1.0
    b[..., 5] = -1.0
    b[..., 2:6] = b[..., 2:6] / base.unsqueeze(-1)
    br = branch.to(f32)
    lines = br[..., [0, 1, 2, 3, 4, 8, 9]].clone()
    lines[..., 5] = torch.where(lines[..., 5] == 0, torch.ones_like(lines[..., 5]), lines[..., 5])
    lines[..., 6] = lines[..., 6] * (math.pi / 180.0)
    g = gen.to(f32)
    gens = torch.cat((g[..., [0, 8, 9, 1, 5, 2]], g[..., 1:2]), dim=-1).clone()
    gens[..., [1, 2, 3, 5, 6]] = gens[..., [1, 2, 3, 5, 6]] / base.unsqueeze(-1)
    if single:
        return b[0], lines[0], gens[0]
    return b, lines, gens
