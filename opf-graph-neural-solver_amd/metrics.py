"""Batched evaluation metrics of the reference's evaluation script (``GNS/evaluate.py:15-18,101-129``), as torch ops on
whatever device the solver outputs live on (the reference computes them per grid with numpy)."""
from __future__ import annotations

import torch


def active_line_flow(v, theta, x, src, dst):
    """``1/x * (V[src] V[dst] sin(theta[src] - theta[dst]))`` per line (``evaluate.py:15-18``).

    ``v, theta``: ``[B,N]`` (or ``[N]``); ``x``: ``[B,E]`` (or ``[E]``); ``src, dst``: 1-based bus ids ``[E]`` (or ``[B,E]``)."""
    s = (src.long() - 1).expand(x.shape) if src.dim() < x.dim() else src.long() - 1
    d = (dst.long() - 1).expand(x.shape) if dst.dim() < x.dim() else dst.long() - 1
    vs, vd = torch.gather(v, -1, s), torch.gather(v, -1, d)
    ts, td = torch.gather(theta, -1, s), torch.gather(theta, -1, d)
    return 1.0 / x * (vs * vd * torch.sin(ts - td))


def solution_errors(v, theta, v_ref, theta_ref):
    """Mean / std of the absolute differences and the percentage errors against a reference solution
    (Newton-Raphson in the reference, ``evaluate.py:101-115``; population std like ``np.std``)."""
    dv, dth = (v - v_ref).abs(), (theta - theta_ref).abs()
    return {
        'v_abs_mean': dv.mean(), 'v_abs_std': dv.std(unbiased=False),
        'theta_abs_mean': dth.mean(), 'theta_abs_std': dth.std(unbiased=False),
        'v_pct_error': ((v - v_ref) / v_ref).abs() * 100, 'theta_pct_error': ((theta - theta_ref) / theta_ref).abs() * 100,
        'v_diff_per_bus_mean': (v_ref - v).mean(dim=0), 'v_diff_per_bus_std': (v_ref - v).std(dim=0, unbiased=False),
        'theta_diff_per_bus_mean': dth.mean(dim=0), 'theta_diff_per_bus_std': dth.std(dim=0, unbiased=False),
    }


def line_flow_percentiles(flow, flow_ref):
    """20th / 50th / 80th percentile of the lower half of ``|flow_ref - flow| / |flow_ref| * 100`` (``evaluate.py:117-125``)."""
    pct = ((flow_ref - flow) / flow_ref).abs() * 100
    low = torch.sort(pct.reshape(-1)).values[: pct.numel() // 2]
    q = torch.quantile(low.double(), torch.tensor([0.2, 0.5, 0.8], dtype=torch.float64, device=low.device))   # linear interpolation = np.percentile
    return {'p20': q[0], 'median': q[1], 'p80': q[2]}
