"""CPU oracle for the GNS K-step hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement, op by op on ONE grid at a time, of the
algorithm in the reference's ``GNS/main.py`` (LeonOrou/OPF-Graph-Neural-Solver).
It is the checker for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product package never does.

Pinning status: the reference ships no tests, no golden vectors and no checkpoints
(SURVEY.md section 4), so parity is unpinned by the reference's own fixtures.  This
restatement is pinned instead against outputs of the reference's own ``main.GNS``
executed in the build container (``oracle/make_goldens.py`` -> ``tests/golden/*.npz``,
checked by ``tests/test_oracle_vs_golden.py``).  ``torch_scatter`` (third party, not
vendored, version not pinned by the reference) is restated from its published
``scatter_add`` semantics: ``out.scatter_add_(dim, broadcast(index), src)``.

What is restated (reference file:line):
  * ``learning_block``               GNS/main.py:17-31   (Linear-LeakyReLU(0.01)-Linear-LeakyReLU-Linear)
  * ``active_compensation``          GNS/main.py:34-78   (global_active_compensation)
  * ``power_imbalance``              GNS/main.py:80-104  (local_power_imbalance)
  * ``gns_forward``                  GNS/main.py:140-202 (GNS.forward)
  * ``param_spec`` / ``init_params`` GNS/main.py:107-138 (state_dict key space and shapes)
  * column maps                      GNS/utils.py:4-13   (get_BLG)

Three behaviours of the reference are reproduced on purpose:
per-line arrays are gathered with BUS indices, phi messages use the latent of the
destination bus and are summed at that same bus, and with a single phi network only
column 0 of the message sum is non-zero.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

# GNS/utils.py:4-13
BUS_COL = {'bus_i': 0, 'type': 1, 'Pd': 2, 'Qd': 3, 'Gs': 4, 'Bs': 5}
LINE_COL = {'f_bus': 0, 't_bus': 1, 'r': 2, 'x': 3, 'b': 4, 'tau': 5, 'theta': 6}
GEN_COL = {'bus_i': 0, 'Pmax': 1, 'Pmin': 2, 'Pg_set': 3, 'vg': 4, 'qg': 5, 'Pg': 6}

LEAKY_SLOPE = 0.01  # torch.nn.LeakyReLU default, GNS/main.py:23


def param_spec(latent_dim: int, hidden_dim: int, K: int, multiple_phi: bool):
    """(name, shape) pairs in the registration order of GNS/main.py:113-134."""
    d, h = latent_dim, hidden_dim
    families = (['phi_v', 'phi_theta', 'phi_m'] if multiple_phi else ['phi']) + ['L_theta', 'L_v', 'L_m']
    spec = []
    for fam in families:
        if fam.startswith('phi'):
            din, dout = d + 5, (d if multiple_phi else 1)
        else:
            din, dout = 4 + 2 * d, (d if fam == 'L_m' else 1)
        for k in range(K):
            for lin, (o, i) in (('linear1', (h, din)), ('linear2', (h, h)), ('linear4', (dout, h))):
                spec.append((f'{fam}.{k}.{lin}.weight', (o, i)))
                spec.append((f'{fam}.{k}.{lin}.bias', (o,)))
    return spec


def init_params(latent_dim, hidden_dim, K, multiple_phi, seed=0, dtype=torch.float32):
    """nn.Linear's default init law (uniform +-1/sqrt(fan_in)) drawn from a private generator."""
    g = torch.Generator().manual_seed(seed)
    params = OrderedDict()
    for name, shape in param_spec(latent_dim, hidden_dim, K, multiple_phi):
        fan_in = shape[1] if len(shape) == 2 else None
        if fan_in is None:  # bias follows its weight: bound 1/sqrt(fan_in of that layer)
            fan_in = params[name.replace('.bias', '.weight')].shape[1]
        bound = 1.0 / math.sqrt(fan_in)
        params[name] = ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
    return params


def flatten_params(params) -> torch.Tensor:
    return torch.cat([p.reshape(-1) for p in params.values()])


def unflatten_params(flat, latent_dim, hidden_dim, K, multiple_phi):
    out, off = OrderedDict(), 0
    for name, shape in param_spec(latent_dim, hidden_dim, K, multiple_phi):
        n = math.prod(shape)
        out[name] = flat[off:off + n].reshape(shape)
        off += n
    assert off == flat.numel()
    return out


def _bucket_sum(values, index, n):
    """torch_scatter.scatter_add(values, index, out=zeros(n...), dim=0): sequential in index order."""
    if values.dim() == 1:
        return torch.zeros(n, dtype=values.dtype).index_add(0, index, values)
    out = torch.zeros((n, values.shape[1]), dtype=values.dtype)
    return out.index_add(0, index, values)


def learning_block(x, params, prefix):
    """GNS/main.py:25-31."""
    lr = torch.nn.functional.leaky_relu
    a = lr(torch.addmm(params[prefix + '.linear1.bias'], x, params[prefix + '.linear1.weight'].t()), LEAKY_SLOPE)
    a = lr(torch.addmm(params[prefix + '.linear2.bias'], a, params[prefix + '.linear2.weight'].t()), LEAKY_SLOPE)
    return torch.addmm(params[prefix + '.linear4.bias'], a, params[prefix + '.linear4.weight'].t())


def _line_terms(v, theta, lines, src, dst):
    """Shared per-line quantities of GNS/main.py:38-41,66-72,87-92,98-99.

    The reference indexes the per-LINE arrays y, tau, shift, b and delta with BUS ids
    (``y_ij[src]`` is the admittance of line number ``src[e]``); ``at_s``/``at_t`` below are
    those line numbers.
    """
    r, x = lines[:, LINE_COL['r']], lines[:, LINE_COL['x']]
    y = 1 / torch.sqrt(r.pow(2) + x.pow(2))
    tau, shift, b = lines[:, LINE_COL['tau']], lines[:, LINE_COL['theta']], lines[:, LINE_COL['b']]
    d_ij = theta[src] - theta[dst]
    d_ji = theta[dst] - theta[src]
    at_s, at_t = src, dst
    return dict(
        vs=v[src], vt=v[dst], ths=theta[src], tht=theta[dst],
        y_s=y[at_s], tau_s=tau[at_s], sh_s=shift[at_s], b_s=b[at_s], dl_s=d_ij[at_s],
        y_t=y[at_t], tau_t=tau[at_t], sh_t=shift[at_t], b_t=b[at_t], dl_t=d_ji[at_t])


def active_compensation(v, theta, buses, lines, gens, src, dst):
    """GNS/main.py:34-78 -> (Pg_new[Gn], qg_new[N], lambda)."""
    n = buses.shape[0]
    q = _line_terms(v, theta, lines, src, dst)
    ang = q['ths'] - q['tht'] - q['dl_s'] - q['sh_s']
    # :41  note v_s / tau^2 (not (v_s/tau)^2) in the middle term
    joule_e = torch.abs(
        q['vs'] * q['vt'] * q['y_s'] / q['tau_s']
        * (torch.sin(ang) + torch.sin(q['tht'] - q['ths'] - q['dl_s'] + q['sh_s']))
        + (q['vs'] / q['tau_s'].pow(2)) * q['y_s'] * torch.sin(q['dl_s'])
        + q['vt'].pow(2) * q['y_s'] * torch.sin(q['dl_s']))
    p_joule = torch.sum(_bucket_sum(joule_e, dst, n))                      # :42-43
    p_global = torch.sum(buses[:, BUS_COL['Pd']]) + torch.sum(v.pow(2) * buses[:, BUS_COL['Gs']]) + p_joule  # :45
    pset, pmin, pmax = gens[:, GEN_COL['Pg_set']], gens[:, GEN_COL['Pmin']], gens[:, GEN_COL['Pmax']]
    if p_global < pset.sum():                                               # :47-51
        lam = (p_global - pmin.sum()) / (2 * (pset.sum() - pmin.sum()))
    else:
        lam = (p_global - 2 * pset.sum() + pmax.sum()) / (2 * (pmax.sum() - pset.sum()))
    if lam < 0.5:                                                           # :53-57
        pg_new = pmin + 2 * (pset - pmin) * lam
    else:
        pg_new = 2 * pset - pmax + 2 * (pmax - pset) * lam
    q_start = buses[:, BUS_COL['Qd']] - buses[:, BUS_COL['Bs']] * v.pow(2)  # :64
    m_from = (-q['vs'] * q['vt'] * q['y_s'] / q['tau_s'] * torch.cos(ang)
              + (q['vs'] / q['tau_s']).pow(2) * (q['y_s'] * torch.cos(q['dl_s']) - q['b_s'] / 2))      # :68-69
    m_to = (-q['vt'] * q['vs'] * q['y_t'] / q['tau_t'] * torch.cos(q['tht'] - q['ths'] - q['dl_t'] - q['sh_t'])
            + q['vt'].pow(2) * (q['y_t'] * torch.sin(q['dl_t']) - q['b_t'] / 2))                     # :70-72
    qg_new = q_start - _bucket_sum(m_from, dst, n) - _bucket_sum(m_to, src, n)                           # :74-76
    return pg_new, qg_new, lam


def power_imbalance(v, theta, buses, lines, gens, pg_k, qg_k, src, dst, gen_bus):
    """GNS/main.py:80-104 -> (delta_p[N], delta_q[N])."""
    n = buses.shape[0]
    q = _line_terms(v, theta, lines, src, dst)
    dp0 = _bucket_sum(pg_k, gen_bus, n) - buses[:, BUS_COL['Pd']] - buses[:, BUS_COL['Gs']] * v.pow(2)   # :81-82
    dq0 = qg_k - buses[:, BUS_COL['Qd']] + buses[:, BUS_COL['Bs']] * v.pow(2)                           # :83
    ang_f = q['ths'] - q['tht'] - q['dl_s'] - q['sh_s']
    ang_t = q['tht'] - q['ths'] - q['dl_t'] - q['sh_t']
    p_from = (q['vs'] * q['vt'] * q['y_s'] / q['tau_s'] * torch.sin(ang_f)
              + (q['vs'] / q['tau_s']).pow(2) * q['y_s'] * torch.sin(q['dl_s']))                       # :91
    p_to = (q['vt'] * q['vs'] * q['y_t'] / q['tau_t'] * torch.sin(ang_t)
            + q['vt'].pow(2) * q['y_t'] * torch.sin(q['dl_t']))                                        # :92
    dp = dp0 + _bucket_sum(p_from, dst, n) + _bucket_sum(p_to, src, n)                                   # :94-96
    q_from = (-q['vs'] * q['vt'] * q['y_s'] / q['tau_s'] * torch.cos(ang_f)
              + (q['vs'] / q['tau_s']).pow(2) * (q['y_s'] * torch.cos(q['dl_s']) - q['b_s'] / 2))      # :98
    q_to = (-q['vt'] * q['vs'] * q['y_t'] / q['tau_t'] * torch.cos(ang_t)
            + q['vt'].pow(2) * (q['y_t'] * torch.sin(q['dl_t']) - q['b_t'] / 2))                       # :99
    dq = dq0 + _bucket_sum(q_from, dst, n) + _bucket_sum(q_to, src, n)                                   # :101-103
    return dp, dq


def gns_forward(params, buses, lines, gens, *, latent_dim, K, gamma=0.9, multiple_phi=False, trace=None):
    """GNS/main.py:140-202 for one grid (2-D tensors).  Returns (v, theta, total_loss, last_loss).

    ``trace`` (a list) receives one dict of per-step intermediates per k, for stage-level goldens.
    """
    dt = buses.dtype
    n, d = buses.shape[0], latent_dim
    src = lines[:, LINE_COL['f_bus']].to(torch.int32).long() - 1      # :35
    dst = lines[:, LINE_COL['t_bus']].to(torch.int32).long() - 1      # :36,153
    gen_bus = gens[:, GEN_COL['bus_i']].long() - 1                     # :144
    m = torch.zeros((n, d), dtype=dt)                                  # :141
    theta = torch.zeros(n, dtype=dt)                                   # :142
    v = _bucket_sum(gens[:, GEN_COL['vg']], gen_bus, n)                # :146
    v = torch.where(v == 0, torch.ones_like(v), v)                     # :147
    dp = _bucket_sum(gens[:, GEN_COL['Pg']], gen_bus, n) - buses[:, BUS_COL['Pd']] - buses[:, BUS_COL['Gs']] * v.pow(2)  # :149-150
    dq = _bucket_sum(gens[:, GEN_COL['qg']], gen_bus, n) - buses[:, BUS_COL['Qd']] + buses[:, BUS_COL['Bs']] * v.pow(2)  # :151-152
    free = torch.ones(n, dtype=torch.bool)
    free[gen_bus] = False                                              # :184-185
    feat = lines[:, 2:]                                                # :155 (hard-coded slice)
    total = torch.zeros((), dtype=dt)
    for k in range(K):
        edge_in = torch.cat((m[dst], feat), dim=1)                     # :155
        head = torch.stack((v, theta, dp, dq), dim=1)
        if multiple_phi:                                               # :156-167
            sums = {fam: _bucket_sum(learning_block(edge_in, params, f'phi_{fam}.{k}'), dst, n)
                    for fam in ('v', 'theta', 'm')}
        else:                                                          # :169-171 -- [E,1] into column 0 of [N,d]
            col0 = _bucket_sum(learning_block(edge_in, params, f'phi.{k}')[:, 0], dst, n)
            one = torch.cat((col0.unsqueeze(1), torch.zeros((n, d - 1), dtype=dt)), dim=1)
            sums = {'v': one, 'theta': one, 'm': one}
        upd = {fam: learning_block(torch.cat((head, m, sums[fam]), dim=1), params, f'L_{fam}.{k}')
               for fam in ('theta', 'v', 'm')}                         # :173-180
        theta = theta + upd['theta'][:, 0]                             # :182
        v = torch.where(free, v + upd['v'][:, 0], v)                   # :186
        m = m + upd['m']                                               # :188
        pg_new, qg_new, lam = active_compensation(v, theta, buses, lines, gens, src, dst)        # :190
        dp, dq = power_imbalance(v, theta, buses, lines, gens, pg_new, qg_new, src, dst, gen_bus)  # :192
        step_loss = torch.sum(dp.pow(2) + dq.pow(2)) / n
        total = total + gamma ** (K - k) * step_loss                   # :198
        if trace is not None:
            trace.append(dict(m=m.detach().clone(), v=v.detach().clone(), theta=theta.detach().clone(),
                              dp=dp.detach().clone(), dq=dq.detach().clone(), pg_new=pg_new.detach().clone(),
                              qg_new=qg_new.detach().clone(), lam=torch.as_tensor(lam).detach().clone(),
                              phi_sum_v=sums['v'].detach().clone(), phi_sum_theta=sums['theta'].detach().clone(),
                              phi_sum_m=sums['m'].detach().clone()))
    last = torch.sum(dp.pow(2) + dq.pow(2)) / n                        # :199
    v = torch.where(v < 0, torch.zeros_like(v), v)                     # :201
    return v, theta, total, last


def gns_forward_backward(flat, buses, lines, gens, *, latent_dim, hidden_dim, K, gamma=0.9, multiple_phi=False):
    """Batch-mean loss and its parameter gradient, one grid per forward call as GNS/main.py:279-288.

    ``buses/lines/gens`` are [B,N,6]/[B,E,7]/[B,Gn,7].  Returns (v[B,N], theta[B,N], total[B], last[B], grad_flat)
    where ``grad_flat`` is d(mean_b total_b)/d(flat).
    """
    flat = flat.detach().clone().requires_grad_(True)
    params = unflatten_params(flat, latent_dim, hidden_dim, K, multiple_phi)
    vs, ths, tots, lasts = [], [], [], []
    for b in range(buses.shape[0]):
        v, th, tot, last = gns_forward(params, buses[b], lines[b], gens[b], latent_dim=latent_dim, K=K,
                                       gamma=gamma, multiple_phi=multiple_phi)
        vs.append(v.detach()); ths.append(th.detach()); tots.append(tot); lasts.append(last.detach())
    tot_t = torch.stack(tots)
    tot_t.mean().backward()
    return torch.stack(vs), torch.stack(ths), tot_t.detach(), torch.stack(lasts), flat.grad.detach()
