"""The CPU oracle (oracle/gns_oracle.py) against vectors produced by the reference's own main.GNS
(oracle/make_goldens.py).  This is what pins the oracle; CPU only."""
import numpy as np
import pytest
import torch

from helpers import assert_close, cfg_of, golden_names, load_golden, t
from oracle import gns_oracle as orc


@pytest.mark.parametrize('name', golden_names())
def test_param_layout_matches_reference_state_dict(name):
    g = load_golden(name)
    c = cfg_of(g)
    spec = orc.param_spec(c['latent_dim'], c['hidden_dim'], c['K'], c['multiple_phi'])
    assert [n for n, _ in spec] == [str(s) for s in g['param_names']]
    assert sum(int(np.prod(s)) for _, s in spec) == g['params'].size


@pytest.mark.parametrize('name', golden_names())
def test_forward_and_steps(name):
    g = load_golden(name)
    c = cfg_of(g)
    params = orc.unflatten_params(t(g['params']), c['latent_dim'], c['hidden_dim'], c['K'], c['multiple_phi'])
    for b in range(int(g['batch'])):
        trace = []
        v, th, tot, last = orc.gns_forward(params, t(g['buses'][b]), t(g['lines'][b]), t(g['generators'][b]),
                                           latent_dim=c['latent_dim'], K=c['K'], gamma=c['gamma'],
                                           multiple_phi=c['multiple_phi'], trace=trace)
        # same torch ops in the same order as the reference -> expect agreement to rounding
        assert_close(v, g['v'][b], 1e-6, what='v')
        assert_close(th, g['theta'][b], 1e-6, what='theta')
        assert_close(tot, g['total_loss'][b], 1e-6, what='total_loss')
        assert_close(last, g['last_loss'][b], 1e-6, what='last_loss')
        for k, st in enumerate(trace):
            assert_close(st['v'], g['step_v'][b, k], 1e-6, what=f'v@{k}')
            assert_close(st['theta'], g['step_theta'][b, k], 1e-6, what=f'theta@{k}')
            assert_close(st['dp'], g['step_dp'][b, k], 1e-6, what=f'dp@{k}')
            assert_close(st['pg_new'], g['step_pg_new'][b, k], 1e-6, what=f'pg_new@{k}')
            assert_close(st['qg_new'], g['step_qg_new'][b, k], 1e-6, what=f'qg_new@{k}')
            # delta_q is identically ~0 (cancellation noise): absolute check only
            assert np.max(np.abs(st['dq'].numpy() - g['step_dq'][b, k])) < 5e-6


@pytest.mark.parametrize('name', golden_names())
def test_parameter_gradient(name):
    g = load_golden(name)
    c = cfg_of(g)
    _, _, tot, last, grad = orc.gns_forward_backward(
        t(g['params']), t(g['buses']), t(g['lines']), t(g['generators']), latent_dim=c['latent_dim'],
        hidden_dim=c['hidden_dim'], K=c['K'], gamma=c['gamma'], multiple_phi=c['multiple_phi'])
    assert_close(grad, g['grad_params'], 2e-5, abs_floor=1e-7, what='grad_params')
    # parameters the reference leaves at grad None get exactly zero here
    spec = orc.param_spec(c['latent_dim'], c['hidden_dim'], c['K'], c['multiple_phi'])
    none = set(str(s) for s in g['none_grad_names'])
    off = 0
    for n, shape in spec:
        sz = int(np.prod(shape))
        if n in none:
            assert float(grad[off:off + sz].abs().max()) == 0.0, n
        off += sz
    assert none == {n for n, _ in spec if (n.startswith(f'L_m.{c["K"] - 1}.') or n.startswith(f'phi_m.{c["K"] - 1}.'))}


def test_lowload_goldens_cover_the_low_lambda_branch():
    """GNS/main.py:48,54 only fire when p_global < sum(Pg_set); the *_lowload goldens must reach them."""
    hit = 0
    for name in golden_names():
        if 'lowload' not in name:
            continue
        g = load_golden(name)
        c = cfg_of(g)
        params = orc.unflatten_params(t(g['params']), c['latent_dim'], c['hidden_dim'], c['K'], c['multiple_phi'])
        trace = []
        orc.gns_forward(params, t(g['buses'][0]), t(g['lines'][0]), t(g['generators'][0]), latent_dim=c['latent_dim'],
                        K=c['K'], gamma=c['gamma'], multiple_phi=c['multiple_phi'], trace=trace)
        hit += sum(float(st['lam']) < 0.5 for st in trace)
    assert hit > 0


def test_float64_oracle_is_close_to_float32():
    g = load_golden('c118_b2_K4_d20_multi')
    c = cfg_of(g)
    p64 = orc.unflatten_params(t(g['params']).double(), c['latent_dim'], c['hidden_dim'], c['K'], c['multiple_phi'])
    v, th, tot, last = orc.gns_forward(p64, t(g['buses'][0]).double(), t(g['lines'][0]).double(),
                                       t(g['generators'][0]).double(), latent_dim=c['latent_dim'], K=c['K'],
                                       gamma=c['gamma'], multiple_phi=c['multiple_phi'])
    assert_close(v, g['v'][0], 1e-5, what='v64')
    assert_close(th, g['theta'][0], 1e-5, what='theta64')
