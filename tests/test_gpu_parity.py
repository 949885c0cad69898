"""HIP path against the reference-generated goldens and against the CPU oracle (-m gpu).
Tolerance: 1e-5 relative (max|diff|/max|ref| per tensor, BASELINE.json north_star) with a 1e-6 absolute floor."""
import numpy as np
import pytest
import torch

from helpers import assert_close, cfg_of, golden_names, load_golden, t

pytestmark = pytest.mark.gpu

REL = 1e-5


def _model(g, device='cuda'):
    import opf_graph_neural_solver_amd as amd
    c = cfg_of(g)
    m = amd.GNS(latent_dim=c['latent_dim'], hidden_dim=c['hidden_dim'], K=c['K'], gamma=c['gamma'],
                multiple_phi=c['multiple_phi'])
    names = [n for n, _ in m.named_parameters()]
    assert names == [str(s) for s in g['param_names']]          # state_dict key space of the reference
    flat, off, sd = t(g['params']), 0, {}
    for n, p in m.named_parameters():
        sd[n] = flat[off:off + p.numel()].view(p.shape).clone()
        off += p.numel()
    m.load_state_dict(sd)
    return m.to(device)


@pytest.mark.parametrize('name', golden_names())
def test_forward_matches_reference_golden(name):
    g = load_golden(name)
    m = _model(g)
    B, L, G = __import__('opf_graph_neural_solver_amd').get_BLG()
    with torch.no_grad():
        v, th, tot, last = m(t(g['buses']).cuda(), t(g['lines']).cuda(), t(g['generators']).cuda(), B, L, G)
    assert_close(v.cpu(), g['v'], REL, what='v')
    assert_close(th.cpu(), g['theta'], REL, what='theta')
    assert_close(tot.cpu(), g['total_loss'], REL, what='total_loss')
    assert_close(last.cpu(), g['last_loss'], REL, what='last_loss')


@pytest.mark.parametrize('name', ['c14_b1_K4_d20_multi', 'c118_b2_K4_d20_single'])
def test_single_grid_2d_api_and_cpu_inputs(name):
    """The reference's call shape: 2-D CPU tensors, keyword arguments (GNS/main.py:281)."""
    g = load_golden(name)
    m = _model(g)
    B, L, G = __import__('opf_graph_neural_solver_amd').get_BLG()
    with torch.no_grad():
        v, th, tot, last = m(buses=t(g['buses'][0]), lines=t(g['lines'][0]), generators=t(g['generators'][0]), B=B, L=L, G=G)
    assert v.device.type == 'cpu' and v.shape == (g['buses'].shape[1],) and tot.dim() == 0
    assert_close(v, g['v'][0], REL, what='v')
    assert_close(th, g['theta'][0], REL, what='theta')
    assert_close(tot, g['total_loss'][0], REL, what='total')
