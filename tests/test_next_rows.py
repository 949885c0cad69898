"""The rows either side of the hot path: input producer, evaluation metrics, training loop."""
import os

import numpy as np
import pytest
import torch

import opf_graph_neural_solver_amd as amd
from helpers import load_golden, t


@pytest.mark.parametrize('name', ['prepare_c14_b4', 'prepare_c118_b2'])
def test_prepare_grids_matches_reference_prepare_grid(name):
    """Goldens come from the reference's own utils.prepare_grid (GNS/utils.py:17-41) run on synthetic PYPOWER dicts."""
    g = load_golden(name)
    b, l, ge = amd.prepare_grids(t(g['bus']), t(g['branch']), t(g['gen']), 100.0)
    assert b.dtype == torch.float32 and b.shape == g['buses'].shape
    np.testing.assert_allclose(b.numpy(), g['buses'], rtol=1e-7, atol=0)
    np.testing.assert_allclose(l.numpy(), g['lines'], rtol=1e-7, atol=0)
    np.testing.assert_allclose(ge.numpy(), g['generators'], rtol=1e-7, atol=0)
    assert float((g['branch'][:, :, 8] == 0).sum()) > 0 and float((l[:, :, 5] == 0).sum()) == 0     # tau == 0 -> 1 (utils.py:33)
    b1, l1, g1 = amd.prepare_grids(t(g['bus'][0]), t(g['branch'][0]), t(g['gen'][0]))               # 2-D form
    assert torch.equal(b1, b[0]) and torch.equal(l1, l[0]) and torch.equal(g1, ge[0])
    # per-grid baseMVA
    b2, _, _ = amd.prepare_grids(t(g['bus']), t(g['branch']), t(g['gen']), torch.full((g['bus'].shape[0],), 100.0))
    assert torch.equal(b2, b)


def test_active_line_flow_and_percentiles_against_numpy_restatement():
    """evaluate.py cannot be imported (module-level script needing PYPOWER); its two formulas are restated in numpy
    here (evaluate.py:15-18 and :117-125) - parity unpinned by reference execution."""
    rng = np.random.default_rng(0)
    B, N, E = 5, 14, 20
    v, th = rng.uniform(0.9, 1.1, (B, N)), rng.uniform(-0.3, 0.3, (B, N))
    x = rng.uniform(0.05, 0.5, (B, E))
    f, tt, _ = amd.synth.case_topology(14)
    ref = np.stack([1 / x[b] * (v[b][f - 1] * v[b][tt - 1] * np.sin(th[b][f - 1] - th[b][tt - 1])) for b in range(B)])
    got = amd.metrics.active_line_flow(t(v), t(th), t(x), t(f.astype(np.float64)), t(tt.astype(np.float64)))
    np.testing.assert_allclose(got.numpy(), ref, rtol=1e-12)
    flow_ref = ref * rng.uniform(0.8, 1.2, ref.shape)
    pct = np.abs((flow_ref - ref) / flow_ref) * 100
    low = np.sort(pct, axis=None)[: int(pct.size / 2)]
    q = amd.metrics.line_flow_percentiles(t(ref), t(flow_ref))
    np.testing.assert_allclose([float(q['p20']), float(q['median']), float(q['p80'])],
                               [np.percentile(low, 20), np.median(low), np.percentile(low, 80)], rtol=1e-10)
    e = amd.metrics.solution_errors(t(v), t(th), t(v * 1.01), t(th + 0.01))
    np.testing.assert_allclose(float(e['v_abs_mean']), np.mean(np.abs(v - v * 1.01)), rtol=1e-10)
    np.testing.assert_allclose(float(e['theta_abs_std']), np.std(np.abs(th - (th + 0.01))), rtol=1e-6, atol=1e-12)


def _check_metrics_against_reference_statements(dev):
    """f4 pinned: the expected values were produced by the reference's OWN statements (``evaluate.py:15-18`` and the
    statistics assignments of ``evaluate.py:93-157``, compiled from the parsed file by ``oracle/make_goldens.py``)."""
    g = load_golden('metrics_c14_s64')
    dt = lambda a: torch.as_tensor(np.asarray(a)).to(dev)
    v_nr, th_nr = dt(g['in_v_nr']), torch.deg2rad(dt(g['in_theta_nr_deg']))
    v, th = dt(g['in_v_gns']), dt(g['in_theta_gns'])
    f_nr = amd.metrics.active_line_flow(v_nr.double(), th_nr.double(), dt(g['in_x']), dt(g['in_src']), dt(g['in_dst']))
    f = amd.metrics.active_line_flow(v.double(), th.double(), dt(g['in_x']), dt(g['in_src']), dt(g['in_dst']))
    np.testing.assert_allclose(f_nr.cpu().numpy(), g['alf_nr'], rtol=2e-6, atol=1e-7)       # the reference stores float32
    np.testing.assert_allclose(f.cpu().numpy(), g['alf_gns'], rtol=2e-6, atol=1e-7)
    e = amd.metrics.solution_errors(v, th, v_nr, th_nr)
    c = lambda x: x.detach().cpu().numpy()
    np.testing.assert_allclose(c(e['theta_abs_mean']), g['ref_mean_diff_theta_gns'], rtol=1e-5)
    np.testing.assert_allclose(c(e['theta_abs_std']), g['ref_std_diff_theta_gns'], rtol=1e-4)
    np.testing.assert_allclose(c(e['v_abs_mean']), g['ref_mean_diff_v_gns'], rtol=1e-5)
    np.testing.assert_allclose(c(e['v_abs_std']), g['ref_std_diff_v_gns'], rtol=1e-4)
    np.testing.assert_allclose(c(e['theta_pct_error']), g['ref_theta_error_gns_nr'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(c(e['v_pct_error']), g['ref_v_error_gns_nr'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(c(e['v_diff_per_bus_mean']), g['ref_mean_diffs_v'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(c(e['v_diff_per_bus_std']), g['ref_std_diffs_v'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(c(e['theta_diff_per_bus_mean']), g['ref_mean_diffs_theta'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(c(e['theta_diff_per_bus_std']), g['ref_std_diffs_theta'], rtol=1e-4, atol=1e-7)
    q = amd.metrics.line_flow_percentiles(dt(g['alf_gns']), dt(g['alf_nr']))
    np.testing.assert_allclose([float(q['p20']), float(q['median']), float(q['p80'])],
                               [float(g['ref_twenty_percentile_diff_alf_gns_nr']), float(g['ref_median_diff_alf_gns_nr']),
                                float(g['ref_eighty_percentile_diff_alf_gns_nr'])], rtol=1e-5)


def test_metrics_match_the_reference_evaluation_statements():
    _check_metrics_against_reference_statements('cpu')


@pytest.mark.gpu
def test_metrics_match_the_reference_evaluation_statements_on_the_device():
    _check_metrics_against_reference_statements('cuda')


class _FakeGNS(torch.nn.Module):
    """Stands in for the GPU model in the CPU test of the loop logic: loss follows a scripted sequence per epoch."""

    def __init__(self, script):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(3))
        self.K, self.latent_dim, self.hidden_dim, self.multiple_phis = 4, 20, 10, True
        self.script, self.calls = script, 0

    def forward(self, buses, lines, generators, B, L, G):
        bt = buses.shape[0]
        val = self.script[min(self.calls // 2, len(self.script) - 1)]       # two batches per epoch
        self.calls += 1
        loss = (self.w.sum() * 0 + val) * torch.ones(bt)
        return None, None, loss, loss.detach()


def test_fit_early_stopping_and_checkpoint_name(tmp_path):
    """Early stop after the epoch loss failed to improve more than twice in a row (main.py:296-304)."""
    script = [5.0, 4.0, 4.5, 4.2, 4.1, 3.0, 2.0]        # epochs 2,3,4 do not beat 4.0 -> third miss breaks the loop
    m = _FakeGNS(script)
    bu, li, ge = amd.synth.synth_grids(14, 8, seed=0)
    logs = []
    hist = amd.training.fit(m, bu, li, ge, epochs=10, batch_size=4, case_nr=14, checkpoint_dir=str(tmp_path), log=logs.append)
    assert np.allclose(hist, [5.0, 4.0, 4.5, 4.2, 4.1]) and len(hist) == 5 and logs[-1] == 'Loss is increasing'
    name = amd.training.checkpoint_name(14, 4, 20, 10, True, 'Adam')
    assert name == 'best_model_c14_K4_L20_H10_True_optimAdam.pth' and os.path.exists(os.path.join(str(tmp_path), name))
    assert isinstance(amd.training.make_optimizer(m, 'Adagrad'), torch.optim.Adagrad)
    assert amd.training.make_optimizer(m, 'Adagrad').defaults['lr'] == 0.01 and amd.training.make_optimizer(m).defaults['lr'] == 0.001


@pytest.mark.gpu
def test_fit_trains_on_device_and_checkpoint_round_trips(tmp_path):
    torch.manual_seed(0)
    m = amd.GNS(20, 10, 4, 0.9, True).cuda()
    bus, br, gen = amd.synth.raw_case_arrays(14, 256, seed=3, zero_tau_fraction=0.0)
    buses, lines, gens = amd.prepare_grids(bus.cuda(), br.cuda(), gen.cuda())      # device-side input producer feeding the hot path
    hist = amd.training.fit(m, buses, lines, gens, epochs=6, batch_size=128, lr=1e-3, case_nr=14, checkpoint_dir=str(tmp_path), log=lambda s: None)
    assert len(hist) == 6 and all(np.isfinite(hist)) and hist[-1] < hist[0]
    path = os.path.join(str(tmp_path), amd.training.checkpoint_name(14, 4, 20, 10, True, 'Adam'))
    m2 = amd.GNS(20, 10, 4, 0.9, True)
    m2.load_state_dict(torch.load(path, weights_only=True))
    m2 = m2.cuda()
    with torch.no_grad():
        a, b = m(buses[:64], lines[:64], gens[:64]), m2(buses[:64], lines[:64], gens[:64])
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    f, tt, _ = amd.synth.case_topology(14)
    flow = amd.metrics.active_line_flow(a[0], a[1], lines[:64, :, 3], lines[0, :, 0], lines[0, :, 1])
    assert flow.shape == (64, 20) and bool(torch.isfinite(flow).all())


def test_counter_based_synthetic_grids_are_keyed_by_global_grid_index():
    """every GPU count must see the same grids: a shard [lo, hi) of the data set equals those rows of the whole."""
    import opf_graph_neural_solver_amd as amd
    full = amd.synth.synth_grids(30, 50, seed=9)
    for lo, n in ((0, 7), (7, 30), (37, 13)):
        part = amd.synth.synth_grids(30, n, seed=9, first_index=lo)
        for a, b in zip(full, part):
            assert torch.equal(a[lo:lo + n], b)
    other = amd.synth.synth_grids(30, 50, seed=10)
    assert not torch.equal(full[0], other[0])
    u = amd.synth.counter_uniform(3, 1, 0, 20000, 16, 'cpu')
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 5e-3
    # balance rescale of augment_grids.py:51: sum(Pd) == sum(Pg) for every grid
    assert torch.allclose(full[0][:, :, 2].sum(1), full[2][:, :, 3].sum(1), rtol=1e-5)
