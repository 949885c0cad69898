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


def test_flat_optimizer_is_a_torch_optimizer_and_takes_a_scheduler():
    """lr schedulers (the reference keeps a LambdaLR warm-up commented out, main.py:245-252) must be able to drive the
    flat optimiser; its param_groups / state are the inner torch optimiser's."""
    torch.manual_seed(0)
    m = amd.GNS(10, 10, 2, 0.9, True)
    opt = amd.training.FlatOptimizer(m, torch.optim.Adam, lr=1e-3)
    assert isinstance(opt, torch.optim.Optimizer)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: 0.5 ** e)
    w0 = m.flat_parameters().clone()
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step(); sched.step()
    assert torch.allclose(w0 - m.flat_parameters(), torch.full_like(w0, 1e-3), rtol=1e-4)      # Adam's first step moves every weight by lr
    assert opt.param_groups[0]['lr'] == 5e-4 and opt.inner.param_groups[0]['lr'] == 5e-4
    w1 = m.flat_parameters().clone()
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert torch.allclose(w1 - m.flat_parameters(), torch.full_like(w0, 5e-4), rtol=1e-3)
    sd = opt.state_dict()
    opt.load_state_dict(sd)
    assert opt.param_groups is opt.inner.param_groups and opt.state is opt.inner.state
    # frozen parameters: no flat gradient leaf (it would span them), a plain torch optimiser over the trainable ones
    m2 = amd.GNS(10, 10, 2, 0.9, True)
    m2.L_v['0'].linear1.weight.requires_grad_(False)
    o2 = amd.training.make_optimizer(m2, flat=True)
    assert not isinstance(o2, amd.training.FlatOptimizer) and not m2.flat_grad


@pytest.mark.gpu
def test_graphed_step_replays_equal_eager_steps_at_the_reference_operating_point():
    """the reference's own run (case14, batch 128, K=15, latent 10, three phis: main.py:209-254) as ONE captured
    HIP graph per step - forward, mean, backward, reduce / unfold, Adam.  Ten replays leave exactly the weights of ten eager steps."""
    import time
    bu, li, ge = amd.synth.synth_grids(14, 128 * 10, seed=5, device='cuda')
    res = {}
    for mode in ('eager', 'graph'):
        torch.manual_seed(0)
        m = amd.GNS(10, 10, 15, 0.9, True).cuda()
        opt = amd.training.make_optimizer(m)
        assert isinstance(opt, amd.training.FlatOptimizer)
        if mode == 'eager':
            opt.capturable = True                          # the same device-side step counter the captured step uses
            m.topology_check = 'first'
            for i in range(10):
                sl = slice(128 * i, 128 * (i + 1))
                amd.training.train_step(m, opt, bu[sl], li[sl], ge[sl])
            torch.cuda.synchronize()
            res['eager10'] = m.flat_parameters().clone()
            t0 = time.perf_counter()
            for i in range(100):
                amd.training.train_step(m, opt, bu[:128], li[:128], ge[:128])
            torch.cuda.synchronize()
            res['ms_per_eager_step'] = (time.perf_counter() - t0) / 100 * 1e3
        else:
            step = amd.training.GraphedStep(m, opt, bu[:128], li[:128], ge[:128])
            for i in range(10):
                sl = slice(128 * i, 128 * (i + 1))
                tot, last = step.run(bu[sl], li[sl], ge[sl])
            assert step.replays == 10 and bool(torch.isfinite(tot)) and bool(torch.isfinite(last))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(200):
                step.run(bu[:128], li[:128], ge[:128])
            torch.cuda.synchronize()
            res['ms_per_replayed_step'] = (time.perf_counter() - t0) / 200 * 1e3
    # (the graph model ran 200 more replays for the timing: compare a fresh capture's first ten instead)
    torch.manual_seed(0)
    m = amd.GNS(10, 10, 15, 0.9, True).cuda()
    opt = amd.training.make_optimizer(m)
    step = amd.training.GraphedStep(m, opt, bu[:128], li[:128], ge[:128])
    for i in range(10):
        sl = slice(128 * i, 128 * (i + 1))
        step.run(bu[sl], li[sl], ge[sl])
    torch.cuda.synchronize()
    assert torch.equal(m.flat_parameters(), res['eager10'])
    print(f"\\n[training step, case14 x 128, K=15, d=10, three phis] eager {res['ms_per_eager_step']:.3f} ms, "
          f"captured graph {res['ms_per_replayed_step']:.3f} ms per step")
    # (at K=15 the step is bound by its kernels' serial chain - 15 steps x 15 phases of one small grid per workgroup - not by the
    #  host: a replay then costs what the eager step costs; the capture pays where the host does, e.g. K=4: 0.41 -> 0.31 ms)
    assert res['ms_per_replayed_step'] < 1.15 * res['ms_per_eager_step']


@pytest.mark.gpu
@pytest.mark.parametrize('wait', ['stream', 'device', 'none'])
def test_graphed_step_replays_equal_eager_steps_when_the_host_waits_in_between(wait):
    """Replays separated by a host-side wait on the stream (what any ``float(loss)`` / ``torch.cuda.synchronize()`` / checkpoint in a
    training loop is).  With ROCm 7's default graph replay path (pre-built AQL packets) exactly this pattern computed wrong gradients
    and NaN parameters a few steps later; the package switches that path off at import and GraphedStep verifies its first replays
    (opf_graph_neural_solver_amd/__init__.py, training.GraphedStep).  Eight steps on alternating batches, bit for bit the eager loop's."""
    bu, li, ge = amd.synth.synth_grids(14, 256, seed=3, device='cuda')

    def run(graphed):
        torch.manual_seed(0)
        m = amd.GNS(20, 10, 4, 0.9, True).cuda()
        opt = amd.training.make_optimizer(m, 'Adam', lr=1e-3)
        st = None
        for step in range(8):
            sl = slice(128 * (step % 2), 128 * (step % 2) + 128)
            if wait == 'stream':
                torch.cuda.current_stream().synchronize()
            elif wait == 'device':
                torch.cuda.synchronize()
            if graphed:
                st = st or amd.training.GraphedStep(m, opt, bu[sl], li[sl], ge[sl])
                st.run(bu[sl], li[sl], ge[sl])
            else:
                amd.training.train_step(m, opt, bu[sl], li[sl], ge[sl])
        torch.cuda.synchronize()
        return m.flat_parameters().detach().clone()

    eager, graphed = run(False), run(True)
    assert bool(torch.isfinite(graphed).all())
    assert torch.equal(eager, graphed)


@pytest.mark.gpu
def test_fit_reads_a_bound_dataset_without_repacking_and_matches_per_call_packing():
    """fit() packs the resident data set once (GNS.bind_dataset); every 64-aligned batch of the epochs is read
    from that copy.  Same bits as the loop that packs each batch on every call."""
    old = amd.get_option('train_mapping')
    amd.set_option('train_mapping', 1)                     # the lane-per-grid kernels are the ones that read the packed layout
    try:
        bu, li, ge = amd.synth.synth_grids(30, 4 * 512, seed=11, device='cuda')
        torch.manual_seed(0)
        a = amd.GNS(20, 10, 3, 0.9, True).cuda()
        hits = {}
        orig_unbind = a.unbind_dataset
        def unbind():
            hits['n'] = a._resident['hits']
            orig_unbind()
        a.unbind_dataset = unbind
        amd.training.fit(a, bu, li, ge, epochs=3, batch_size=512, log=lambda s: None, graph=False)
        assert hits['n'] == 4 * 3 * 1 or hits['n'] == 4 * 3                     # one lookup per training forward
        torch.manual_seed(0)
        b = amd.GNS(20, 10, 3, 0.9, True).cuda()
        b.topology_check = 'first'
        opt = amd.training.make_optimizer(b)
        for epoch in range(3):
            for lo in range(0, 2048, 512):
                amd.training.train_step(b, opt, bu[lo:lo + 512], li[lo:lo + 512], ge[lo:lo + 512])
        torch.cuda.synchronize()
        assert torch.equal(a.flat_parameters(), b.flat_parameters())
        # a batch that is not a 64-aligned slice of the bound set is packed per call as before
        a.bind_dataset(bu, li, ge)
        n0 = a._resident['hits']
        with torch.enable_grad():
            a(bu[32:32 + 512], li[32:32 + 512], ge[32:32 + 512])[2].mean().backward()
            a(bu[64:64 + 512].clone(), li[64:64 + 512].clone(), ge[64:64 + 512].clone())[2].mean().backward()
        assert a._resident['hits'] == n0
        a(bu[64:64 + 100], li[64:64 + 100], ge[64:64 + 100])[2].mean().backward()   # ragged tail inside the set: lanes beyond it are dead
        assert a._resident['hits'] == n0 + 1
        a.unbind_dataset()
    finally:
        amd.set_option('train_mapping', old)


@pytest.mark.gpu
def test_default_topology_check_compares_the_whole_batch_and_inputs_are_version_checked():
    """a batch whose FIRST grid matches the cached case but whose other grids do not must raise (default 'always');
    inputs modified in place between forward and backward must raise like autograd would."""
    torch.manual_seed(0)
    m = amd.GNS(20, 10, 2, 0.9, True).cuda()
    assert m.topology_check == 'always'
    bu, li, ge = amd.synth.synth_grids(14, 8, seed=2, device='cuda')
    m(bu, li, ge)
    bad = li.clone()
    bad[5, 3, 0], bad[5, 3, 1] = li[5, 3, 1], li[5, 3, 0]        # one line of one grid reversed: another topology
    with pytest.raises(ValueError):
        m(bu, bad, ge)
    m.topology_check = 'grid0'                                   # explicit opt-in: only the first grid is compared
    m(bu, bad, ge)
    m.topology_check = 'always'
    out = m(bu, li, ge)
    bu.mul_(1.0)                                                 # in-place write bumps the version counter
    with pytest.raises(amd.GNSError):
        out[2].mean().backward()


def _check_synth_against_augment_golden(device):
    """the on-device synthetic generator against the REFERENCE'S OWN perturbation statements (GNS/augment_grids.py:
    12-20, 30-53, compiled from the parsed file by oracle/make_goldens.run_augment and run 4096 times on the IEEE-14 base data):
    ranges exact, the balance identity sum(Pd) == sum(Pg) of :51, means and spreads of every column statistically equal."""
    import math
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'augment_c14.npz'))
    n = 4096
    assert int(g['n_draws']) == n
    bu, li, ge = amd.synth.synth_grids(14, n, seed=77, device=device)
    bu, li, ge = bu.double().cpu().numpy(), li.double().cpu().numpy(), ge.double().cpu().numpy()
    mva = amd.synth.BASE_MVA
    cols = {'r': li[:, :, 2], 'x': li[:, :, 3], 'b': li[:, :, 4], 'tau': li[:, :, 5], 'shift': li[:, :, 6] * 180.0 / math.pi,
            'vg': ge[:, :, 4], 'pg': ge[:, :, 3] * mva, 'pd': bu[:, :, 2] * mva, 'qd': bu[:, :, 3] * mva}
    eps = 2e-6
    # multiplicative perturbations: value / base inside the reference's range, exact zeros where the base is zero
    for k, rk, base in (('r', 'r_range', g['base_r']), ('x', 'x_range', g['base_x']), ('b', 'b_range', g['base_b']),
                        ('vg', 'vg_range', g['base_vg']), ('qd', 'qd_range', g['base_qd'])):
        lo, hi = g[rk]
        nz = base != 0
        ratio = cols[k][:, nz] / base[nz]
        assert ratio.min() >= lo * (1 - eps) and ratio.max() <= hi * (1 + eps), (k, ratio.min(), ratio.max())
        assert np.all(cols[k][:, ~nz] == 0)
        assert ratio.min() < lo + 0.02 * (hi - lo) and ratio.max() > hi - 0.02 * (hi - lo)       # ... and the range is used
    for k, rk in (('tau', 'tau_range'), ('shift', 'theta_shift_range')):
        lo, hi = g[rk]
        assert cols[k].min() >= lo - eps and cols[k].max() <= hi + eps and cols[k].min() < lo + 0.01 * (hi - lo)
    span = g['base_pmax'] - g['base_pmin']
    lo_pg, hi_pg = g['base_pmin'] + span * g['pg_range'][0], span * g['pg_range'][1]             # augment_grids.py:45-47
    assert np.all(cols['pg'] >= lo_pg * (1 - eps) - eps) and np.all(cols['pg'] <= hi_pg * (1 + eps) + eps)
    # the reference's draws obey the same bounds (the golden is the reference, not a restatement of it)
    assert np.all(g['pg_min'] >= lo_pg - 1e-9) and np.all(g['pg_max'] <= hi_pg + 1e-9)
    # balance (augment_grids.py:51): the reference reaches 1e-15 in float64, the fp32 generator 1e-5
    assert float(g['balance_max_rel_err']) < 1e-12
    assert np.abs(cols['pd'].sum(1) / cols['pg'].sum(1) - 1.0).max() < 1e-5
    # every column: mean within 5 standard errors of the difference of two independent samples of n (138 comparisons: a 3-sigma
    # bar would fail one of them by chance in every third data set; both samples are seeded, so the outcome is deterministic),
    # spread within 6 %
    for k in cols:
        mu, sd = g[k + '_mean'], g[k + '_std']
        se = np.sqrt(2.0) * sd / math.sqrt(n)
        assert np.all(np.abs(cols[k].mean(0) - mu) <= 5.0 * se + 1e-6 * (1 + np.abs(mu))), (k, np.abs(cols[k].mean(0) - mu).max())
        live = sd > 1e-12
        assert np.all(np.abs(cols[k].std(0)[live] / sd[live] - 1.0) < 0.06), k


def test_synthetic_grids_follow_the_reference_augmentation_statements():
    _check_synth_against_augment_golden('cpu')


@pytest.mark.gpu
def test_synthetic_grids_follow_the_reference_augmentation_statements_on_the_device():
    _check_synth_against_augment_golden('cuda')
