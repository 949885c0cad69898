"""HIP path against the reference-generated goldens and against the CPU oracle (-m gpu).
Tolerance: 1e-5 relative (max|diff|/max|ref| per tensor, BASELINE.json north_star) with a 1e-6 absolute floor."""
import numpy as np
import pytest
import torch

from helpers import assert_close, cfg_of, deep_golden_names, golden_names, load_golden, rel_err, shallow_golden_names, t

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


@pytest.fixture(params=['lane-per-grid', 'grid-per-workgroup'])
def fwd_mapping(request):
    """Both forward mappings (gns_set_option "fwd_mapping"): 1 = lane per grid, state streamed through HBM;
    2 = grid per workgroup, state on chip (evaluation-mode calls)."""
    import opf_graph_neural_solver_amd as amd
    old = amd.get_option('fwd_mapping')
    amd.set_option('fwd_mapping', 1 if request.param == 'lane-per-grid' else 2)
    yield request.param
    amd.set_option('fwd_mapping', old)


@pytest.mark.parametrize('name', shallow_golden_names())
def test_forward_matches_reference_golden(name, fwd_mapping):
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


@pytest.fixture(params=['matrix-pipe', 'split-kernels', 'split-kernels-per-family', 'split-kernels-bus-major', 'matrix-pipe-background-chains',
                        'matrix-pipe-wide-records', 'packed-fma'])
def dw_engine(request):
    """The weight-gradient paths of the lane-per-grid backward kernel (gns_set_option "dw_mfma", "bwd_variant"): matrix pipe
    with the layer-wise sweep and sub-record windows, the same sweep as one kernel sequence per reverse step (bwd_variant 4,
    gns_backward_split.hip), the same with the contraction chains issued behind the weight streams, matrix pipe with wide
    half-wave records, packed-FMA tiles."""
    import opf_graph_neural_solver_amd as amd
    old = amd.get_option('dw_mfma'), amd.get_option('bwd_variant'), amd.get_option('train_mapping'), amd.get_option('bwds_mode')
    amd.set_option('train_mapping', 1)                 # (small batches would otherwise go to the grid-per-workgroup pair)
    amd.set_option('dw_mfma', 0 if request.param == 'packed-fma' else 1)
    split = request.param.startswith('split-kernels')  # bwds_mode: sweep kernels per step {m}{theta+v} | {m}{theta}{v} | {m+theta+v}
    amd.set_option('bwd_variant', 4 if split else {'matrix-pipe-wide-records': 1, 'matrix-pipe-background-chains': 3}.get(request.param, 2))
    if split:
        amd.set_option('bwds_mode', {'split-kernels': 1, 'split-kernels-per-family': 0, 'split-kernels-bus-major': 2}[request.param])
    yield request.param
    amd.set_option('dw_mfma', old[0]); amd.set_option('bwd_variant', old[1]); amd.set_option('train_mapping', old[2]); amd.set_option('bwds_mode', old[3])


@pytest.mark.parametrize('name', shallow_golden_names())
def test_parameter_gradient_matches_reference_autograd(name, dw_engine):
    """d(mean total_loss)/d(params) against the reference's own .backward() (GNS/main.py:284-288).
    Gradient tolerance 5e-5 of max|grad| (fp32 products of ~1e5 terms summed in a different order)."""
    g = load_golden(name)
    m = _model(g)
    v, th, tot, last = m(t(g['buses']).cuda(), t(g['lines']).cuda(), t(g['generators']).cuda())
    tot.mean().backward()
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    assert_close(grad, g['grad_params'], 5e-5, abs_floor=1e-7, what='grad_params')
    none = set(str(s) for s in g['none_grad_names'])       # L_m.{K-1}, phi_m.{K-1}: no gradient in the reference
    for n, p in m.named_parameters():
        if n in none:
            assert float(p.grad.abs().max()) == 0.0, n


@pytest.fixture
def lane_mapping():
    """Pin both the evaluation and the training calls to the lane-per-grid kernels (small batches default to the other pair)."""
    import opf_graph_neural_solver_amd as amd
    old = amd.get_option('fwd_mapping'), amd.get_option('train_mapping'), amd.get_option('team')
    amd.set_option('fwd_mapping', 1); amd.set_option('train_mapping', 1)
    yield amd
    amd.set_option('fwd_mapping', old[0]); amd.set_option('train_mapping', old[1]); amd.set_option('team', old[2])


@pytest.mark.parametrize('team', [1, 2, 4])
@pytest.mark.parametrize('name', ['c14_b3_K4_d20_multi_lowload', 'c118_b2_K4_d20_multi', 'c118_b2_K4_d20_single', 'c300_b1_K10_d20_multi'])
def test_teams_of_workgroups_match_reference_goldens(name, team, lane_mapping):
    """A batch with fewer 64-grid groups than CUs is worked by teams of 2 or 4 workgroups per group (gns_device.h, "teams":
    barrier through an HBM counter, partial sums through HBM).  Forward outputs and parameter gradients against the
    reference's, for every team size; team 1 is the single-workgroup kernel."""
    lane_mapping.set_option('team', team)
    g = load_golden(name)
    m = _model(g)
    bu, li, ge = t(g['buses']).cuda(), t(g['lines']).cuda(), t(g['generators']).cuda()
    with torch.no_grad():
        v, th, tot, last = m(bu, li, ge)
    assert_close(v.cpu(), g['v'], REL, what='v')
    assert_close(th.cpu(), g['theta'], REL, what='theta')
    assert_close(tot.cpu(), g['total_loss'], REL, what='total_loss')
    v, th, tot, last = m(bu, li, ge)
    assert_close(last.detach().cpu(), g['last_loss'], REL, what='last_loss (training mode)')
    tot.mean().backward()
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    assert_close(grad, g['grad_params'], 5e-5, abs_floor=1e-7, what='grad_params')


@pytest.mark.parametrize('case,bt,d,multi,K', [(118, 4099, 20, True, 4), (300, 130, 20, True, 10), (30, 8000, 10, False, 3), (118, 8192, 10, True, 6)])
def test_teams_agree_with_single_workgroups_on_partial_chip_batches(case, bt, d, multi, K, lane_mapping):
    """Many teams at once (65 groups x 2, 3 x 4, 125 x 2, 128 x 2 workgroups), ragged last group: the team kernels give the
    single-workgroup kernels' outputs and gradients up to the order of the per-wave partial sums."""
    amd = lane_mapping
    torch.manual_seed(4)
    m = amd.GNS(d, 10, K, 0.9, multi).cuda()
    bu, li, ge = amd.synth.synth_grids(case, bt, seed=21, device='cuda')
    res = {}
    for team in (1, 0):
        amd.set_option('team', team)
        m.zero_grad()
        with torch.no_grad():
            ev = m(bu, li, ge)
        out = m(bu, li, ge)
        out[2].mean().backward()
        res[team] = [x.detach().clone() for x in ev] + [x.detach().clone() for x in out] + [torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()]
    for a, b, what in zip(res[1], res[0], ['v', 'theta', 'total', 'last', 'v (train)', 'theta (train)', 'total (train)', 'last (train)', 'grad']):
        assert torch.isfinite(b).all(), what
        assert_close(b.cpu(), a.cpu(), 2e-5 if what == 'grad' else 2e-6, abs_floor=1e-7, what=what)


@pytest.mark.parametrize('case,bt,d,multi,K,pack', [(118, 4099, 20, True, 4, 1), (118, 1030, 20, True, 4, 4), (30, 777, 10, False, 3, 3),
                                                    (14, 2001, 10, True, 2, 16), (300, 130, 20, True, 10, 1), (118, 257, 10, True, 15, 2)])
def test_forward_mappings_agree_on_large_batches(case, bt, d, multi, K, pack):
    """Every lane of many workgroups live, ragged last pack, several grids per workgroup: the grid-per-workgroup forward
    must reproduce the lane-per-grid forward (same arithmetic, other summation order for the per-grid sums: 2e-6)."""
    import opf_graph_neural_solver_amd as amd
    bu, li, ge = amd.synth.synth_grids(case, bt, seed=11, device='cuda')
    torch.manual_seed(0)
    m = amd.GNS(latent_dim=d, hidden_dim=10, K=K, gamma=0.9, multiple_phi=multi).cuda()
    old = amd.get_option('fwd_mapping'), amd.get_option('gw_pack')
    outs = []
    try:
        for mapping in (1, 2):
            amd.set_option('fwd_mapping', mapping)
            amd.set_option('gw_pack', pack)
            with torch.no_grad():
                outs.append([o.double().cpu() for o in m(bu, li, ge)])
        amd.set_option('fwd_mapping', 2)
        with torch.no_grad():
            again = [o.double().cpu() for o in m(bu, li, ge)]
    finally:
        amd.set_option('fwd_mapping', old[0])
        amd.set_option('gw_pack', old[1])
    for a, b, what in zip(outs[1], outs[0], ('v', 'theta', 'total', 'last')):
        assert_close(a, b, 2e-6 if K <= 10 else 1e-4, what=what)
    for a, b in zip(again, outs[1]):
        assert torch.equal(a, b)                                 # bitwise run-to-run


@pytest.fixture
def gw_training():
    """Training-mode forward + backward on the grid-per-workgroup kernels (gns_set_option "train_mapping" = 2)."""
    import opf_graph_neural_solver_amd as amd
    old = amd.get_option('train_mapping'), amd.get_option('gw_pack')
    amd.set_option('train_mapping', 2)
    yield amd
    amd.set_option('train_mapping', old[0])
    amd.set_option('gw_pack', old[1])


@pytest.mark.parametrize('name', shallow_golden_names())
def test_parameter_gradient_matches_reference_autograd_grid_per_workgroup(name, gw_training):
    """Same check as above for the on-chip mapping: outputs of the training-mode forward and d(mean total_loss)/d(params)
    against the reference's own forward / .backward() (GNS/main.py:281-288)."""
    g = load_golden(name)
    m = _model(g)
    v, th, tot, last = m(t(g['buses']).cuda(), t(g['lines']).cuda(), t(g['generators']).cuda())
    assert_close(v.detach().cpu(), g['v'], REL, what='v')
    assert_close(th.detach().cpu(), g['theta'], REL, what='theta')
    assert_close(tot.detach().cpu(), g['total_loss'], REL, what='total_loss')
    tot.mean().backward()
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    assert_close(grad, g['grad_params'], 5e-5, abs_floor=1e-7, what='grad_params')
    none = set(str(s) for s in g['none_grad_names'])
    for n, p in m.named_parameters():
        if n in none:
            assert float(p.grad.abs().max()) == 0.0, n


@pytest.mark.parametrize('case,bt,d,multi,K,pack', [(118, 1031, 20, True, 4, 1), (118, 259, 20, True, 4, 2), (30, 777, 10, False, 3, 3),
                                                    (14, 2001, 10, True, 2, 8), (300, 67, 20, True, 3, 1)])
def test_training_mappings_agree_on_large_batches(case, bt, d, multi, K, pack, gw_training):
    """Ragged last pack, several grids per workgroup, every upstream gradient (v, theta, total, last) non-zero: the
    grid-per-workgroup pair must give the lane-per-grid pair's gradient up to fp32 summation order."""
    amd = gw_training
    bu, li, ge = amd.synth.synth_grids(case, bt, seed=13, device='cuda')
    gen = torch.Generator(device='cuda').manual_seed(3)
    wv, wt = torch.randn(bt, bu.shape[1], device='cuda', generator=gen), torch.randn(bt, bu.shape[1], device='cuda', generator=gen)
    wl = torch.rand(bt, device='cuda', generator=gen)
    grads = []
    for mapping in (1, 2):
        amd.set_option('train_mapping', mapping)
        amd.set_option('gw_pack', pack)
        torch.manual_seed(0)
        m = amd.GNS(latent_dim=d, hidden_dim=10, K=K, gamma=0.9, multiple_phi=multi).cuda()
        v, th, tot, last = m(bu, li, ge)
        (tot.mean() + (v * wv).mean() + (th * wt).mean() + (last * wl).mean()).backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).double().cpu())
    scale = float(grads[0].abs().max())
    assert scale > 0
    assert float((grads[0] - grads[1]).abs().max()) <= 3e-6 * scale


@pytest.mark.parametrize('case,bt,d,multi,K', [(118, 4099, 20, True, 4), (30, 777, 10, False, 3), (14, 20000, 10, True, 2)])
def test_weight_gradient_engines_agree_on_large_batches(case, bt, d, multi, K):
    """The goldens hold <= 4 grids; here every lane of many waves is live: the matrix-pipe contraction and the
    packed-FMA register tiles must give the same gradient up to fp32 summation order (1e-6 of max|grad|)."""
    import opf_graph_neural_solver_amd as amd
    bu, li, ge = amd.synth.synth_grids(case, bt, seed=5, device='cuda')
    grads = []
    old = amd.get_option('dw_mfma')
    old_map = amd.get_option('train_mapping')
    amd.set_option('train_mapping', 1)
    for flag in (1, 0):
        amd.set_option('dw_mfma', flag)
        torch.manual_seed(0)
        m = amd.GNS(latent_dim=d, hidden_dim=10, K=K, gamma=0.9, multiple_phi=multi).cuda()
        m(bu, li, ge)[2].mean().backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).double().cpu())
    amd.set_option('dw_mfma', old)
    amd.set_option('train_mapping', old_map)
    scale = float(grads[1].abs().max())
    assert scale > 0
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-6 * scale


@pytest.fixture(params=['lane-per-grid', 'grid-per-workgroup'])
def train_mapping(request):
    """Pin the training-mode kernels (the default picks per batch size: gns_api.hip, gw_train_pack)."""
    import opf_graph_neural_solver_amd as amd
    old = amd.get_option('train_mapping')
    amd.set_option('train_mapping', 1 if request.param == 'lane-per-grid' else 2)
    yield request.param
    amd.set_option('train_mapping', old)


def test_full_size_batch_against_oracle_sample_and_properties(train_mapping):
    """BASELINE config 2/3 scale: case30 batch 4096 and case118 batch 16384 through the fused path, under either
    training mapping.
    Checked (a) against the CPU oracle on a sample of grids, (b) bitwise run-to-run reproducibility,
    (c) batch-composition independence: a grid's outputs do not depend on its neighbours in the batch,
    (d) the batch gradient is the mean of shard gradients (the identity data parallelism relies on)."""
    import opf_graph_neural_solver_amd as amd
    from oracle import gns_oracle as orc
    for case, bt in ((30, 4096), (118, 16384)):
        torch.manual_seed(1)
        m = amd.GNS(20, 10, 4, 0.9, True).cuda()
        bu, li, ge = amd.synth.synth_grids(case, bt, seed=5, device='cuda')
        v, th, tot, last = m(bu, li, ge)
        tot.mean().backward()
        g_all = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
        flat = m.flat_parameters().detach().cpu()
        params = orc.unflatten_params(flat, 20, 10, 4, True)
        for b in (0, 1, 63, 64, bt // 2 + 7, bt - 1):
            vo, tho, toto, lasto = orc.gns_forward(params, bu[b].cpu(), li[b].cpu(), ge[b].cpu(), latent_dim=20, K=4,
                                                  gamma=0.9, multiple_phi=True)
            assert_close(v[b].detach().cpu(), vo, REL, what=f'v[{b}]')
            assert_close(th[b].detach().cpu(), tho, REL, what=f'theta[{b}]')
            assert_close(tot[b].detach().cpu(), toto, REL, what=f'total[{b}]')
        # (b) bitwise reproducibility (no atomics anywhere)
        m.zero_grad()
        v2, th2, tot2, _ = m(bu, li, ge)
        tot2.mean().backward()
        g_again = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        assert torch.equal(v, v2) and torch.equal(th, th2) and torch.equal(tot, tot2) and torch.equal(g_all, g_again)
        # (c) a slice of the batch, alone, gives bitwise the same per-grid outputs
        # (the grid-per-workgroup kernel reproduces itself bitwise on any batch composition; which kernel an evaluation call takes by
        #  itself depends on the batch - the full batch fills the chip on the lane-per-grid kernel, the slice stays on chip - and the
        #  two agree to fp32 summation order)
        old_fwd = amd.get_option('fwd_mapping')
        try:
            amd.set_option('fwd_mapping', 2)
            with torch.no_grad():
                ve, the, tote, _ = m(bu, li, ge)
                vs, ths, tots, _ = m(bu[100:229], li[100:229], ge[100:229])
            assert torch.equal(vs, ve[100:229]) and torch.equal(ths, the[100:229]) and torch.equal(tots, tote[100:229])
            amd.set_option('fwd_mapping', 0)
            with torch.no_grad():
                va, tha, tota, _ = m(bu, li, ge)
                vsa, thsa, totsa, _ = m(bu[100:229], li[100:229], ge[100:229])
        finally:
            amd.set_option('fwd_mapping', old_fwd)
        assert_close(vsa.cpu(), va[100:229].cpu(), 2e-6, what='slice vs batch v (automatic mapping)')
        assert_close(thsa.cpu(), tha[100:229].cpu(), 2e-6, what='slice vs batch theta (automatic mapping)')
        assert_close(totsa.cpu(), tota[100:229].cpu(), 2e-6, what='slice vs batch total (automatic mapping)')
        assert_close(va.cpu(), ve.cpu(), 2e-6, what='automatic vs grid-per-workgroup v')
        assert_close(ve.cpu(), v.detach().cpu(), 2e-6, what='eval vs train v')
        assert_close(the.cpu(), th.detach().cpu(), 2e-6, what='eval vs train theta')
        # training-mode slice: bitwise with the same number of workgroups per 64-grid group; a small batch alone is worked by
        # teams of workgroups (other split of the buses -> other order of the per-wave partial sums of lambda and the loss)
        old_team = amd.get_option('team')
        groups = (bt + 63) // 64
        amd.set_option('team', 4 if groups * 4 <= 256 else (2 if groups * 2 <= 256 else 1))    # what the whole batch ran with
        vt, tht, tott, _ = m(bu[100:229], li[100:229], ge[100:229])
        amd.set_option('team', old_team)
        assert torch.equal(vt, v[100:229]) and torch.equal(tht, th[100:229]) and torch.equal(tott, tot[100:229])
        vt, tht, tott, _ = m(bu[100:229], li[100:229], ge[100:229])
        assert_close(vt.detach().cpu(), v[100:229].detach().cpu(), 2e-6, what='slice in teams v')
        assert_close(tott.detach().cpu(), tot[100:229].detach().cpu(), 2e-6, what='slice in teams total')
        # (d) mean of two half-batch gradients == full-batch gradient
        halves = []
        for lo, hi in ((0, bt // 2), (bt // 2, bt)):
            m.zero_grad()
            _, _, t_h, _ = m(bu[lo:hi], li[lo:hi], ge[lo:hi])
            t_h.mean().backward()
            halves.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone())
        assert_close((0.5 * (halves[0] + halves[1])).cpu(), g_all.cpu(), 2e-5, abs_floor=1e-7, what='half-batch mean')


@pytest.mark.parametrize('name', deep_golden_names())
@pytest.mark.parametrize('mapping', ['lane-per-grid', 'grid-per-workgroup'])
def test_reference_run_configurations_pinned_by_goldens(name, mapping, capsys):
    """The reference's own configurations - K=15, three phis (main.py:209-213) and the constructor defaults K=30, one phi
    (main.py:108) - against goldens produced by the reference itself.  The measured error of the HIP path against the
    reference's fp32 outputs and gradients is printed next to the conditioning bound (distance of the reference-order fp32
    oracle from the fp64 oracle on the same inputs: what fp32 arithmetic in ANY order costs at this depth), and must stay
    within max(1e-5, 2 x bound) for outputs and max(5e-5, 2 x bound) for gradients.  The gradient check also pins the
    analytically dropped delta_q adjoint (gns_backward.hip:13-14) at depth."""
    import opf_graph_neural_solver_amd as amd
    from oracle import gns_oracle as orc
    g = load_golden(name)
    c = cfg_of(g)
    old = amd.get_option('fwd_mapping'), amd.get_option('train_mapping')
    code = 1 if mapping == 'lane-per-grid' else 2
    amd.set_option('fwd_mapping', code); amd.set_option('train_mapping', code)
    try:
        m = _model(g)
        bu, li, ge = t(g['buses']).cuda(), t(g['lines']).cuda(), t(g['generators']).cuda()
        with torch.no_grad():
            ev = [o.cpu().numpy() for o in m(bu, li, ge)]
        v, th, tot, last = m(bu, li, ge)
        tot.mean().backward()
        grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    finally:
        amd.set_option('fwd_mapping', old[0]); amd.set_option('train_mapping', old[1])
    okw = dict(latent_dim=c['latent_dim'], K=c['K'], gamma=c['gamma'], multiple_phi=c['multiple_phi'])
    flat = t(g['params'])
    nb = g['buses'].shape[0]
    o32, o64, g32, g64 = [], [], torch.zeros_like(flat), torch.zeros_like(flat, dtype=torch.float64)
    for b in range(nb):
        f32, f64 = flat.clone().requires_grad_(True), flat.double().requires_grad_(True)
        a = orc.gns_forward(orc.unflatten_params(f32, c['latent_dim'], c['hidden_dim'], c['K'], c['multiple_phi']),
                            t(g['buses'][b]), t(g['lines'][b]), t(g['generators'][b]), **okw)
        d_ = orc.gns_forward(orc.unflatten_params(f64, c['latent_dim'], c['hidden_dim'], c['K'], c['multiple_phi']),
                             t(g['buses'][b]).double(), t(g['lines'][b]).double(), t(g['generators'][b]).double(), **okw)
        (a[2] / nb).backward(); (d_[2] / nb).backward()
        g32 += f32.grad; g64 += f64.grad
        o32.append([x.detach().numpy() for x in a]); o64.append([x.detach().numpy() for x in d_])
    report = []
    for i, (key, mine_ev, mine_tr) in enumerate((('v', ev[0], v), ('theta', ev[1], th), ('total_loss', ev[2], tot), ('last_loss', ev[3], last))):
        ref = g[key]
        bound = rel_err(np.stack([o[i] for o in o32]), np.stack([o[i] for o in o64]))
        e_ev, e_tr = rel_err(mine_ev, ref), rel_err(mine_tr.detach().cpu().numpy(), ref)
        report.append(f'{key}: eval {e_ev:.2e} train {e_tr:.2e} (fp32 conditioning bound {bound:.2e})')
        assert max(e_ev, e_tr) <= max(REL, 2.0 * bound), report[-1]
    gbound = rel_err(g32.numpy(), g64.numpy())
    e_g = rel_err(grad, g['grad_params'])
    report.append(f'grad_params: {e_g:.2e} (fp32 conditioning bound {gbound:.2e})')
    assert e_g <= max(5e-5, 2.0 * gbound), report[-1]
    with capsys.disabled():
        print(f'\n[{name} | {mapping}] error vs the REFERENCE\'s fp32 outputs: ' + '; '.join(report))


def test_resident_batch_is_packed_once_and_repacked_when_it_changes():
    """cache_packed_inputs: the same tensors passed again are not re-packed (gns_prepack); results are the same bits as with
    per-call packing, and an in-place change of the inputs (version counter) invalidates the cached layout."""
    import opf_graph_neural_solver_amd as amd
    torch.manual_seed(0)
    m = amd.GNS(20, 10, 3, 0.9, True).cuda()
    bu, li, ge = amd.synth.synth_grids(118, 300, seed=4, device='cuda')
    old_map = amd.get_option('train_mapping')
    amd.set_option('train_mapping', 1)                 # the lane-per-grid kernels are the ones that read the packed layout

    def run():
        m.zero_grad()
        out = m(bu, li, ge)
        out[2].mean().backward()
        return [o.detach().clone() for o in out] + [torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()]

    base = run()
    m.cache_packed_inputs = True
    first, again = run(), run()
    assert m._pack_cache is not None
    for a, b, c in zip(base, first, again):
        assert torch.equal(a, b) and torch.equal(a, c)
    bu[:, :, 2] *= 1.01                                          # in place: same storage, new version
    changed = run()
    m.cache_packed_inputs = False
    fresh = run()
    amd.set_option('train_mapping', old_map)
    assert not torch.equal(changed[2], base[2])
    for a, b in zip(changed, fresh):
        assert torch.equal(a, b)
    # the grid-per-workgroup kernels read the caller's tensors in place: nothing is packed for them
    m._pack_cache = None
    m.cache_packed_inputs = True
    amd.set_option('train_mapping', 2)
    try:
        m(bu, li, ge)[2].mean().backward()
    finally:
        amd.set_option('train_mapping', old_map)
    assert m._pack_cache is None


@pytest.mark.parametrize('mapping', ['lane-per-grid', 'grid-per-workgroup'])
def test_mid_size_case_between_118_and_300_buses(mapping):
    """A 200-bus synthetic case: the lane-per-grid forward keeps its (v, theta) plane in 100 KB of dynamic LDS (the > 64 KB
    opt-in path that case118 and case300 both miss); four waves per grid in the grid-per-workgroup kernels.  Outputs and
    gradients against the CPU oracle."""
    import opf_graph_neural_solver_amd as amd
    from oracle import gns_oracle as orc
    code = 1 if mapping == 'lane-per-grid' else 2
    old = amd.get_option('fwd_mapping'), amd.get_option('train_mapping')
    amd.set_option('fwd_mapping', code); amd.set_option('train_mapping', code)
    try:
        torch.manual_seed(6)
        m = amd.GNS(20, 10, 3, 0.9, True).cuda()
        bu, li, ge = amd.synth.synth_grids(200, 70, seed=8, device='cuda')
        with torch.no_grad():
            ev = m(bu, li, ge)
        v, th, tot, last = m(bu, li, ge)
        tot.mean().backward()
        grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu()
    finally:
        amd.set_option('fwd_mapping', old[0]); amd.set_option('train_mapping', old[1])
    flat = m.flat_parameters().detach().cpu()
    g_o = torch.zeros_like(flat)
    for b in (0, 33, 69):
        f = flat.clone().requires_grad_(True)
        o = orc.gns_forward(orc.unflatten_params(f, 20, 10, 3, True), bu[b].cpu(), li[b].cpu(), ge[b].cpu(), latent_dim=20, K=3, gamma=0.9, multiple_phi=True)
        for mine, ref, what in ((v[b], o[0], 'v'), (th[b], o[1], 'theta'), (tot[b], o[2], 'total'), (ev[0][b], o[0], 'v eval'), (ev[1][b], o[1], 'theta eval')):
            assert_close(mine.detach().cpu(), ref.detach(), REL, what=f'{what}[{b}]')
    for b in range(70):
        f = flat.clone().requires_grad_(True)
        (orc.gns_forward(orc.unflatten_params(f, 20, 10, 3, True), bu[b].cpu(), li[b].cpu(), ge[b].cpu(), latent_dim=20, K=3, gamma=0.9, multiple_phi=True)[2] / 70.0).backward()
        g_o += f.grad
    assert_close(grad, g_o, 5e-5, abs_floor=1e-7, what='grad_params')


def test_flat_optimizer_is_the_reference_optimizer_in_one_launch():
    """training.make_optimizer on a GPU model runs torch.optim.Adam on the ONE flat parameter buffer: the same weights as
    torch.optim.Adam(model.parameters()) after a few steps, the in-place guard still fires between forward and backward."""
    import opf_graph_neural_solver_amd as amd
    bu, li, ge = amd.synth.synth_grids(30, 64, seed=12, device='cuda')
    ws = []
    for flat in (True, False):
        torch.manual_seed(7)
        m = amd.GNS(20, 10, 3, 0.9, True).cuda()
        opt = amd.training.make_optimizer(m, flat=flat)
        assert isinstance(opt, amd.training.FlatOptimizer) == flat
        if flat:
            # the flattening itself is checked with torch's own kernel on the flat tensor: identical arithmetic, so even the
            # weights whose gradient is rounding noise (Adam turns those into +-lr) must agree; the library's kernel is
            # checked on given gradients in test_library_adam_kernel_is_torch_adam_and_shares_its_state_dict
            assert opt._use_native
            opt = amd.training.FlatOptimizer(m, torch.optim.Adam, native=False, lr=1e-3, fused=True)
        for _ in range(4):
            opt.zero_grad()
            m(bu, li, ge)[2].mean().backward()
            opt.step()
        ws.append(torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone())
    assert float((ws[0] - ws[1]).abs().max()) <= 1e-7 * float(ws[1].abs().max())
    tot = m(bu, li, ge)[2].mean()
    torch.manual_seed(7)
    m2 = amd.GNS(20, 10, 3, 0.9, True).cuda()
    o2 = amd.training.make_optimizer(m2, flat=True)
    m2(bu, li, ge)[2].mean().backward(); o2.step()
    t2 = m2(bu, li, ge)[2].mean()
    t2.backward(); o2.step()                               # normal order works
    t3 = m2(bu, li, ge)[2].mean()
    o2.step()                                              # a step between forward and backward must be refused
    with pytest.raises(amd.GNSError, match='modified in place'):
        t3.backward()


def test_library_adam_kernel_is_torch_adam_and_shares_its_state_dict():
    """gns_adam_step (include/gns_hip.h) against torch.optim.Adam (the optimiser of GNS/main.py:241-243,290) on the same flat
    tensor and the same GIVEN gradient sequence (zeros, tiny and ordinary values): weights and both moment estimates after 6
    steps; then the two optimisers swap their state_dicts and continue to the same weights."""
    import opf_graph_neural_solver_amd as amd
    torch.manual_seed(3)
    m0 = amd.GNS(10, 10, 2, 0.9, True).cuda()
    n = m0.flat_parameters().numel()
    gen = torch.Generator(device='cuda').manual_seed(11)
    grads = []
    for t_ in range(7):
        g = torch.randn(n, device='cuda', generator=gen) * torch.logspace(-12, 0, n, device='cuda')
        g[::7] = 0.0
        grads.append(g)
    w0 = m0.flat_parameters().detach().clone()

    def give(m, g):
        off = 0
        for p in m.parameters():
            p.grad = g[off:off + p.numel()].view(p.shape)
            off += p.numel()

    runs = {}
    for native in (True, False):
        torch.manual_seed(3)
        m = amd.GNS(10, 10, 2, 0.9, True).cuda()
        assert torch.equal(m.flat_parameters().detach(), w0)
        opt = amd.training.FlatOptimizer(m, torch.optim.Adam, native=native, lr=1e-3)
        assert opt._use_native == native
        for t_ in range(6):
            give(m, grads[t_]); opt.step()
        st = opt.state_dict()['state'][0]
        runs[native] = (m.flat_parameters().detach().clone(), st['exp_avg'].clone(), st['exp_avg_sq'].clone(), float(st['step']), m, opt)
    assert float((runs[True][0] - w0).abs().max()) > 1e-3                         # the steps did move the weights
    for a, b, what in zip(runs[True][:3], runs[False][:3], ('weights', 'exp_avg', 'exp_avg_sq')):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), what
        # element-wise too (the moments span 24 orders of magnitude here); torch contracts its update into FMAs, the library
        # is built with -ffp-contract=off, so single elements differ by a few ulp per step
        assert bool(((a - b).abs() <= 1e-4 * b.abs() + 1e-7 * b.abs().max()).all()) or what == 'weights', what
    assert runs[True][3] == runs[False][3] == 6.0
    sd_native, sd_torch = runs[True][5].state_dict(), runs[False][5].state_dict()
    runs[True][5].load_state_dict(sd_torch); runs[False][5].load_state_dict(sd_native)
    out = []
    for native in (True, False):
        m, opt = runs[native][4], runs[native][5]
        give(m, grads[6]); opt.step()
        out.append(m.flat_parameters().detach().clone())
    assert float((out[0] - out[1]).abs().max()) <= 2e-6 * float(out[1].abs().max())


def test_flat_grad_mode_delivers_the_same_gradient_as_one_tensor():
    """GNS.flat_grad (switched on by training.make_optimizer): the backward hands ONE tensor, flat_leaf().grad, bitwise equal to
    the per-parameter .grad views of the default mode; gradients accumulate across backward calls like .grad does; zero_grad
    clears it; a train_step with the flat optimiser moves the weights exactly like the default mode with the same optimiser."""
    import opf_graph_neural_solver_amd as amd
    bu, li, ge = amd.synth.synth_grids(30, 70, seed=4, device='cuda')
    torch.manual_seed(2)
    m = amd.GNS(20, 10, 3, 0.9, True).cuda()
    m(bu, li, ge)[2].mean().backward()
    g_views = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    m.zero_grad()
    m.flat_grad = True
    out = m(bu, li, ge)
    (out[2].mean() + 0.5 * out[3].sum()).backward()
    m.flat_grad = False
    out = m(bu, li, ge)
    (out[2].mean() + 0.5 * out[3].sum()).backward()
    g_ref2 = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    assert torch.equal(m.flat_leaf().grad, g_ref2)
    assert all(p.grad is not None for p in m.parameters())
    m.zero_grad()
    assert m.flat_leaf().grad is None
    m.flat_grad = True
    m(bu, li, ge)[2].mean().backward()
    assert torch.equal(m.flat_leaf().grad, g_views) and all(p.grad is None for p in m.parameters())
    m(bu, li, ge)[2].mean().backward()                      # accumulates
    assert torch.equal(m.flat_leaf().grad, g_views + g_views)
    assert torch.equal(amd.dist.flat_gradient(m), m.flat_leaf().grad)
    m.zero_grad()
    # training step: flat mode against the default mode, same optimiser arithmetic (torch's Adam on the flat buffer)
    ws = []
    for mode in (True, False):
        torch.manual_seed(2)
        mm = amd.GNS(20, 10, 3, 0.9, True).cuda()
        mm.flat_grad = mode
        opt = amd.training.FlatOptimizer(mm, torch.optim.Adam, native=False, lr=1e-3)
        for _ in range(3):
            amd.training.train_step(mm, opt, bu, li, ge)
        ws.append(mm.flat_parameters().detach().clone())
    assert torch.equal(ws[0], ws[1])
    mo = amd.GNS(20, 10, 3, 0.9, True).cuda()
    assert not mo.flat_grad
    amd.training.make_optimizer(mo)
    assert mo.flat_grad                                      # the package's own optimiser switches it on


def test_in_place_parameter_update_between_forward_and_backward_raises():
    """forward / optimizer.step() (or any in-place parameter write) / backward mixes weights packed by the forward with the
    live buffer; torch autograd raises in that situation and so must the fused path."""
    import opf_graph_neural_solver_amd as amd
    torch.manual_seed(1)
    m = amd.GNS(10, 10, 2, 0.9, False).cuda()
    bu, li, ge = amd.synth.synth_grids(14, 4, seed=2, device='cuda')
    tot = m(bu, li, ge)[2].mean()
    with torch.no_grad():
        next(m.parameters()).add_(1.0)
    with pytest.raises(amd.GNSError, match='modified in place'):
        tot.backward()
    m.zero_grad()
    m(bu, li, ge)[2].mean().backward()                       # an untouched forward/backward pair still works
    assert all(p.grad is not None for p in m.parameters())


@pytest.mark.parametrize('optim', ['Adam', 'SGD'])
def test_reference_training_loop_runs_unchanged_on_a_cpu_resident_model(optim):
    """The literal loop of GNS/main.py:274-291 - model built on the CPU and never moved (main.py:227, the .to('cuda') at
    :230-233 is commented out), CPU tensors from load_all_grids, per-grid keyword calls, `losses[i % batch_size] = loss` into a
    zeros tensor and torch.mean(losses) exactly as main.py:277-284 writes them,
    torch.optim.Adam(model.parameters()) - with only the import changed.  The kernels run on the GPU on a mirror of the
    parameters.  After two epochs the weights are compared with the same loop on the CPU oracle (torch autograd + the same
    optimiser), tolerance 1e-5 of max|w|:
      * SGD: every weight.
      * Adam (the reference's optimiser): the normalised update lr * m / (sqrt(v) + eps) turns a gradient element that is
        rounding noise in fp32 (cancellation, or the linear1 column fed by delta_q, which is noise in the reference too:
        main.py:83,103) into +-lr whatever its size, in ANY two implementations; so at least 99 % of the weights must
        agree to 1e-5, none may differ by more than the 4 steps x lr an opposite sign can produce, and the gradient of
        the first step is compared directly (5e-5 of max|g|)."""
    import opf_graph_neural_solver_amd as amd
    from oracle import gns_oracle as orc
    d, h, K, multi, nr, bs, epochs = 20, 10, 3, True, 8, 4, 2
    B, L, G = amd.get_BLG()
    bu, li, ge = amd.synth.synth_grids(14, nr, seed=31)                        # CPU tensors
    torch.manual_seed(5)
    model = amd.GNS(latent_dim=d, hidden_dim=h, K=K, gamma=0.9, multiple_phi=multi)     # stays on the CPU
    flat0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    mk = (lambda ps: torch.optim.Adam(ps, lr=0.001)) if optim == 'Adam' else (lambda ps: torch.optim.SGD(ps, lr=0.01))
    optimizer = mk(model.parameters())
    first_grad = None
    for epoch in range(epochs):
        for batch in range(0, nr, bs):
            losses = torch.zeros(bs)                       # the reference's own collection: index assignment into a zeros tensor
            last_losses = torch.zeros(bs)                  # (main.py:277-284), not a stacked list
            for i in range(batch, batch + bs):
                v, theta, loss, last_loss = model(buses=bu[i], lines=li[i], generators=ge[i], B=B, L=L, G=G)
                losses[i % bs] = loss
                last_losses[i % bs] = last_loss.data
            total_loss = torch.mean(losses)
            total_loss.backward()
            if first_grad is None:
                first_grad = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
            optimizer.step()
            optimizer.zero_grad()
    assert all(p.device.type == 'cpu' for p in model.parameters()) and v.device.type == 'cpu'
    mine = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    fo = flat0.clone().requires_grad_(True)                                     # the same loop on the oracle
    opt_o = mk([fo])
    first_grad_o = None
    for epoch in range(epochs):
        for batch in range(0, nr - bs + 1, bs):
            losses = []
            for i in range(batch, batch + bs):
                losses.append(orc.gns_forward(orc.unflatten_params(fo, d, h, K, multi), bu[i], li[i], ge[i], latent_dim=d, K=K,
                                              gamma=0.9, multiple_phi=multi)[2])
            torch.mean(torch.stack(losses)).backward()
            if first_grad_o is None:
                first_grad_o = fo.grad.clone()
            opt_o.step(); opt_o.zero_grad()
    ref = fo.detach()
    assert float((mine - flat0).abs().max()) > 1e-4                            # the weights really moved
    assert_close(first_grad.numpy(), first_grad_o.numpy(), 5e-5, abs_floor=1e-7, what='gradient of the first step')
    tol = 1e-7 + 1e-5 * float(ref.abs().max())
    diff = (mine - ref).abs()
    if optim == 'SGD':
        assert float(diff.max()) <= tol, f'SGD weights: max|diff| {float(diff.max()):.3e} > {tol:.3e}'
    else:
        frac = float((diff <= tol).double().mean())
        assert frac >= 0.99, f'Adam: only {frac:.4f} of the weights within 1e-5'
        assert float(diff.max()) <= 2.0 * 4 * 0.001 + tol


def test_input_producer_prepare_grids_on_device_matches_reference_goldens(golden_dir):
    """the batched device-side prepare_grid (GNS/utils.py:17-41) against the reference's own function."""
    import glob, os
    import opf_graph_neural_solver_amd as amd
    files = sorted(glob.glob(os.path.join(golden_dir, 'prepare_*.npz')))
    assert files
    for f in files:
        z = np.load(f, allow_pickle=False)
        bu, li, ge = amd.prepare_grids(t(z['bus']).cuda(), t(z['branch']).cuda(), t(z['gen']).cuda())
        assert bu.is_cuda and bu.dtype == torch.float32
        for mine, key in ((bu, 'buses'), (li, 'lines'), (ge, 'generators')):
            assert_close(mine.cpu(), z[key], 1e-6, abs_floor=1e-7, what=f'{os.path.basename(f)} {key}')


def test_counter_based_synthetic_grids_shards_and_devices_agree():
    """SURVEY 8(d,f3): grids keyed by (seed, global grid index): on the device a shard reproduces the rows of the unsharded
    batch bit for bit (the CPU synthesis draws the same numbers and agrees to rounding); the forward on a shard therefore
    reproduces its rows bitwise."""
    import opf_graph_neural_solver_amd as amd
    full = amd.synth.synth_grids(118, 200, seed=77, device='cuda')
    part = amd.synth.synth_grids(118, 64, seed=77, device='cuda', first_index=100)
    cpu = amd.synth.synth_grids(118, 64, seed=77, device='cpu', first_index=100)
    for a, b, c in zip(full, part, cpu):
        assert torch.equal(a[100:164], b)                         # shard == rows of the whole, bit for bit
        assert_close(b.cpu(), c, 1e-6, abs_floor=1e-7, what='device vs CPU synthesis')   # same draws; float ops may round differently
    torch.manual_seed(0)
    m = amd.GNS(20, 10, 4, 0.9, True).cuda()
    with torch.no_grad():
        of, op = m(*full), m(*part)
    for a, b in zip(of, op):
        assert torch.equal(a[100:164], b)


@pytest.mark.parametrize('kw', [dict(), dict(latent_dim=10, hidden_dim=10, K=15, gamma=0.9, multiple_phi=True)],
                         ids=['ctor-defaults-K30-single-phi', 'main.py-run-config-K15-multi-phi'])
def test_reference_default_configurations_against_oracle(kw):
    """`GNS()` as the reference constructs it by default (main.py:108: d=10, h=10, K=30, one phi) and as main.py:210-214
    runs it (K=15, three phis), on case14 / case30 grids.  No golden holds these K (the oracle is pinned by the goldens
    at K<=10).  With random weights the K=30 recursion amplifies fp32 rounding: the reference's OWN fp32 result is up
    to 3e-5 from the exact (fp64) one (tools/gpu_depth_conditioning.py), so the 1e-5 bar of the K=4 configs cannot be
    asked against an fp32 answer.  Checked instead: the HIP result is no further from the fp64 oracle than
    max(1e-5, 4 x the distance of the reference-order fp32 oracle from it); gradients likewise with 1e-4.  (Both
    configurations are pinned directly against the reference by test_reference_run_configurations_pinned_by_goldens.)"""
    import opf_graph_neural_solver_amd as amd
    from oracle import gns_oracle as orc
    torch.manual_seed(3)
    m = amd.GNS(**kw).cuda()
    d, K, multi = m.latent_dim, m.K, m.multiple_phis
    okw = dict(latent_dim=d, K=K, gamma=0.9, multiple_phi=multi)

    def dist(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max())

    for case in (14, 30):
        bu, li, ge = amd.synth.synth_grids(case, 3, seed=11, device='cuda')
        m.zero_grad()
        v, th, tot, last = m(bu, li, ge)
        tot.mean().backward()
        grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu()
        flat = m.flat_parameters().detach().cpu()
        g32, g64 = torch.zeros_like(flat), torch.zeros_like(flat, dtype=torch.float64)
        for b in range(3):
            f32, f64 = flat.clone().requires_grad_(True), flat.double().requires_grad_(True)
            o32 = orc.gns_forward(orc.unflatten_params(f32, d, 10, K, multi), bu[b].cpu(), li[b].cpu(), ge[b].cpu(), **okw)
            o64 = orc.gns_forward(orc.unflatten_params(f64, d, 10, K, multi), bu[b].cpu().double(), li[b].cpu().double(),
                                  ge[b].cpu().double(), **okw)
            (o32[2] / 3.0).backward()
            (o64[2] / 3.0).backward()
            g32 += f32.grad
            g64 += f64.grad
            for name, mine, i in (('v', v[b], 0), ('theta', th[b], 1), ('total', tot[b], 2), ('last', last[b], 3)):
                ref_noise = dist(o32[i].detach(), o64[i].detach())
                mine_err = dist(mine.detach().cpu(), o64[i].detach())
                assert mine_err <= max(REL, 4.0 * ref_noise), f'case{case} {name}[{b}]: {mine_err:.2e} vs fp32 reference noise {ref_noise:.2e}'
        ref_noise, mine_err = dist(g32, g64), dist(grad, g64)
        assert mine_err <= max(1e-4, 4.0 * ref_noise), f'case{case} grad: {mine_err:.2e} vs fp32 reference noise {ref_noise:.2e}'


def test_evaluation_mode_saves_nothing_and_matches_training_mode():
    """Under torch.no_grad() (evaluate.py:79) the forward saves nothing for a backward (by default it runs the
    grid-per-workgroup kernel: state on chip); in training mode it saves the per-step states.  Same results to fp32
    summation order - the same bits when both run the lane-per-grid kernels - and the evaluation call must not
    allocate the saved-state workspace."""
    import opf_graph_neural_solver_amd as amd
    torch.manual_seed(2)
    m = amd.GNS(20, 10, 6, 0.9, True).cuda()
    bu, li, ge = amd.synth.synth_grids(118, 1500, seed=3, device='cuda')
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated()
    with torch.no_grad():
        ev = m(bu, li, ge)
    torch.cuda.synchronize(); peak_eval = torch.cuda.max_memory_allocated() - base
    assert not ev[2].requires_grad
    torch.cuda.reset_peak_memory_stats()
    old_train = amd.get_option('train_mapping')
    amd.set_option('train_mapping', 1)
    try:
        tr = m(bu, li, ge)
    finally:
        amd.set_option('train_mapping', old_train)
    torch.cuda.synchronize(); peak_train = torch.cuda.max_memory_allocated() - base
    assert tr[2].requires_grad
    for a, b in zip(ev, tr):
        assert_close(a.cpu(), b.detach().cpu(), 2e-6, what='evaluation vs training mode')
    assert peak_eval < 0.7 * peak_train, (peak_eval, peak_train)
    # with both modes on the lane-per-grid kernels the outputs are the same bits
    old = amd.get_option('fwd_mapping')
    amd.set_option('fwd_mapping', 1)
    try:
        with torch.no_grad():
            ev1 = m(bu, li, ge)
    finally:
        amd.set_option('fwd_mapping', old)
    for a, b in zip(ev1, tr):
        assert torch.equal(a, b.detach())


def test_reference_style_per_grid_training_loop_equals_the_batched_step():
    """main.py:277-291 as written: one model call per grid (2-D CPU tensors, keyword arguments), the losses collected in
    a Python list, `torch.mean(torch.stack(losses)).backward()`.  The gradients must equal those of ONE batched call."""
    import opf_graph_neural_solver_amd as amd
    B, L, G = amd.get_BLG()
    torch.manual_seed(4)
    m = amd.GNS(latent_dim=20, hidden_dim=10, K=4, gamma=0.9, multiple_phi=True).cuda()
    bu, li, ge = amd.synth.synth_grids(14, 6, seed=9)            # CPU tensors, like utils.load_all_grids returns
    losses, last_losses = [], []
    for i in range(6):
        v, theta, loss, last_loss = m(buses=bu[i], lines=li[i], generators=ge[i], B=B, L=L, G=G)
        assert loss.requires_grad and loss.dim() == 0
        losses.append(loss)
        last_losses.append(last_loss)
    torch.mean(torch.stack(losses)).backward()
    g_loop = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    m.zero_grad()
    _, _, tot, _ = m(bu.cuda(), li.cuda(), ge.cuda(), B, L, G)
    tot.mean().backward()
    g_batch = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert_close(torch.stack(losses).detach().cpu(), tot.detach().cpu(), 1e-6, what='per-grid losses')
    assert_close(g_loop.cpu(), g_batch.cpu(), 2e-5, abs_floor=1e-7, what='loop vs batched gradient')


def test_ragged_batches_and_upstream_gradients():
    """Batch sizes that do not fill a 64-grid wave (1, 63, 65, 130) and gradients flowing in through v, theta and
    last_loss - checked against autograd on the CPU oracle."""
    import opf_graph_neural_solver_amd as amd
    from oracle import gns_oracle as orc
    torch.manual_seed(7)
    m = amd.GNS(10, 10, 3, 0.9, False).cuda()
    flat = m.flat_parameters().detach().cpu()
    for bt in (1, 63, 65, 130):
        bu, li, ge = amd.synth.synth_grids(14, bt, seed=bt)
        wv, wth = torch.randn(bt, 14), torch.randn(bt, 14)
        m.zero_grad()
        v, th, tot, last = m(bu.cuda(), li.cuda(), ge.cuda())
        loss = (v * wv.cuda()).sum() + (th * wth.cuda()).sum() + 0.3 * tot.sum() + 0.7 * last.sum()
        loss.backward()
        grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu()
        fo = flat.clone().requires_grad_(True)
        po = orc.unflatten_params(fo, 10, 10, 3, False)
        acc = 0.
        for b in range(bt):
            vo, tho, toto, lasto = orc.gns_forward(po, bu[b], li[b], ge[b], latent_dim=10, K=3, gamma=0.9, multiple_phi=False)
            acc = acc + (vo * wv[b]).sum() + (tho * wth[b]).sum() + 0.3 * toto + 0.7 * lasto
            assert_close(v[b].detach().cpu(), vo.detach(), REL, what='v')
        acc.backward()
        assert_close(grad, fo.grad, 5e-5, abs_floor=1e-6, what=f'grad bt={bt}')


def test_error_behaviour_on_device():
    import opf_graph_neural_solver_amd as amd
    m = amd.GNS(20, 10, 2, 0.9, True).cuda()
    bu, li, ge = amd.synth.synth_grids(14, 4, seed=0, device='cuda')
    bad = li.clone(); bad[2, 3, 1] = 5.0          # one grid with a different t_bus
    with pytest.raises(ValueError):
        m(bu, bad, ge)
    bad = li.clone(); bad[:, 0, 0] = 99.0         # bus id out of range
    with pytest.raises(ValueError):
        m(bu, bad, ge)
    with pytest.raises(ValueError):
        m(bu.double(), li, ge)
    torch.manual_seed(0)
    cpu_model = amd.GNS(20, 10, 2, 0.9, True)      # CPU-resident, like the reference's (main.py:227): runs on the GPU via a mirror
    torch.manual_seed(0)
    gpu_model = amd.GNS(20, 10, 2, 0.9, True).cuda()
    with torch.no_grad():
        oc, og = cpu_model(bu.cpu(), li.cpu(), ge.cpu()), gpu_model(bu, li, ge)
    assert oc[0].device.type == 'cpu' and torch.equal(oc[0], og[0].cpu()) and torch.equal(oc[2], og[2].cpu())
    with pytest.raises(amd.GNSError):
        amd.GNS(24, 10, 2, 0.9, True).cuda()(bu, li, ge)   # no compiled kernel holds latent_dim 24 (narrower models run zero-padded)


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()


def test_more_groups_than_backward_workgroups():
    """20 000 case14 grids = 313 wave-groups > the 256 persistent backward workgroups: slabs accumulate across groups."""
    import opf_graph_neural_solver_amd as amd
    torch.manual_seed(3)
    m = amd.GNS(10, 10, 2, 0.9, True).cuda()
    bt = 20000
    bu, li, ge = amd.synth.synth_grids(14, bt, seed=9, device='cuda')
    _, _, tot, _ = m(bu, li, ge)
    tot.mean().backward()
    g_all = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    parts = []
    for lo, hi in ((0, 8000), (8000, 20000)):
        m.zero_grad()
        _, _, t_h, _ = m(bu[lo:hi], li[lo:hi], ge[lo:hi])
        t_h.sum().backward()
        parts.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone())
    assert_close(((parts[0] + parts[1]) / bt).cpu(), g_all.cpu(), 2e-5, abs_floor=1e-7, what='20000-grid gradient')


def test_config5_shape_case300_K10_batched():
    """BASELINE config 5 shape (case300, K=10, multiple_phi) at a batch spanning several wave-groups, forward and
    gradient against the CPU oracle on a sample (autograd through the oracle costs ~1 s per grid at this size)."""
    import opf_graph_neural_solver_amd as amd
    from oracle import gns_oracle as orc
    torch.manual_seed(5)
    m = amd.GNS(20, 10, 10, 0.9, True).cuda()
    bt = 200
    bu, li, ge = amd.synth.synth_grids(300, bt, seed=12)
    v, th, tot, last = m(bu.cuda(), li.cuda(), ge.cuda())
    w = torch.zeros(bt)
    sample = [0, 63, 64, 199]
    w[sample] = 1.0
    (tot * w.cuda()).sum().backward()
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu()
    flat = m.flat_parameters().detach().cpu()
    fo = flat.clone().requires_grad_(True)
    po = orc.unflatten_params(fo, 20, 10, 10, True)
    acc = 0.
    for b in sample:
        vo, tho, toto, lasto = orc.gns_forward(po, bu[b], li[b], ge[b], latent_dim=20, K=10, gamma=0.9, multiple_phi=True)
        assert_close(v[b].detach().cpu(), vo.detach(), REL, what=f'v[{b}]')
        assert_close(th[b].detach().cpu(), tho.detach(), REL, what=f'theta[{b}]')
        assert_close(last[b].detach().cpu(), lasto.detach(), REL, what=f'last[{b}]')
        acc = acc + toto
    acc.backward()
    assert_close(grad, fo.grad, 5e-5, abs_floor=1e-6, what='grad case300 K=10')


def test_config5_per_gpu_batch_case300_x_8192_K10():
    """BASELINE config 5 at its per-GPU size (65 536 / 8 = 8 192 case300 grids, K=10, three phis): 128 wave-groups worked by
    teams of two workgroups.  Forward outputs and the gradient of a weighted loss against the CPU oracle on a sample of grids
    from both ends and the middle of the batch; run-to-run bitwise reproducibility; the batch gradient as the mean of the two
    half-batch gradients (which run as 64 groups x 4 workgroups)."""
    import opf_graph_neural_solver_amd as amd
    from oracle import gns_oracle as orc
    torch.manual_seed(6)
    m = amd.GNS(20, 10, 10, 0.9, True).cuda()
    bt = 8192
    bu, li, ge = amd.synth.synth_grids(300, bt, seed=15, device='cuda')
    v, th, tot, last = m(bu, li, ge)
    assert all(torch.isfinite(x).all() for x in (v, th, tot, last))
    sample = [0, 4097, bt - 1]
    w = torch.zeros(bt, device='cuda')
    w[sample] = 1.0
    (tot * w).sum().backward()
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu()
    fo = m.flat_parameters().detach().cpu().clone().requires_grad_(True)
    po = orc.unflatten_params(fo, 20, 10, 10, True)
    acc = 0.
    for b in sample:
        vo, tho, toto, lasto = orc.gns_forward(po, bu[b].cpu(), li[b].cpu(), ge[b].cpu(), latent_dim=20, K=10, gamma=0.9, multiple_phi=True)
        assert_close(v[b].detach().cpu(), vo.detach(), REL, what=f'v[{b}]')
        assert_close(th[b].detach().cpu(), tho.detach(), REL, what=f'theta[{b}]')
        assert_close(tot[b].detach().cpu(), toto.detach(), REL, what=f'total[{b}]')
        acc = acc + toto
    acc.backward()
    assert_close(grad, fo.grad, 5e-5, abs_floor=1e-6, what='grad case300 K=10, 8192 grids')
    m.zero_grad()
    v1, th1, tot1, _ = m(bu, li, ge)
    tot1.mean().backward()
    g_all = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    m.zero_grad()
    v2, th2, tot2, _ = m(bu, li, ge)
    tot2.mean().backward()
    g_again = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    assert torch.equal(v1, v2) and torch.equal(th1, th2) and torch.equal(tot1, tot2) and torch.equal(g_all, g_again)
    halves = []
    for lo, hi in ((0, bt // 2), (bt // 2, bt)):
        m.zero_grad()
        _, _, t_h, _ = m(bu[lo:hi], li[lo:hi], ge[lo:hi])
        t_h.mean().backward()
        halves.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone())
    assert_close((0.5 * (halves[0] + halves[1])).cpu(), g_all.cpu(), 2e-5, abs_floor=1e-7, what='half-batch mean')


def test_c_abi_calls_are_graph_capturable():
    """include/gns_hip.h promises no allocation / synchronisation / host copies inside gns_forward: capture an inference
    call into a HIP graph (torch.cuda.graph), replay it on new inputs and compare with the eager result."""
    import opf_graph_neural_solver_amd as amd
    torch.manual_seed(2)
    m = amd.GNS(20, 10, 4, 0.9, True).cuda()
    m.topology_check = 'first'
    bu, li, ge = amd.synth.synth_grids(30, 320, seed=1, device='cuda')
    bu2, li2, ge2 = amd.synth.synth_grids(30, 320, seed=2, device='cuda')
    sb, sl, sg = bu.clone(), li.clone(), ge.clone()
    with torch.no_grad():
        m(sb, sl, sg)                                  # warm-up: topology blob, library load
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = m(sb, sl, sg)
        sb.copy_(bu2); sl.copy_(li2); sg.copy_(ge2)
        graph.replay()
        torch.cuda.synchronize()
        ref = m(bu2, li2, ge2)
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2])


def test_a_team_that_gave_up_is_reported_to_the_host_before_any_gradient_is_computed(lane_mapping):
    """VERDICT r2 item 5 / ADVICE r2: a team barrier that gives up used to leave NaN losses with rc 0.  The failing workgroup now
    sets a status word in the forward workspace; the host wrapper reads it (gns_team_status) before it launches the backward of a
    training call, and at the next call / check_status() for an evaluation call, and raises GNSError.  The word is injected here -
    a real give-up needs a foreign kernel squatting on a partner's CU for seconds."""
    import ctypes
    import opf_graph_neural_solver_amd as amd
    lib = amd.load_library()
    lane_mapping.set_option('team', 2)
    torch.manual_seed(0)
    m = amd.GNS(20, 10, 2, 0.9, True).cuda()
    bu, li, ge = amd.synth.synth_grids(30, 4096, seed=3, device='cuda')
    cfg = m._config(30, 41, 6)
    off = ctypes.c_size_t()
    assert lib.gns_team_status_offset(ctypes.byref(cfg), 4096, 1, ctypes.byref(off)) == 0 and off.value != ctypes.c_size_t(-1).value
    # a healthy call: status 0, backward runs
    out = m(bu, li, ge)
    out[2].mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    # training call with the word set between forward and backward
    m.zero_grad()
    out = m(bu, li, ge)
    ws = out[2].grad_fn.ws
    ws[off.value:off.value + 4] = torch.tensor([1, 0, 0, 0], dtype=torch.uint8, device='cuda')
    with pytest.raises(amd.GNSError, match='team of workgroups gave up'):
        out[2].mean().backward()
    assert all(p.grad is None for p in m.parameters())
    # evaluation call: reported at the next call of the module (and by check_status)
    lib.gns_team_status_offset(ctypes.byref(cfg), 4096, 0, ctypes.byref(off))
    if off.value != ctypes.c_size_t(-1).value:
        with torch.no_grad():
            m(bu, li, ge)
        pend = m._pending_status
        assert pend is not None
        pend[2][off.value:off.value + 4] = torch.tensor([1, 0, 0, 0], dtype=torch.uint8, device='cuda')
        with pytest.raises(amd.GNSError):
            m.check_status()
        m.check_status()                                   # reported once
    # without teams there is nothing to check and nothing is kept
    lane_mapping.set_option('team', 1)
    assert lib.gns_team_status_offset(ctypes.byref(cfg), 4096, 1, ctypes.byref(off)) == 0 and off.value == ctypes.c_size_t(-1).value
    with torch.no_grad():
        m(bu, li, ge)
    assert m.__dict__.get('_pending_status') is None


@pytest.mark.parametrize('case,bt,d,K,multi', [(118, 130, 20, 4, True), (30, 40000, 20, 3, True), (14, 70000, 10, 2, True), (300, 70, 20, 5, True),
                                               (118, 130, 10, 6, False), (30, 40000, 20, 3, False)])
def test_split_backward_modes_agree_with_the_persistent_kernel(case, bt, d, K, multi, lane_mapping):
    """The split backward (bwd_variant 4) in its three sweep modes against the persistent kernel (variant 2) on batches the goldens do
    not reach: ragged (130, 70 grids), many groups per sweep workgroup (40 000 / 70 000 grids = 625 / 1 094 groups: the accumulator
    tiles are carried across the groups of a workgroup, R = 2 / 4), both compiled model widths, non-trivial upstream gradients; and
    single-phi models (the reference's constructor default, main.py:108), which always take the bus-major kernel.
    Same arithmetic, other summation orders: gradients to 5e-6 of max|grad|; each mode bitwise reproducible."""
    import opf_graph_neural_solver_amd as amd
    torch.manual_seed(7)
    m = amd.GNS(d, 10, K, 0.9, multi).cuda()
    m.topology_check = 'first'
    bu, li, ge = amd.synth.synth_grids(case, bt, seed=13, device='cuda')
    gen = torch.Generator(device='cuda').manual_seed(1)
    wt, wl = torch.rand(bt, device='cuda', generator=gen), torch.rand(bt, device='cuda', generator=gen)
    wv = torch.randn(bt, bu.shape[1], device='cuda', generator=gen) * 1e-3

    def grads():
        m.zero_grad()
        v, th, tot, last = m(bu, li, ge)
        ((tot * wt).sum() / bt + (last * wl).mean() + (v * wv).sum() + (th * wv).sum() * 0.5).backward()
        return torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()

    old = amd.get_option('bwd_variant'), amd.get_option('bwds_mode')
    try:
        amd.set_option('bwd_variant', 2 if multi else 1)
        ref = grads()
        scale = float(ref.abs().max())
        for mode in ((1, 0, 2) if multi else (1,)):
            amd.set_option('bwd_variant', 4); amd.set_option('bwds_mode', mode)
            g1, g2 = grads(), grads()
            assert torch.equal(g1, g2), f'mode {mode} is not run-to-run reproducible'
            err = float((g1 - ref).abs().max()) / scale
            assert err < 5e-6, f'mode {mode}: {err:.2e} of max|grad|'
    finally:
        amd.set_option('bwd_variant', old[0]); amd.set_option('bwds_mode', old[1])


@pytest.mark.parametrize('d,h,multi', [(13, 12, True), (6, 7, False), (20, 14, True)])
def test_widths_between_the_compiled_kernels_on_ragged_batches_against_oracle(d, h, multi):
    """A model narrower than a compiled (latent_dim, hidden_dim) pair runs on it zero-padded (gns_common.h, GnsFamilies): (13, 12) on the
    (20, 14) kernels, (6, 7) on (10, 10); (20, 14) is a compiled pair itself.  130 case30 grids (three 64-grid groups, the last
    ragged), default mapping selection, non-trivial upstream gradients: outputs of three grids and the summed gradient of all of
    them against the CPU oracle; the gradient comes back in the MODEL's flat layout."""
    import opf_graph_neural_solver_amd as amd
    from oracle import gns_oracle as orc
    K, bt = 3, 130
    torch.manual_seed(11)
    m = amd.GNS(d, h, K, 0.9, multi).cuda()
    assert sum(p.numel() for p in m.parameters()) == m.flat_parameters().numel()
    bu, li, ge = amd.synth.synth_grids(30, bt, seed=5, device='cuda')
    gen = torch.Generator(device='cuda').manual_seed(2)
    wt = torch.rand(bt, device='cuda', generator=gen) + 0.5
    with torch.no_grad():
        ev = m(bu, li, ge)
    v, th, tot, last = m(bu, li, ge)
    ((tot * wt).sum() / bt).backward()
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu()
    flat = m.flat_parameters().detach().cpu()
    g_o = torch.zeros_like(flat)
    for b in range(bt):
        f = flat.clone().requires_grad_(True)
        o = orc.gns_forward(orc.unflatten_params(f, d, h, K, multi), bu[b].cpu(), li[b].cpu(), ge[b].cpu(), latent_dim=d, K=K, gamma=0.9, multiple_phi=multi)
        (o[2] * float(wt[b]) / bt).backward()
        g_o += f.grad
        if b in (0, 64, 129):
            for mine, ref, what in ((v[b], o[0], 'v'), (th[b], o[1], 'theta'), (tot[b], o[2], 'total'), (ev[0][b], o[0], 'v eval'), (ev[2][b], o[2], 'total eval')):
                assert_close(mine.detach().cpu(), ref.detach(), REL, what=f'{what}[{b}]')
    assert_close(grad, g_o, 5e-5, abs_floor=1e-7, what='grad_params')
