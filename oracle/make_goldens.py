"""Generate tests/golden/*.npz from the REFERENCE'S OWN forward/backward.  Runs only in the build
container (needs /root/reference); the .npz files it writes are data and travel to the GPU box, the
reference's code does not.

How the reference is executed: ``GNS/main.py`` is imported from where it lies.  Two third-party modules it
imports at module level are absent from this image and stay absent: ``wandb`` (logging only, never touched
by the hot path) is registered as an empty module; ``torch_scatter`` is registered with ONE function,
``scatter_add``, restating the package's published semantics (index broadcast along ``dim``, then
``out.scatter_add_``).  ``main()`` is never called.  Inputs are synthetic (``synth.py``): the reference's
shipped case14 pickles are refused by every non-executing loader (torch.load(weights_only=True), numpy.load),
so they are not used.

Each golden holds: config, inputs, flat parameters (state_dict order), outputs of main.GNS.forward per grid,
d(mean total_loss)/d(params) from autograd, the names of parameters whose .grad stayed None, and per-step
intermediates captured by wrapping the two module-level physics functions.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference/GNS'
sys.dont_write_bytecode = True


def _load_pkg():
    spec = importlib.util.spec_from_file_location(
        'opf_graph_neural_solver_amd_synth', os.path.join(ROOT, 'opf-graph-neural-solver_amd', 'synth.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _import_reference():
    ts = types.ModuleType('torch_scatter')

    def scatter_add(src, index, dim=-1, out=None, dim_size=None):
        if dim < 0:
            dim += src.dim()
        idx = index
        for _ in range(dim):
            idx = idx.unsqueeze(0)
        while idx.dim() < src.dim():
            idx = idx.unsqueeze(-1)
        idx = idx.expand_as(src)
        if out is None:
            size = list(src.shape)
            size[dim] = int(dim_size if dim_size is not None else index.max() + 1)
            out = torch.zeros(size, dtype=src.dtype)
        return out.scatter_add_(dim, idx, src)

    ts.scatter_add = scatter_add
    sys.modules['torch_scatter'] = ts
    sys.modules['wandb'] = types.ModuleType('wandb')
    sys.path.insert(0, REF)
    import main as ref_main  # noqa
    return ref_main


def run_case(ref, synth, name, case_nr, batch, K, d, h, multi, seed, load_scale=1.0, gamma=0.9, inputs=None):
    """``inputs`` = (buses, lines, generators) replaces the synthetic case ``case_nr`` (the odd_* goldens: hand-made topologies)."""
    import warnings
    warnings.filterwarnings('ignore')
    torch.manual_seed(seed)
    model = ref.GNS(latent_dim=d, hidden_dim=h, K=K, gamma=gamma, multiple_phi=multi)
    B, L, G = ref.get_BLG()
    buses, lines, gens = inputs if inputs is not None else synth.synth_grids(case_nr, batch, seed=100 + seed, load_scale=load_scale)
    trace = []
    orig_gac, orig_lpi = ref.global_active_compensation, ref.local_power_imbalance

    def gac(v, theta, *a, **k):
        out = orig_gac(v, theta, *a, **k)
        trace.append(dict(v=v.detach().numpy().copy(), theta=theta.detach().numpy().copy(),
                          pg_new=out[0].detach().numpy().copy(), qg_new=out[1].detach().numpy().copy()))
        return out

    def lpi(*a, **k):
        out = orig_lpi(*a, **k)
        trace[-1]['dp'] = out[0].detach().numpy().copy()
        trace[-1]['dq'] = out[1].detach().numpy().copy()
        return out

    ref.global_active_compensation, ref.local_power_imbalance = gac, lpi
    try:
        vs, ths, tots, lasts = [], [], [], []
        for b in range(batch):
            v, th, tot, last = model(buses=buses[b], lines=lines[b], generators=gens[b], B=B, L=L, G=G)
            vs.append(v.detach().numpy()); ths.append(th.detach().numpy()); tots.append(tot); lasts.append(float(last))
        torch.stack(tots).mean().backward()
    finally:
        ref.global_active_compensation, ref.local_power_imbalance = orig_gac, orig_lpi
    names = [n for n, _ in model.named_parameters()]
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy()
    none_grad = [n for n, p in model.named_parameters() if p.grad is None]
    grad = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                      for p in model.parameters()]).numpy()
    steps = {}
    for key in ('v', 'theta', 'pg_new', 'qg_new', 'dp', 'dq'):
        steps['step_' + key] = np.stack([t[key] for t in trace]).reshape(batch, K, -1)
    out = dict(case_nr=case_nr, batch=batch, K=K, latent_dim=d, hidden_dim=h, multiple_phi=int(multi), gamma=gamma,
               seed=seed, load_scale=load_scale,
               buses=buses.numpy(), lines=lines.numpy(), generators=gens.numpy(), params=flat,
               param_names=np.array(names), none_grad_names=np.array(none_grad),
               v=np.stack(vs), theta=np.stack(ths), total_loss=np.array([float(t) for t in tots], dtype=np.float32),
               last_loss=np.array(lasts, dtype=np.float32), grad_params=grad, **steps)
    path = os.path.join(ROOT, 'tests', 'golden', name + '.npz')
    np.savez_compressed(path, **out)
    lam_lo = 'n/a'
    print(f'{name}: wrote {os.path.getsize(path) / 1024:.0f} KiB  total_loss[0]={out["total_loss"][0]:.6g} '
          f'none_grad={len(none_grad)}')


def run_prepare(synth, name, case_nr, batch, seed):
    """Golden for the input producer: the reference's OWN utils.prepare_grid (GNS/utils.py:17-41) executed on synthetic
    PYPOWER-format case dicts.  Its file access is the only thing replaced: ``open`` / ``pkl.load`` inside the module
    are pointed at the in-memory dict (the shipped pickles are not readable with a non-executing loader)."""
    import utils as ref_utils
    bus, br, ge = synth.raw_case_arrays(case_nr, batch, seed=seed)
    outs = []
    holder = {}
    ref_utils.open = lambda *a, **k: None
    orig = ref_utils.pkl.load
    ref_utils.pkl.load = lambda f: holder['case']
    try:
        for b in range(batch):
            holder['case'] = {'baseMVA': 100.0, 'bus': bus[b].numpy(), 'branch': br[b].numpy(), 'gen': ge[b].numpy()}
            outs.append([t.numpy() for t in ref_utils.prepare_grid(case_nr, b)])
    finally:
        ref_utils.pkl.load = orig
        del ref_utils.open
    path = os.path.join(ROOT, 'tests', 'golden', name + '.npz')
    np.savez_compressed(path, kind='prepare', case_nr=case_nr, bus=bus.numpy(), branch=br.numpy(), gen=ge.numpy(),
                        buses=np.stack([o[0] for o in outs]), lines=np.stack([o[1] for o in outs]),
                        generators=np.stack([o[2] for o in outs]))
    print(f'{name}: wrote {os.path.getsize(path) / 1024:.0f} KiB')


def run_metrics(name, n_samples, n_bus, n_line, seed):
    """Goldens for the evaluation metrics.  ``GNS/evaluate.py`` is a module-level script that needs PYPOWER,
    unshipped grids and an unshipped checkpoint, so it cannot be imported; what CAN be executed are its own statements:
    the file is parsed (never copied), the ``active_line_flow`` function (``evaluate.py:15-18``) and the module-level
    assignments that compute the error statistics from the six result arrays (``evaluate.py:93-125,150-156``: only ``np``
    calls on those arrays) are compiled from the parsed tree and run on synthetic solver / Newton-Raphson results."""
    import ast
    src = open(os.path.join(REF, 'evaluate.py')).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'active_line_flow']
    assert len(fn) == 1
    inputs = {'NR_theta_out', 'GNS_theta_out', 'NR_v_out', 'GNS_v_out', 'NR_active_line_flow', 'GNS_active_line_flow'}
    known = set(inputs) | {'np', 'int'}
    stats = []
    for n in tree.body:
        if not isinstance(n, ast.Assign) or n.lineno < 93:
            continue
        names = {x.id for x in ast.walk(n.value) if isinstance(x, ast.Name)}
        if names and names <= known and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name):
            stats.append(n)
            known.add(n.targets[0].id)
    ns = {'np': np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), 'evaluate.py', 'exec'), ns)
    rng = np.random.default_rng(seed)
    v_nr = rng.uniform(0.94, 1.06, (n_samples, n_bus)).astype(np.float32)
    th_nr_deg = rng.uniform(-25.0, 25.0, (n_samples, n_bus)).astype(np.float32)      # PYPOWER reports degrees (evaluate.py:94)
    v_gns = (v_nr * rng.uniform(0.98, 1.02, v_nr.shape)).astype(np.float32)
    th_gns = (np.deg2rad(th_nr_deg) + rng.normal(0, 0.02, v_nr.shape)).astype(np.float32)
    src_bus = rng.integers(1, n_bus + 1, n_line).astype(np.float64)
    dst_bus = ((src_bus - 1 + rng.integers(1, n_bus, n_line)) % n_bus + 1).astype(np.float64)
    x = rng.uniform(0.02, 0.3, (n_samples, n_line))
    alf = ns['active_line_flow']
    alf_nr = np.stack([alf(v_nr[i], np.deg2rad(th_nr_deg[i]), x[i], src_bus, dst_bus) for i in range(n_samples)]).astype(np.float32)
    alf_gns = np.stack([alf(v_gns[i], th_gns[i], x[i], src_bus, dst_bus) for i in range(n_samples)]).astype(np.float32)
    ns.update(NR_theta_out=th_nr_deg.copy(), GNS_theta_out=th_gns.copy(), NR_v_out=v_nr.copy(), GNS_v_out=v_gns.copy(),
              NR_active_line_flow=alf_nr.copy(), GNS_active_line_flow=alf_gns.copy())
    exec(compile(ast.Module(body=stats, type_ignores=[]), 'evaluate.py', 'exec'), ns)
    out = {'in_v_nr': v_nr, 'in_theta_nr_deg': th_nr_deg, 'in_v_gns': v_gns, 'in_theta_gns': th_gns, 'in_x': x,
           'in_src': src_bus, 'in_dst': dst_bus, 'alf_nr': alf_nr, 'alf_gns': alf_gns,
           'stat_lines': np.array([n.lineno for n in stats])}
    for n in stats:
        k = n.targets[0].id
        if k not in inputs:
            out['ref_' + k] = np.asarray(ns[k])
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', name + '.npz'), **out)
    print(name, 'statements', [n.lineno for n in stats], sorted(k for k in out if k.startswith('ref_')))


def run_augment(synth, name, case_nr, n_draws, seed):
    """Goldens for the synthetic-grid generator.  ``GNS/augment_grids.py`` is a module-level script that imports
    PYPOWER (absent) and writes 10 000 pickles, so it cannot be run; what CAN be executed are its own statements: the file is
    parsed (never copied), the nine ``*_range`` assignments (``augment_grids.py:12-20``) and the perturbation statements inside its
    loop (``:30-53``: only ``np`` calls on ``augmented_case``) are compiled from the parsed tree and run ``n_draws`` times with a
    seeded numpy generator on a PYPOWER-layout case built from the public IEEE-14 base data.  Stored: the range constants,
    per-column statistics of the draws relative to the base case, the balance identity of ``:51`` and the first draws."""
    import ast
    import copy
    tree = ast.parse(open(os.path.join(REF, 'augment_grids.py')).read())
    ranges = [n for n in tree.body if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name)
              and n.targets[0].id.endswith('_range')]
    loop = [n for n in tree.body if isinstance(n, ast.For)]
    assert len(ranges) == 9 and len(loop) >= 1
    body = []
    for n in loop[0].body:
        # the statements that touch augmented_case['bus' | 'branch' | 'gen'] or define pg_min / pg_max; not the list bookkeeping
        if isinstance(n, ast.Assign) or isinstance(n, ast.AugAssign):
            tgt = n.targets[0] if isinstance(n, ast.Assign) else n.target
            names = {x.id for x in ast.walk(tgt) if isinstance(x, ast.Name)}
            if names & {'augmented_case', 'pg_max', 'pg_min'} and not (isinstance(n, ast.Assign) and isinstance(n.value, ast.Call)
                                                                   and getattr(n.value.func, 'attr', '') == 'deepcopy'):
                body.append(n)
    ns = {'np': np}
    exec(compile(ast.Module(body=ranges, type_ignores=[]), 'augment_grids.py', 'exec'), ns)
    code = compile(ast.Module(body=body, type_ignores=[]), 'augment_grids.py', 'exec')
    c = synth.base_case(case_nr)
    n, e, gn = synth.CASE_SHAPES[case_nr]
    bus = np.zeros((n, 13)); bus[:, 0] = np.arange(1, n + 1); bus[:, 2] = c['Pd']; bus[:, 3] = c['Qd']
    br = np.zeros((e, 13)); br[:, 0] = c['f_bus']; br[:, 1] = c['t_bus']; br[:, 2] = c['r']; br[:, 3] = c['x']; br[:, 4] = c['b']
    ge = np.zeros((gn, 21)); ge[:, 0] = c['gen_bus']; ge[:, 1] = c['Pg']; ge[:, 5] = c['Vg']; ge[:, 8] = c['Pmax']; ge[:, 9] = c['Pmin']
    case = {'bus': bus, 'branch': br, 'gen': ge}
    np.random.seed(seed)
    cols = {k: [] for k in ('r', 'x', 'b', 'tau', 'shift', 'vg', 'pg', 'pd', 'qd')}
    for _ in range(n_draws):
        ns['augmented_case'] = copy.deepcopy(case)
        exec(code, ns)
        a = ns['augmented_case']
        cols['r'].append(a['branch'][:, 2]); cols['x'].append(a['branch'][:, 3]); cols['b'].append(a['branch'][:, 4])
        cols['tau'].append(a['branch'][:, 8]); cols['shift'].append(a['branch'][:, 9]); cols['vg'].append(a['gen'][:, 5])
        cols['pg'].append(a['gen'][:, 1]); cols['pd'].append(a['bus'][:, 2]); cols['qd'].append(a['bus'][:, 3])
    out = {'n_draws': np.array(n_draws), 'stmt_lines': np.array([x.lineno for x in body]),
           'base_r': c['r'], 'base_x': c['x'], 'base_b': c['b'], 'base_vg': c['Vg'], 'base_pd': c['Pd'], 'base_qd': c['Qd'],
           'base_pmax': c['Pmax'], 'base_pmin': c['Pmin']}
    for r_ in ranges:
        out[r_.targets[0].id] = np.asarray(ns[r_.targets[0].id], dtype=np.float64)
    for k, v in cols.items():
        v = np.stack(v)
        out[k + '_min'], out[k + '_max'], out[k + '_mean'], out[k + '_std'] = v.min(0), v.max(0), v.mean(0), v.std(0)
        out[k + '_first8'] = v[:8]
    pd_, pg_ = np.stack(cols['pd']), np.stack(cols['pg'])
    out['balance_max_rel_err'] = np.array(np.abs(pd_.sum(1) / pg_.sum(1) - 1.0).max())       # augment_grids.py:51
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', name + '.npz'), **out)
    print(name, 'statements', [x.lineno for x in body], 'ranges', {r_.targets[0].id: ns[r_.targets[0].id] for r_ in ranges},
          'balance', float(out['balance_max_rel_err']))


def main():
    ref = _import_reference()
    synth = _load_pkg()
    run_prepare(synth, 'prepare_c14_b4', 14, 4, 21)
    run_prepare(synth, 'prepare_c118_b2', 118, 2, 22)
    run_metrics('metrics_c14_s64', 64, 14, 20, 31)
    run_augment(synth, 'augment_c14', 14, 4096, 41)
    os.makedirs(os.path.join(ROOT, 'tests', 'golden'), exist_ok=True)
    # name, case, batch, K, d, h, multi, seed, load_scale
    cases = [
        ('c14_b1_K4_d20_multi', 14, 1, 4, 20, 10, True, 0, 1.0),          # BASELINE config 1
        ('c14_b4_K4_d20_single', 14, 4, 4, 20, 10, False, 1, 1.0),
        ('c14_b3_K4_d20_multi_lowload', 14, 3, 4, 20, 10, True, 2, 0.2),  # lambda < 0.5 branches (main.py:48,54)
        ('c14_b2_K1_d20_multi', 14, 2, 1, 20, 10, True, 3, 1.0),
        ('c14_b2_K4_d10_multi', 14, 2, 4, 10, 10, True, 4, 1.0),
        ('c14_b2_K4_d10_single', 14, 2, 4, 10, 10, False, 5, 1.0),
        ('c30_b3_K4_d20_multi', 30, 3, 4, 20, 10, True, 6, 1.0),
        ('c30_b2_K4_d20_single_lowload', 30, 2, 4, 20, 10, False, 7, 0.2),
        ('c118_b2_K4_d20_multi', 118, 2, 4, 20, 10, True, 8, 1.0),         # BASELINE config 3 shape
        ('c118_b2_K4_d20_single', 118, 2, 4, 20, 10, False, 9, 1.0),
        ('c300_b1_K10_d20_multi', 300, 1, 10, 20, 10, True, 10, 1.0),       # BASELINE config 5 shape
        ('c14_b2_K15_d10_multi', 14, 2, 15, 10, 10, True, 11, 1.0),         # the reference's own run configuration (main.py:209-213)
        ('c14_b2_K30_d10_single', 14, 2, 30, 10, 10, False, 12, 1.0),       # the reference's constructor defaults (main.py:108)
        # widths between the compiled kernels': the HIP path runs them zero-padded on the next wider kernel ("maybe also try more or
        # less hidden dim", main.py:215); odd widths included
        ('c14_b2_K3_d16_h8_multi', 14, 2, 3, 16, 8, True, 13, 1.0),
        ('c14_b2_K4_d7_h5_single', 14, 2, 4, 7, 5, False, 14, 1.0),
        ('c30_b2_K2_d20_h6_multi', 30, 2, 2, 20, 6, True, 15, 1.0),
        ('c14_b2_K4_d3_h10_multi_lowload', 14, 2, 4, 3, 10, True, 16, 0.2),
        # "more hidden dim": the (20, 14) kernels, exactly and zero-padded
        ('c14_b2_K3_d20_h14_multi', 14, 2, 3, 20, 14, True, 17, 1.0),
        ('c14_b2_K4_d10_h12_single', 14, 2, 4, 10, 12, False, 18, 1.0),
        ('c30_b2_K2_d12_h11_multi', 30, 2, 2, 12, 11, True, 19, 1.0),
    ]
    for c in cases:
        run_case(ref, synth, *c)
    # shapes no case file has (tests/helpers.py::odd_topologies): parallel lines, a hub of in-degree 40, buses without lines,
    # one-way chains, a generator on every bus / on one bus only / two on one bus.  name suffix, batch, K, d, multi, seed
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers
    topo = helpers.odd_topologies()
    odd = [('pair', 3, 3, 20, True, 40), ('hub_all_gens', 2, 4, 10, False, 41), ('ring_isolated_dupgen', 3, 4, 20, True, 42),
           ('chain_one_way', 2, 2, 20, False, 43), ('hub_indegree_40', 2, 3, 10, True, 44), ('random_40_one_gen', 2, 4, 20, True, 45),
           ('random_33_many_gens', 2, 4, 20, False, 46)]
    for tname, batch, K, d, multi, seed in odd:
        n, f, t_, gb = topo[tname]
        inputs = helpers.grids_on_topology(n, f, t_, gb, batch, seed)
        run_case(ref, synth, f"odd_{tname}_b{batch}_K{K}_d{d}_{'multi' if multi else 'single'}", 0, batch, K, d, 10, multi, seed, inputs=inputs)


if __name__ == '__main__':
    main()
