"""Ad-hoc: HIP forward+backward on every golden; prints gradient error per family, then a quick timing."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from helpers import golden_names, load_golden, cfg_of, t, rel_err
import opf_graph_neural_solver_amd as amd

def build(g):
    c = cfg_of(g)
    m = amd.GNS(latent_dim=c['latent_dim'], hidden_dim=c['hidden_dim'], K=c['K'], gamma=c['gamma'], multiple_phi=c['multiple_phi'])
    flat, off, sd = t(g['params']), 0, {}
    for n, p in m.named_parameters():
        sd[n] = flat[off:off + p.numel()].view(p.shape).clone(); off += p.numel()
    m.load_state_dict(sd)
    return m.cuda()

for name in golden_names():
    g = load_golden(name); m = build(g)
    v, th, tot, last = m(t(g['buses']).cuda(), t(g['lines']).cuda(), t(g['generators']).cuda())
    tot.mean().backward()
    torch.cuda.synchronize()
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    ref = g['grad_params']
    worst = ''
    off = 0; fam_err = {}
    for n, p in m.named_parameters():
        sz = p.numel(); fam = n.split('.')[0]
        d = np.max(np.abs(grad[off:off+sz] - ref[off:off+sz]))
        fam_err[fam] = max(fam_err.get(fam, 0.0), d); off += sz
    print(f"{name:34s} grad rel {rel_err(grad, ref):.2e} max|ref| {np.max(np.abs(ref)):.2e} " +
          ' '.join(f'{k}:{v_:.1e}' for k, v_ in fam_err.items()), flush=True)

# timing on the BASELINE workload shape
for case, bt in ((118, 16384),):
    m = amd.GNS(20, 10, 4, 0.9, True).cuda(); m.topology_check = 'first'
    bu, li, ge = amd.synth.synth_grids(case, bt, seed=1, device='cuda')
    for mode in ('fwd', 'fwd+bwd'):
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if mode == 'fwd':
                with torch.no_grad():
                    out = m(bu, li, ge)
            else:
                out = m(bu, li, ge); out[2].mean().backward(); m.zero_grad()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f'case{case} batch {bt} {mode}: {dt*1e3:.2f} ms  {bt/dt/1e6:.3f} M grids/s', flush=True)
