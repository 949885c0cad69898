"""Ad-hoc: run the HIP forward on every golden and print the error per tensor."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from helpers import golden_names, load_golden, cfg_of, t, rel_err
import opf_graph_neural_solver_amd as amd
for name in golden_names():
    g = load_golden(name); c = cfg_of(g)
    m = amd.GNS(latent_dim=c['latent_dim'], hidden_dim=c['hidden_dim'], K=c['K'], gamma=c['gamma'], multiple_phi=c['multiple_phi'])
    flat, off, sd = t(g['params']), 0, {}
    for n, p in m.named_parameters():
        sd[n] = flat[off:off + p.numel()].view(p.shape).clone(); off += p.numel()
    m.load_state_dict(sd); m = m.cuda()
    with torch.no_grad():
        v, th, tot, last = m(t(g['buses']).cuda(), t(g['lines']).cuda(), t(g['generators']).cuda())
    torch.cuda.synchronize()
    print(f"{name:36s} v {rel_err(v.cpu(), g['v']):.2e} theta {rel_err(th.cpu(), g['theta']):.2e} "
          f"total {rel_err(tot.cpu(), g['total_loss']):.2e} last {rel_err(last.cpu(), g['last_loss']):.2e}", flush=True)
