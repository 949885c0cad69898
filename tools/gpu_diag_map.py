"""Ad-hoc: per-grid differences between the two forward mappings, and both against the fp64 oracle for the worst grid."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opf_graph_neural_solver_amd as amd
from oracle import gns_oracle as orc
case, bt, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
m = amd.GNS(20, 10, K, 0.9, True).cuda()
bu, li, ge = amd.synth.synth_grids(case, bt, seed=1, device='cuda')
outs = []
for mapping in (1, 2):
    amd.set_option('fwd_mapping', mapping); amd.set_option('gw_pack', 1)
    with torch.no_grad():
        outs.append([o.double().cpu() for o in m(bu, li, ge)])
for name, a, b in zip(('v', 'theta', 'total', 'last'), outs[0], outs[1]):
    d = (a - b).abs()
    per = d.reshape(bt, -1).max(dim=1).values
    w = int(per.argmax())
    print(name, 'max abs diff', float(d.max()), 'scale', float(a.abs().max()), 'worst grid', w, 'grids with diff > 1e-5*scale:', int((per > 1e-5 * a.abs().max()).sum()))
w = int((outs[0][0] - outs[1][0]).abs().reshape(bt, -1).max(dim=1).values.argmax())
flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
for dt in (torch.float64, torch.float32):
    params = orc.unflatten_params(flat.to(dt), 20, 10, K, True)
    with torch.no_grad():
        v, th, tot, last = orc.gns_forward(params, bu[w].cpu().to(dt), li[w].cpu().to(dt), ge[w].cpu().to(dt), latent_dim=20, K=K, gamma=0.9, multiple_phi=True)
    print(str(dt), 'oracle grid', w, 'max|v|', float(v.abs().max()), 'v diff lane', float((outs[0][0][w] - v.double()).abs().max()), 'v diff lds', float((outs[1][0][w] - v.double()).abs().max()),
          'theta diff lane', float((outs[0][1][w] - th.double()).abs().max()), 'lds', float((outs[1][1][w] - th.double()).abs().max()))
