"""Diagnostic: error of the HIP forward against the fp32 and fp64 CPU oracle as the number of steps K grows
(reference default K=30).  Shows whether a K=30 deviation is fp32 conditioning of the algorithm or a defect."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import opf_graph_neural_solver_amd as amd
from oracle import gns_oracle as orc

def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())

for multi, d in ((False, 10), (True, 10), (True, 20)):
    for K in (4, 10, 15, 20, 30):
        torch.manual_seed(3)
        m = amd.GNS(latent_dim=d, hidden_dim=10, K=K, gamma=0.9, multiple_phi=multi).cuda()
        flat = m.flat_parameters().detach().cpu()
        worst = [0.0] * 6
        for case in (14, 30):
            bu, li, ge = amd.synth.synth_grids(case, 3, seed=11, device='cuda')
            with torch.no_grad():
                v, th, tot, last = m(bu, li, ge)
            for b in range(3):
                p32 = orc.unflatten_params(flat, d, 10, K, multi)
                p64 = orc.unflatten_params(flat.double(), d, 10, K, multi)
                kw = dict(latent_dim=d, K=K, gamma=0.9, multiple_phi=multi)
                with torch.no_grad():
                    o32 = orc.gns_forward(p32, bu[b].cpu(), li[b].cpu(), ge[b].cpu(), **kw)
                    o64 = orc.gns_forward(p64, bu[b].cpu().double(), li[b].cpu().double(), ge[b].cpu().double(), **kw)
                e = [rel(v[b].cpu(), o32[0]), rel(th[b].cpu(), o32[1]), rel(v[b].cpu(), o64[0]), rel(th[b].cpu(), o64[1]),
                     rel(o32[0], o64[0]), rel(o32[1], o64[1])]
                worst = [max(a, c) for a, c in zip(worst, e)]
        print(f'multi={multi!s:5} d={d} K={K:2d}  hip-vs-cpu32 v {worst[0]:.1e} th {worst[1]:.1e} | hip-vs-fp64 v {worst[2]:.1e} th {worst[3]:.1e} | '
              f'cpu32-vs-fp64 v {worst[4]:.1e} th {worst[5]:.1e}', flush=True)
