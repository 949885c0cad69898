"""Where do two backward variants differ?  usage: python tools/gpu_debug_split.py case bt d K"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import opf_graph_neural_solver_amd as amd
case, bt, d, K = (int(x) for x in sys.argv[1:5])
amd.set_option('train_mapping', 1)
torch.manual_seed(7)
m = amd.GNS(d, 10, K, 0.9, True).cuda(); m.topology_check = 'first'
bu, li, ge = amd.synth.synth_grids(case, bt, seed=13, device='cuda')
gen = torch.Generator(device='cuda').manual_seed(1)
wt, wl = torch.rand(bt, device='cuda', generator=gen), torch.rand(bt, device='cuda', generator=gen)
wv = torch.randn(bt, bu.shape[1], device='cuda', generator=gen) * 1e-3
def grads(simple):
    m.zero_grad()
    v, th, tot, last = m(bu, li, ge)
    (tot.mean() if simple else ((tot * wt).sum() / bt + (last * wl).mean() + (v * wv).sum() + (th * wv).sum() * 0.5)).backward()
    return {n: p.grad.clone() for n, p in m.named_parameters()}
for simple in (True, False):
    amd.set_option('bwd_variant', 2); ref = grads(simple)
    scale = max(float(g.abs().max()) for g in ref.values())
    for mode in (1, 2):
        amd.set_option('bwd_variant', 4); amd.set_option('bwds_mode', mode)
        g = grads(simple)
        errs = sorted(((float((g[n] - ref[n]).abs().max()) / scale, n, float(ref[n].abs().max()) / scale) for n in g), reverse=True)
        print(f'simple={simple} mode {mode}: scale {scale:.3e}; worst blocks:', [(f'{e:.2e}', n, f'{r:.1e}') for e, n, r in errs[:4]], flush=True)
