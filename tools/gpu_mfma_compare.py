"""Ad-hoc: packed-FMA vs GNS_DW_MFMA=1 weight-gradient contraction: gradient agreement on big batches + kernel time."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import opf_graph_neural_solver_amd as amd
lib = amd.load_library()

def run(case, bt, d, multi, K, mfma, reps=4):
    os.environ['GNS_DW_MFMA'] = '1' if mfma else '0'
    torch.manual_seed(0)
    m = amd.GNS(d, 10, K, 0.9, multi).cuda(); m.topology_check = 'first'
    bu, li, ge = amd.synth.synth_grids(case, bt, seed=1, device='cuda')
    for it in range(2):
        m.zero_grad(); out = m(bu, li, ge); out[2].mean().backward()
    lib.gns_profile_enable(8)
    for it in range(reps):
        m.zero_grad(); out = m(bu, li, ge); out[2].mean().backward()
    torch.cuda.synchronize()
    a, n = ctypes.c_float(), ctypes.c_int()
    lib.gns_profile_read(1, ctypes.byref(a), ctypes.byref(n)); b = a.value / max(n.value, 1)
    lib.gns_profile_enable(0)
    return torch.cat([p.grad.reshape(-1) for p in m.parameters()]).double().cpu(), b

for case, bt, d, multi, K in ((118, 16384, 20, True, 4), (118, 4099, 20, False, 4), (14, 20000, 10, True, 4), (30, 777, 10, False, 3), (300, 2048, 20, True, 10)):
    g0, t0 = run(case, bt, d, multi, K, False)
    g1, t1 = run(case, bt, d, multi, K, True)
    rel = float((g0 - g1).abs().max() / g0.abs().max())
    print(f"case{case} bt {bt} d {d} multi {multi} K {K}: max rel diff {rel:.2e}   bwd valu {t0:.3f} ms   mfma {t1:.3f} ms", flush=True)
