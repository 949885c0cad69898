"""Kernel times of the other BASELINE configurations and of the reference's default model shapes."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import opf_graph_neural_solver_amd as amd
lib = amd.load_library()
MAC = {  # MLP MACs per grid per step: lines x phi + buses x L   
}
for case, bt, d, multi, K in ((14, 1, 20, True, 4), (30, 4096, 20, True, 4), (118, 16384, 20, True, 4), (300, 8192, 20, True, 10),
                              (118, 16384, 10, False, 30), (118, 16384, 10, True, 15), (14, 65536, 10, False, 30)):
    torch.manual_seed(0)
    m = amd.GNS(d, 10, K, 0.9, multi).cuda(); m.topology_check = 'first'
    bu, li, ge = amd.synth.synth_grids(case, bt, seed=1, device='cuda')
    for it in range(2):
        m.zero_grad(); out = m(bu, li, ge); out[2].mean().backward()
    lib.gns_profile_enable(8)
    for it in range(4):
        m.zero_grad(); out = m(bu, li, ge); out[2].mean().backward()
    torch.cuda.synchronize()
    a, n = ctypes.c_float(), ctypes.c_int()
    lib.gns_profile_read(0, ctypes.byref(a), ctypes.byref(n)); f = a.value / max(n.value, 1)
    lib.gns_profile_read(1, ctypes.byref(a), ctypes.byref(n)); b = a.value / max(n.value, 1)
    lib.gns_profile_enable(0)
    N, E, Gn = amd.synth.CASE_SHAPES[case]
    phi_out = d if multi else 1
    macs = K * (E * (3 if multi else 1) * ((d + 5) * 10 + 100 + 10 * phi_out) + N * (2 * ((4 + 2 * d) * 10 + 100 + 10) + (4 + 2 * d) * 10 + 100 + 10 * d))
    tf = lambda ms, mult: macs * 2 * mult * bt / (ms * 1e-3) / 1e12
    print(f'case{case:<3d} batch {bt:6d} d={d} K={K:2d} multi={multi!s:5}: fwd {f:8.3f} ms ({tf(f,1):5.1f} TF nominal)  bwd {b:8.3f} ms ({tf(b,2):5.1f} TF nominal)  '
          f'{bt / ((f + b) * 1e-3) / 1e6:7.3f} M grids/s fwd+bwd kernels', flush=True)
    del m, bu, li, ge, out
    torch.cuda.empty_cache()
