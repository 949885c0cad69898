"""Kernel and wall time per training step at the reference's own operating point (GNS/main.py:209-254: case14, batch 128, K=15,
latent 10, three phis) and neighbours, eager vs captured graph.  usage: python tools/gpu_small_batch.py"""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import opf_graph_neural_solver_amd as amd
lib = amd.load_library()
for case, bt, d, K, mapping in ((14, 128, 10, 15, 0), (14, 128, 20, 4, 0), (14, 1024, 10, 15, 0), (118, 128, 20, 4, 0), (118, 1024, 20, 4, 0)):
    amd.set_option('train_mapping', mapping)
    torch.manual_seed(0)
    m = amd.GNS(d, 10, K, 0.9, True).cuda(); m.topology_check = 'first'
    opt = amd.training.make_optimizer(m)
    bu, li, ge = amd.synth.synth_grids(case, bt, seed=1, device='cuda')
    for _ in range(3):
        amd.training.train_step(m, opt, bu, li, ge)
    lib.gns_profile_enable(64)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        amd.training.train_step(m, opt, bu, li, ge)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 50 * 1e3
    a, n = ctypes.c_float(), ctypes.c_int()
    lib.gns_profile_read(0, ctypes.byref(a), ctypes.byref(n)); f = a.value / max(n.value, 1)
    lib.gns_profile_read(1, ctypes.byref(a), ctypes.byref(n)); b = a.value / max(n.value, 1)
    lib.gns_profile_enable(0)
    g = amd.training.GraphedStep(m, opt, bu, li, ge)
    for _ in range(5):
        g.run(bu, li, ge)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        g.run(bu, li, ge)
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 200 * 1e3
    print(f'case{case} x {bt} d={d} K={K} train_mapping {mapping}: fwd kernel {f:.3f} ms  bwd kernels {b:.3f} ms  eager step {eager:.3f} ms  captured step {graph:.3f} ms', flush=True)
