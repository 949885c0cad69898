"""A/B of forward launch options on one box: LDS planes (fwd_plane 0/1/2) x waves per workgroup (fwd_waves 8/16), case118 x 16384."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import opf_graph_neural_solver_amd as amd
lib = amd.load_library()
m = amd.GNS(20, 10, 4, 0.9, True).cuda(); m.topology_check = 'first'; m.cache_packed_inputs = True
bu, li, ge = amd.synth.synth_grids(118, 16384, seed=1, device='cuda')
amd.set_option('train_mapping', 1)
for rep in range(2):
    for waves in (16, 8):
        for pl in (2, 1):
            amd.set_option('fwd_plane', pl); amd.set_option('fwd_waves', waves)
            for it in range(3):
                out = m(bu, li, ge); out[2].mean().backward(); m.zero_grad()
            lib.gns_profile_enable(16)
            for it in range(10):
                out = m(bu, li, ge); out[2].mean().backward(); m.zero_grad()
            torch.cuda.synchronize()
            a, n = ctypes.c_float(), ctypes.c_int()
            lib.gns_profile_read(0, ctypes.byref(a), ctypes.byref(n)); f = a.value / max(n.value, 1)
            lib.gns_profile_read(1, ctypes.byref(a), ctypes.byref(n)); b = a.value / max(n.value, 1)
            lib.gns_profile_enable(0)
            print(f'fwd_waves {waves} fwd_plane {pl}: fwd(train) {f:.3f} ms  bwd {b:.3f} ms', flush=True)
