import sys, os, time, ctypes
sys.path.insert(0, '/root/repo')
import torch
import opf_graph_neural_solver_amd as amd
lib = amd.load_library()
m = amd.GNS(20, 10, 4, 0.9, True).cuda(); m.topology_check = 'first'
bu, li, ge = amd.synth.synth_grids(118, 16384, seed=1, device='cuda')
amd.set_option('train_mapping', 1)
for rep in range(2):
    for pl in (1, 2):
        amd.set_option('fwd_plane', pl)
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
        print(f'fwd_plane {pl}: fwd(train) {f:.3f} ms  bwd {b:.3f} ms', flush=True)
