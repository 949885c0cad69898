import sys, os, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import opf_graph_neural_solver_amd as amd
lib = amd.load_library()
m = amd.GNS(20, 10, 4, 0.9, True).cuda(); m.topology_check = 'first'; m.cache_packed_inputs = True
bu, li, ge = amd.synth.synth_grids(118, 16384, seed=1, device='cuda')
amd.set_option('train_mapping', 1)
for it in range(4):
    out = m(bu, li, ge); out[2].mean().backward(); m.zero_grad()
lib.gns_profile_enable(16)
for it in range(12):
    out = m(bu, li, ge); out[2].mean().backward(); m.zero_grad()
torch.cuda.synchronize()
a, n = ctypes.c_float(), ctypes.c_int()
lib.gns_profile_read(0, ctypes.byref(a), ctypes.byref(n)); f = a.value / max(n.value, 1)
print(f"GNS_UCOST_EXP={os.environ.get('GNS_UCOST_EXP')}: fwd(train) {f:.3f} ms", flush=True)
