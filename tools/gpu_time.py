"""Ad-hoc kernel timing through the library's own HIP-event hooks: prints fwd/bwd kernel ms for case118 x 16384."""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import opf_graph_neural_solver_amd as amd
lib = amd.load_library()
case, bt = 118, 16384
m = amd.GNS(20, 10, 4, 0.9, True).cuda(); m.topology_check = 'first'
bu, li, ge = amd.synth.synth_grids(case, bt, seed=1, device='cuda')
for it in range(2):
    out = m(bu, li, ge); out[2].mean().backward(); m.zero_grad()
lib.gns_profile_enable(8)
for it in range(5):
    out = m(bu, li, ge); out[2].mean().backward(); m.zero_grad()
torch.cuda.synchronize()
a, n = ctypes.c_float(), ctypes.c_int()
lib.gns_profile_read(0, ctypes.byref(a), ctypes.byref(n)); f = a.value / max(n.value, 1)
lib.gns_profile_read(1, ctypes.byref(a), ctypes.byref(n)); b = a.value / max(n.value, 1)
print(f"{os.environ.get('GNS_LIB', 'default'):40s} fwd(train) {f:.3f} ms   bwd {b:.3f} ms", flush=True)
