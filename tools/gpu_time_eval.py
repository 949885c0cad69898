"""Ad-hoc: forward kernel time in evaluation mode (library HIP-event hooks) and the host-inclusive rate of the call loop."""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import opf_graph_neural_solver_amd as amd
lib = amd.load_library()
m = amd.GNS(20, 10, 4, 0.9, True).cuda(); m.topology_check = 'first'
bu, li, ge = amd.synth.synth_grids(118, 16384, seed=1, device='cuda')
with torch.no_grad():
    for it in range(3): m(bu, li, ge)
    lib.gns_profile_enable(64)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(50): m(bu, li, ge)
    t1 = time.perf_counter()            # host time to ENQUEUE 50 calls
    torch.cuda.synchronize(); t2 = time.perf_counter()
a, n = ctypes.c_float(), ctypes.c_int()
lib.gns_profile_read(0, ctypes.byref(a), ctypes.byref(n))
print(f"{os.environ.get('GNS_LIB', 'default'):24s} fwd(eval) kernel {a.value / max(n.value, 1):.3f} ms   loop {(t2 - t0) / 50 * 1e3:.3f} ms/call   host enqueue {(t1 - t0) / 50 * 1e3:.3f} ms/call")
