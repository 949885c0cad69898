"""Ad-hoc: time of gns_forward's three kernels together in evaluation mode via CUDA events around the module call."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import opf_graph_neural_solver_amd as amd
m = amd.GNS(20, 10, 4, 0.9, True).cuda(); m.topology_check = 'first'
bu, li, ge = amd.synth.synth_grids(118, 16384, seed=1, device='cuda')
with torch.no_grad():
    for it in range(5): m(bu, li, ge)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for it in range(50): m(bu, li, ge)
    e1.record(); torch.cuda.synchronize()
print(f"{os.environ.get('GNS_LIB', 'default'):24s} forward call (pack + kernel, GPU time) {e0.elapsed_time(e1) / 50:.3f} ms")
