"""Ad-hoc: a few training steps (forward + backward) with the mapping chosen by GNS_TRAIN_MAPPING / GNS_GW_PACK (for rocprofv3 passes)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import opf_graph_neural_solver_amd as amd
case = int(sys.argv[1]) if len(sys.argv) > 1 else 118
bt = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
m = amd.GNS(20, 10, K, 0.9, True).cuda(); m.topology_check = 'first'
bu, li, ge = amd.synth.synth_grids(case, bt, seed=1, device='cuda')
for it in range(4):
    out = m(bu, li, ge); out[2].mean().backward(); m.zero_grad()
torch.cuda.synchronize()
