"""Diagnostic: run the STAMPS build of the backward through the raw C-ABI and print per-phase cycle shares."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GNS_LIB', os.path.join(ROOT, 'tools', 'abl_STAMPS.so'))
import numpy as np, torch
import opf_graph_neural_solver_amd as amd
from opf_graph_neural_solver_amd._lib import GnsConfig
lib = amd.load_library()
case, bt = 118, 16384
N, E, Gn = amd.synth.CASE_SHAPES[case]
m = amd.GNS(20, 10, 4, 0.9, True).cuda()
flat = m.flat_parameters()
bu, li, ge = amd.synth.synth_grids(case, bt, seed=1, device='cuda')
topo = m._topology(li, ge, N)
cfg = GnsConfig(N, E, Gn, 4, 20, 10, 1, 0.9)
fw, bw = ctypes.c_size_t(), ctypes.c_size_t()
lib.gns_workspace_bytes(ctypes.byref(cfg), bt, 1, ctypes.byref(fw), ctypes.byref(bw))
ws = torch.empty(fw.value, dtype=torch.uint8, device='cuda'); bws = torch.zeros(bw.value, dtype=torch.uint8, device='cuda')
v = torch.empty(bt, N, device='cuda'); th = torch.empty_like(v); tot = torch.empty(bt, device='cuda'); last = torch.empty_like(tot)
gt = torch.full((bt,), 1.0 / bt, device='cuda'); grad = torch.zeros_like(flat)
st = torch.cuda.current_stream().cuda_stream
amd.set_option('train_mapping', 1)
for it in range(2):
    assert lib.gns_forward(ctypes.byref(cfg), topo.blob.data_ptr(), flat.data_ptr(), bu.data_ptr(), li.data_ptr(), ge.data_ptr(), bt, None,
                           v.data_ptr(), th.data_ptr(), tot.data_ptr(), last.data_ptr(), ws.data_ptr(), ws.numel(), 1, st) == 0
    assert lib.gns_backward(ctypes.byref(cfg), topo.blob.data_ptr(), flat.data_ptr(), bu.data_ptr(), li.data_ptr(), ge.data_ptr(), bt, None,
                            ws.data_ptr(), ws.numel(), gt.data_ptr(), None, None, None, grad.data_ptr(), bws.data_ptr(), bws.numel(), st) == 0
torch.cuda.synchronize()
groups = (bt + 63) // 64
RB = 1 + 5
off_slots = (groups * N * (RB + 1) * 64 * 16 + 255) // 256 * 256
blocks = min(groups, 256)
st_ = bws[off_slots: off_slots + blocks * 8 * 10 * 4].view(torch.float32).view(blocks, 8, 10).cpu().numpy()
names = ['Pb0', 'barrier1', 'Pb-edge', 'barrier2', 'gather G', 'pass theta(l=0)', 'pass v(l=1)', 'pass m(l=2)', 'finalize', '-']
tot_c = st_.sum(axis=2)
print('cycles per wave: mean %.0f  min %.0f  max %.0f' % (tot_c.mean(), tot_c.min(), tot_c.max()))
for i, nme in enumerate(names[:9]):
    x = st_[:, :, i]
    print(f'{nme:18s} share {100 * x.sum() / st_.sum():5.1f}%   per-wave mean {x.mean():10.0f}  min {x.min():10.0f}  max {x.max():10.0f}')
print('per-wave totals of block 0 (wave 0..7):', ' '.join(f'{t:.0f}' for t in tot_c[0]))
print('wave-by-wave family-pass cycles, block 0:')
for w in range(8):
    print('  wave', w, ' '.join(f'{names[i]}={st_[0, w, i]:.0f}' for i in (4, 5, 6, 7)))

sub = bws[off_slots + blocks * 8 * 10 * 4: off_slots + blocks * 8 * 10 * 4 + blocks * 8 * 12 * 4].view(torch.float32).view(blocks, 8, 12).cpu().numpy()
snames = ['row loads until complete', 'L\' recompute (mlp2_fwd)', 'output layer: W4^T rows + dW4 passes', 'hidden layer: W2^T rows + dW2 pass',
          'dW1 windows (3 passes)', 'input adjoints (W1^T stream)', 'phi head', 'lines: tail + W2^T rows', 'lines: dW1/dW2 passes',
          'G1 -> m adjoints + latent dW1 passes', 'row stores until complete', 'outside the bus loops (flush, physics phases)']
print('V2 family sweeps, drained segments (shares of the stamped total; every stamp waits for the wave\'s memory operations):')
for i, nme in enumerate(snames):
    x = sub[:, :, i]
    print(f'  {nme:46s} {100 * x.sum() / sub.sum():5.1f}%   per-wave mean {x.mean():10.0f} cycles')
