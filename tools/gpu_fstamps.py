"""Diagnostic: per-phase cycle shares of the forward kernel (STAMPS build, eval mode, raw C-ABI)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GNS_LIB', os.path.join(ROOT, 'tools', 'abl_FSTAMPS.so'))
import torch
import opf_graph_neural_solver_amd as amd
from opf_graph_neural_solver_amd._lib import GnsConfig
lib = amd.load_library()
case, bt, K = 118, 16384, 4
N, E, Gn = amd.synth.CASE_SHAPES[case]
m = amd.GNS(20, 10, K, 0.9, True).cuda()
flat = m.flat_parameters()
bu, li, ge = amd.synth.synth_grids(case, bt, seed=1, device='cuda')
topo = m._topology(li, ge, N)
cfg = GnsConfig(N, E, Gn, K, 20, 10, 1, 0.9)
fw = ctypes.c_size_t()
lib.gns_workspace_bytes(ctypes.byref(cfg), bt, 0, ctypes.byref(fw), None)
ws = torch.zeros(fw.value, dtype=torch.uint8, device='cuda')
v = torch.empty(bt, N, device='cuda'); th = torch.empty_like(v); tot = torch.empty(bt, device='cuda'); last = torch.empty_like(tot)
st = torch.cuda.current_stream().cuda_stream
for it in range(2):
    assert lib.gns_forward(ctypes.byref(cfg), topo.blob.data_ptr(), flat.data_ptr(), bu.data_ptr(), li.data_ptr(), ge.data_ptr(), bt,
                           v.data_ptr(), th.data_ptr(), tot.data_ptr(), last.data_ptr(), ws.data_ptr(), ws.numel(), 0, st) == 0
torch.cuda.synchronize()
# layout (gns_common.h, case118 K=4 d=20 h=10 multi): off_lam = off_pt + off_pn + packed inputs
off_lam = 45824 + 44032 + 256 * 1099 * 1024
nw = 16
st_ = ws[off_lam: off_lam + 256 * nw * 6 * 4].view(torch.float32).view(256, nw, 6).cpu().numpy()
names = ['phase U', 'barrier U', 'phase P', 'barrier P', 'lambda+fixup+loss', 'loop head']
tot_c = st_.sum(axis=2)
print('cycles per wave: mean %.0f  min %.0f  max %.0f' % (tot_c.mean(), tot_c.min(), tot_c.max()))
for i, nme in enumerate(names):
    x = st_[:, :, i]
    print(f'{nme:20s} share {100 * x.sum() / st_.sum():5.1f}%   per-wave mean {x.mean():10.0f}  min {x.min():10.0f}  max {x.max():10.0f}')
print('block 0 phase-U cycles per wave:', ' '.join(f'{t:.0f}' for t in st_[0, :, 0]))
print('block 0 phase-P cycles per wave:', ' '.join(f'{t:.0f}' for t in st_[0, :, 2]))
