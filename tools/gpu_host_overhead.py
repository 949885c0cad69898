"""Host time per training step (python + autograd + ctypes) against the GPU time of its kernels, for a small batch where the
host is the limiter.  usage: python tools/gpu_host_overhead.py [case] [batch] [K]"""
import sys, os, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import opf_graph_neural_solver_amd as amd
case = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bt = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
m = amd.GNS(20, 10, K, 0.9, True).cuda()
bu, li, ge = amd.synth.synth_grids(case, bt, seed=1, device='cuda')
m.cache_packed_inputs = True
opt = amd.training.make_optimizer(m)
for it in range(5):
    amd.training.train_step(m, opt, bu, li, ge)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for it in range(n):
    amd.training.train_step(m, opt, bu, li, ge)
t1 = time.perf_counter()           # host has issued everything
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'case{case} x {bt} K={K}: host issue {(t1 - t0) / n * 1e3:.3f} ms/step, wall {(t2 - t0) / n * 1e3:.3f} ms/step', flush=True)
pr = cProfile.Profile(); pr.enable()
for it in range(50):
    amd.training.train_step(m, opt, bu, li, ge)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print(s.getvalue()[:3500])
