"""Eager vs captured training step on a resident, bound batch (bench.py's step).  usage: python tools/gpu_graph_step.py [case bt K]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import opf_graph_neural_solver_amd as amd
case = int(sys.argv[1]) if len(sys.argv) > 1 else 118
bt = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.manual_seed(0)
m = amd.GNS(20, 10, K, 0.9, True).cuda()
opt = amd.training.make_optimizer(m)
bu, li, ge = amd.synth.synth_grids(case, bt, seed=1, device='cuda')
m.bind_dataset(bu, li, ge); m.topology_check = 'first'
for _ in range(5):
    amd.training.train_step(m, opt, bu, li, ge)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        amd.training.train_step(m, opt, bu, li, ge)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 50 * 1e3
    print(f'eager   {eager:.4f} ms/step', flush=True)
g = amd.training.GraphedStep(m, opt, bu, li, ge, copy_inputs=False)
for _ in range(5):
    g.run(bu, li, ge)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        g.run(bu, li, ge)
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 50 * 1e3
    print(f'graph   {graph:.4f} ms/step   resident hits {m._resident["hits"]}', flush=True)
