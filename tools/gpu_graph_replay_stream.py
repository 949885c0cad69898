"""Ad-hoc: a captured training step replayed on the default stream of an idle GPU against the same graph replayed on a stream of
its own (what training.GraphedStep.run does), both against eager steps.  usage: python tools/gpu_graph_replay_stream.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import opf_graph_neural_solver_amd as amd
from opf_graph_neural_solver_amd import training as T
bus, br, gen = amd.synth.raw_case_arrays(14, 256, seed=3, zero_tau_fraction=0.0)
buses, lines, gens = amd.prepare_grids(bus.cuda(), br.cuda(), gen.cuda())


def dosync(kind):
    if kind == 'device': torch.cuda.synchronize()
    elif kind == 'stream': torch.cuda.current_stream().synchronize()
    elif kind == 'event':
        e = torch.cuda.Event(); e.record(); e.synchronize()
    elif kind == 'item': probe.sum().item()
    elif kind == 'sleep':
        import time; time.sleep(0.05)


probe = torch.ones(4, device='cuda')


def run(mode, sync=False):
    torch.manual_seed(0)
    m = amd.GNS(20, 10, 4, 0.9, True).cuda()
    opt = T.make_optimizer(m, 'Adam', lr=1e-3)
    st = None
    for step in range(8):
        sl = slice(128 * (step % 2), 128 * (step % 2) + 128)
        if mode == 'eager':
            T.train_step(m, opt, buses[sl], lines[sl], gens[sl])
        elif mode == 'run':
            if st is None: st = T.GraphedStep(m, opt, buses[sl], lines[sl], gens[sl])
            dosync(sync)
            st.run(buses[sl], lines[sl], gens[sl])
        else:                                   # the graph launched on the current (default) stream
            if st is None: st = T.GraphedStep(m, opt, buses[sl], lines[sl], gens[sl])
            for dst, src in zip(st.static, (buses[sl], lines[sl], gens[sl])): dst.copy_(src, non_blocking=True)
            dosync(sync)
            st.graph.replay()
    torch.cuda.synchronize()
    return m.flat_parameters().detach().clone()


p0 = run('eager')
for rep in range(1):
    for mode, sync in (('run', False), ('run', 'device'), ('run', 'stream'), ('run', 'event'), ('run', 'item'), ('run', 'sleep'), ('run', False)):
        p = run(mode, sync)
        print(f'{mode:15s} host sync before each replay {sync!s:5s}: parameters after 8 steps equal the eager loop\'s: {bool(torch.equal(p, p0))}  max diff {float((p - p0).abs().max()):.3e}', flush=True)
