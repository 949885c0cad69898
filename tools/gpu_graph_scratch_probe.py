"""Ad-hoc: forward + backward captured as one HIP graph, replayed with and without a host wait on the stream in between; prints
whether consecutive replays deliver the same gradient.  Run under DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 with / without
HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 to see which runtime mechanism the wrong replays need (profiles/r03/graph_replay_packet_capture.txt)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import opf_graph_neural_solver_amd as amd
bu, li, ge = amd.synth.synth_grids(14, 128, seed=3, device='cuda')
for kind in ('none', 'stream'):
    torch.manual_seed(0)
    m = amd.GNS(20, 10, 4, 0.9, True).cuda()
    m.topology_check = 'first'
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = m(bu, li, ge); out[2].mean().backward(); del out
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    m.zero_grad(set_to_none=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out2 = m(bu, li, ge)
        out2[2].mean().backward()
        g_static = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    res = []
    for rep in range(4):
        if kind == 'stream': torch.cuda.current_stream().synchronize()
        graph.replay()
        res.append(g_static.clone())
    torch.cuda.synchronize()
    print('host wait before each replay:', kind, '| replays deliver one gradient:', all(bool(torch.equal(res[0], r)) for r in res[1:]), '| NaN', [int(torch.isnan(g).sum()) for g in res], flush=True)
