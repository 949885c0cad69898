"""world_size-2 run of the grid-sharded data-parallel path over gloo (CPU): the flat-gradient all-reduce of
dist.allreduce_gradients must reproduce the single-process gradient of the mean over the global batch.
The per-shard gradients come from the CPU oracle here (the HIP kernels need a GPU); what is under test is the
sharding + ONE flat all-reduce + rescale logic that bench.py runs over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import opf_graph_neural_solver_amd as amd
    from oracle import gns_oracle as orc
    torch.manual_seed(0)
    model = amd.GNS(10, 10, 2, 0.9, True)
    bu, li, ge = amd.synth.synth_grids(14, 7, seed=11)          # 7 grids: uneven shards (4 + 3)
    lo, hi = amd.dist.shard_range(7, rank, world)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    _, _, _, _, g_local = orc.gns_forward_backward(flat, bu[lo:hi], li[lo:hi], ge[lo:hi], latent_dim=10, hidden_dim=10, K=2,
                                                   gamma=0.9, multiple_phi=True)     # gradient of the LOCAL mean
    off = 0
    gbuf = g_local.clone()
    for p in model.parameters():
        p.grad = gbuf[off:off + p.numel()].view(p.shape)        # views of one flat buffer, like the fused backward returns
        off += p.numel()
    red = amd.dist.allreduce_gradients(model, global_batch=7, local_batch=hi - lo)
    assert red.data_ptr() == gbuf.data_ptr()                     # zero-copy: the flat buffer itself was reduced
    if rank == 0:
        _, _, _, _, g_ref = orc.gns_forward_backward(flat, bu, li, ge, latent_dim=10, hidden_dim=10, K=2, gamma=0.9, multiple_phi=True)
        np.save(os.path.join(out_dir, 'err.npy'), np.array([float((red - g_ref).abs().max()), float(g_ref.abs().max())]))
    # non-flat gradients (some other producer) take the fallback path and still reduce correctly
    for p in model.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    red2 = amd.dist.allreduce_gradients(model)
    assert torch.all(red2 == 3.0) and all(torch.all(p.grad == 3.0) for p in model.parameters())
    dist.destroy_process_group()


def test_two_rank_flat_gradient_allreduce(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    err, scale = np.load(os.path.join(str(tmp_path), 'err.npy'))
    assert err <= 2e-6 * scale + 1e-8, (err, scale)
