"""CPU-only checks of everything around the kernels: the C-ABI library loads and exports what include/gns_hip.h
declares, the topology builder, parameter bookkeeping (state_dict compatibility with the reference), the synthetic
generator, sharding helpers.  No compute call touches a GPU here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import opf_graph_neural_solver_amd as amd
from opf_graph_neural_solver_amd import _lib
from helpers import ROOT, golden_names, load_golden, cfg_of
from oracle import gns_oracle as orc


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'gns_hip.h')).read()
    declared = set(re.findall(r'\b(gns_[a-z_]+)\s*\(', hdr))
    lib = amd.load_library()
    for sym in declared:
        assert hasattr(lib, sym), f'{sym} declared in gns_hip.h but not exported'
    assert set(_lib.EXPORTS) <= declared
    assert lib.gns_version().startswith(b'gns_hip')


@pytest.mark.parametrize('d,h,K,multi', [(20, 10, 4, True), (20, 10, 4, False), (10, 10, 30, False), (20, 10, 10, True)])
def test_param_count_and_state_dict_keys_match_reference_layout(d, h, K, multi):
    lib = amd.load_library()
    cfg = _lib.GnsConfig(118, 186, 54, K, d, h, int(multi), 0.9)
    n = ctypes.c_int64()
    assert lib.gns_param_count(ctypes.byref(cfg), ctypes.byref(n)) == 0
    m = amd.GNS(d, h, K, 0.9, multi)
    assert n.value == sum(p.numel() for p in m.parameters())
    spec = orc.param_spec(d, h, K, multi)          # restates GNS/main.py:113-134
    assert [k for k, _ in spec] == list(m.state_dict().keys())
    assert [tuple(s) for _, s in spec] == [tuple(v.shape) for v in m.state_dict().values()]
    assert m.multiple_phis == multi and m.K == K and m.latent_dim == d and m.gamma == 0.9


def test_same_seed_same_initial_weights_as_reference_golden():
    """Construction order equals the reference's, so torch.manual_seed(s) reproduces its initial weights."""
    g = load_golden('c14_b1_K4_d20_multi')
    torch.manual_seed(int(g['seed']))
    m = amd.GNS(20, 10, 4, 0.9, True)
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).numpy()
    assert np.array_equal(flat, g['params'])


def test_flat_parameter_views_survive_load_state_dict_and_optimizer():
    m = amd.GNS(10, 10, 2, 0.9, False)
    flat = m.flat_parameters()
    assert flat.numel() == sum(p.numel() for p in m.parameters())
    sd = {k: torch.randn_like(v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m._ensure_flat()
    assert m.flat_parameters().data_ptr() == flat.data_ptr()       # in-place copy keeps the views
    assert torch.equal(m.flat_parameters(), torch.cat([v.reshape(-1) for v in sd.values()]))
    opt = torch.optim.Adam(m.parameters(), lr=0.1)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert torch.equal(m.flat_parameters(), torch.cat([p.detach().reshape(-1) for p in m.parameters()]))
    m2 = amd.GNS(10, 10, 2, 0.9, False)
    m2.load_state_dict(m.state_dict())                               # checkpoint round trip (main.py:308, evaluate.py:66)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def _py_topology(src, dst, N):
    E = len(src)
    in_order = sorted(range(E), key=lambda e: (dst[e], e))
    out_order = sorted(range(E), key=lambda e: (src[e], e))
    return in_order, out_order


@pytest.mark.parametrize('case', [14, 30, 118, 300])
def test_topology_blob(case):
    lib = amd.load_library()
    f, t, gb = amd.synth.case_topology(case)
    N, E, Gn = amd.synth.CASE_SHAPES[case]
    src, dst, gen = (f - 1).astype(np.int32), (t - 1).astype(np.int32), (gb - 1).astype(np.int32)
    nb = ctypes.c_size_t()
    assert lib.gns_topology_bytes(N, E, Gn, ctypes.byref(nb)) == 0
    blob = np.zeros(nb.value // 4, dtype=np.int32)
    assert lib.gns_prepare_topology(N, E, Gn, src.ctypes.data, dst.ctypes.data, gen.ctypes.data, blob.ctypes.data, blob.nbytes) == 0
    H = {k: i for i, k in enumerate(['MAGIC', 'N', 'E', 'GN', 'IN_PTR', 'IN_EID', 'IN_SRC', 'IN_A', 'IN_B', 'OUT_PTR', 'OUT_EID',
                                     'OUT_DST', 'OUT_C', 'OUT_D', 'IS_GEN', 'GEN_PTR', 'GEN_IDX', 'PART', 'P2Q', 'Q2P', 'EPART',
                                     'INCD_PTR', 'INCD', 'IN_DST', 'UPART', 'PPART', 'TOTAL'])}
    arr = lambda k, n: blob[blob[H[k]]:blob[H[k]] + n]
    assert blob[H['N']] == N and blob[H['E']] == E and blob[H['GN']] == Gn and blob[H['TOTAL']] <= blob.size
    in_order, out_order = _py_topology(src, dst, N)
    assert list(arr('IN_EID', E)) == in_order and list(arr('OUT_EID', E)) == out_order
    in_ptr, out_ptr = arr('IN_PTR', N + 1), arr('OUT_PTR', N + 1)
    assert list(np.diff(in_ptr)) == list(np.bincount(dst, minlength=N)) and list(np.diff(out_ptr)) == list(np.bincount(src, minlength=N))
    for p, e in enumerate(in_order):
        s = src[e]
        assert arr('IN_SRC', E)[p] == s and arr('IN_DST', E)[p] == dst[e]
        assert arr('IN_A', E)[p] == src[s] and arr('IN_B', E)[p] == dst[s]          # line NUMBER s (reference quirk)
        assert out_order[arr('P2Q', E)[p]] == e
    for q, e in enumerate(out_order):
        tt = dst[e]
        assert arr('OUT_DST', E)[q] == tt and arr('OUT_C', E)[q] == src[tt] and arr('OUT_D', E)[q] == dst[tt]
        assert in_order[arr('Q2P', E)[q]] == e
    assert set(np.nonzero(arr('IS_GEN', N))[0]) == set(gen.tolist())
    for wi, W in enumerate((1, 2, 4, 8, 16, 32)):
        part = blob[blob[H['PART']] + wi * 33: blob[H['PART']] + wi * 33 + 33]
        assert part[0] == 0 and part[W] == N and np.all(np.diff(part) >= 0)
    incd_ptr = arr('INCD_PTR', N + 1)
    assert incd_ptr[N] == 4 * E
    for wi, W in enumerate((1, 2, 4, 8, 16, 32)):
        up = blob[blob[H['UPART']] + wi * 33: blob[H['UPART']] + wi * 33 + 33]
        pp = blob[blob[H['PPART']] + wi * 33: blob[H['PPART']] + wi * 33 + 33]
        assert up[0] == 0 and up[W] == 2 * N and np.all(np.diff(up) >= 0)        # (family group, bus) units of the forward update phase
        assert pp[0] == 0 and pp[W] == N and np.all(np.diff(pp) >= 0)


def test_topology_rejects_bad_ids():
    lib = amd.load_library()
    src, dst, gen = np.array([0, 1, 5], np.int32), np.array([1, 2, 0], np.int32), np.array([0], np.int32)
    blob = np.zeros(4096, dtype=np.int32)
    rc = lib.gns_prepare_topology(6, 3, 1, src.ctypes.data, dst.ctypes.data, gen.ctypes.data, blob.ctypes.data, blob.nbytes)
    assert rc == 3      # bus id 5 is not a valid line index (E = 3): the reference would raise IndexError at main.py:41


def test_workspace_sizes_are_consistent():
    lib = amd.load_library()
    cfg = _lib.GnsConfig(118, 186, 54, 4, 20, 10, 1, 0.9)
    a, b, c = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.gns_workspace_bytes(ctypes.byref(cfg), 16384, 0, ctypes.byref(a), None) == 0
    assert lib.gns_workspace_bytes(ctypes.byref(cfg), 16384, 1, ctypes.byref(b), ctypes.byref(c)) == 0
    assert b.value > a.value > 0 and c.value > 0
    assert b.value < 3 * 2**30      # K+1 saved states + K hidden-sum sets of 16384 case118 grids stay below 3 GiB
    assert lib.gns_config_supported(ctypes.byref(cfg)) == 1
    assert lib.gns_config_supported(ctypes.byref(_lib.GnsConfig(118, 186, 54, 4, 12, 10, 1, 0.9))) == 1     # runs zero-padded on the (20, 10) kernels
    assert lib.gns_config_supported(ctypes.byref(_lib.GnsConfig(118, 186, 54, 4, 24, 10, 1, 0.9))) == 0
    assert lib.gns_config_supported(ctypes.byref(_lib.GnsConfig(118, 186, 54, 4, 10, 12, 1, 0.9))) == 1     # zero-padded on (20, 14)
    assert lib.gns_config_supported(ctypes.byref(_lib.GnsConfig(118, 186, 54, 4, 10, 15, 1, 0.9))) == 0


def test_synthetic_grids_follow_the_reference_layout_and_ranges():
    bu, li, ge = amd.synth.synth_grids(118, 32, seed=0)
    assert bu.shape == (32, 118, 6) and li.shape == (32, 186, 7) and ge.shape == (32, 54, 7)
    assert torch.all(li[:, :, 0:2] == li[0:1, :, 0:2]) and torch.all(ge[:, :, 0] == ge[0:1, :, 0])
    assert torch.allclose(bu[:, :, 2].sum(1), ge[:, :, 6].sum(1), rtol=1e-4)      # sum Pd == sum Pg (augment_grids.py:51)
    assert float(li[:, :, 5].min()) >= 0.8 and float(li[:, :, 5].max()) <= 1.2     # tau
    assert float(li[:, :, 6].abs().max()) <= 0.2 * np.pi / 180 + 1e-7               # shift in radians (utils.py:35)
    assert torch.all(bu[:, :, 4] == 0.01) and torch.all(bu[:, :, 5] == -0.01)     # Gs, Bs forced (utils.py:25-30)
    assert torch.equal(ge[:, :, 3], ge[:, :, 6])                                   # Pg_set duplicate (utils.py:38)


def test_column_remap_and_shape_errors_without_gpu():
    m = amd.GNS(20, 10, 2, 0.9, True)
    bu, li, ge = amd.synth.synth_grids(14, 2, seed=0)
    with pytest.raises(amd.GNSError):           # parameters on CPU: loud failure, no fallback
        m(bu, li, ge)
    assert amd.GNS._remap(bu, None, amd.get_BLG()[0]) is bu
    B2 = {'bus_i': 0, 'type': 1, 'Qd': 2, 'Pd': 3, 'Gs': 4, 'Bs': 5}
    sw = bu[..., [0, 1, 3, 2, 4, 5]]
    assert torch.equal(amd.GNS._remap(sw, B2, amd.get_BLG()[0]), bu)


def test_shard_range_partitions_the_batch():
    for total, world in ((131072, 8), (10, 3), (7, 8)):
        spans = [amd.dist.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
