"""``GNS`` nn.Module: the reference's operator interface (GNS/main.py:107-202) over the HIP hot path.

Host side only - parameter bookkeeping, validation, topology caching, autograd glue.  All arithmetic of
the K-step loop happens in ``libgns_hip.so``.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.nn as nn

from ._lib import GNS_ERRORS, GnsConfig, get_option, load_library


class GNSError(RuntimeError):
    pass


def get_BLG():
    """Column maps of the bus / line / generator tensors (GNS/utils.py:4-13)."""
    B = {'bus_i': 0, 'type': 1, 'Pd': 2, 'Qd': 3, 'Gs': 4, 'Bs': 5}
    L = {'f_bus': 0, 't_bus': 1, 'r': 2, 'x': 3, 'b': 4, 'tau': 5, 'theta': 6}
    G = {'bus_i': 0, 'Pmax': 1, 'Pmin': 2, 'Pg_set': 3, 'vg': 4, 'qg': 5, 'Pg': 6}
    return B, L, G


_B0, _L0, _G0 = get_BLG()


class LearningBlock(nn.Module):
    """Parameter container with the reference's layer names ``linear1, linear2, linear4`` (GNS/main.py:17-31).
    ``forward`` is plain torch and exists for callers that use a block on its own; the fused GNS path reads the
    parameters directly."""

    def __init__(self, dim_in, hidden_dim, dim_out):
        super().__init__()
        self.linear1 = nn.Linear(dim_in, hidden_dim)
        self.linear2 = nn.Linear(hidden_dim, hidden_dim)
        self.linear4 = nn.Linear(hidden_dim, dim_out)
        self.lrelu = nn.LeakyReLU()

    def forward(self, x):
        return self.linear4(self.lrelu(self.linear2(self.lrelu(self.linear1(x)))))


def _check(rc, what):
    if rc != 0:
        raise GNSError(f'{what} failed: {GNS_ERRORS.get(rc, rc)}')


class _Topology:
    """Device-resident topology blob of one case + the id columns it was built from."""

    def __init__(self, lib, src, dst, gen_bus, device):
        n_line, n_gen = int(src.size), int(gen_bus.size)
        self.src, self.dst, self.gen_bus = src, dst, gen_bus
        self.n_line, self.n_gen = n_line, n_gen

    @staticmethod
    def build(lib, n_bus, src, dst, gen_bus, device):
        t = _Topology(lib, src, dst, gen_bus, device)
        t.n_bus = n_bus
        nbytes = ctypes.c_size_t()
        _check(lib.gns_topology_bytes(n_bus, t.n_line, t.n_gen, ctypes.byref(nbytes)), 'gns_topology_bytes')
        host = np.zeros(nbytes.value // 4, dtype=np.int32)
        s32, d32 = np.ascontiguousarray(src, dtype=np.int32), np.ascontiguousarray(dst, dtype=np.int32)
        g32 = np.ascontiguousarray(gen_bus if gen_bus.size else np.zeros(1), dtype=np.int32)
        rc = lib.gns_prepare_topology(n_bus, t.n_line, t.n_gen, s32.ctypes.data, d32.ctypes.data, g32.ctypes.data,
                                      host.ctypes.data, host.nbytes)
        if rc == 3:
            raise ValueError('invalid topology: bus ids must be 1..N and, because the reference gathers per-line '
                             'arrays with bus ids (GNS/main.py:41), every connected bus id must also be <= E')
        _check(rc, 'gns_prepare_topology')
        t.blob = torch.from_numpy(host).to(device)
        return t


POISON_WORKSPACES = False     # test hook: workspaces are handed to the library filled with 0xFF bytes (NaN as floats) instead of
                              # uninitialised, so that a kernel reading a word nothing wrote shows up as NaN instead of by chance


def _workspace(nbytes, dev):
    if POISON_WORKSPACES:
        return torch.full((nbytes,), 0xFF, dtype=torch.uint8, device=dev)
    return torch.empty(nbytes, dtype=torch.uint8, device=dev)


def _raise_if_team_failed(lib, cfg, Bt, ws, save_state, dev):
    st = ctypes.c_int()
    with torch.cuda.device(dev):
        _check(lib.gns_team_status(ctypes.byref(cfg), Bt, ws.data_ptr(), ws.numel(), int(save_state), ctypes.byref(st),
                                   torch.cuda.current_stream(dev).cuda_stream), 'gns_team_status')
    if st.value:
        raise GNSError('a team of workgroups gave up at a barrier: a kernel of another stream or process held a partner\'s compute unit, '
                       'so the losses of this call are NaN and no gradient was computed from them.  Teams need the device to themselves: '
                       'opf_graph_neural_solver_amd.set_option("team", 1) runs one workgroup per 64-grid group instead')


class _GNSFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, topo, want_grad, buses, lines, gens, *params):
        lib = load_library()
        Bt, N = buses.shape[0], buses.shape[1]
        cfg = mod._config(N, lines.shape[1], gens.shape[1])
        dev = buses.device
        need_grad = bool(want_grad)                     # evaluation (torch.no_grad()) keeps 2 state slots and saves nothing for a backward
        ctx.set_materialize_grads(False)                # unused outputs (v, theta, last_loss) arrive as None in backward, not as zero-filled tensors
        fwd_b, bwd_b = ctypes.c_size_t(), ctypes.c_size_t()
        _check(lib.gns_workspace_bytes(ctypes.byref(cfg), Bt, int(need_grad), ctypes.byref(fwd_b), ctypes.byref(bwd_b)),
               'gns_workspace_bytes')
        ws = _workspace(fwd_b.value, dev)
        v = torch.empty((Bt, N), dtype=torch.float32, device=dev)
        theta = torch.empty_like(v)
        total = torch.empty(Bt, dtype=torch.float32, device=dev)
        last = torch.empty_like(total)
        flat = mod._exec_flat(dev)                       # the parameters themselves, or their device mirror for a CPU-resident model
        with torch.cuda.device(dev):                     # the launch must land on the tensors' device, whatever the current device is
            stream = torch.cuda.current_stream(dev).cuda_stream
            packed = mod._packed_inputs(lib, cfg, topo, buses, lines, gens, stream, need_grad)
            _check(lib.gns_forward(ctypes.byref(cfg), topo.blob.data_ptr(), flat.data_ptr(), buses.data_ptr(), lines.data_ptr(),
                                   gens.data_ptr(), Bt, None if packed is None else packed.data_ptr(),
                                   v.data_ptr(), theta.data_ptr(), total.data_ptr(), last.data_ptr(),
                                   ws.data_ptr(), ws.numel(), int(need_grad), stream), 'gns_forward')
        # Teams of workgroups (lane-per-grid kernels on a batch that leaves CUs idle) can give up at a barrier when another kernel
        # holds a partner's CU: the losses are then NaN and the workspace carries a status word.  A training call is checked before
        # its backward is launched (no gradient of invalid losses reaches an optimiser); an evaluation call at the next call of
        # the module or by ``GNS.check_status()``.
        off = ctypes.c_size_t()
        _check(lib.gns_team_status_offset(ctypes.byref(cfg), Bt, int(need_grad), ctypes.byref(off)), 'gns_team_status_offset')
        uses_teams = off.value != ctypes.c_size_t(-1).value
        ctx.team_status = uses_teams and need_grad
        if uses_teams and not need_grad and not torch.cuda.is_current_stream_capturing():
            mod.__dict__['_pending_status'] = (cfg, Bt, ws, dev)
        if need_grad:
            ctx.cfg, ctx.topo, ctx.ws, ctx.flat, ctx.Bt, ctx.bwd_bytes = cfg, topo, ws, flat, Bt, bwd_b.value
            ctx.params = params
            ctx.mod_flat = mod._flat
            ctx.param_versions = tuple(p._version for p in params) + (mod._flat._version,)   # (a flat optimiser writes through the buffer)
            ctx.inputs = (buses, lines, gens)          # the backward of the grid-per-workgroup mapping re-reads them
            ctx.input_versions = (buses._version, lines._version, gens._version)
            ctx.packed = packed
            ctx.shapes = [p.shape for p in params]
        return v, theta, total, last

    @staticmethod
    def backward(ctx, gv, gth, gtot, glast):
        lib = load_library()
        flat = ctx.flat
        # the backward mixes weights packed by the forward with the live buffer: an in-place update in between (an
        # optimizer.step(), p.add_()) would give silently inconsistent gradients where torch autograd raises
        if tuple(p._version for p in ctx.params) + (ctx.mod_flat._version,) != ctx.param_versions:
            raise GNSError('parameters were modified in place between forward and backward')
        if tuple(t._version for t in ctx.inputs) != ctx.input_versions:
            raise GNSError('buses / lines / generators were modified in place between forward and backward')
        dev = flat.device
        if ctx.team_status and not torch.cuda.is_current_stream_capturing():     # (the check synchronises: not inside a graph capture)
            _raise_if_team_failed(lib, ctx.cfg, ctx.Bt, ctx.ws, 1, dev)
        grad = torch.zeros_like(flat)
        bws = _workspace(ctx.bwd_bytes, dev)

        def ptr(t):
            return None if t is None else t.contiguous().data_ptr()

        keep = [t.to(dev).contiguous() if t is not None else None for t in (gtot, glast, gv, gth)]
        bu, li_, ge = ctx.inputs
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _check(lib.gns_backward(ctypes.byref(ctx.cfg), ctx.topo.blob.data_ptr(), flat.data_ptr(), bu.data_ptr(), li_.data_ptr(),
                                    ge.data_ptr(), ctx.Bt, None if ctx.packed is None else ctx.packed.data_ptr(), ctx.ws.data_ptr(),
                                    ctx.ws.numel(), ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), grad.data_ptr(),
                                    bws.data_ptr(), bws.numel(), stream), 'gns_backward')
        # The persistent backward kernels (bwd_variant 1-3 or the packed-FMA engine: opt-ins) run in teams too; one that gave up leaves
        # NaN in the first gradient element (gns_backward.hip).  The default split backward has no teams and is not checked.
        if ctx.team_status and not torch.cuda.is_current_stream_capturing() and (get_option('bwd_variant') != 4 or get_option('dw_mfma') == 0):
            if bool(torch.isnan(grad[0])):
                raise GNSError('a team of workgroups gave up at a barrier of the backward kernel: no gradient is delivered '
                               '(opf_graph_neural_solver_amd.set_option("team", 1) or the default bwd_variant 4 run without teams)')
        pdev = ctx.params[0].device
        if pdev != dev:
            grad = grad.to(pdev)                          # CPU-resident model: 59 KB back to the host, like the reference's .grad
        out, off = [], 0
        for shp in ctx.shapes:
            n = int(np.prod(shp))
            out.append(grad[off:off + n].view(shp))
            off += n
        return (None, None, None, None, None, None, *out)


class GNS(nn.Module):
    """Drop-in for the reference's ``GNS`` (GNS/main.py:107-202) with a batched extension.

    ``forward(buses[N,6], lines[E,7], generators[Gn,7], B, L, G)`` returns ``(v[N], theta[N], total_loss[], last_loss[])``
    like the reference.  Additionally 3-D inputs ``[Bt,N,6], [Bt,E,7], [Bt,Gn,7]`` (what ``utils.load_all_grids``
    returns) give ``(v[Bt,N], theta[Bt,N], total_loss[Bt], last_loss[Bt])``; the topology columns must be identical
    across the batch.  ``state_dict`` keys and shapes are the reference's.
    """

    def __init__(self, latent_dim=10, hidden_dim=10, K=30, gamma=0.9, multiple_phi=False):
        super().__init__()
        self.multiple_phis = multiple_phi                      # attribute name of the reference (main.py:111)
        if self.multiple_phis:
            self.phi_v = nn.ModuleDict()
            self.phi_theta = nn.ModuleDict()
            self.phi_m = nn.ModuleDict()
        else:
            self.phi = nn.ModuleDict()
        self.L_theta = nn.ModuleDict()
        self.L_v = nn.ModuleDict()
        self.L_m = nn.ModuleDict()
        for k in range(K):                                     # same construction order => same RNG stream as the reference
            if self.multiple_phis:
                self.phi_v[str(k)] = LearningBlock(5 + latent_dim, hidden_dim, latent_dim)
                self.phi_theta[str(k)] = LearningBlock(5 + latent_dim, hidden_dim, latent_dim)
                self.phi_m[str(k)] = LearningBlock(5 + latent_dim, hidden_dim, latent_dim)
            else:
                self.phi[str(k)] = LearningBlock(5 + latent_dim, hidden_dim, 1)
            self.L_theta[str(k)] = LearningBlock(4 + 2 * latent_dim, hidden_dim, 1)
            self.L_v[str(k)] = LearningBlock(4 + 2 * latent_dim, hidden_dim, 1)
            self.L_m[str(k)] = LearningBlock(4 + 2 * latent_dim, hidden_dim, latent_dim)
        self.latent_dim = latent_dim
        self.hidden_dim = hidden_dim
        self.gamma = gamma
        self.K = K
        # host-side state (not parameters / buffers: they must not appear in state_dict)
        self.__dict__['_flat'] = None
        self.__dict__['_topo_cache'] = {}
        # 'always' (default): every call compares the id columns of EVERY grid of the batch with the cached case - one fused
        # device compare, one flag, one sync - so a batch that mixes topologies raises like it does when a case is first seen.
        # Opt-ins for loops that have validated their data set themselves (``training.fit`` does): 'grid0' compares the first
        # grid only, 'first' trusts the cached case of that shape.
        self.topology_check = 'always'
        self.__dict__['_mirror'] = None
        # True: a batch that is passed again unchanged (same tensors, same versions) is brought into the kernels' input
        # layout once instead of on every call (gns_prepack).  Off by default: the cache keeps the last batch alive.
        self.cache_packed_inputs = False
        self.__dict__['_pack_cache'] = None
        self.__dict__['_resident'] = None                      # a data set bound by bind_dataset(): packed once, batches are slices of it
        self.__dict__['_plist'] = None                         # cached list(self.parameters()): walking 430 sub-modules costs 0.25 ms
        # True: the backward delivers d loss / d parameters as ONE tensor, ``flat_leaf().grad`` (state_dict order), instead of
        # one ``.grad`` view per parameter: the 6K x 6 AccumulateGrad nodes and views of a step cost more host time than a
        # small batch's kernels take (1.2 of 1.9 ms per step at case30 x 4096).  ``training.make_optimizer`` switches it on
        # for its flat optimiser; the reference's own loop (``torch.optim.Adam(model.parameters())``) needs it off (default).
        self.flat_grad = False
        self.__dict__['_leaf'] = None

    # ---- flat parameter storage -------------------------------------------------------------------
    def _config(self, n_bus, n_line, n_gen):
        return GnsConfig(n_bus, n_line, n_gen, self.K, self.latent_dim, self.hidden_dim, int(self.multiple_phis),
                         float(self.gamma))

    def _param_list(self):
        """``list(self.parameters())``, cached: the module tree (3 Linear layers in each of up to 6K LearningBlocks) is fixed after
        construction, and the training step asks for this list half a dozen times.  Dropped whenever ``_apply`` (``.to()``,
        ``.cuda()``, ``.float()``) or ``load_state_dict`` ran; re-validated against the first and last registered parameter on every
        use (a parameter replaced in the middle of the tree by hand needs ``model._plist = None``)."""
        pl = self._plist
        if pl is not None:
            # parameters replaced outside _apply / load_state_dict (``block.linear1.weight = nn.Parameter(...)``) would leave the
            # list stale: the first and the last registered parameter are looked up directly (no tree walk) and compared
            first = (self.phi_v if self.multiple_phis else self.phi)['0'].linear1.weight
            if pl[0] is first and pl[-1] is self.L_m[str(self.K - 1)].linear4.bias:
                return pl
        pl = list(self.parameters())
        self.__dict__['_plist'] = pl
        return pl

    def _apply(self, fn, *args, **kwargs):
        self.__dict__['_plist'] = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.__dict__['_plist'] = None
        return super().load_state_dict(*args, **kwargs)

    def _ensure_flat(self):
        """All parameters are views of ONE contiguous fp32 buffer in state_dict order: it is what the kernels read
        and what a data-parallel all-reduce sends.  Re-established after .to()/.load_state_dict() replaced storages."""
        params = self._param_list()
        flat = self._flat
        ok = flat is not None and flat.device == params[0].device
        if ok:
            off, base, esz = 0, flat.data_ptr(), 4
            for p in params:
                if p.dtype != torch.float32 or p.data_ptr() != base + off * esz or not p.is_contiguous():
                    ok = False
                    break
                off += p.numel()
        if not ok:
            if any(p.dtype != torch.float32 for p in params):
                raise GNSError('GNS parameters must be float32')
            flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
            off = 0
            for p in params:
                p.data = flat[off:off + p.numel()].view(p.shape)
                off += p.numel()
            self.__dict__['_flat'] = flat
        return params

    def _exec_flat(self, dev):
        """The flat parameter buffer on the device the kernels run on.  A model that lives on a ROCm device is read in
        place; a CPU-resident model (the reference never calls .to('cuda'): GNS/main.py:227,230-233) is mirrored, 59 KB
        host-to-device per call, and its gradient is copied back by the backward."""
        flat = self._flat
        if flat.device == dev:
            return flat
        mir = self._mirror
        if mir is None or mir.device != dev or mir.shape != flat.shape:
            mir = torch.empty_like(flat, device=dev)
            self.__dict__['_mirror'] = mir
        mir.copy_(flat, non_blocking=False)
        return mir

    # ---- inputs in the kernels' layout ----------------------------------------------------------------
    def bind_dataset(self, all_buses, all_lines, all_generators):
        """Bring a device-resident data set ``[S,N,6] [S,E,7] [S,Gn,7]`` (what ``utils.load_all_grids`` returns, GNS/utils.py:57-68,
        ``main.py:255``) into the lane-per-grid kernels' input layout ONCE.  Afterwards a batch that is a slice
        ``all_*[lo:lo+bs]`` of these very tensors with ``lo`` a multiple of 64 (the reference's batches of 128 in order,
        ``main.py:276-281``) is read from the packed copy - 64-grid groups ``lo/64 ...`` of it - and no input packing kernel runs
        for it; any other input is packed per call as before.  The id columns of the whole set must be identical (checked).
        ``unbind_dataset()`` releases the copy (17.6 KB per case118 grid)."""
        lib = load_library()
        if not (all_buses.is_cuda and all_lines.is_cuda and all_generators.is_cuda):
            raise ValueError('bind_dataset needs device-resident tensors')
        for t in (all_buses, all_lines, all_generators):
            if t.dim() != 3 or t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError('bind_dataset needs contiguous float32 [S,...] tensors')
        S, N = all_buses.shape[0], all_buses.shape[1]
        if not (all_lines.shape[0] == all_generators.shape[0] == S) or S == 0:
            raise ValueError('batch sizes of buses, lines, generators differ (or are zero)')
        if all_buses.shape[-1] != 6 or all_lines.shape[-1] != 7 or all_generators.shape[-1] != 7:
            raise ValueError('expected buses[...,6], lines[...,7], generators[...,7] (GNS/utils.py:4-13)')
        self.__dict__['_resident'] = None
        old = self.topology_check
        self.topology_check = 'always'                # the whole set is compared here, once
        try:
            topo = self._topology(all_lines, all_generators, N)
        finally:
            self.topology_check = old
        cfg = self._config(N, all_lines.shape[1], all_generators.shape[1])
        nbytes, gbytes = ctypes.c_size_t(), ctypes.c_size_t()
        _check(lib.gns_prepack_bytes(ctypes.byref(cfg), S, ctypes.byref(nbytes)), 'gns_prepack_bytes')
        _check(lib.gns_prepack_bytes(ctypes.byref(cfg), 64, ctypes.byref(gbytes)), 'gns_prepack_bytes')
        packed = torch.empty(nbytes.value, dtype=torch.uint8, device=all_buses.device)
        with torch.cuda.device(all_buses.device):
            _check(lib.gns_prepack(ctypes.byref(cfg), topo.blob.data_ptr(), all_buses.data_ptr(), all_lines.data_ptr(),
                                   all_generators.data_ptr(), S, packed.data_ptr(), packed.numel(),
                                   torch.cuda.current_stream(all_buses.device).cuda_stream), 'gns_prepack')
        tens = (all_buses, all_lines, all_generators)
        self.__dict__['_resident'] = dict(tensors=tens, versions=tuple(t._version for t in tens), topo=topo, packed=packed,
                                          group_bytes=gbytes.value, S=S, hits=0)

    def unbind_dataset(self):
        self.__dict__['_resident'] = None

    def _resident_slice(self, topo, buses, lines, gens):
        """The packed 64-grid groups of a bound data set that hold this batch, or None when the batch is not an aligned slice of it."""
        R = self._resident
        if R is None or R['topo'] is not topo:
            return None
        if tuple(t._version for t in R['tensors']) != R['versions']:
            self.__dict__['_resident'] = None         # the data set was written to: its packed copy is stale
            return None
        lo = None
        for t, r in zip((buses, lines, gens), R['tensors']):
            if t.device != r.device or t.untyped_storage().data_ptr() != r.untyped_storage().data_ptr() or t.shape[1:] != r.shape[1:]:
                return None
            per = r.shape[1] * r.shape[2]
            d = t.storage_offset() - r.storage_offset()
            if d < 0 or d % per or not t.is_contiguous():
                return None
            if lo is None:
                lo = d // per
            elif lo != d // per:
                return None
        if lo % 64 or lo + buses.shape[0] > R['S']:
            return None
        R['hits'] += 1
        return R['packed'][(lo // 64) * R['group_bytes']:]

    def _packed_inputs(self, lib, cfg, topo, buses, lines, gens, stream, need_grad):
        if not lib.gns_uses_packed_inputs(ctypes.byref(cfg), buses.shape[0], int(need_grad)):
            return None                              # (the grid-per-workgroup kernels read the caller's tensors in place)
        res = self._resident_slice(topo, buses, lines, gens)
        if res is not None:
            return res
        if not self.cache_packed_inputs:
            return None
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (buses, lines, gens)) + (id(topo),)
        ent = self._pack_cache
        if ent is not None and ent[0] == key:
            return ent[2]
        nbytes = ctypes.c_size_t()
        _check(lib.gns_prepack_bytes(ctypes.byref(cfg), buses.shape[0], ctypes.byref(nbytes)), 'gns_prepack_bytes')
        packed = torch.empty(nbytes.value, dtype=torch.uint8, device=buses.device)
        _check(lib.gns_prepack(ctypes.byref(cfg), topo.blob.data_ptr(), buses.data_ptr(), lines.data_ptr(), gens.data_ptr(),
                               buses.shape[0], packed.data_ptr(), packed.numel(), stream), 'gns_prepack')
        # the entry holds the tensors themselves: their storage cannot be freed and handed to other data while it is cached
        self.__dict__['_pack_cache'] = (key, (buses, lines, gens, topo), packed)
        return packed

    def check_status(self):
        """Raise ``GNSError`` if the last evaluation-mode call ran teams of workgroups and one of them gave up at a barrier (its losses
        are NaN).  Synchronises with the device; called automatically at the next call of the module.  Training-mode calls are
        checked before their backward is launched."""
        pend = self.__dict__.get('_pending_status')
        if pend is not None:
            self.__dict__['_pending_status'] = None
            cfg, Bt, ws, dev = pend
            _raise_if_team_failed(load_library(), cfg, Bt, ws, 0, dev)

    def flat_leaf(self):
        """A leaf tensor (``requires_grad``) that aliases the flat parameter buffer: with ``flat_grad = True`` the autograd graph
        hangs on it and its ``.grad`` is the flat gradient.  It shares storage and version counter with the parameters."""
        self._ensure_flat()
        leaf, flat = self._leaf, self._flat
        if leaf is None or leaf.data_ptr() != flat.data_ptr() or leaf.device != flat.device or leaf.numel() != flat.numel():
            leaf = flat.detach().requires_grad_(True)
            self.__dict__['_leaf'] = leaf
        return leaf

    def zero_grad(self, set_to_none: bool = True):
        leaf = self._leaf
        if leaf is not None and leaf.grad is not None:
            if set_to_none:
                leaf.grad = None
            else:
                leaf.grad.zero_()
        return super().zero_grad(set_to_none)

    def flat_parameters(self):
        """The flat fp32 parameter buffer (state_dict order); parameters are views into it."""
        self._ensure_flat()
        return self._flat

    # ---- topology -----------------------------------------------------------------------------------
    def _topology(self, lines3, gens3, n_bus):
        dev = lines3.device
        key = (n_bus, lines3.shape[1], gens3.shape[1], str(dev))
        ids_l = lines3[0, :, 0:2]
        ids_g = gens3[0, :, 0]
        ent = self._topo_cache.get(key)
        if ent is not None and self.topology_check == 'first':
            return ent[0]
        if ent is not None:
            if self.topology_check == 'grid0':
                same = bool(((ids_l == ent[1]).all() & (ids_g == ent[2]).all()).item())
            else:                                        # the whole batch against the cached ids: one flag, one sync
                same = bool(((lines3[:, :, 0:2] == ent[1]).all() & (gens3[:, :, 0] == ent[2]).all()).item())
            if same:
                return ent[0]
        if not (bool((lines3[:, :, 0:2] == ids_l).all()) and bool((gens3[:, :, 0] == ids_g).all())):
            raise ValueError('f_bus / t_bus / generator bus columns differ across the batch: the fused path needs one '
                             'topology per call (group grids by topology on the host)')
        l_np, g_np = ids_l.detach().cpu().numpy().astype(np.float64), ids_g.detach().cpu().numpy().astype(np.float64)
        if not (np.all(l_np == np.round(l_np)) and np.all(g_np == np.round(g_np))):
            raise ValueError('bus id columns must hold integers')
        src, dst, gb = l_np[:, 0].astype(np.int64) - 1, l_np[:, 1].astype(np.int64) - 1, g_np.astype(np.int64) - 1
        if src.min(initial=0) < 0 or dst.min(initial=0) < 0 or max(src.max(initial=0), dst.max(initial=0)) >= n_bus \
                or (gb.size and (gb.min() < 0 or gb.max() >= n_bus)):
            raise ValueError(f'bus ids must lie in 1..{n_bus}')
        topo = _Topology.build(load_library(), n_bus, src, dst, gb, dev)
        self._topo_cache[key] = (topo, ids_l.clone(), ids_g.clone())
        return topo

    # ---- forward --------------------------------------------------------------------------------------
    @staticmethod
    def _remap(t, cols, default):
        if cols is None or dict(cols) == default:
            return t
        missing = [k for k in default if k not in cols]
        if missing:
            raise ValueError(f'column map lacks {missing}')
        order = [cols[k] for k in sorted(default, key=default.get)]
        return t[..., order]

    def forward(self, buses, lines, generators, B=None, L=None, G=None):
        if self.__dict__.get('_pending_status') is not None and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            self.check_status()                             # (it synchronises: never inside a graph capture)
        params = self._ensure_flat()
        dev = params[0].device
        if dev.type != 'cuda':
            # a CPU-resident model, as the reference builds it (GNS/main.py:227): the kernels still run on the GPU, on a
            # mirror of the parameters; there is no CPU implementation to fall back to
            if not torch.cuda.is_available():
                raise GNSError('the GNS hot path runs on a ROCm device only and none is visible (there is no CPU fallback)')
            dev = torch.device('cuda', torch.cuda.current_device())
        single = buses.dim() == 2
        if single:
            if lines.dim() != 2 or generators.dim() != 2:
                raise ValueError('buses, lines, generators must all be 2-D (one grid) or all 3-D (a batch)')
            buses, lines, generators = buses.unsqueeze(0), lines.unsqueeze(0), generators.unsqueeze(0)
        if not (buses.dim() == lines.dim() == generators.dim() == 3):
            raise ValueError('buses, lines, generators must all be 2-D (one grid) or all 3-D (a batch)')
        if not (buses.shape[0] == lines.shape[0] == generators.shape[0]) or buses.shape[0] == 0:
            raise ValueError('batch sizes of buses, lines, generators differ (or are zero)')
        in_dev = buses.device
        buses, lines, generators = self._remap(buses, B, _B0), self._remap(lines, L, _L0), self._remap(generators, G, _G0)
        if buses.shape[-1] != 6 or lines.shape[-1] != 7 or generators.shape[-1] != 7:
            raise ValueError('expected buses[...,6], lines[...,7], generators[...,7] (GNS/utils.py:4-13)')
        for name, t in (('buses', buses), ('lines', lines), ('generators', generators)):
            if t.dtype != torch.float32:
                raise ValueError(f'{name} must be float32')
        buses = buses.to(dev).contiguous()
        lines = lines.to(dev).contiguous()
        generators = generators.to(dev).contiguous()
        topo = self._topology(lines, generators, buses.shape[1])
        # whether a backward pass can follow is decided HERE: inside Function.forward grad mode is always off, and
        # ctx.needs_input_grad ignores torch.no_grad()
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if self.flat_grad and want_grad and params[0].device.type == 'cuda':
            params = [self.flat_leaf()]                     # one differentiable input: the gradient comes back as one tensor
        v, theta, total, last = _GNSFunction.apply(self, topo, want_grad, buses, lines, generators, *params)
        if in_dev != dev:
            v, theta, total, last = v.to(in_dev), theta.to(in_dev), total.to(in_dev), last.to(in_dev)
        if single:
            return v[0], theta[0], total[0], last[0]
        return v, theta, total, last
