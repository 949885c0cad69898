"""Batched training loop around the fused forward/backward: what ``main()`` does at ``GNS/main.py:235-309`` without the
per-grid Python loop and without wandb.  Data-parallel when ``torch.distributed`` is initialised (one flat-gradient
all-reduce per step, ``dist.allreduce_gradients``)."""
from __future__ import annotations

import os

import torch
import torch.distributed as tdist

from . import dist as gdist
from .gns import GNSError, get_BLG


def checkpoint_name(case_nr, K, latent_dim, hidden_dim, multiple_phi, optimizer_name):
    """File name of the reference's checkpoints (``main.py:308-309``) so that files interchange."""
    return f'best_model_c{case_nr}_K{K}_L{latent_dim}_H{hidden_dim}_{multiple_phi}_optim{optimizer_name}.pth'


class FlatOptimizer(torch.optim.Optimizer):
    """The reference's optimiser (``torch.optim.Adam`` / ``Adagrad``, ``main.py:236-243``) run on the model's ONE flat
    parameter buffer instead of its 144 tensors: the parameters are views of that buffer (``GNS.flat_parameters``) and the
    fused backward hands their gradients as views of one flat gradient (``dist.flat_gradient``), so the element-wise update
    is the same arithmetic in one kernel launch instead of six multi-tensor chunks (34 -> 6 us per step on MI355X).
    For Adam on a GPU model that launch is the library's own kernel (``gns_adam_step``, ``include/gns_hip.h``: torch's update
    rule with its defaults, 3 us instead of torch's two launches of 20 us); the moment estimates live in the inner optimiser's
    state, so ``state_dict`` / ``load_state_dict`` are those of ``torch.optim.Adam`` over the flat tensor either way
    (the model's own ``state_dict`` - what the reference checkpoints, ``main.py:308`` - is unaffected).

    It IS a ``torch.optim.Optimizer`` (``param_groups`` / ``state`` are the inner optimiser's own objects), so
    ``torch.optim.lr_scheduler.*`` - the warm-up ``LambdaLR`` the reference keeps commented out at ``main.py:245-252`` - drives it
    like any other.  Per-parameter ``.grad`` tensors do not exist while ``GNS.flat_grad`` is on (the gradient is ``flat_leaf().grad``):
    clip or log through ``dist.flat_gradient(model)``."""

    def __init__(self, model, inner_cls, native=None, capturable=False, **kw):
        self.model = model
        self._cls, self._kw = inner_cls, kw
        self._flat = None
        self.inner = None
        self._native = native
        # True: the native Adam keeps its step counter on the device (``gns_adam_step_dev``), so that a whole step can be captured
        # into a HIP graph and replayed (``GraphedStep``); ``state['step']`` is then a 0-dim device tensor like torch's capturable Adam
        self.capturable = bool(capturable)
        self._dev_state = None
        self._base_init = False
        self._bind()

    def _bind(self):
        flat = self.model.flat_parameters()
        if self._flat is None or self._flat.data_ptr() != flat.data_ptr() or self._flat.device != flat.device:
            # (re)built when .to() / load_state_dict replaced the storage; the moment estimates restart with it
            if getattr(self.model, 'flat_grad', False) and flat.is_cuda:
                self._flat = self.model.flat_leaf()                                # the leaf the backward hands its one gradient to
            else:
                self._flat = torch.nn.Parameter(flat.detach(), requires_grad=True)     # shares the storage
            kw = dict(self._kw)
            native = self._native
            if native is None:
                native = self._cls is torch.optim.Adam and flat.is_cuda and flat.dtype == torch.float32
            self._use_native = bool(native)
            if self._use_native:
                kw.pop('fused', None)           # the inner object only keeps hyper-parameters and state
            old_groups = self.inner.param_groups if self.inner is not None else None
            self.inner = self._cls([self._flat], **kw)
            if old_groups is not None:          # hyper-parameters a scheduler has moved survive a rebind
                for k, v in old_groups[0].items():
                    if k != 'params':
                        self.inner.param_groups[0][k] = v
            if not self._base_init:
                torch.optim.Optimizer.__init__(self, [self._flat], dict(self.inner.defaults))
                self._base_init = True
            # one set of hyper-parameters and one state: the inner optimiser's
            self.param_groups = self.inner.param_groups
            self.state = self.inner.state
            self.defaults = self.inner.defaults

    def _native_step(self):
        from ._lib import load_library, GNS_ERRORS
        grp = self.inner.param_groups[0]
        if grp.get('weight_decay', 0) or grp.get('amsgrad', False) or grp.get('maximize', False):
            raise ValueError('FlatOptimizer(native=True) implements torch.optim.Adam without weight decay / amsgrad / maximize')
        st = self.inner.state[self._flat]
        if 'exp_avg' not in st:
            st['step'] = torch.tensor(0.0)                                         # host counter: no launch to advance it
            st['exp_avg'] = torch.zeros_like(self._flat.data)
            st['exp_avg_sq'] = torch.zeros_like(self._flat.data)
        if self.capturable:
            dst = self._dev_state
            if dst is None or dst.device != self._flat.device:
                dst = self._dev_state = torch.zeros(4, dtype=torch.float32, device=self._flat.device)
                dst[0] = float(st['step'])
            if st['step'].data_ptr() != dst.data_ptr():                            # a host counter so far, or a loaded state_dict
                dst[0] = float(st['step'])
                st['step'] = dst[0]                                                # 0-dim view: the kernel advances it
        else:
            if st['step'].is_cuda:                                                 # a state_dict of a fused / capturable Adam was loaded
                st['step'] = st['step'].cpu().clone()
            st['step'] += 1
        grad = self._flat.grad
        if grad.dtype != torch.float32 or not grad.is_contiguous() or grad.device != self._flat.device:
            raise ValueError('FlatOptimizer: the flat gradient must be a contiguous float32 tensor on the parameters\' device')
        dev = self._flat.device
        lr = grp['lr']
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if self.capturable:
                rc = load_library().gns_adam_step_dev(self._flat.data_ptr(), grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(),
                                                      self._flat.numel(), float(lr), float(grp['betas'][0]), float(grp['betas'][1]),
                                                      float(grp['eps']), self._dev_state.data_ptr(), stream)
            else:
                rc = load_library().gns_adam_step(self._flat.data_ptr(), grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(),
                                                  self._flat.numel(), float(lr), float(grp['betas'][0]), float(grp['betas'][1]),
                                                  float(grp['eps']), int(st['step'].item()), stream)
        if rc != 0:
            raise RuntimeError(f'gns_adam_step failed: {GNS_ERRORS.get(rc, rc)}')

    def step(self, closure=None):
        from . import dist as gdist
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._bind()
        g = gdist.flat_gradient(self.model)
        if self._flat.grad is not g:
            self._flat.grad = g
        if self._use_native:
            self._native_step()
        else:
            self.inner.step()
        # the update went through an alias of the buffer: tell autograd's bookkeeping (the fused backward refuses to mix
        # weights packed before an update with the live buffer, like torch autograd would)
        torch.autograd.graph.increment_version(self.model.flat_parameters())
        return loss

    def zero_grad(self, set_to_none=True):
        for p in (self.model._param_list() if hasattr(self.model, '_param_list') else self.model.parameters()):
            p.grad = None
        if self._flat is not None:
            self._flat.grad = None

    def state_dict(self):
        return self.inner.state_dict()

    def load_state_dict(self, sd):
        self.inner.load_state_dict(sd)
        self.param_groups = self.inner.param_groups
        self.state = self.inner.state


def make_optimizer(model, optimizer_name='Adam', lr=None, flat=None):
    """``main.py:236-243``: Adagrad with lr 0.01, otherwise Adam with lr 0.001.  ``flat`` (default: on for a model that
    lives on the GPU) runs the same update on the flat parameter buffer in one launch (``FlatOptimizer``)."""
    on_gpu = all(p.is_cuda for p in model.parameters())
    if flat is None:
        flat = on_gpu and hasattr(model, 'flat_parameters')
    if optimizer_name == 'Adagrad':
        kw, cls = dict(lr=0.01 if lr is None else lr), torch.optim.Adagrad
    else:
        kw, cls = dict(lr=0.001 if lr is None else lr, fused=on_gpu), torch.optim.Adam
    if flat and not all(p.requires_grad for p in model.parameters()):
        # the flat buffer (and the flat gradient leaf) spans every parameter: a model with frozen parameters
        # (requires_grad = False) gets a plain torch optimiser over the trainable ones, so the frozen ones are never touched
        return cls([p for p in model.parameters() if p.requires_grad], **kw)
    if flat:
        if on_gpu and hasattr(model, 'flat_leaf'):
            model.flat_grad = True                  # one gradient tensor per step instead of a view per parameter (GNS.flat_grad)
        return FlatOptimizer(model, cls, **kw)
    return cls(model.parameters(), **kw)


class GraphedStep:
    """One training step on one batch - forward, mean of the per-grid losses, backward, ``optimizer.step()``, i.e. ``main.py:281-291`` -
    captured ONCE into a HIP graph and replayed.  At the reference's own operating point (case14, batch 128, K=15: ``main.py:209-254``)
    the kernels of a step take ~0.1 ms and its Python / autograd / ctypes work ~0.7 ms; a replay costs one launch.

    Needs a GPU-resident model, a ``FlatOptimizer`` running the library's Adam (it is switched to its device-side step counter,
    ``gns_adam_step_dev``) and one process (no collective is captured).  Batches must keep the captured shape; their values are
    copied into the graph's static input tensors before each replay.  The capture itself runs no kernel: one eager warm-up step
    before it (lazy initialisation of topology, kernel attributes, allocator) is undone by restoring parameters and optimiser
    state, so N replays equal N eager steps bit for bit.  ``run`` returns (mean total_loss, mean last_loss) as tensors that the
    NEXT replay overwrites."""

    def __init__(self, model, optimizer, buses, lines, generators, copy_inputs=True):
        """``copy_inputs=False``: capture on the given tensors themselves instead of private copies - for a resident batch that is
        trained on repeatedly or refilled in place (a batch that is a slice of a data set bound with ``GNS.bind_dataset`` then keeps
        reading the set's packed copy: no input packing kernel in the graph)."""
        if not isinstance(optimizer, FlatOptimizer) or not optimizer._use_native:
            raise ValueError('GraphedStep needs a FlatOptimizer running the library Adam (training.make_optimizer on a GPU model)')
        if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
            raise ValueError('GraphedStep captures no collective: one process only')
        flat = model.flat_parameters()
        if not flat.is_cuda:
            raise ValueError('GraphedStep needs a GPU-resident model')
        dev = flat.device
        self.model, self.optimizer = model, optimizer
        if copy_inputs:
            self.static = tuple(t.detach().to(dev).contiguous().clone() for t in (buses, lines, generators))
        else:
            if not all(t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 for t in (buses, lines, generators)):
                raise ValueError('GraphedStep(copy_inputs=False) needs contiguous float32 device tensors')
            self.static = (buses, lines, generators)
        B, L, G = get_BLG()
        optimizer.capturable = True
        # the id columns are validated on the warm-up call; replays trust them like ``topology_check = 'first'`` does (a host
        # comparison is not capturable): feed one topology, as ``fit`` - which compares the whole data set first - does
        snap_p = flat.clone()
        snap_o = None if not optimizer.inner.state else {k: (v.clone() if torch.is_tensor(v) else v) for k, v in optimizer.inner.state[optimizer._flat].items()}
        snap_dev = None if optimizer._dev_state is None else optimizer._dev_state.clone()

        def restore():
            with torch.no_grad():
                flat.copy_(snap_p)
                st = optimizer.inner.state[optimizer._flat]
                if snap_o is None:
                    st['exp_avg'].zero_(); st['exp_avg_sq'].zero_(); optimizer._dev_state.zero_()
                else:
                    st['exp_avg'].copy_(snap_o['exp_avg']); st['exp_avg_sq'].copy_(snap_o['exp_avg_sq'])
                    optimizer._dev_state.copy_(snap_dev) if snap_dev is not None else optimizer._dev_state.__setitem__(0, float(snap_o['step']))
            torch.autograd.graph.increment_version(flat)

        side = torch.cuda.Stream(dev)
        self._stream = side
        side.wait_stream(torch.cuda.current_stream(dev))
        refs = []
        with torch.cuda.stream(side):
            optimizer.zero_grad(set_to_none=True)
            for _ in range(2):                                    # eager warm-up (full topology check included); the parameters after
                train_step(model, optimizer, *self.static)        # each step are what the first two replays must reproduce (below)
                refs.append(flat.detach().clone())
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        restore()
        saved = getattr(model, 'topology_check', None)
        model.topology_check = 'first'
        try:
            optimizer.zero_grad(set_to_none=True)                 # the captured backward must ASSIGN the gradient, not accumulate into one
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                v, theta, losses, last_losses = model(*self.static, B, L, G)
                total = losses.mean()
                total.backward()
                optimizer.step()
                self.total, self.last = total.detach(), last_losses.detach().mean()
        finally:
            model.topology_check = saved
        # Self-test.  ROCm 7's default way of replaying a graph (pre-built AQL packets, DEBUG_CLR_GRAPH_PACKET_CAPTURE) was measured
        # to go wrong for this graph once the host has waited on the stream (hipStreamSynchronize / hipDeviceSynchronize) between
        # replays: a step's forward right, its gradient not, NaN parameters a few steps later (tools/gpu_graph_replay_stream.py).  The
        # package switches that path off at import (__init__.py) - which only takes if the HIP runtime was not initialised before.
        # So: two replays, the host waiting on the stream before each, must leave exactly the parameters of the two eager warm-up
        # steps; otherwise no GraphedStep is handed out (``fit`` then runs its steps eagerly).
        ok = True
        for ref in refs:
            torch.cuda.current_stream(dev).synchronize()
            self.graph.replay()
            torch.cuda.synchronize(dev)
            ok = ok and bool(torch.equal(flat.detach(), ref))
        restore()
        if not ok:
            raise GNSError('a replayed HIP graph of the training step does not reproduce the eager step on this runtime (ROCm graph packet '
                           'capture): set DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment before the first GPU call, or train without graph=True')
        self.replays = 0

    def run(self, buses, lines, generators):
        for dst, src in zip(self.static, (buses, lines, generators)):
            if src is not dst:
                if src.shape != dst.shape:
                    raise ValueError(f'GraphedStep was captured for inputs of shape {tuple(dst.shape)}, got {tuple(src.shape)}')
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        self.replays += 1
        # the replay updated the parameters through the captured kernels: keep autograd's version bookkeeping honest
        torch.autograd.graph.increment_version(self.model.flat_parameters())
        return self.total, self.last


def train_step(model, optimizer, buses, lines, generators, global_batch=None):
    """One optimiser step on one batch: mean of the per-grid losses (``main.py:284``), backward, step, zero_grad
    (``:288-291``).  Returns ``(mean total_loss, mean last_loss)`` as 0-dim tensors (no host sync)."""
    B, L, G = get_BLG()
    v, theta, losses, last_losses = model(buses, lines, generators, B, L, G)
    total = losses.mean()
    total.backward()
    if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
        gdist.allreduce_gradients(model, global_batch=global_batch or buses.shape[0] * tdist.get_world_size(), local_batch=buses.shape[0])
    optimizer.step()
    optimizer.zero_grad()
    return total.detach(), last_losses.detach().mean()


def _graph_pays(model, optimizer, all_buses, all_lines, all_generators, batch_size):
    """A captured step is used where a step is host-bound: a GPU model under the library's Adam, one process, and a batch small
    enough that the on-chip (grid-per-workgroup) kernels run it.  Larger batches are kernel-bound; they take the resident
    data set in the kernels' layout instead (``GNS.bind_dataset``)."""
    if not (isinstance(optimizer, FlatOptimizer) and optimizer._use_native and all_buses.is_cuda):
        return False
    if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
        return False
    import ctypes
    from ._lib import load_library
    cfg = model._config(all_buses.shape[1], all_lines.shape[1], all_generators.shape[1])
    return not load_library().gns_uses_packed_inputs(ctypes.byref(cfg), batch_size, 1)


def fit(model, all_buses, all_lines, all_generators, *, epochs=101, batch_size=128, optimizer_name='Adam', lr=None, case_nr=14,
        print_every=1, checkpoint_dir=None, log=print, graph=None):
    """Epoch loop of ``main.py:274-309``: batches in order, early stop once the epoch's mean final loss has failed to
    improve more than twice in a row (``:296-304``), checkpoint every ``print_every`` epochs (``:306-309``).

    Like the reference (where ``best_model = model`` is an alias, ``main.py:303``) the checkpoint holds the CURRENT
    parameters.  Returns the list of epoch mean final losses.

    ``graph``: replay one captured HIP graph per step (``GraphedStep``) instead of running the step from Python; default
    (None) = where it pays (``_graph_pays``).  Same arithmetic, same order: the weights are those of the eager loop bit for bit."""
    optimizer = make_optimizer(model, optimizer_name, lr)
    nr_samples = all_buses.shape[0]
    best, bad, history = float('inf'), 0, []
    # the id columns of the whole data set are compared once here; the per-call comparison (four device->host syncs per
    # step) is switched off for the loop: the model then builds / validates the topology on the first batch only
    if not (bool((all_lines[:, :, 0:2] == all_lines[0, :, 0:2]).all()) and bool((all_generators[:, :, 0] == all_generators[0, :, 0]).all())):
        raise ValueError('f_bus / t_bus / generator bus columns differ across the data set: train one topology at a time')
    saved_check = getattr(model, 'topology_check', None)
    if saved_check is not None:
        model.topology_check = 'first'
        model._topo_cache.clear()          # a cached topology of the same shape from an earlier data set must not be reused unchecked
    # The data set is resident (``main.py:255`` loads it whole): it is brought into the kernels' input layout once, and every
    # batch of the epochs - a 64-aligned slice of it - is read from that copy instead of being packed again on every step.
    bound = False
    if hasattr(model, 'bind_dataset') and all_buses.is_cuda and batch_size % 64 == 0 and all(
            t.is_contiguous() and t.dtype == torch.float32 for t in (all_buses, all_lines, all_generators)):
        model.bind_dataset(all_buses, all_lines, all_generators)
        bound = True
    if graph is None:
        graph = 'auto' if nr_samples >= batch_size and _graph_pays(model, optimizer, all_buses, all_lines, all_generators, batch_size) else False
    try:
        return _fit_loop(model, optimizer, all_buses, all_lines, all_generators, nr_samples, epochs, batch_size, optimizer_name,
                         case_nr, print_every, checkpoint_dir, log, best, bad, history, graph)
    finally:
        if bound:
            model.unbind_dataset()
        if saved_check is not None:
            model.topology_check = saved_check


def _fit_loop(model, optimizer, all_buses, all_lines, all_generators, nr_samples, epochs, batch_size, optimizer_name, case_nr,
              print_every, checkpoint_dir, log, best, bad, history, graph=False):
    stepper = None
    for epoch in range(epochs):
        finals = []
        for lo in range(0, nr_samples - batch_size + 1, batch_size):
            sl = slice(lo, lo + batch_size)
            if graph and stepper is None:
                try:
                    stepper = GraphedStep(model, optimizer, all_buses[sl], all_lines[sl], all_generators[sl])
                except GNSError as e:
                    if graph is True:                # asked for explicitly: the caller hears about it
                        raise
                    log(f'captured training step not used: {e}')
                    graph = False                    # 'auto': the eager loop computes the same thing
            if graph:
                _, last = stepper.run(all_buses[sl], all_lines[sl], all_generators[sl])
                last = last.clone()                  # the next replay overwrites the graph's output
            else:
                _, last = train_step(model, optimizer, all_buses[sl], all_lines[sl], all_generators[sl])
            finals.append(last)
        epoch_final = float(torch.stack(finals).mean()) if finals else float('nan')
        history.append(epoch_final)
        if epoch_final >= best:
            bad += 1
            if bad > 2:
                log('Loss is increasing')
                break
        else:
            best, bad = epoch_final, 0
        if epoch % print_every == 0:
            log(f'Epoch: {epoch}, Final Loss: {epoch_final}, best loss: {best}')
            if checkpoint_dir is not None:
                os.makedirs(checkpoint_dir, exist_ok=True)
                torch.save(model.state_dict(), os.path.join(checkpoint_dir, checkpoint_name(
                    case_nr, model.K, model.latent_dim, model.hidden_dim, model.multiple_phis, optimizer_name)))
    return history
