"""Batched training loop around the fused forward/backward: what ``main()`` does at ``GNS/main.py:235-309`` without the
per-grid Python loop and without wandb.  Data-parallel when ``torch.distributed`` is initialised (one flat-gradient
all-reduce per step, ``dist.allreduce_gradients``)."""
from __future__ import annotations

import os

import torch
import torch.distributed as tdist

from . import dist as gdist
from .gns import get_BLG


def checkpoint_name(case_nr, K, latent_dim, hidden_dim, multiple_phi, optimizer_name):
    """File name of the reference's checkpoints (``main.py:308-309``) so that files interchange."""
    return f'best_model_c{case_nr}_K{K}_L{latent_dim}_H{hidden_dim}_{multiple_phi}_optim{optimizer_name}.pth'


class FlatOptimizer:
    """The reference's optimiser (``torch.optim.Adam`` / ``Adagrad``, ``main.py:236-243``) run on the model's ONE flat
    parameter buffer instead of its 144 tensors: the parameters are views of that buffer (``GNS.flat_parameters``) and the
    fused backward hands their gradients as views of one flat gradient (``dist.flat_gradient``), so the element-wise update
    is the same arithmetic in one kernel launch instead of six multi-tensor chunks (34 -> 6 us per step on MI355X).
    For Adam on a GPU model that launch is the library's own kernel (``gns_adam_step``, ``include/gns_hip.h``: torch's update
    rule with its defaults, 3 us instead of torch's two launches of 20 us); the moment estimates live in the inner optimiser's
    state, so ``state_dict`` / ``load_state_dict`` are those of ``torch.optim.Adam`` over the flat tensor either way
    (the model's own ``state_dict`` - what the reference checkpoints, ``main.py:308`` - is unaffected)."""

    def __init__(self, model, inner_cls, native=None, **kw):
        self.model = model
        self._cls, self._kw = inner_cls, kw
        self._flat = None
        self.inner = None
        self._native = native
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
            self.inner = self._cls([self._flat], **kw)

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
        if st['step'].is_cuda:                                                     # a state_dict of a fused torch Adam was loaded
            st['step'] = st['step'].cpu()
        st['step'] += 1
        grad = self._flat.grad
        if grad.dtype != torch.float32 or not grad.is_contiguous() or grad.device != self._flat.device:
            raise ValueError('FlatOptimizer: the flat gradient must be a contiguous float32 tensor on the parameters\' device')
        dev = self._flat.device
        with torch.cuda.device(dev):
            rc = load_library().gns_adam_step(self._flat.data_ptr(), grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(),
                                              self._flat.numel(), float(grp['lr']), float(grp['betas'][0]), float(grp['betas'][1]),
                                              float(grp['eps']), int(st['step'].item()), torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise RuntimeError(f'gns_adam_step failed: {GNS_ERRORS.get(rc, rc)}')

    def step(self):
        from . import dist as gdist
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

    def zero_grad(self, set_to_none=True):
        for p in (self.model._param_list() if hasattr(self.model, '_param_list') else self.model.parameters()):
            p.grad = None
        if self._flat is not None:
            self._flat.grad = None

    def state_dict(self):
        return self.inner.state_dict()

    def load_state_dict(self, sd):
        self.inner.load_state_dict(sd)

    @property
    def param_groups(self):
        return self.inner.param_groups


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
    if flat:
        if on_gpu and hasattr(model, 'flat_leaf'):
            model.flat_grad = True                  # one gradient tensor per step instead of a view per parameter (GNS.flat_grad)
        return FlatOptimizer(model, cls, **kw)
    return cls(model.parameters(), **kw)


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


def fit(model, all_buses, all_lines, all_generators, *, epochs=101, batch_size=128, optimizer_name='Adam', lr=None, case_nr=14,
        print_every=1, checkpoint_dir=None, log=print):
    """Epoch loop of ``main.py:274-309``: batches in order, early stop once the epoch's mean final loss has failed to
    improve more than twice in a row (``:296-304``), checkpoint every ``print_every`` epochs (``:306-309``).

    Like the reference (where ``best_model = model`` is an alias, ``main.py:303``) the checkpoint holds the CURRENT
    parameters.  Returns the list of epoch mean final losses."""
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
    try:
        return _fit_loop(model, optimizer, all_buses, all_lines, all_generators, nr_samples, epochs, batch_size, optimizer_name,
                         case_nr, print_every, checkpoint_dir, log, best, bad, history)
    finally:
        if saved_check is not None:
            model.topology_check = saved_check


def _fit_loop(model, optimizer, all_buses, all_lines, all_generators, nr_samples, epochs, batch_size, optimizer_name, case_nr,
              print_every, checkpoint_dir, log, best, bad, history):
    for epoch in range(epochs):
        finals = []
        for lo in range(0, nr_samples - batch_size + 1, batch_size):
            sl = slice(lo, lo + batch_size)
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
