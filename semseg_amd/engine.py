"""Data-parallel training plumbing around the forward/backward path: what Lightning's
``DDPStrategy(find_unused_parameters=False)`` (train.py:306) and ``torch_optimizer.AdaBound`` (train.py:53-58) do implicitly
in the reference, laid out for one process per MI355X:

* ``FlatParams``  re-homes every parameter of a model into ONE contiguous fp32 buffer and every gradient into another
  (``p.data`` / ``p.grad`` become views), in reverse registration order so the gradients the backward produces first sit
  at the front of the buffer;
* ``GradReducer`` all-reduces that gradient buffer in contiguous ~25 MB buckets over ``torch.distributed`` (backend
  ``nccl`` == RCCL over xGMI on the GPU box, ``gloo`` in the CPU tests).  A bucket is launched from the parameters'
  post-accumulate-grad hooks the moment its last gradient has been produced, on a side stream, so the exchange of bucket
  k overlaps the backward kernels of bucket k+1; ``wait()`` joins them before the optimizer.  SUM is used and the 1/world
  mean is folded into the optimizer kernel's ``grad_scale``;
* ``AdaBound``  is one HIP launch over the flat buffers (``ss_adabound_step``).
"""
import math
import torch
import torch.distributed as dist

from . import ops, weightprep
from ._lib import lib, check


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _host_staged(t):
    """gloo cannot move device memory between processes that share a GPU: stage through the host for that backend
    (rehearsals of the N-rank path on one box; RCCL takes device pointers directly)."""
    return t.is_cuda and dist.get_backend() == 'gloo'


def all_reduce_sum(t, async_op=False):
    if _host_staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


class FlatParams:
    def __init__(self, model, align=64):
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError('FlatParams: model has no trainable parameters')
        dev = params[0].device
        order = list(reversed(params))                         # backward produces gradients roughly in reverse order
        offs, n = [], 0
        for p in order:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError('FlatParams: parameters must be float32 on one device')
            offs.append(n)
            n += (p.numel() + align - 1) // align * align
        self.numel = n
        self.data = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.params, self.offsets = order, offs
        weightprep.clear()                                     # the parameters move: remembered weight copies point at the old storage
        with torch.no_grad():
            for p, o in zip(order, offs):
                self.data[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.data[o:o + p.numel()].view(p.shape)
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
                # the backward nodes of this package add their gradients straight into the slot (runtime.deliver_grads): one
                # launch per node instead of one AccumulateGrad add per parameter
                p._ss_grad_sink = p.grad

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):           # a set_to_none elsewhere must not detach the views
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
                p._ss_grad_sink = p.grad


class GradReducer:
    """Bucketed gradient all-reduce over the flat gradient buffer (the C1 collective of SURVEY.md §2.2).  A bucket travels as soon
    as every gradient contribution of THIS backward to its parameters exists.  Contributions are counted, not assumed to be one
    per parameter: the forward of every graph node of this package announces (``runtime.expect_grads``) how many gradients it
    will deliver to each parameter, so a module called twice before one ``backward()`` (or a shared module) holds its bucket
    back until the second contribution has been added.  A gradient that still arrives after its bucket has been reduced -- a
    second ``backward()`` before ``wait()``, i.e. gradient accumulation -- raises (torch DDP raises there too); accumulate
    with ``overlap=False``, which reduces everything inside ``wait()``."""

    def __init__(self, flat: FlatParams, bucket_bytes=25 << 20, overlap=True, force_comm=False):
        self.flat, self.world = flat, _world()
        # force_comm: run the hooks, the side stream and the collectives even in a world of one (a one-rank all-reduce is the
        # identity), so the real backend's path can be exercised on a single GPU
        self.active = self.world > 1 or (force_comm and dist.is_available() and dist.is_initialized())
        self.buckets = []                                      # (start, end, [param indices])
        start, members = 0, []
        for i, (p, o) in enumerate(zip(flat.params, flat.offsets)):
            members.append(i)
            end = flat.offsets[i + 1] if i + 1 < len(flat.params) else flat.numel
            if (end - start) * 4 >= bucket_bytes or i + 1 == len(flat.params):
                self.buckets.append((start, end, members))
                start, members = end, []
        self.bucket_of = {}
        for b, (_, _, mem) in enumerate(self.buckets):
            for i in mem:
                self.bucket_of[i] = b
        self.overlap = overlap and self.active
        self.stream = torch.cuda.Stream() if (self.overlap and flat.data.is_cuda) else None
        self.comm_events = []                                  # (start, end) event pairs of the side stream, when timing is on
        self.time_comm = False
        self._rearm()
        self._hooks = []
        if self.active:
            for i, p in enumerate(flat.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_autograd_hook(i)))   # gradients through autograd
                p._ss_grad_ready = self._make_delivery_hook(i)                                           # ... through runtime.deliver_grads
                p._ss_grad_expect = self._make_expect_hook(i)                                            # announced by the forward

    def _rearm(self):
        n = len(self.flat.params)
        self.pending = [len(m) for _, _, m in self.buckets]
        self.fired, self.expected, self.got = [False] * n, [0] * n, [0] * n
        self.works, self.launched = [], [False] * len(self.buckets)

    def _make_expect_hook(self, i):
        def expect(_p):
            self.expected[i] += 1
        return expect

    def _late(self, i):
        return RuntimeError(f'GradReducer: parameter {i} received a gradient after its bucket had been all-reduced in this step '
                            '(a second backward() before wait()?).  Accumulate gradients with GradReducer(..., overlap=False).')

    def _ready(self, i):
        self.fired[i] = True
        b = self.bucket_of[i]
        self.pending[b] -= 1
        if self.pending[b] == 0 and self.overlap:
            self._launch(b)

    def _make_delivery_hook(self, i):
        def delivered(_p):                                     # one gradient contribution has been queued into the flat buffer
            if self.launched[self.bucket_of[i]]:
                raise self._late(i)
            self.got[i] += 1
            if not self.fired[i] and self.got[i] >= max(self.expected[i], 1):
                self._ready(i)
        return delivered

    def _make_autograd_hook(self, i):
        def hook(_p):
            # torch also fires this hook for a parameter whose node returned None because runtime.deliver_grads had already
            # added the gradient: deliveries are counted above, so only gradients that came through autograd count here
            # (AccumulateGrad runs once per backward, after autograd has summed all uses)
            if self.got[i] or self.expected[i]:
                return
            if self.launched[self.bucket_of[i]]:
                raise self._late(i)
            if not self.fired[i]:
                self._ready(i)
        return hook

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        s, e, _ = self.buckets[b]
        view = self.flat.grad[s:e]
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                ev = None
                if self.time_comm:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                work = all_reduce_sum(view, async_op=True)
                if ev is not None:
                    if work is not None:
                        work.wait()                            # stream-side wait: orders the end event behind the collective
                        work = None
                    ev[1].record()
                    self.comm_events.append(ev)
        else:
            work = all_reduce_sum(view, async_op=True)
        if work is not None:
            self.works.append(work)

    def wait(self):
        """Launch whatever has not been launched (unused parameters, overlap off), join, re-arm.  -> grad_scale."""
        if self.active:
            for b in range(len(self.buckets)):
                self._launch(b)
            for w in self.works:
                w.wait()
            if self.stream is not None:
                torch.cuda.current_stream().wait_stream(self.stream)
        self._rearm()
        return 1.0 / self.world

    def close(self):
        """Detach from the parameters (before another reducer takes over the same FlatParams)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.flat.params:
            for a in ('_ss_grad_ready', '_ss_grad_expect'):
                if hasattr(p, a):
                    delattr(p, a)

    def comm_ms(self):
        """Sum of the bucket all-reduce durations recorded since the last call (``time_comm = True``; synchronises)."""
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.comm_events)
        self.comm_events = []
        return ms


def broadcast_module_state(model, src=0):
    """DDP start-up semantics: rank-``src`` parameters and buffers win."""
    if _world() == 1:
        return
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            if _host_staged(t):
                h = t.data.cpu()
                dist.broadcast(h, src)
                t.data.copy_(h)
            else:
                dist.broadcast(t.data, src)
    weightprep.bump()


class AdaBound:
    """torch_optimizer.AdaBound(lr, betas, final_lr, gamma, eps, weight_decay, amsbound=False) semantics on FlatParams
    (train.py:53-58).  The parameters themselves are the fp32 master copy: the bf16 operands the kernels read are derived from
    them by ``weightprep`` after every step, so no second set of weights exists to drift.  ``lr`` and ``betas`` are plain
    attributes a scheduler may move (``OneCycleLR`` below); ``base_lr`` is the constructor's lr, as in the package, so the
    dynamic bounds scale with lr / base_lr."""

    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), final_lr=0.1, gamma=1e-3, eps=1e-8, weight_decay=0.0):
        self.flat, self.lr, self.base_lr = flat, lr, lr
        self.betas, self.final_lr, self.gamma, self.eps, self.wd = tuple(betas), final_lr, gamma, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.t = 0

    def step_scalars(self):
        b1, b2 = self.betas
        t = self.t
        step_size = self.lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        flr = self.final_lr * self.lr / self.base_lr
        return step_size, flr * (1 - 1 / (self.gamma * t + 1)), flr * (1 + 1 / (self.gamma * t))

    def step(self, grad_scale=1.0):
        self.t += 1
        f = self.flat
        if not f.data.is_cuda:
            raise RuntimeError('semseg_amd.AdaBound: parameters must live on a HIP device (no CPU fallback)')
        step_size, lower, upper = self.step_scalars()
        check(lib().ss_adabound_step(f.data.data_ptr(), f.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                     f.numel, self.betas[0], self.betas[1], self.eps, step_size, lower, upper, self.wd,
                                     grad_scale, torch.cuda.current_stream().cuda_stream))
        weightprep.bump()                                      # the kernel wrote the parameters through a raw pointer

    # ---- checkpointing, in torch.optim's layout (what Lightning stores under 'optimizer_states'): per-parameter 'step',
    # 'exp_avg', 'exp_avg_sq' indexed in model.parameters() order, one param group with the hyper-parameters
    def _slots(self):
        f = self.flat
        n = len(f.params)
        return [(n - 1 - j, f.params[j], f.offsets[j]) for j in range(n)]        # flat order is reversed registration order

    def state_dict(self):
        state = {}
        for idx, p, o in self._slots():
            sl = slice(o, o + p.numel())
            state[idx] = {'step': self.t, 'exp_avg': self.exp_avg[sl].view(p.shape).clone(),
                          'exp_avg_sq': self.exp_avg_sq[sl].view(p.shape).clone()}
        group = {'lr': self.lr, 'betas': tuple(self.betas), 'final_lr': self.final_lr, 'gamma': self.gamma, 'eps': self.eps,
                 'weight_decay': self.wd, 'amsbound': False, 'params': list(range(len(self.flat.params)))}
        return {'state': dict(sorted(state.items())), 'param_groups': [group], 'base_lrs': [self.base_lr]}

    def load_state_dict(self, sd):
        group = sd['param_groups'][0]
        if group.get('amsbound'):
            raise ValueError('AdaBound.load_state_dict: amsbound=True is not built (train.py:57 uses False)')
        self.lr, self.betas = float(group['lr']), tuple(group['betas'])
        self.final_lr, self.gamma, self.eps, self.wd = group['final_lr'], group['gamma'], group['eps'], group['weight_decay']
        if 'base_lrs' in sd:                                   # torch_optimizer keeps base_lrs as a constructor attribute, not in state_dict:
            self.base_lr = float(sd['base_lrs'][0])            # a torch / Lightning checkpoint leaves the constructor's lr in place
        steps = set()
        with torch.no_grad():
            for idx, p, o in self._slots():
                st = sd['state'].get(idx, sd['state'].get(str(idx)))
                sl = slice(o, o + p.numel())
                if st is None:                                     # a parameter that has not been stepped yet
                    self.exp_avg[sl].zero_(), self.exp_avg_sq[sl].zero_()
                    continue
                if tuple(st['exp_avg'].shape) != tuple(p.shape):
                    raise ValueError(f'AdaBound.load_state_dict: state {idx} has shape {tuple(st["exp_avg"].shape)}, parameter {tuple(p.shape)}')
                self.exp_avg[sl].copy_(st['exp_avg'].reshape(-1))
                self.exp_avg_sq[sl].copy_(st['exp_avg_sq'].reshape(-1))
                steps.add(int(st['step']))
        if len(steps) > 1:
            raise ValueError(f'AdaBound.load_state_dict: parameters at different steps {sorted(steps)}; one flat step count is kept')
        self.t = steps.pop() if steps else 0


class OneCycleLR:
    """torch.optim.lr_scheduler.OneCycleLR (train.py:144: max_lr=0.01, steps_per_epoch, epochs) for an optimizer that keeps ``lr``
    and ``betas`` as attributes (``AdaBound`` above): host arithmetic only, same formulas and defaults as torch (two phases, cosine
    annealing, momentum cycled on betas[0] between 0.85 and 0.95 in the opposite direction)."""

    def __init__(self, optimizer, max_lr, total_steps=None, epochs=None, steps_per_epoch=None, pct_start=0.3, anneal_strategy='cos',
                 cycle_momentum=True, base_momentum=0.85, max_momentum=0.95, div_factor=25.0, final_div_factor=1e4, three_phase=False):
        if total_steps is None:
            if epochs is None or steps_per_epoch is None or epochs <= 0 or steps_per_epoch <= 0:
                raise ValueError('OneCycleLR: give total_steps, or positive epochs and steps_per_epoch')
            total_steps = epochs * steps_per_epoch
        if total_steps <= 0 or not 0 <= pct_start <= 1 or anneal_strategy not in ('cos', 'linear'):
            raise ValueError('OneCycleLR: bad total_steps / pct_start / anneal_strategy')
        self.opt, self.total_steps = optimizer, int(total_steps)
        self.anneal = self._cos if anneal_strategy == 'cos' else self._lin
        init_lr, min_lr = max_lr / div_factor, max_lr / div_factor / final_div_factor
        hi_m, lo_m = max_momentum, base_momentum
        if three_phase:
            self.phases = [(float(pct_start * total_steps) - 1, init_lr, max_lr, hi_m, lo_m),
                           (float(2 * pct_start * total_steps) - 2, max_lr, init_lr, lo_m, hi_m),
                           (total_steps - 1, init_lr, min_lr, hi_m, hi_m)]
        else:
            self.phases = [(float(pct_start * total_steps) - 1, init_lr, max_lr, hi_m, lo_m),
                           (total_steps - 1, max_lr, min_lr, lo_m, hi_m)]
        self.cycle_momentum = cycle_momentum
        self.last_epoch = 0
        self._apply()

    @staticmethod
    def _cos(start, end, pct):
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1)

    @staticmethod
    def _lin(start, end, pct):
        return (end - start) * pct + start

    def values(self, step_num):
        if step_num > self.total_steps:
            raise ValueError(f'Tried to step {step_num} times. The specified number of total steps is {self.total_steps}')
        start = 0.0
        for i, (end, lr0, lr1, m0, m1) in enumerate(self.phases):
            if step_num <= end or i == len(self.phases) - 1:
                pct = (step_num - start) / (end - start)
                return self.anneal(lr0, lr1, pct), self.anneal(m0, m1, pct)
            start = end

    def _apply(self):
        lr, mom = self.values(self.last_epoch)
        self.opt.lr = lr
        if self.cycle_momentum:
            self.opt.betas = (mom,) + tuple(self.opt.betas[1:])

    def step(self):
        self.last_epoch += 1
        self._apply()

    def state_dict(self):
        return {'last_epoch': self.last_epoch, 'total_steps': self.total_steps}

    def load_state_dict(self, sd):
        self.last_epoch = int(sd['last_epoch'])
        self._apply()
