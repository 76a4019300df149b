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
    """Bucketed gradient all-reduce over the flat gradient buffer (the C1 collective of SURVEY.md §2.2).  One backward pass per
    ``wait()``: a bucket travels as soon as its last gradient of THIS backward exists (accumulating several backward passes before
    the exchange needs ``overlap=False``)."""

    def __init__(self, flat: FlatParams, bucket_bytes=25 << 20, overlap=True):
        self.flat, self.world = flat, _world()
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
        self.pending = [len(m) for _, _, m in self.buckets]
        self.fired = [False] * len(flat.params)
        self.works, self.launched = [], [False] * len(self.buckets)
        self.overlap = overlap and self.world > 1
        self.stream = torch.cuda.Stream() if (self.overlap and flat.data.is_cuda) else None
        self._hooks = []
        if self.world > 1:
            for i, p in enumerate(flat.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))    # gradients that arrive through autograd
                p._ss_grad_ready = self._make_hook(i)                                             # ... and through runtime.deliver_grads

    def _make_hook(self, i):
        def hook(_p):
            # a parameter can be announced twice in one backward: by runtime.deliver_grads when its gradient has been queued into
            # the flat buffer, and by autograd's post-accumulate hook (torch fires it even though the node returned None for it)
            if self.fired[i]:
                return
            self.fired[i] = True
            b = self.bucket_of[i]
            self.pending[b] -= 1
            if self.pending[b] == 0 and self.overlap:
                self._launch(b)
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
                work = all_reduce_sum(view, async_op=True)
        else:
            work = all_reduce_sum(view, async_op=True)
        if work is not None:
            self.works.append(work)

    def wait(self):
        """Launch whatever has not been launched (unused parameters, overlap off), join, re-arm.  -> grad_scale."""
        if self.world > 1:
            for b in range(len(self.buckets)):
                self._launch(b)
            for w in self.works:
                w.wait()
            if self.stream is not None:
                torch.cuda.current_stream().wait_stream(self.stream)
        self.works, self.launched = [], [False] * len(self.buckets)
        self.pending = [len(m) for _, _, m in self.buckets]
        self.fired = [False] * len(self.flat.params)
        return 1.0 / self.world


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
    """torch_optimizer.AdaBound(lr, betas, final_lr, gamma, eps, weight_decay, amsbound=False) semantics on FlatParams."""

    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), final_lr=0.1, gamma=1e-3, eps=1e-8, weight_decay=0.0):
        self.flat, self.lr, self.base_lr = flat, lr, lr
        self.betas, self.final_lr, self.gamma, self.eps, self.wd = betas, final_lr, gamma, eps, weight_decay
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
