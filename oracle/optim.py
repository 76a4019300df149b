"""CPU oracle: the optimizer step and learning-rate schedule that follow the backward pass in the reference's training loop.
TEST INFRASTRUCTURE ONLY.

``torch_optimizer.AdaBound`` (/root/reference/train.py:5,53-58: lr=1e-3, final_lr=0.1, amsbound=False; defaults betas=(0.9, 0.999),
gamma=1e-3, eps=1e-8, weight_decay=0) lives in a third-party package that is absent from this image and from /root/reference
(no requirements file pins its version; 0.3.0 is the last release).  The reference holds no fixture for it => **parity unpinned**:
``adabound_step`` restates the published algorithm (Luo et al., "Adaptive Gradient Methods with Dynamic Bound of Learning Rate",
ICLR 2019, Algorithm 2, in the operation order of the package's ``step()``) with plain torch tensor ops.

``torch.optim.lr_scheduler.OneCycleLR`` (/root/reference/train.py:144: max_lr=0.01, steps_per_epoch, epochs; torch defaults
pct_start=0.3, anneal 'cos', cycle_momentum=True with base/max 0.85/0.95 applied to betas[0], div_factor=25, final_div_factor=1e4,
three_phase=False) IS importable here: ``one_cycle_reference`` drives torch's own class, so the schedule is pinned against the real thing.
"""
import math

import torch


def adabound_step(p, g, state, lr=1e-3, betas=(0.9, 0.999), final_lr=0.1, gamma=1e-3, eps=1e-8, weight_decay=0.0, base_lr=None):
    """One AdaBound step (amsbound=False) on tensors p (updated in place) and g; state = {'step', 'exp_avg', 'exp_avg_sq'}."""
    base_lr = lr if base_lr is None else base_lr
    if not state:
        state.update(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
    state['step'] += 1
    t = state['step']
    b1, b2 = betas
    grad = g if weight_decay == 0 else g.add(p, alpha=weight_decay)
    state['exp_avg'].mul_(b1).add_(grad, alpha=1 - b1)
    state['exp_avg_sq'].mul_(b2).addcmul_(grad, grad, value=1 - b2)
    denom = state['exp_avg_sq'].sqrt().add_(eps)
    step_size = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    flr = final_lr * lr / base_lr
    lower, upper = flr * (1 - 1 / (gamma * t + 1)), flr * (1 + 1 / (gamma * t))
    upd = torch.full_like(denom, step_size).div_(denom).clamp_(lower, upper).mul_(state['exp_avg'])
    p.add_(-upd)
    return p


def one_cycle_reference(total_steps, max_lr=0.01, lr0=1e-3, betas=(0.9, 0.999), **kw):
    """(lr, beta1) that torch's OneCycleLR leaves in the optimizer's param group at construction and after each of
    total_steps - 1 scheduler.step() calls -> list of total_steps pairs."""
    w = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([w], lr=lr0, betas=betas)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=max_lr, total_steps=total_steps, **kw)
    out = [(opt.param_groups[0]['lr'], opt.param_groups[0]['betas'][0])]
    for _ in range(total_steps - 1):
        opt.step()
        sched.step()
        out.append((opt.param_groups[0]['lr'], opt.param_groups[0]['betas'][0]))
    return out
