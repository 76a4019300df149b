"""The optimizer step inside the headline number: ss_adabound_step over flat buffers against
oracle.optim.adabound_step, the plain-torch restatement of torch_optimizer.AdaBound as train.py:53-58 configures it.
**Parity unpinned** against the package itself (absent from the image, no fixture in the reference): the restatement follows the
published algorithm; fp32 agreement asserted at 1e-6 of the parameter scale."""
import pytest
import torch

from _parity import rel

pytestmark = pytest.mark.gpu


class _Flat:
    """the two flat buffers engine.AdaBound works on, without a model"""
    def __init__(self, sizes, seed):
        g = torch.Generator().manual_seed(seed)
        self.params, self.offsets, n = [], [], 0
        for s in sizes:
            self.offsets.append(n)
            n += (s + 63) // 64 * 64
        self.numel = n
        self.data = torch.randn(n, generator=g).cuda()
        self.grad = torch.zeros(n).cuda()
        for s, o in zip(sizes, self.offsets):
            self.params.append(self.data[o:o + s])


@pytest.mark.parametrize('wd,gscale,sizes', [(0.0, 1.0, [1000, 4096, 77]), (1e-2, 0.25, [4097, 3, 1]), (5e-4, 1.0, [257 * 131])])
def test_adabound_steps_match_restatement(wd, gscale, sizes):
    from semseg_amd.engine import AdaBound
    from oracle.optim import adabound_step
    flat = _Flat(sizes, 3)
    if len(sizes) == 1:                          # an odd, unpadded length: exercises the scalar tail (n % 4 != 0)
        flat.numel = sizes[0]
        flat.data, flat.grad = flat.data[:sizes[0]].clone(), flat.grad[:sizes[0]].clone()
    opt = AdaBound(flat, lr=1e-3, final_lr=0.1, gamma=1e-3, weight_decay=wd)
    p_ref, state = flat.data.cpu().clone(), {}
    g = torch.Generator().manual_seed(11)
    for step in range(5):
        grad = torch.randn(flat.numel, generator=g) * (10.0 ** (step - 2))      # gradients from 1e-2 to 1e2: both clamp bounds bind
        flat.grad.copy_(grad)
        opt.step(grad_scale=gscale)
        adabound_step(p_ref, grad * gscale, state, lr=1e-3, final_lr=0.1, gamma=1e-3, weight_decay=wd)
        assert rel(flat.data, p_ref) < 1e-6, step
        assert rel(opt.exp_avg, state['exp_avg']) < 1e-6 and rel(opt.exp_avg_sq, state['exp_avg_sq']) < 1e-6, step
    # the step must have moved every element, and by different amounts under the two bounds
    assert float((flat.data.cpu() - _Flat(sizes, 3).data.cpu()[:flat.numel]).abs().min()) > 0


def test_adabound_with_one_cycle_schedule_matches_restatement():
    """lr and beta1 move every step (train.py:144); final_lr scales with lr / base_lr as in the package."""
    from semseg_amd.engine import AdaBound, OneCycleLR
    from oracle.optim import adabound_step, one_cycle_reference
    flat = _Flat([5000], 5)
    opt = AdaBound(flat, lr=1e-3, final_lr=0.1)
    sched = OneCycleLR(opt, max_lr=0.01, total_steps=8)
    ref_sched = one_cycle_reference(8, max_lr=0.01)
    p_ref, state = flat.data.cpu().clone(), {}
    g = torch.Generator().manual_seed(2)
    for step in range(7):
        grad = torch.randn(flat.numel, generator=g)
        flat.grad.copy_(grad)
        lr, b1 = ref_sched[step]
        assert (opt.lr, opt.betas[0]) == (lr, b1)
        opt.step()
        sched.step()
        adabound_step(p_ref, grad, state, lr=lr, betas=(b1, 0.999), final_lr=0.1, base_lr=1e-3)
        assert rel(flat.data, p_ref) < 1e-6, step


def test_adabound_state_dict_round_trip():
    """save -> load into a fresh optimizer -> step == uninterrupted run, bit for bit (retrain.py resumes this way)."""
    from semseg_amd.engine import AdaBound, FlatParams
    torch.manual_seed(0)
    def make():
        net = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5)).cuda()
        with torch.no_grad():
            for i, p in enumerate(net.parameters()):
                p.copy_(torch.linspace(-1, 1, p.numel()).view(p.shape) * (i + 1))
        return net, FlatParams(net)
    g = torch.Generator().manual_seed(4)
    grads = [torch.randn(10000, generator=g) for _ in range(4)]
    net_a, flat_a = make()
    opt_a = AdaBound(flat_a, weight_decay=1e-3)
    for k in range(2):
        flat_a.grad.copy_(grads[k][:flat_a.numel]); opt_a.step()
    sd = opt_a.state_dict()
    assert sorted(sd['state']) == [0, 1, 2, 3] and sd['state'][0]['exp_avg'].shape == (19, 37) and sd['state'][3]['step'] == 2
    net_b, flat_b = make()
    flat_b.data.copy_(flat_a.data)
    opt_b = AdaBound(flat_b)
    opt_b.load_state_dict(sd)
    assert opt_b.t == 2 and opt_b.wd == 1e-3
    for k in range(2, 4):
        for f, o in ((flat_a, opt_a), (flat_b, opt_b)):
            f.grad.copy_(grads[k][:f.numel]); o.step()
    for pa, pb in zip(net_a.parameters(), net_b.parameters()):          # the parameters proper (alignment padding between them is not state)
        assert torch.equal(pa, pb)
    sa, sb = opt_a.state_dict(), opt_b.state_dict()
    for i in sa['state']:
        assert torch.equal(sa['state'][i]['exp_avg_sq'], sb['state'][i]['exp_avg_sq']) and sa['state'][i]['step'] == sb['state'][i]['step'] == 4


def test_adabound_load_torch_layout_checkpoint_keeps_constructor_base_lr():
    """A torch / Lightning 'optimizer_states' entry has no 'base_lrs' (torch_optimizer keeps it as a constructor attribute): after a
    resume with the scheduler's lr in the group, the bounds must still scale with lr / 1e-3, the constructor's lr (train.py:53-58)."""
    from semseg_amd.engine import AdaBound
    from oracle.optim import adabound_step
    flat = _Flat([3000], 9)
    opt = AdaBound(flat, lr=1e-3, final_lr=0.1)
    p_ref, state = flat.data.cpu().clone(), {}
    g = torch.Generator().manual_seed(3)
    grad = torch.randn(flat.numel, generator=g)
    flat.grad.copy_(grad); opt.step()
    adabound_step(p_ref, grad, state, lr=1e-3, final_lr=0.1, base_lr=1e-3)
    sd = opt.state_dict()
    del sd['base_lrs']
    sd['param_groups'][0]['lr'] = 7e-3                      # where OneCycleLR had moved it when the checkpoint was written
    flat2 = _Flat([3000], 9)
    flat2.data.copy_(flat.data)
    flat2.params = [flat2.data[:3000]]
    opt2 = AdaBound(flat2, lr=1e-3, final_lr=0.1)
    opt2.load_state_dict({'state': {0: {'step': 1, 'exp_avg': opt.exp_avg[:3000].clone(), 'exp_avg_sq': opt.exp_avg_sq[:3000].clone()}},
                          'param_groups': sd['param_groups']})
    assert opt2.lr == 7e-3 and opt2.base_lr == 1e-3
    grad = torch.randn(flat.numel, generator=g) * 30
    flat2.grad.copy_(grad); opt2.step()
    adabound_step(p_ref, grad, state, lr=7e-3, final_lr=0.1, base_lr=1e-3)
    assert rel(flat2.data[:3000], p_ref[:3000]) < 1e-6
