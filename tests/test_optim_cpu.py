"""Host side of the optimizer step: engine.OneCycleLR against torch.optim.lr_scheduler.OneCycleLR itself
(train.py:144), driven through oracle.optim.one_cycle_reference.  Pinned: the real scheduler is importable here."""
import pytest

from semseg_amd.engine import OneCycleLR
from oracle.optim import one_cycle_reference


class _Opt:
    def __init__(self):
        self.lr, self.betas = 1e-3, (0.9, 0.999)


@pytest.mark.parametrize('kw', [{}, {'three_phase': True}, {'anneal_strategy': 'linear', 'pct_start': 0.25}, {'cycle_momentum': False},
                                {'div_factor': 10.0, 'final_div_factor': 100.0, 'base_momentum': 0.8, 'max_momentum': 0.9}])
@pytest.mark.parametrize('total', [10, 37, 1000])
def test_one_cycle_lr_equals_torch(kw, total):
    ref = one_cycle_reference(total, max_lr=0.01, **kw)
    opt = _Opt()
    sched = OneCycleLR(opt, 0.01, total_steps=total, **kw)
    got = [(opt.lr, opt.betas[0])]
    for _ in range(total - 1):
        sched.step()
        got.append((opt.lr, opt.betas[0]))
    assert got == ref                                  # same float64 arithmetic in the same order: bit-identical
    assert opt.betas[1] == 0.999
    with pytest.raises(ValueError):                    # torch raises when stepped past total_steps; so do we
        sched.step(), sched.step()


def test_one_cycle_lr_epochs_form_and_resume():
    opt = _Opt()
    s = OneCycleLR(opt, 0.01, epochs=5, steps_per_epoch=7)
    assert s.total_steps == 35
    for _ in range(11):
        s.step()
    want = (opt.lr, opt.betas)
    opt2 = _Opt()
    s2 = OneCycleLR(opt2, 0.01, epochs=5, steps_per_epoch=7)
    s2.load_state_dict(s.state_dict())
    assert (opt2.lr, opt2.betas) == want
    with pytest.raises(ValueError):
        OneCycleLR(_Opt(), 0.01)
