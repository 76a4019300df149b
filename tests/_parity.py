"""Error measures shared by the parity tests, with a record of what was measured.

``rel`` / ``rel_l2`` return a ``Measured`` float: comparing it with a tolerance (``assert rel(a, b) < 1e-4``) stores the measured
value, the tolerance and the call site.  At the end of a pytest session ``dump()`` (called from conftest.py) writes the worst value
per call site to ``gpurun_out/parity_errors.json`` (replacing the file: one session, one record); DESIGN.md's table of measured
errors is that file.
"""
import inspect
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LOG = {}


def _site():
    for fr in inspect.stack()[2:]:
        fn = os.path.basename(fr.filename)
        if fn.startswith('test_') or fn.startswith('wip_'):
            return f'{fn}:{fr.lineno}', fr.function
    return '?', '?'


class Measured(float):
    metric = 'max'

    def _rec(self, tol):
        site, func = _site()
        e = _LOG.setdefault(site, {'test': func, 'metric': self.metric, 'worst': 0.0, 'tol': float(tol), 'n': 0})
        e['worst'] = max(e['worst'], float(self))
        e['tol'] = max(e['tol'], float(tol))
        e['n'] += 1

    def __lt__(self, tol):
        self._rec(tol)
        return float(self) < float(tol)

    def __le__(self, tol):
        self._rec(tol)
        return float(self) <= float(tol)


def _m(v, metric):
    m = Measured(v)
    m.metric = metric
    return m


def _f(t):
    return torch.as_tensor(t).detach().float().cpu()


def rel(a, b, floor=1e-6):
    """max |a - b| / max |b|"""
    a, b = _f(a), _f(b)
    return _m(float((a - b).abs().max() / max(float(b.abs().max()), floor)), 'max|d|/max|ref|')


def rel_l2(a, b):
    """||a - b|| / ||b||"""
    a, b = _f(a), _f(b)
    return _m(float((a - b).norm() / b.norm()), 'rel-L2')


def measured(v, metric='value'):
    return _m(float(v), metric)


def dump(path=None):
    if not _LOG:
        return
    path = path or os.path.join(ROOT, 'gpurun_out', 'parity_errors.json')
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        old = {}
        # one pytest session = one record: a previous file is replaced, not merged (stale worst cases of older trees would survive);
        # PARITY_MERGE=1 keeps it for a suite that is run file by file
        if os.environ.get('PARITY_MERGE') == '1' and os.path.exists(path):
            with open(path) as f:
                old = json.load(f)
        old.update(_LOG)
        with open(path, 'w') as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except OSError:
        pass
