"""CPU oracle for the MI355X hot path (TEST INFRASTRUCTURE ONLY).

Plain-PyTorch (CPU, fp32) functional restatement of the reference's
forward/backward hot path: MiT encoder, Lawin head, OHEM-CE / Dice losses and
the confusion-matrix metrics.  Every function cites the reference file:line
it follows.  Gradients come from torch autograd on these functions.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package, and only as the checker.  The
product package ``semseg_amd`` never imports it.

Pinning: the restatement is checked against the reference itself (imported on
CPU from /root/reference in the build container) by
``tests/golden/make_golden.py``; the resulting vectors are committed under
``tests/golden/*.npz`` and re-checked by ``tests/test_oracle_golden.py``.
The metrics restatement follows torchmetrics<=0.10 documented behaviour; the
reference has no fixtures for it and torchmetrics is absent here, so that
part is "parity unpinned" against the reference (it is cross-checked against
sklearn/torch.bincount instead).
"""
