"""Gradient delivery into engine.FlatParams' flat buffer (runtime.deliver_grads + ss_multi_add_f32) against plain autograd."""
import pytest
import torch

from semseg_amd.detinit import det_fill_, det_input, det_labels

pytestmark = pytest.mark.gpu


def _step(net, crit, x, lab):
    loss = crit(net(x), lab)
    loss.backward()
    return loss


def test_flat_gradients_equal_autograd_gradients():
    from semseg_amd.model import SegFormerLawin
    from semseg_amd.loss import OhemCrossEntropyLoss
    from semseg_amd.engine import FlatParams, AdaBound
    net = SegFormerLawin('B0', 128, 6)
    det_fill_(net.encoder), det_fill_(net.decoder)
    net = net.cuda().eval()                                   # eval: no DropPath / Dropout draws, the two passes see the same graph
    x, lab = det_input('eng.x', (2, 3, 64, 64)).cuda(), det_labels('eng.lab', (2, 64, 64), 6, cell=8).cuda()
    _step(net, OhemCrossEntropyLoss(0, 0.7, 0.1), x, lab)
    ref = {n: p.grad.clone() for n, p in net.named_parameters()}
    assert all(g is not None for g in ref.values())
    for p in net.parameters():
        p.grad = None
    flat = FlatParams(net)
    fired = []
    for p in flat.params:
        p._ss_grad_ready = fired.append
    flat.zero_grad()
    _step(net, OhemCrossEntropyLoss(0, 0.7, 0.1), x, lab)
    for n, p in net.named_parameters():
        assert p.grad.data_ptr() == p._ss_grad_sink.data_ptr()                  # still the view of the flat buffer
        assert torch.equal(p.grad, ref[n]), n
    assert len(fired) == len(flat.params) and {id(p) for p in fired} == {id(p) for p in flat.params}
    _step(net, OhemCrossEntropyLoss(0, 0.7, 0.1), x, lab)                       # no zero_grad in between: accumulates like autograd
    for n, p in net.named_parameters():
        assert torch.allclose(p.grad, 2 * ref[n], rtol=1e-6, atol=0), n
    # the flat optimizer consumes the same buffer
    before = flat.data.clone()
    AdaBound(flat).step()
    assert not torch.equal(before, flat.data)


def test_multi_add_ragged_sizes():
    from semseg_amd import ops
    torch.manual_seed(0)
    sizes = [1, 3, 4, 4095, 4096, 4097, 70000, 5] * 9                           # > 48 tensors: two launches; odd sizes: scalar tails
    pool = torch.randn(sum(sizes) + 8, device='cuda')
    srcs, dsts, o = [], [], 1                                                   # offset 1: unaligned slots
    for n in sizes:
        srcs.append(torch.randn(n, device='cuda'))
        dsts.append(pool[o:o + n])
        o += n
    slab = torch.randn(5, 64, 65, device='cuda')                                # strided sources: [64][64] rows 65 apart + the 65th column
    for h in range(5):
        for src in (slab[h, :, :64], slab[h, :, 64]):
            srcs.append(src)
            dsts.append(torch.randn(src.numel(), device='cuda'))
    want = [d + s.reshape(-1) for s, d in zip(srcs, dsts)]
    edge = (pool[0].item(), pool[o:].clone())
    ops.multi_add(srcs, dsts)
    for d, w in zip(dsts, want):
        assert torch.equal(d, w)
    assert pool[0].item() == edge[0] and torch.equal(pool[o:], edge[1])


# ---------------------------------------------------------------------------- world size 2 on ONE GPU (gloo, host-staged collectives)
def _ddp_worker(rank, world, port, out):
    import os, sys
    import torch.distributed as dist
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from semseg_amd.model import SegFormerLawin
    from semseg_amd.engine import FlatParams, GradReducer, broadcast_module_state
    net = SegFormerLawin('B0', 128, 6)
    det_fill_(net.encoder), det_fill_(net.decoder)
    net = net.cuda().eval()
    if rank == 1:
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.5)
    broadcast_module_state(net)                      # rank 0's parameters win
    flat = FlatParams(net)
    red = GradReducer(flat, bucket_bytes=1 << 20)    # several buckets, launched while backward is still running
    x, proj = det_input('ddp.x', (4, 3, 64, 64)).cuda(), det_input('ddp.p', (4, 6, 64, 64), -1, 1).cuda()
    res = {'buckets': len(red.buckets)}
    for step in range(2):                            # the reducer re-arms itself
        flat.zero_grad()
        lo = rank * 2
        (net(x[lo:lo + 2]) * proj[lo:lo + 2]).sum().backward()
        red.wait()
        res[f'grad{step}'] = flat.grad.cpu().clone()
    # the network called twice before ONE backward: every parameter receives two deliveries (runtime.deliver_grads); a bucket
    # must not travel after the first one.  Eval mode: images are independent, so the sum equals the two-image batch above.
    flat.zero_grad()
    lo = rank * 2
    ((net(x[lo:lo + 1]) * proj[lo:lo + 1]).sum() + (net(x[lo + 1:lo + 2]) * proj[lo + 1:lo + 2]).sum()).backward()
    assert all(e == 2 for e in red.expected) and all(g == 2 for g in red.got)
    red.wait()
    res['grad_twice'] = flat.grad.cpu().clone()
    res['launched_all'] = True
    torch.save(res, os.path.join(out, f'r{rank}.pt'))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2_on_one_gpu(tmp_path):
    """The bucket hooks fired from runtime.deliver_grads (not from autograd) must all-reduce every bucket: SUM over two ranks' half
    batches == the single-process gradient of the whole batch (eval mode: images independent), on both ranks, twice."""
    import os, socket
    import torch.multiprocessing as mp
    from semseg_amd.model import SegFormerLawin
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f'r{r}.pt'), weights_only=True) for r in (0, 1))
    net = SegFormerLawin('B0', 128, 6)
    det_fill_(net.encoder), det_fill_(net.decoder)
    net = net.cuda().eval()
    x, proj = det_input('ddp.x', (4, 3, 64, 64)).cuda(), det_input('ddp.p', (4, 6, 64, 64), -1, 1).cuda()
    (net(x) * proj).sum().backward()
    ref = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1), (0, (-p.numel()) % 64)) for p in reversed(list(net.parameters()))]).cpu()
    assert r0['buckets'] >= 3
    for r in (r0, r1):
        for step in range(2):
            g = r[f'grad{step}']
            assert g.shape == ref.shape
            assert float((g - ref).abs().max()) <= 1e-4 * float(ref.abs().max()), step
        assert float((r['grad_twice'] - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    assert torch.equal(r0['grad1'], r1['grad1']) and torch.equal(r0['grad_twice'], r1['grad_twice'])


# ---------------------------------------------------------------------------- the real backend (RCCL) in a world of one
def _nccl_worker(rank, world, port, out):
    import os, sys
    import torch.distributed as dist
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    from semseg_amd.model import SegFormerLawin
    from semseg_amd.loss import OhemCrossEntropyLoss
    from semseg_amd.metrics import SegmentationMetrics
    from semseg_amd.engine import FlatParams, GradReducer, AdaBound
    net = SegFormerLawin('B0', 128, 6)
    det_fill_(net.encoder), det_fill_(net.decoder)
    net = net.cuda().train()
    x, lab = det_input('nccl.x', (2, 3, 64, 64)).cuda(), det_labels('nccl.lab', (2, 64, 64), 6, cell=8).cuda()
    flat = FlatParams(net)
    res = {'backend': dist.get_backend()}

    def one_step(reducer):
        torch.manual_seed(11)                                 # the same DropPath / Dropout2d draws in both runs
        met = SegmentationMetrics(6, ignore_index=0).cuda()
        flat.zero_grad()
        logits = net(x)
        loss = OhemCrossEntropyLoss(0, 0.7, 0.1)(logits, lab)
        met.update(logits, lab)
        loss.backward()
        launched_in_backward = sum(reducer.launched) if reducer is not None else 0
        scale = reducer.wait() if reducer is not None else 1.0
        torch.cuda.synchronize()
        return flat.grad.clone(), met.compute()['Confusion_Matrix'].clone(), launched_in_backward, scale

    g_ref, cm_ref, _, _ = one_step(None)
    red = GradReducer(flat, bucket_bytes=1 << 20, overlap=True, force_comm=True)
    res['buckets'], res['side_stream'] = len(red.buckets), red.stream is not None
    g1, cm1, n1, scale = one_step(red)
    with torch.no_grad():                                     # a validation forward between two steps must not hold the buckets back
        net(x)
    g2, cm2, n2, _ = one_step(red)
    res.update(grad_equal=bool(torch.equal(g_ref, g1) and torch.equal(g_ref, g2)), cm_equal=bool(torch.equal(cm_ref, cm1) and torch.equal(cm_ref, cm2)),
               launched_in_backward=(n1, n2), scale=scale, grad_nonzero=float(g_ref.abs().max()) > 0)
    before = flat.data.clone()
    AdaBound(flat).step(scale)
    res['stepped'] = not torch.equal(before, flat.data)
    torch.save(res, os.path.join(out, 'nccl.pt'))
    dist.destroy_process_group()


def test_rccl_world1_bucketed_allreduce_on_side_stream(tmp_path):
    """train.py:306-311 (DDP over RCCL): a fresh child process creates the `nccl` process group (world size 1), GradReducer keeps its
    hook / side-stream / collective path on (force_comm), and one MiT-B0 + Lawin training step plus the confusion-matrix read-out
    must equal the run without a reducer bit for bit (a one-rank SUM is the identity); every bucket must have been launched from the
    backward's delivery hooks, also in the step after a no_grad forward."""
    import os, socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_nccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(os.path.join(tmp_path, 'nccl.pt'), weights_only=True)
    assert r['backend'] == 'nccl' and r['side_stream'] and r['buckets'] >= 3
    assert r['grad_nonzero'] and r['grad_equal'] and r['cm_equal'] and r['stepped'] and r['scale'] == 1.0
    assert r['launched_in_backward'] == (r['buckets'], r['buckets'])
