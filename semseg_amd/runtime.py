"""Host-side runtime policy: compute dtype selection, layout adoption, DropPath multipliers."""
import threading
import warnings
import torch

from . import ops

_FORCED = None


def set_compute_dtype(dtype):
    """Force the activation dtype of every semseg_amd module (torch.float32 / torch.bfloat16), or None to follow
    torch.autocast / the input dtype."""
    global _FORCED
    if dtype not in (None, torch.float32, torch.bfloat16):
        raise ValueError('compute dtype must be float32, bfloat16 or None')
    _FORCED = dtype


def compute_dtype_for(x):
    if _FORCED is not None:
        return _FORCED
    if torch.is_autocast_enabled():
        adt = torch.get_autocast_dtype('cuda')
        if adt == torch.float16:
            warnings.warn('semseg_amd computes reduced precision in bfloat16 (fp32 accumulation); fp16 autocast is mapped to bf16')
            return torch.bfloat16
        return adt
    return x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32


def as_tokens_nhwc(x, dtype):
    """[B, C, H, W] (any strides) -> channels-last token matrix [B*H*W, C] in `dtype`; zero-copy when x already is the
    permuted view of a channels-last map (what SegFormer stages hand to each other)."""
    b, c, h, w = x.shape
    nhwc = x.permute(0, 2, 3, 1)
    if x.dtype == dtype and nhwc.is_contiguous():
        return nhwc.reshape(b * h * w, c)
    if x.dtype == torch.float32 and x.is_contiguous() and not x.requires_grad:
        return ops.nchw_to_nhwc(x, dtype)
    return nhwc.contiguous().to(dtype).reshape(b * h * w, c)     # rare layouts: torch copy keeps autograd intact


def drop_path_keep(b, drop_prob, dtype, device):
    """Per-sample DropPath multiplier floor(kp + U[0,1)) / kp drawn exactly like modules/helpers.py:20-24 (same torch RNG
    call, shape and dtype), returned as fp32 [B] for the GEMM epilogue."""
    kp = 1 - drop_prob
    return ops.droppath_keep(torch.rand((b, 1, 1), dtype=dtype, device=device), kp)


_tls = threading.local()


class GradAwareFunction(torch.autograd.Function):
    """autograd.Function whose forward can tell whether a graph is being recorded.  Inside ``Function.forward`` grad mode is always
    off and ``ctx.needs_input_grad`` is True for parameters under ``torch.no_grad()`` too, so ``apply`` notes the caller's grad mode
    for ``expect_grads``: a validation forward must not announce gradients that no backward will deliver."""

    @classmethod
    def apply(cls, *args, **kwargs):
        prev = getattr(_tls, 'recording', None)
        _tls.recording = torch.is_grad_enabled()
        try:
            return super().apply(*args, **kwargs)
        finally:
            _tls.recording = prev


def recording():
    """Inside a ``GradAwareFunction.forward``: False when the caller runs under ``torch.no_grad()`` (nothing needs to be kept for a
    backward), True otherwise (also when unknown)."""
    return getattr(_tls, 'recording', None) is not False


def expect_grads(ctx, params, first):
    """Forward-side twin of ``deliver_grads``: tells whoever tracks gradient arrival (engine.GradReducer, through the parameter's
    ``_ss_grad_expect`` callback) that this graph node will deliver one gradient to each of its parameters, ``params`` being the
    inputs ``first``, ``first + 1``, ... of the autograd Function whose forward is running.  Nothing is announced when the caller
    runs under ``torch.no_grad()`` (``GradAwareFunction.apply``)."""
    if getattr(_tls, 'recording', None) is False:
        return
    for i, p in enumerate(params):
        cb = getattr(p, '_ss_grad_expect', None)
        if cb is not None and ctx.needs_input_grad[first + i]:
            cb(p)


def deliver_grads(params, grads):
    """What a backward node returns for its parameters.  A parameter that engine.FlatParams has given a slot in the flat
    gradient buffer (``_ss_grad_sink``) receives its gradient there -- ALL such gradients of the node in one ``ss_multi_add_f32``
    launch, with accumulate semantics like autograd's -- and autograd is handed None for it; its ``_ss_grad_ready`` callback
    (the bucketed all-reduce of engine.GradReducer) fires once the launch is queued.  Everything else goes back to autograd."""
    out, srcs, dsts, ready = [], [], [], []
    for p, g in zip(params, grads):
        sink = getattr(p, '_ss_grad_sink', None)
        if g is None or sink is None or g.dtype != torch.float32 or g.numel() != sink.numel() or g.numel() == 0 \
                or ops.multi_add_layout(g) is None or p.grad is None or p.grad.data_ptr() != sink.data_ptr():
            out.append(g)
            continue
        srcs.append(g), dsts.append(sink), out.append(None)
        cb = getattr(p, '_ss_grad_ready', None)
        if cb is not None:
            ready.append((cb, p))
    if srcs:
        ops.multi_add(srcs, dsts)
        for cb, p in ready:
            cb(p)
    return out
