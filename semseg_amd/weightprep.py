"""Compute-dtype / transposed / packed copies of the fp32 master weights, kept across steps and refreshed in ONE launch.

The reference leaves this to autocast, which casts every weight at every use (``torch.autocast`` around ``core/igniter.py:76-80``).
On this path a weight is needed as a cast copy (forward GEMM), as a transposed copy (data-gradient GEMM) or as a packed conv
matrix; per step that was ~280 launches of ~5 us.  Here every conversion is remembered the first time it is asked for (source
view, destination buffer, kind); when a lookup finds its copy stale -- the parameter's version counter moved (any in-place
update through torch), or ``bump()`` was called (engine.AdaBound updates the flat buffer through a raw pointer) -- ALL remembered
conversions are replayed by ``ss_weight_prep_batch`` from a table in device memory, and the rest of the step is served from
the refreshed buffers.  Only ``nn.Parameter`` sources are remembered: derived tensors (scaled or concatenated weights) change
address every step and go through the single-tensor kernels.

Updating a parameter through ``p.data`` bypasses the version counter: call ``bump()`` after doing that.
"""
import torch
from torch import nn

from . import ops
from ._lib import lib, check

MAX_ENTRIES = 4096
_EPOCH = 0
_state = {}                 # device index -> _DeviceState


class _Entry:
    __slots__ = ('src', 'out', 'row', 'epoch', 'version')


class _DeviceState:
    def __init__(self):
        self.entries, self.order, self.table, self.table_n, self.blocks = {}, [], None, 0, 0
        self.gathered = {}          # (source pointers, dtype) -> persistent concatenation buffer


def bump():
    """Every remembered copy is stale from now on (the optimizer wrote the parameters behind torch's back)."""
    global _EPOCH
    _EPOCH += 1


def clear():
    _state.clear()
    _PADDED.clear()


def _refresh(st, device):
    n = len(st.order)
    if st.table is None or st.table_n != n:
        rows, blocks = [], 0
        for e in st.order:
            kind, a, b, c = e.row
            n_out = e.out.numel()
            code = kind[0] | (ops._DT[e.out.dtype] << 8) | (kind[1] << 16)
            rows.append([e.src.data_ptr(), e.out.data_ptr(), n_out, code, a, b, c, blocks])
            blocks += ((a + 31) // 32) * ((b + 31) // 32) if kind[0] == 1 else (n_out + 1023) // 1024
        st.table = torch.tensor(rows, dtype=torch.int64).to(device)
        st.table_n, st.blocks = n, blocks
    check(lib().ss_weight_prep_batch(st.table.data_ptr(), n, st.blocks, torch.cuda.current_stream(device).cuda_stream))
    for e in st.order:
        e.epoch, e.version = _EPOCH, e.src._version


def _lookup(p, key, make):
    dev = p.device
    if dev.type != 'cuda':
        raise RuntimeError('semseg_amd: parameters must live on a HIP device (no CPU fallback)')
    st = _state.get(dev.index)
    if st is None:
        st = _state[dev.index] = _DeviceState()
    key = (p.data_ptr(), tuple(p.shape)) + key
    e = st.entries.get(key)
    if e is None:
        if len(st.order) >= MAX_ENTRIES:                       # models came and went: start over
            st = _state[dev.index] = _DeviceState()
        e = _Entry()
        e.src, e.out, e.row = make()
        e.epoch, e.version = _EPOCH, e.src._version
        st.entries[key] = e
        st.order.append(e)
        return e.out
    if e.epoch != _EPOCH or e.version != e.src._version:
        with torch.cuda.device(dev):
            _refresh(st, dev)
    return e.out


def _view2d(p, rows, cols):
    v = p.detach().reshape(rows, cols)
    if not v.is_contiguous() or v.dtype != torch.float32:
        raise ValueError('semseg_amd: parameters must be contiguous float32')
    return v


def cast(p, dtype):
    """Parameter [N, ...] -> [N, K] in the compute dtype (the fp32 view itself when dtype is float32)."""
    rows = p.shape[0]
    cols = p.numel() // rows
    if dtype == torch.float32 or not isinstance(p, nn.Parameter):
        v = p.detach().reshape(rows, cols)
        return v if dtype == torch.float32 else ops.cast(v.contiguous(), dtype)

    def make():
        v = _view2d(p, rows, cols)
        return v, ops.cast(v, dtype), ((0, 0), 0, 0, 0)
    return _lookup(p, ('cast', dtype), make)


def transpose(p, rows, cols, dtype, reverse_cols=False):
    """Parameter viewed as [rows, cols] -> [cols, rows] in dtype (columns optionally read in reverse order)."""
    if not isinstance(p, nn.Parameter):
        return ops.transpose_cast(p.detach().reshape(rows, cols).contiguous(), dtype, reverse_cols=reverse_cols)

    def make():
        v = _view2d(p, rows, cols)
        return v, ops.transpose_cast(v, dtype, reverse_cols=reverse_cols), ((1, int(reverse_cols)), rows, cols, 0)
    return _lookup(p, ('tr', rows, cols, dtype, bool(reverse_cols)), make)


def gather(params, dtype):
    """Several parameters laid end to end in ONE persistent buffer of ``dtype`` -- what ``torch.cat`` / ``torch.stack`` of weights gave
    (modules/lawin.py: the per-head position-mixing Linears as one [heads, 64, 64] operand, phi | g as one projection): each
    parameter is remembered as a cast whose destination is its slice of the buffer, so the buffer is refreshed by the same batched
    launch as every other copy and no concatenation runs in the step.  -> flat [sum numel] tensor."""
    params = list(params)
    dev = params[0].device
    if dev.type != 'cuda':
        raise RuntimeError('semseg_amd: parameters must live on a HIP device (no CPU fallback)')
    if not all(isinstance(p, nn.Parameter) for p in params):
        return torch.cat([p.detach().reshape(-1) for p in params]).to(dtype)     # derived tensors (tests): nothing to remember
    st = _state.get(dev.index)
    if st is None:
        st = _state[dev.index] = _DeviceState()
    key = (tuple(p.data_ptr() for p in params), dtype)
    buf = st.gathered.get(key)
    if buf is None:
        buf = st.gathered[key] = torch.empty(sum(p.numel() for p in params), dtype=dtype, device=dev)
    off = 0
    for p in params:
        n = p.numel()

        def make(p=p, off=off, n=n):
            v = _view2d(p, 1, n)
            out = buf[off:off + n]
            check(lib().ss_cast(0, ops._DT[dtype], v.data_ptr(), out.data_ptr(), n, torch.cuda.current_stream(dev).cuda_stream))
            return v, out, ((0, 0), 0, 0, 0)
        _lookup(p, ('gather', dtype, buf.data_ptr(), off), make)
        off += n
    return buf


_PADDED = {}              # data_ptr of a remembered transpose -> the zero-padded buffer it is the head of


def transpose_padded(p, rows, cols, dtype, pad_rows):
    """``transpose`` whose [cols, rows] result is the head of a persistent [pad_rows, rows] buffer with zero rows behind it (a GEMM
    operand whose row count must be a multiple of 8: the ConvTranspose2d weight of core/decoder.py as [(o, dy, dx)][c])."""
    if not isinstance(p, nn.Parameter):
        full = torch.zeros((pad_rows, rows), dtype=dtype, device=p.device)
        full[:cols] = ops.transpose_cast(p.detach().reshape(rows, cols).contiguous(), dtype)
        return full

    def make():
        v = _view2d(p, rows, cols)
        full = torch.zeros((pad_rows, rows), dtype=dtype, device=p.device)
        out = full[:cols]
        check(lib().ss_transpose_cast(ops._DT[dtype], v.data_ptr(), out.data_ptr(), rows, cols, 0, torch.cuda.current_stream(p.device).cuda_stream))
        _PADDED[out.data_ptr()] = full
        return v, out, ((1, 0), rows, cols, 0)
    out = _lookup(p, ('trp', rows, cols, dtype, pad_rows), make)
    return _PADDED[out.data_ptr()]


def pack(p, dtype, transposed=False, k_multiple=8):
    """Conv weight [O, I, k, k] -> packed GEMM operand [O, Kp] (or its transpose [Kp, O]), column order (ky, kx, i), zero padded
    to a multiple of k_multiple columns."""
    if not isinstance(p, nn.Parameter):
        wp, wpt = ops.pack_conv_weight(p.detach(), dtype, transposed=transposed, k_multiple=k_multiple)
        return wpt if transposed else wp
    O, I, k, _ = p.shape
    Kp = (k * k * I + k_multiple - 1) // k_multiple * k_multiple

    def make():
        v = p.detach()
        if not v.is_contiguous() or v.dtype != torch.float32:
            raise ValueError('semseg_amd: parameters must be contiguous float32')
        wp, wpt = ops.pack_conv_weight(v, dtype, transposed=transposed, k_multiple=k_multiple)
        return v, (wpt if transposed else wp), ((3 if transposed else 2, 0), O, I, k | (Kp << 32))
    return _lookup(p, ('pack', dtype, bool(transposed), k_multiple), make)


def cast_padded(p, dtype, pad_rows):
    """``cast`` whose [rows, cols] result is the head of a persistent [pad_rows, cols] buffer with zero rows behind it (a Linear that
    writes a few always-zero output columns: the padding of a concat buffer to whole k-tiles).  -> the padded buffer."""
    rows = p.shape[0]
    cols = p.numel() // rows
    if not isinstance(p, nn.Parameter):
        full = torch.zeros((pad_rows, cols), dtype=dtype, device=p.device)
        full[:rows] = p.detach().reshape(rows, cols).to(dtype)
        return full

    def make():
        v = _view2d(p, rows, cols)
        full = torch.zeros((pad_rows, cols), dtype=dtype, device=p.device)
        out = full[:rows]
        check(lib().ss_cast(0, ops._DT[dtype], v.data_ptr(), out.data_ptr(), rows * cols, torch.cuda.current_stream(p.device).cuda_stream))
        _PADDED[out.data_ptr()] = full
        return v, out, ((0, 0), 0, 0, 0)
    out = _lookup(p, ('castp', dtype, pad_rows), make)
    return _PADDED[out.data_ptr()]
