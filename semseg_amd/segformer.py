"""MiT / SegFormer encoder on the MI355X kernels.

Same constructors, forward signatures, ``.channels`` and ``state_dict`` layout as the reference
``modules/segformer.py`` (Attention :7-41, DWConv :44-53, MLP :56-64, PatchEmbed :67-84, Block :87-99,
SegFormer :102-200), but forward AND backward are explicit sequences of HIP kernel launches (``ops``) inside coarse
``torch.autograd.Function`` s; torch only allocates.  Activations live channels-last as [B*N, C] token matrices in the
compute dtype (fp32, or bf16 with fp32 accumulation / statistics / parameters).
"""
import torch
from torch import nn, Tensor

from . import ops, weightprep
from .runtime import compute_dtype_for, as_tokens_nhwc, drop_path_keep, deliver_grads, expect_grads, GradAwareFunction, recording
from .shapes import MIT_VARIANTS, MIT_HEADS, MIT_SR, MIT_PATCH


# ------------------------------------------------------------------ weight preparation (cast / repack per step)
def w_fwd(w, dtype):
    """nn.Linear / 1x1 conv weight [N, K, ...] fp32 -> [N, K] in the compute dtype (the fp32 view itself in fp32)."""
    return weightprep.cast(w, dtype)


def w_bwd(w, dtype):
    """[N, K] -> [K, N] in the compute dtype: the B operand of the data-gradient GEMM."""
    n = w.shape[0]
    return weightprep.transpose(w, n, w.numel() // n, dtype)


def w_taps(w, dtype=torch.float32, reverse=False):
    """depthwise conv weight [C, 1, k, k] -> tap-major [k*k, C] (taps reversed for the data gradient)."""
    c = w.shape[0]
    return weightprep.transpose(w, c, w.numel() // c, dtype, reverse_cols=reverse)


def linear_grads(dout, x_in, want_bias=True):
    """-> (dW [N, K] fp32, db [N] fp32)."""
    if want_bias:
        return ops.gemm_tn(dout, x_in, colsum=True)
    return ops.gemm_tn(dout, x_in), None


def branch_head_grads(dout, x_in, w_t, keep, rows):
    """Backward of the LAST Linear of a residual branch under DropPath (modules/segformer.py:96-97): dout is the gradient of the block
    output, the branch sees keep[b] * dout.  -> (dW, db, d(x_in)).  Where the tile kernels take the shape the scaled gradient is never
    materialised: the weight-gradient GEMM skips dropped samples and scales its slabs, the data-gradient GEMM scales rows in its epilogue."""
    if keep is None:
        dW, db = linear_grads(dout, x_in)
        return dW, db, ops.gemm_nt(dout, w_t)
    if ops.gemm_tn_keep_ok(dout, x_in, rows):
        dW, db = ops.gemm_tn_keep(dout, x_in, keep, rows)
        return dW, db, ops.gemm_nt(dout, w_t, rowscale=keep, rows_per_scale=rows)
    g = ops.rowscale(dout, keep, rows)
    dW, db = linear_grads(g, x_in)
    return dW, db, ops.gemm_nt(g, w_t)


# ------------------------------------------------------------------ attention (modules/segformer.py:21-41)
def attn_forward(P, x, B, h, w, head, sr, res=None, keep=None):
    """x: normed tokens [B*N, C].  P: dict of fp32 parameters.  -> out [B*N, C], saved-state dict."""
    T = x.dtype
    N = h * w
    C = x.shape[1]
    S = {'x': x}
    q = ops.gemm_nt(x, w_fwd(P['q.weight'], T), bias=P['q.bias'])
    if sr > 1:
        wp = weightprep.pack(P['sr.weight'], T)
        if ops.conv_patch_ok(x, B, h, w, C, sr, C):             # the GEMM reads the patches out of x itself: no patch matrix
            s = ops.conv_patch_fwd(x, B, h, w, C, sr, wp, bias=P['sr.bias'])
            ho, wo = h // sr, w // sr
            S.update(col=None)
        else:
            col, ho, wo = ops.im2col(x, B, h, w, C, sr, sr, 0)
            s = ops.gemm_nt(col, wp, bias=P['sr.bias'])
            S.update(col=col)
        xr, m, r = ops.layernorm_fwd(s, P['norm.weight'], P['norm.bias'], 1e-5)
        S.update(s=s, m=m, r=r)
        Nk = ho * wo
    else:
        xr, Nk = x, N
    kv = ops.gemm_nt(xr, w_fwd(P['kv.weight'], T), bias=P['kv.bias'])
    ao, lse = ops.mit_attn_fwd(q, kv, B, N, Nk, head)
    out = ops.gemm_nt(ao, w_fwd(P['proj.weight'], T), bias=P['proj.bias'], res=res, rowscale=keep,
                      rows_per_scale=N if keep is not None else 0)
    S.update(q=q, kv=kv, ao=ao, lse=lse, xr=xr, Nk=Nk)
    return out, S


def attn_backward(P, S, dout, B, h, w, head, sr, keep=None):
    """dout: gradient of the block output on this branch; keep: the branch's DropPath multipliers [B] (or None).  -> dx, {param: grad}."""
    T = dout.dtype
    N, C = h * w, dout.shape[1]
    G = {}
    G['proj.weight'], G['proj.bias'], dao = branch_head_grads(dout, S['ao'], w_bwd(P['proj.weight'], T), keep, N)
    dq, dkv = ops.mit_attn_bwd(S['q'], S['kv'], S['ao'], dao, S['lse'], B, N, S['Nk'], head)
    G['kv.weight'], G['kv.bias'] = linear_grads(dkv, S['xr'])
    dxr = ops.gemm_nt(dkv, w_bwd(P['kv.weight'], T))
    if sr > 1:
        ds, dgb = ops.layernorm_bwd(dxr, S['s'], S['m'], S['r'], P['norm.weight'])
        G['norm.weight'], G['norm.bias'] = dgb[0], dgb[1]
        wpt = weightprep.pack(P['sr.weight'], T, transposed=True)
        if S['col'] is None:
            gp, G['sr.bias'] = ops.conv_patch_wgrad(ds, S['x'], B, h, w, C, sr)
            dxa = ops.conv_patch_dgrad(ds, wpt, B, h, w, C, sr)
        else:
            gp, G['sr.bias'] = ops.gemm_tn(ds, S['col'], colsum=True)
            dxa = ops.col2im(ops.gemm_nt(ds, wpt), B, h, w, C, sr, sr, 0)
        G['sr.weight'] = ops.unpack_conv_wgrad(gp, C, C, sr)
    else:
        dxa = dxr
    G['q.weight'], G['q.bias'] = linear_grads(dq, S['x'])
    dx = ops.gemm_nt(dq, w_bwd(P['q.weight'], T), res=dxa)
    return dx, G


# ------------------------------------------------------------------ Mix-FFN (modules/segformer.py:44-64)
FUSED_DW_MIN_PIXELS = 4096        # maps at least this large take the one-pass depthwise backward (measured crossover: 32 x 32 loses, 64 x 64 wins)


FUSED_DW_LANE_MIN_PIXELS = 1024   # the lane-per-column kernel (bf16, C % 64 == 0) has no barrier in its row loop: 32 x 32 wins as well (132 against 170 us)


FUSED_MIXFFN = True               # C = 64 / 128 in bf16 (MiT stages 1-2): fc1 -> depthwise + GELU -> fc2 as ONE kernel, hidden tensors on chip


def mlp_forward(P, x, B, h, w, res=None, keep=None):
    T = x.dtype
    w9c = w_taps(P['dw_conv.dw_conv.weight'])
    if FUSED_MIXFFN and ops.mixffn_ok(x, P['fc1.weight'].shape[0]):
        # one launch; a forward that records a graph also stores h1 and a2 (the backward below reads them), inference stores neither
        args = (x, w_fwd(P['fc1.weight'], T), P['fc1.bias'], w9c, P['dw_conv.dw_conv.bias'], w_fwd(P['fc2.weight'], T), P['fc2.bias'], B, h, w)
        if recording():
            out, h1, a2 = ops.mixffn_fwd(*args, res=res, keep=keep, save_hidden=True)
            return out, {'x': x, 'h1': h1, 'a2': a2, 'w9c': w9c}
        return ops.mixffn_fwd(*args, res=res, keep=keep), {'x': x, 'w9c': w9c}
    h1 = ops.gemm_nt(x, w_fwd(P['fc1.weight'], T), bias=P['fc1.bias'])
    a2 = ops.dwconv3_fwd(h1, w9c, P['dw_conv.dw_conv.bias'], B, h, w, act=1)
    out = ops.gemm_nt(a2, w_fwd(P['fc2.weight'], T), bias=P['fc2.bias'], res=res, rowscale=keep,
                      rows_per_scale=h * w if keep is not None else 0)
    return out, {'x': x, 'h1': h1, 'a2': a2, 'w9c': w9c}


def mlp_backward(P, S, dout, B, h, w, keep=None):
    T = dout.dtype
    G = {}
    Ch = S['h1'].shape[1]
    if ops.dwconv3_bwd_fc2_ok(dout, S['h1'], B, h, w) and (keep is None or ops.gemm_tn_keep_ok(dout, S['a2'], h * w)):
        # stages 1-2: fc2's data gradient inside the depthwise backward (da2, a [pixels, 4C] tensor, is never stored)
        if keep is None:
            G['fc2.weight'], G['fc2.bias'] = linear_grads(dout, S['a2'])
        else:
            G['fc2.weight'], G['fc2.bias'] = ops.gemm_tn_keep(dout, S['a2'], keep, h * w)
        dh1, dwdb = ops.dwconv3_bwd_fc2(dout, w_bwd(P['fc2.weight'], T), keep, S['h1'], S['w9c'], P['dw_conv.dw_conv.bias'], B, h, w)
        da2 = None
    else:
        G['fc2.weight'], G['fc2.bias'], da2 = branch_head_grads(dout, S['a2'], w_bwd(P['fc2.weight'], T), keep, h * w)
    if da2 is None:
        pass
    elif h * w >= (FUSED_DW_LANE_MIN_PIXELS if ops.dwconv3_lane_ok(S['h1']) else FUSED_DW_MIN_PIXELS):
        # one pass: dh1 and the tap / bias gradients together, g = da2 * GELU'(u) stays on chip (3 tensor passes instead of 5)
        dh1, dwdb = ops.dwconv3_bwd_fused(da2, S['h1'], S['w9c'], P['dw_conv.dw_conv.bias'], B, h, w, act=1)
    else:
        # small maps (stage 4 of a 1024^2 tile is 32 x 32): too few column tiles per image for the halo exchange to pay
        g, dwdb = ops.dwconv3_bwd(da2, S['h1'], S['w9c'], P['dw_conv.dw_conv.bias'], B, h, w, act=1)
        dh1 = ops.dwconv3_fwd(g, w_taps(P['dw_conv.dw_conv.weight'], reverse=True), None, B, h, w, act=0)
    G['dw_conv.dw_conv.weight'] = ops.transpose_cast(dwdb[:9], torch.float32).reshape(Ch, 1, 3, 3)
    G['dw_conv.dw_conv.bias'] = dwdb[9]
    G['fc1.weight'], G['fc1.bias'] = linear_grads(dh1, S['x'])
    dx = ops.gemm_nt(dh1, w_bwd(P['fc1.weight'], T))
    return dx, G


# ------------------------------------------------------------------ autograd glue
def _names(module_or_names):
    return [n for n, _ in module_or_names.named_parameters()] if isinstance(module_or_names, nn.Module) else module_or_names


class _ParamFn(GradAwareFunction):
    """Base: forward(ctx, x, meta, names, *params); subclasses implement fwd(P, x, meta) -> (out, S) and
    bwd(P, S, dout, meta) -> (dx, {name: grad})."""

    @staticmethod
    def fwd(P, x, meta):
        raise NotImplementedError

    @staticmethod
    def bwd(P, S, dout, meta):
        raise NotImplementedError

    @classmethod
    def run(cls, x, meta, module):
        names = [n for n, _ in module.named_parameters()]
        return cls.apply(x, meta, names, *[p for _, p in module.named_parameters()])

    @classmethod
    def forward(cls, ctx, x, meta, names, *params):
        P = dict(zip(names, params))
        out, S = cls.fwd(P, x.detach(), meta)
        ctx.S, ctx.meta, ctx.names, ctx.params = S, meta, names, params
        expect_grads(ctx, params, 3)
        return out

    @classmethod
    def backward(cls, ctx, dout):
        P = dict(zip(ctx.names, ctx.params))
        dx, G = cls.bwd(P, ctx.S, dout.contiguous(), ctx.meta)
        ctx.S = None
        grads = [G.get(n) for n in ctx.names]
        grads = deliver_grads(ctx.params, [g if g is None else g.reshape(p.shape) for g, p in zip(grads, ctx.params)])
        return (dx if ctx.needs_input_grad[0] else None, None, None, *grads)


def _sub(P, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in P.items() if k.startswith(prefix)}


def _pre(G, prefix):
    return {prefix + k: v for k, v in G.items()}


class AttentionFn(_ParamFn):
    @staticmethod
    def fwd(P, x, meta):
        B, h, w, head, sr = meta
        return attn_forward(P, x, B, h, w, head, sr)

    @staticmethod
    def bwd(P, S, dout, meta):
        B, h, w, head, sr = meta
        return attn_backward(P, S, dout, B, h, w, head, sr)


class MLPFn(_ParamFn):
    @staticmethod
    def fwd(P, x, meta):
        B, h, w = meta
        return mlp_forward(P, x, B, h, w)

    @staticmethod
    def bwd(P, S, dout, meta):
        B, h, w = meta
        return mlp_backward(P, S, dout, B, h, w)


class BlockFn(_ParamFn):
    """x + DropPath(attn(LN1 x)); then + DropPath(mlp(LN2 .))   (modules/segformer.py:96-99); residual adds and the
    DropPath multipliers are fused into the epilogues of the output-projection GEMMs."""

    @staticmethod
    def fwd(P, x, meta):
        B, h, w, head, sr, keep_a, keep_m = meta
        n1, m1, r1 = ops.layernorm_fwd(x, P['norm1.weight'], P['norm1.bias'], 1e-5)
        x2, Sa = attn_forward(_sub(P, 'attn.'), n1, B, h, w, head, sr, res=x, keep=keep_a)
        n2, m2, r2 = ops.layernorm_fwd(x2, P['norm2.weight'], P['norm2.bias'], 1e-5)
        x3, Sm = mlp_forward(_sub(P, 'mlp.'), n2, B, h, w, res=x2, keep=keep_m)
        return x3, {'x': x, 'm1': m1, 'r1': r1, 'Sa': Sa, 'x2': x2, 'm2': m2, 'r2': r2, 'Sm': Sm}

    @staticmethod
    def bwd(P, S, dx3, meta):
        B, h, w, head, sr, keep_a, keep_m = meta
        N = h * w
        dn2, Gm = mlp_backward(_sub(P, 'mlp.'), S['Sm'], dx3, B, h, w, keep=keep_m)
        dx2, dgb2 = ops.layernorm_bwd(dn2, S['x2'], S['m2'], S['r2'], P['norm2.weight'], res=dx3)
        dn1, Ga = attn_backward(_sub(P, 'attn.'), S['Sa'], dx2, B, h, w, head, sr, keep=keep_a)
        dx, dgb1 = ops.layernorm_bwd(dn1, S['x'], S['m1'], S['r1'], P['norm1.weight'], res=dx2)
        G = {'norm1.weight': dgb1[0], 'norm1.bias': dgb1[1], 'norm2.weight': dgb2[0], 'norm2.bias': dgb2[1]}
        G.update(_pre(Ga, 'attn.'))
        G.update(_pre(Gm, 'mlp.'))
        return dx, G


class PatchEmbedFn(_ParamFn):
    """conv(k, s, pad=k//2) as im2col + GEMM, then LayerNorm (modules/segformer.py:79-84).  x: NHWC tokens."""

    @staticmethod
    def fwd(P, x, meta):
        B, H, W, Cin, k, s = meta
        T = x.dtype
        col, ho, wo = ops.im2col(x, B, H, W, Cin, k, s, k // 2)
        wp = weightprep.pack(P['proj.weight'], T)
        z = ops.gemm_nt(col, wp, bias=P['proj.bias'])
        y, m, r = ops.layernorm_fwd(z, P['norm.weight'], P['norm.bias'], 1e-5)
        return y, {'col': col, 'z': z, 'm': m, 'r': r}

    @staticmethod
    def bwd(P, S, dy, meta):
        B, H, W, Cin, k, s = meta
        T = dy.dtype
        Cout = dy.shape[1]
        dz, dgb = ops.layernorm_bwd(dy, S['z'], S['m'], S['r'], P['norm.weight'])
        G = {'norm.weight': dgb[0], 'norm.bias': dgb[1]}
        gp, G['proj.bias'] = ops.gemm_tn(dz, S['col'], colsum=True)
        G['proj.weight'] = ops.unpack_conv_wgrad(gp, Cout, Cin, k)
        dx = None
        if Cin % 8 == 0:                                      # the RGB image (stage 1) needs no gradient
            wpt = weightprep.pack(P['proj.weight'], T, transposed=True)
            dx = ops.col2im(ops.gemm_nt(dz, wpt), B, H, W, Cin, k, s, k // 2)
        return dx, G


class StageNormFn(GradAwareFunction):
    """A stage's final LayerNorm (modules/segformer.py:180-198).  Its output has two consumers, the head and the next stage's patch
    embedding, so it is handed out twice (second output = a view of the first): backward then receives the two gradients separately and
    ``ss_layernorm_bwd`` sums them while it reads them, instead of autograd adding two full feature maps first."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y, m, r = ops.layernorm_fwd(x.detach(), weight, bias, 1e-5)
        ctx.save_for_backward(x, m, r, weight)
        ctx.set_materialize_grads(False)                       # an unused output (the last stage has no next stage) arrives as None
        ctx.params = (weight, bias)
        expect_grads(ctx, ctx.params, 1)
        return y, y.view_as(y)

    @staticmethod
    def backward(ctx, dy_a, dy_b):
        x, m, r, weight = ctx.saved_tensors
        if dy_a is None:
            dy_a, dy_b = dy_b, None
        if dy_a is None:
            return None, None, None
        dx, dgb = ops.layernorm_bwd(dy_a.contiguous(), x.detach(), m, r, weight, dy2=None if dy_b is None else dy_b.contiguous())
        gw, gb = deliver_grads(ctx.params, [dgb[0], dgb[1]])
        return dx, gw, gb


class LayerNormFn(_ParamFn):
    @staticmethod
    def fwd(P, x, meta):
        y, m, r = ops.layernorm_fwd(x, P['weight'], P['bias'], 1e-5)
        return y, {'x': x, 'm': m, 'r': r}

    @staticmethod
    def bwd(P, S, dy, meta):
        dx, dgb = ops.layernorm_bwd(dy, S['x'], S['m'], S['r'], P['weight'])
        return dx, {'weight': dgb[0], 'bias': dgb[1]}


# ------------------------------------------------------------------ nn.Modules with the reference's API
def _tokens2d(x):
    """[B, N, C] contiguous -> [B*N, C] view."""
    b, n, c = x.shape
    return x.contiguous().view(b * n, c)


class Attention(nn.Module):
    def __init__(self, dim, head, sr_ratio):
        super().__init__()
        self.head, self.sr_ratio = head, sr_ratio
        self.scale = (dim // head) ** -0.5
        self.q = nn.Linear(dim, dim)
        self.kv = nn.Linear(dim, dim * 2)
        self.proj = nn.Linear(dim, dim)
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, sr_ratio, sr_ratio)
            self.norm = nn.LayerNorm(dim)

    def forward(self, x: Tensor, h, w) -> Tensor:
        b, n, c = x.shape
        T = compute_dtype_for(x)
        out = AttentionFn.run(_tokens2d(x.to(T)), (b, h, w, self.head, self.sr_ratio), self)
        return out.view(b, n, c)


class DWConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dw_conv = nn.Conv2d(dim, dim, (3, 3), (1, 1), 1, groups=dim)

    def forward(self, x: Tensor, h, w) -> Tensor:
        b, n, c = x.shape
        T = compute_dtype_for(x)
        return DWConvFn.run(_tokens2d(x.to(T)), (b, h, w), self).view(b, n, c)


class DWConvFn(_ParamFn):
    @staticmethod
    def fwd(P, x, meta):
        B, h, w = meta
        C = x.shape[1]
        w9c = w_taps(P['dw_conv.weight'])
        return ops.dwconv3_fwd(x, w9c, P['dw_conv.bias'], B, h, w, act=0), {'x': x, 'w9c': w9c}

    @staticmethod
    def bwd(P, S, dy, meta):
        B, h, w = meta
        C = dy.shape[1]
        g, dwdb = ops.dwconv3_bwd(dy, S['x'], S['w9c'], P['dw_conv.bias'], B, h, w, act=0)
        flip = w_taps(P['dw_conv.weight'], reverse=True)
        dx = ops.dwconv3_fwd(g, flip, None, B, h, w, act=0)
        return dx, {'dw_conv.weight': ops.transpose_cast(dwdb[:9], torch.float32).reshape(C, 1, 3, 3), 'dw_conv.bias': dwdb[9]}


class MLP(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        self.fc1 = nn.Linear(c1, c2)
        self.dw_conv = DWConv(c2)
        self.fc2 = nn.Linear(c2, c1)

    def forward(self, x: Tensor, h, w) -> Tensor:
        b, n, c = x.shape
        T = compute_dtype_for(x)
        return MLPFn.run(_tokens2d(x.to(T)), (b, h, w), self).view(b, n, c)


class PatchEmbed(nn.Module):
    def __init__(self, c1=3, c2=32, patch_size=7, stride=4):
        super().__init__()
        self.proj = nn.Conv2d(c1, c2, (patch_size, patch_size), (stride, stride), (patch_size // 2))
        self.norm = nn.LayerNorm(c2)
        self.patch_size, self.stride = patch_size, stride

    def forward(self, x: Tensor) -> tuple:
        b, cin, H, W = x.shape
        T = compute_dtype_for(x)
        tok = as_tokens_nhwc(x, T)
        k, s = self.patch_size, self.stride
        y = PatchEmbedFn.run(tok, (b, H, W, cin, k, s), self)
        h, w = ops.conv_out_size(H, k, s, k // 2), ops.conv_out_size(W, k, s, k // 2)
        return y.view(b, h * w, -1), h, w


class Block(nn.Module):
    def __init__(self, dim, head, sr_ratio=1, dpr=0.):
        super().__init__()
        from .helpers import DropPath
        self.norm1 = nn.LayerNorm(dim)
        self.attn = Attention(dim, head, sr_ratio)
        self.drop_path = DropPath(dpr) if dpr > 0. else nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MLP(dim, int(dim * 4))
        self.dpr = dpr

    def forward(self, x: Tensor, h, w) -> Tensor:
        b, n, c = x.shape
        T = compute_dtype_for(x)
        keep_a = keep_m = None
        if self.training and self.dpr > 0.:
            keep_a = drop_path_keep(b, self.dpr, T, x.device)      # same torch RNG calls / order as the reference
            keep_m = drop_path_keep(b, self.dpr, T, x.device)
        meta = (b, h, w, self.attn.head, self.attn.sr_ratio, keep_a, keep_m)
        return BlockFn.run(_tokens2d(x.to(T)), meta, self).view(b, n, c)


class SegFormer(nn.Module):
    __available_configs = MIT_VARIANTS

    def __init__(self, model_name: str = 'B0', in_channels: int = 3):
        super().__init__()
        assert model_name in self.__available_configs.keys(), (
            f"Unknown SegFormer variant '{model_name}'! Supported variants: {list(self.__available_configs.keys())}")
        assert in_channels >= 1, "Input channels must be a positive integer!"
        embed_dims, depths = self.__available_configs[model_name]
        embed_dims, depths = list(embed_dims), list(depths)
        self.channels = embed_dims
        dpr = [v.item() for v in torch.linspace(0, 0.1, sum(depths))]
        cur, cin = 0, in_channels
        for s in range(4):
            k, st = MIT_PATCH[s]
            setattr(self, f'patch_embed{s + 1}', PatchEmbed(cin, embed_dims[s], k, st))
            cin = embed_dims[s]
        for s in range(4):
            setattr(self, f'block{s + 1}', nn.ModuleList(
                [Block(embed_dims[s], MIT_HEADS[s], MIT_SR[s], dpr[cur + i]) for i in range(depths[s])]))
            setattr(self, f'norm{s + 1}', nn.LayerNorm(embed_dims[s]))
            cur += depths[s]

    def forward(self, x: Tensor) -> tuple:
        b = x.shape[0]
        outs = []
        for s in range(4):
            x, h, w = getattr(self, f'patch_embed{s + 1}')(x)
            for blk in getattr(self, f'block{s + 1}'):
                x = blk(x, h, w)
            norm = getattr(self, f'norm{s + 1}')
            y, y2 = StageNormFn.apply(_tokens2d(x), norm.weight, norm.bias)
            # NCHW-shaped views of the channels-last map, as the reference: one for the caller, one for the next stage
            outs.append(y.view(b, h, w, -1).permute(0, 3, 1, 2))
            x = y2.view(b, h, w, -1).permute(0, 3, 1, 2)
        return tuple(outs)
