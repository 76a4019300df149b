// Column-walker kernels of the x2 BatchNorm-of-a-resize entry points (norm.hip includes this file twice, inside namespaces w8 and
// w4, with UN = 8 and UN = 4 channels per thread).  No include guard on purpose.
struct FN { float v[UN]; };
template <typename T> struct RawN {
    T __attribute__((ext_vector_type(UN))) v;
    __device__ __forceinline__ void load(const T* p) { v = *(const T __attribute__((ext_vector_type(UN)))*)p; }
    __device__ __forceinline__ float get(int e) const { return (float)v[e]; }
};
__device__ __forceinline__ FN loadN(const float* p) { FN r;
#pragma unroll
    for (int q = 0; q < UN / 4; ++q) { floatx4 a = *(const floatx4*)(p + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) r.v[4 * q + i] = a[i]; }
    return r; }
template <typename T> __device__ __forceinline__ FN loadN(const T* p) { RawN<T> q; q.load(p); FN r;
#pragma unroll
    for (int i = 0; i < UN; ++i) r.v[i] = q.get(i);
    return r; }
template <typename T> __device__ __forceinline__ void storeN(T* p, const FN& r) {
    T __attribute__((ext_vector_type(UN))) a;
#pragma unroll
    for (int i = 0; i < UN; ++i) a[i] = from_f<T>(r.v[i]);
    *(T __attribute__((ext_vector_type(UN)))*)p = a;
}
struct Up2Geom { int cpb, xl, groups, xtiles, ystrips, tp, chunks; };
static inline Up2Geom up2_geom(int B, int hi, int wi, int C) {
    Up2Geom g; g.chunks = C / UN;
    int cpb = 1; while (cpb < g.chunks && cpb < 64) cpb <<= 1;
    g.cpb = cpb; g.xl = 256 / cpb; g.groups = (int)cdiv(g.chunks, cpb);
    g.xtiles = (int)cdiv(wi, g.xl);
    int tp = 24;                                           // row pairs per strip (a multiple of 3: the window rotates through 3 slots)
    while (tp > 6 && (int64_t)B * g.xtiles * cdiv(hi, tp) * g.groups < 4096) tp -= 6;
    g.tp = tp; g.ystrips = (int)cdiv(hi, tp);
    return g;
}
template <typename T> struct Up2Row { RawN<T> a0, b0, a1, b1; };
template <typename T> struct Up2 {
    const T* base; int64_t rs, oA0, oB0, oA1, oB1; int hi; float l00, l01, l10, l11; float bv[UN];
    // bvec: optional per-image, per-channel constant [B, C] added to the resized value (a term of the pre-resize map that is constant
    // over an image commutes with the resize: the weights of every output pixel sum to one)
    __device__ __forceinline__ void init(const T* x, int64_t ld, int hi_, int wi, int img, int j, int c, const float* bvec, int C) {
        const int jc = min(max(j, 0), wi - 1);
        const LinTap t0 = lin_tap(2 * jc, 0.5f, wi), t1 = lin_tap(2 * jc + 1, 0.5f, wi);
        hi = hi_; rs = (int64_t)wi * ld; base = x + (int64_t)img * hi * rs + c;
        oA0 = t0.i0 * ld; oB0 = t0.i1 * ld; oA1 = t1.i0 * ld; oB1 = t1.i1 * ld;
        l00 = t0.l0; l01 = t0.l1; l10 = t1.l0; l11 = t1.l1;
#pragma unroll
        for (int e = 0; e < UN; ++e) bv[e] = bvec ? bvec[(int64_t)img * C + c + e] : 0.f;
    }
    __device__ __forceinline__ void issue(int row, Up2Row<T>& r) const {
        const T* p = base + (int64_t)min(max(row, 0), hi - 1) * rs;
        r.a0.load(p + oA0); r.b0.load(p + oB0); r.a1.load(p + oA1); r.b1.load(p + oB1);
    }
    __device__ __forceinline__ void hlerp(const Up2Row<T>& r, float (&h)[2][UN]) const {
#pragma unroll
        for (int e = 0; e < UN; ++e) {
            h[0][e] = fmaf(l01, r.b0.get(e), l00 * r.a0.get(e));
            h[1][e] = fmaf(l11, r.b1.get(e), l10 * r.a1.get(e));
        }
    }
};
// the two output rows of pair p from the lerped input rows p-1, p, p+1: z[r][col][e], rounded to T as ss_bilinear_fwd stores them
template <typename T>
__device__ __forceinline__ void up2_rows(int p, int hi, const float (&prev)[2][UN], const float (&cur)[2][UN], const float (&next)[2][UN],
                                         float (&z)[2][2][UN], const float (&bv)[UN]) {
    const LinTap te = lin_tap(2 * p, 0.5f, hi), to = lin_tap(2 * p + 1, 0.5f, hi);
#pragma unroll
    for (int col = 0; col < 2; ++col)
#pragma unroll
        for (int e = 0; e < UN; ++e) {
            z[0][col][e] = to_f(from_f<T>(fmaf(te.l1, cur[col][e], te.l0 * prev[col][e]))) + bv[e];
            z[1][col][e] = to_f(from_f<T>(fmaf(to.l1, next[col][e], to.l0 * cur[col][e]))) + bv[e];
        }
}
// Shared frame of the walker kernels.  body(p, z) runs once per row pair p in [ps, ps + 3 * ceil((pe - ps) / 3)): pairs >= pe are
// computed from clamped rows and must be ignored by the body.
template <typename T, typename F>
__device__ __forceinline__ void up2_walk(const Up2<T>& U, int ps, int pe, F&& body) {
    float H[3][2][UN]; Up2Row<T> ring[3];
    { Up2Row<T> r; U.issue(ps - 1, r); U.hlerp(r, H[0]); U.issue(ps, r); U.hlerp(r, H[1]); U.issue(ps + 1, r); U.hlerp(r, H[2]); }
#pragma unroll
    for (int i = 0; i < 3; ++i) U.issue(ps + 2 + i, ring[i]);
    for (int p0 = ps; p0 < pe; p0 += 3) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int p = p0 + i;
            float z[2][2][UN];
            up2_rows<T>(p, U.hi, H[i % 3], H[(i + 1) % 3], H[(i + 2) % 3], z, U.bv);
            body(p, z);
            U.hlerp(ring[i], H[i % 3]);                      // row p + 2 takes the slot of row p - 1
            U.issue(p + 5, ring[i]);
        }
    }
}

struct Up2Tile { int img, j, ps, pe; bool colok; };
__device__ __forceinline__ Up2Tile up2_tile(int tile, const Up2Geom& g, int hi, int wi, int xs) {
    const int per_img = g.xtiles * g.ystrips;
    Up2Tile t; t.img = tile / per_img; const int rem = tile - t.img * per_img;
    const int ys = rem / g.xtiles, xt = rem - ys * g.xtiles;
    t.j = xt * g.xl + xs; t.colok = t.j < wi;
    t.ps = ys * g.tp; t.pe = min(hi, t.ps + g.tp);
    return t;
}

// partial[blockIdx.x][0][c] = sum (z - shift), [1] = sum (z - shift)^2 over this workgroup's tiles, shift = x[0][c]
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_up2_kernel(const T* __restrict__ x, int64_t ld, const float* __restrict__ bvec, int B, int hi, int wi, int C, Up2Geom g,
                                                           float* __restrict__ partial) {
    __shared__ float red[256 * UN];
    const int tid = threadIdx.x, cl = tid % g.cpb, xs = tid / g.cpb;
    const int chunk_raw = blockIdx.y * g.cpb + cl;
    const bool cok = chunk_raw < g.chunks;
    const int c = (cok ? chunk_raw : g.chunks - 1) * UN;
    float acc[2][UN];
#pragma unroll
    for (int e = 0; e < UN; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
    const FN sh = loadN(x + c);
    const int ntiles = B * g.xtiles * g.ystrips;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const Up2Tile t = up2_tile(tile, g, hi, wi, xs);
        Up2<T> U; U.init(x, ld, hi, wi, t.img, t.j, c, bvec, C);
        up2_walk<T>(U, t.ps, t.pe, [&](int p, const float (&z)[2][2][UN]) {
            const float live = (t.colok && cok && p < t.pe) ? 1.f : 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int col = 0; col < 2; ++col)
#pragma unroll
                    for (int e = 0; e < UN; ++e) { const float d = (z[r][col][e] - sh.v[e]) * live; acc[0][e] += d; acc[1][e] = fmaf(d, d, acc[1][e]); }
        });
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < UN; ++e) red[tid * UN + e] = acc[k][e];
        __syncthreads();
        if (xs == 0 && cok) {
            float* dst = partial + ((int64_t)blockIdx.x * 2 + k) * C + c;
#pragma unroll
            for (int e = 0; e < UN; ++e) { float s = 0.f;
                for (int q = 0; q < g.xl; ++q) s += red[(q * g.cpb + cl) * UN + e];
                dst[e] = s; }
        }
    }
}

// y[2p + r][2j + col] = act(gamma * (z - mean) * rstd + beta)
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_apply_up2_kernel(const T* __restrict__ x, int64_t ld, const float* __restrict__ bvec, int hi, int wi, const float* __restrict__ stats,
        const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y, int64_t ldy, int C, Up2Geom g) {
    const int tid = threadIdx.x, cl = tid % g.cpb, xs = tid / g.cpb;
    const int chunk = blockIdx.y * g.cpb + cl;
    if (chunk >= g.chunks) return;
    const int c = chunk * UN;
    const Up2Tile t = up2_tile(blockIdx.x, g, hi, wi, xs);
    if (!t.colok) return;
    FN sc, sh;
    { FN mu = loadN(stats + c), rs = loadN(stats + C + c), gm = loadN(gamma + c), b = loadN(beta + c);
#pragma unroll
      for (int e = 0; e < UN; ++e) { sc.v[e] = rs.v[e] * gm.v[e]; sh.v[e] = b.v[e] - mu.v[e] * sc.v[e]; } }
    const int wo = 2 * wi;
    T* yb = y + ((int64_t)t.img * 2 * hi * wo + 2 * t.j) * ldy + c;
    Up2<T> U; U.init(x, ld, hi, wi, t.img, t.j, c, bvec, C);
    up2_walk<T>(U, t.ps, t.pe, [&](int p, const float (&z)[2][2][UN]) {
        if (p < t.pe) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int col = 0; col < 2; ++col) {
                    FN o;
#pragma unroll
                    for (int e = 0; e < UN; ++e) { float u = fmaf(z[r][col][e], sc.v[e], sh.v[e]); o.v[e] = ACT == 1 ? mish_f(u) : u; }
                    storeN(yb + ((int64_t)(2 * p + r) * wo + col) * ldy, o);
                }
        }
    });
}

// backward over the resized map: PASS2 = false: partial sums of g and g * xhat; true: dz (the gradient of the resized map)
template <typename T, int ACT, bool PASS2>
__global__ __launch_bounds__(256) void bn_bwd_up2_kernel(const T* __restrict__ dy, int64_t ldd, const T* __restrict__ x, int64_t ld, const float* __restrict__ bvec, int B, int hi, int wi,
        const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ coef,
        T* __restrict__ dz, int64_t ldo, float* __restrict__ partial, int C, Up2Geom g) {
    __shared__ float red[PASS2 ? 1 : 256 * UN];
    const int tid = threadIdx.x, cl = tid % g.cpb, xs = tid / g.cpb;
    const int chunk_raw = blockIdx.y * g.cpb + cl;
    const bool cok = chunk_raw < g.chunks;
    const int c = (cok ? chunk_raw : g.chunks - 1) * UN;
    float acc[2][UN];
#pragma unroll
    for (int e = 0; e < UN; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
    FN xs_, xo, gm, bt, k0, k1, k2;                          // xhat = z * xs + xo;  dz = k0 * g + k1 + xhat * k2
    { FN mu = loadN(stats + c), rs = loadN(stats + C + c);
      gm = loadN(gamma + c); bt = loadN(beta + c);
#pragma unroll
      for (int e = 0; e < UN; ++e) { xs_.v[e] = rs.v[e]; xo.v[e] = -mu.v[e] * rs.v[e]; }
      if (PASS2) { FN c1 = loadN(coef + c), c2 = loadN(coef + C + c);
#pragma unroll
          for (int e = 0; e < UN; ++e) { k0.v[e] = gm.v[e] * rs.v[e]; k1.v[e] = -k0.v[e] * c1.v[e]; k2.v[e] = -k0.v[e] * c2.v[e]; } } }
    const int wo = 2 * wi, ntiles = B * g.xtiles * g.ystrips;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const Up2Tile t = up2_tile(tile, g, hi, wi, xs);
        const int jc = min(t.j, wi - 1);
        const int64_t ob = (int64_t)t.img * 2 * hi * wo + 2 * jc;
        const T* db = dy + ob * ldd + c;
        T* zb = dz + ob * ldo + c;
        const bool live_t = t.colok && cok;
        Up2<T> U; U.init(x, ld, hi, wi, t.img, t.j, c, bvec, C);
        RawN<T> dn[2][2];                                    // dy of the pair about to be consumed
        { const int pp = min(t.ps, hi - 1);
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
              for (int col = 0; col < 2; ++col) dn[r][col].load(db + ((int64_t)(2 * pp + r) * wo + col) * ldd); }
        up2_walk<T>(U, t.ps, t.pe, [&](int p, const float (&z)[2][2][UN]) {
            RawN<T> d[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int col = 0; col < 2; ++col) d[r][col] = dn[r][col];
            { const int pp = min(p + 1, hi - 1);
#pragma unroll
              for (int r = 0; r < 2; ++r)
#pragma unroll
                  for (int col = 0; col < 2; ++col) dn[r][col].load(db + ((int64_t)(2 * pp + r) * wo + col) * ldd); }
            const bool live = live_t && p < t.pe;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int col = 0; col < 2; ++col) {
                    FN o;
#pragma unroll
                    for (int e = 0; e < UN; ++e) {
                        const float xh = fmaf(z[r][col][e], xs_.v[e], xo.v[e]);
                        float gg = d[r][col].get(e);
                        if (ACT == 1) gg *= mish_grad(fmaf(xh, gm.v[e], bt.v[e]));
                        if (PASS2) o.v[e] = fmaf(k0.v[e], gg, fmaf(xh, k2.v[e], k1.v[e]));
                        else {
                            // pass 1 with a destination: g = dy * act'(.) is kept (rounded to the storage dtype, and summed as
                            // rounded) so that the second pass does not evaluate the activation's derivative again
                            if (dz) { gg = (float)(T)gg; o.v[e] = gg; }
                            if (live) { acc[0][e] += gg; acc[1][e] = fmaf(gg, xh, acc[1][e]); }
                        }
                    }
                    if ((PASS2 || dz) && live) storeN(zb + ((int64_t)(2 * p + r) * wo + col) * ldo, o);
                }
        });
    }
    if (!PASS2) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < UN; ++e) red[tid * UN + e] = acc[k][e];
            __syncthreads();
            if (xs == 0 && cok) {
                float* dst = partial + ((int64_t)blockIdx.x * 2 + k) * C + c;
#pragma unroll
                for (int e = 0; e < UN; ++e) { float s = 0.f;
                    for (int q = 0; q < g.xl; ++q) s += red[(q * g.cpb + cl) * UN + e];
                    dst[e] = s; }
            }
        }
    }
}
// Pass 2 with the transpose of the resize folded in: dx[b, i, j] = sum over the output pixels whose taps touch (i, j) of their weight
// times dz, dz never written.  Same walker; a workgroup's outer two columns are halo (computed, not stored).  Per row pair a thread
// has dz of its 2 x 2 output pixels, exchanges the outer columns with its column neighbours through LDS (one barrier per step, two
// buffers), forms the horizontal sum of its input column for both output rows, and folds the rows vertically: input row p takes
// output rows 2p-1 (carried over from the previous step), 2p, 2p+1 and 2p+2 (added one step later, when the row is stored).
// Weights are those of lin_tap, so the clamped border taps fold onto the edge pixels exactly as in ss_bilinear_bwd.
__device__ __forceinline__ float tap_weight(int o, int n_out, int i, int n_in) {
    if (o < 0 || o >= n_out) return 0.f;
    const LinTap t = lin_tap(o, 0.5f, n_in);
    return (t.i0 == i ? t.l0 : 0.f) + (t.i1 == i ? t.l1 : 0.f);
}
struct Up2DxGeom { int cpb, xl, groups, xtiles, ystrips, tp, chunks; };
static inline Up2DxGeom up2dx_geom(int B, int hi, int wi, int C) {
    Up2DxGeom g; g.chunks = C / UN;
    int cpb = 1; while (cpb < g.chunks && cpb < 32) cpb <<= 1;
    g.cpb = cpb; g.xl = 512 / cpb; g.groups = (int)cdiv(g.chunks, cpb);
    g.xtiles = (int)cdiv(wi, g.xl - 2);
    int tp = 22;                                           // input rows per strip; tp + 2 steps, a multiple of 3
    while (tp > 4 && (int64_t)B * g.xtiles * cdiv(hi, tp) * g.groups < 2048) tp -= 6;
    g.tp = tp; g.ystrips = (int)cdiv(hi, tp);
    return g;
}
template <typename T, int ACT>
__global__ __launch_bounds__(512) void bn_bwd_up2_dx_kernel(const T* __restrict__ dy, int64_t ldd, const T* __restrict__ x, int64_t ld, const float* __restrict__ bvec, int hi, int wi,
        const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ coef,
        T* __restrict__ dx, int64_t ldo, int C, Up2DxGeom g) {
    __shared__ float ex[2][512][2][2][UN];                    // [buffer][thread][output row][output column][channel]
    const int tid = threadIdx.x, cl = tid % g.cpb, xs = tid / g.cpb;
    const int chunk_raw = blockIdx.y * g.cpb + cl;
    const bool cok = chunk_raw < g.chunks;
    const int c = (cok ? chunk_raw : g.chunks - 1) * UN;
    FN xs_, xo, gm, bt, k0, k1, k2;
    { FN mu = loadN(stats + c), rs = loadN(stats + C + c), c1 = loadN(coef + c), c2 = loadN(coef + C + c);
      gm = loadN(gamma + c); bt = loadN(beta + c);
#pragma unroll
      for (int e = 0; e < UN; ++e) { xs_.v[e] = rs.v[e]; xo.v[e] = -mu.v[e] * rs.v[e];
          k0.v[e] = gm.v[e] * rs.v[e]; k1.v[e] = -k0.v[e] * c1.v[e]; k2.v[e] = -k0.v[e] * c2.v[e]; } }
    const int per_img = g.xtiles * g.ystrips;
    const int img = blockIdx.x / per_img, rem = blockIdx.x - img * per_img;
    const int ys = rem / g.xtiles, xt = rem - ys * g.xtiles;
    const int j = xt * (g.xl - 2) - 1 + xs, ps = ys * g.tp, pe = min(hi, ps + g.tp);
    const bool colok = j >= 0 && j < wi;
    const bool inner = colok && cok && xs > 0 && xs + 1 < g.xl;
    const int jc = min(max(j, 0), wi - 1), wo = 2 * wi, ho = 2 * hi;
    const float wxl = tap_weight(2 * j - 1, wo, j, wi), wx0 = tap_weight(2 * j, wo, j, wi), wx1 = tap_weight(2 * j + 1, wo, j, wi),
                wxr = tap_weight(2 * j + 2, wo, j, wi);
    const int tl = xs > 0 ? tid - g.cpb : tid, tr = xs + 1 < g.xl ? tid + g.cpb : tid;
    const int64_t ob = (int64_t)img * ho * wo + 2 * jc;
    const T* db = dy + ob * ldd + c;
    T* xb = dx + ((int64_t)img * hi * wi + jc) * ldo + c;
    Up2<T> U; U.init(x, ld, hi, wi, img, j, c, bvec, C);
    RawN<T> dn[2][2];
    { const int pp = min(max(ps - 1, 0), hi - 1);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int col = 0; col < 2; ++col) dn[r][col].load(db + ((int64_t)(2 * pp + r) * wo + col) * ldd); }
    float accp[UN], carry[UN];
#pragma unroll
    for (int e = 0; e < UN; ++e) { accp[e] = 0.f; carry[e] = 0.f; }
    int par = 0;
    up2_walk<T>(U, ps - 1, pe + 1, [&](int p, const float (&z)[2][2][UN]) {
        RawN<T> d[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int col = 0; col < 2; ++col) d[r][col] = dn[r][col];
        { const int pp = min(max(p + 1, 0), hi - 1);
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
              for (int col = 0; col < 2; ++col) dn[r][col].load(db + ((int64_t)(2 * pp + r) * wo + col) * ldd); }
        const float live = (colok && p >= 0 && p < hi) ? 1.f : 0.f;         // dz is zero outside the image
        float gz[2][2][UN];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int col = 0; col < 2; ++col)
#pragma unroll
                for (int e = 0; e < UN; ++e) {
                    const float xh = fmaf(z[r][col][e], xs_.v[e], xo.v[e]);
                    float gg = d[r][col].get(e);
                    if (ACT == 1) gg *= mish_grad(fmaf(xh, gm.v[e], bt.v[e]));
                    gz[r][col][e] = fmaf(k0.v[e], gg, fmaf(xh, k2.v[e], k1.v[e])) * live;
                    ex[par][tid][r][col][e] = gz[r][col][e];
                }
        __syncthreads();
        float hx[2][UN];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < UN; ++e)
                hx[r][e] = fmaf(wxr, ex[par][tr][r][0][e], fmaf(wx1, gz[r][1][e], fmaf(wx0, gz[r][0][e], wxl * ex[par][tl][r][1][e])));
        par ^= 1;
        // vertical fold (wave-uniform weights)
        const float w_prev = tap_weight(2 * p, ho, p - 1, hi), w_c0 = tap_weight(2 * p, ho, p, hi), w_c1 = tap_weight(2 * p + 1, ho, p, hi),
                    w_next = tap_weight(2 * p + 1, ho, p + 1, hi);
        FN o;
#pragma unroll
        for (int e = 0; e < UN; ++e) {
            o.v[e] = fmaf(w_prev, hx[0][e], accp[e]);                           // input row p - 1 is complete
            accp[e] = fmaf(w_c1, hx[1][e], fmaf(w_c0, hx[0][e], carry[e]));
            carry[e] = w_next * hx[1][e];
        }
        if (inner && p - 1 >= ps && p - 1 < pe) storeN(xb + (int64_t)(p - 1) * wi * ldo, o);
    });
}

static inline int up2_blocks(const Up2Geom& g, int B) {
    const int64_t t = (int64_t)B * g.xtiles * g.ystrips;
    return (int)(t < SS_COLRED_MAX_BLOCKS ? t : SS_COLRED_MAX_BLOCKS);
}

