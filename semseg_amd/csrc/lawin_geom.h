// Window / pooled-grid geometry of the Lawin head kernels (lawin.hip, lawin_mix_mfma.hip).
#pragma once
#include "common.h"

struct WinGeom { int B, H, W, nh, nw, r, st, pad, gh, gw; };
static inline WinGeom win_geom(int B, int H, int W, int r) {
    WinGeom g; g.B = B; g.H = H; g.W = W; g.nh = H / 8; g.nw = W / 8; g.r = r; g.st = 8 / r; g.pad = 4 * (r - 1);
    g.gh = (g.nh - 1) * g.st + 8; g.gw = (g.nw - 1) * g.st + 8;
    return g;
}

// bf16 matrix-core position mixing (lawin_mix_mfma.hip).  cg = C / heads in {8, 16, 32, 64, 128}; C a multiple of 128.
bool lawin_mix_mfma_ok(int C, int heads, const void* src, const void* out);
int64_t lawin_mix_mfma_workspace(int heads);
int lawin_mix_mfma(const void* G, const void* xwin, const float* Wm, const float* bm, void* out, int B, int H, int W, int C, int r, int heads,
                   int transpose, void* workspace, hipStream_t stream);
int64_t lawin_mix_wgrad_mfma_slabs(int B, int H, int W, int C, int heads);
int lawin_mix_wgrad_mfma(const void* G, const void* dwin, float* dWb, int B, int H, int W, int C, int r, int heads, float* workspace,
                         hipStream_t stream);
