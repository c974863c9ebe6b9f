// HBM-bound row kernels of the DiT path: fused RMSNorm+modulation, attention preparation
// (window gather + q/k RMSNorm + RoPE), text mean, patch in / one-step endpoint.  All loads and
// stores are 16 bytes per lane (8 bf16); reductions use wave shuffles; arithmetic is fp32.
#include "common.h"

namespace {

// ----------------------------------------------------------------------------- rmsnorm + modulation
// One wave per row, 4 rows per 256-thread workgroup.  dim <= 8192 -> at most 16 vectors per lane.
template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_mod_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, long long rows, int dim,
                                                          float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = dim >> 3;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        const bf16x8* xr = (const bf16x8*)(x + row * dim);
        bf16x8 v[MAXV];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                v[i] = xr[c];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = bf2f(v[i][e]);
                    ss += f * f;
                }
            }
        }
        ss = wave_sum(ss);
        const float rstd = 1.0f / sqrtf(ss / (float)dim + eps);
        bf16x8* yr = (bf16x8*)(y + row * dim);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                const f32x4 s0 = *(const f32x4*)(scale + c * 8), s1 = *(const f32x4*)(scale + c * 8 + 4);
                f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
                if (shift) {
                    h0 = *(const f32x4*)(shift + c * 8);
                    h1 = *(const f32x4*)(shift + c * 8 + 4);
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = f2bf(bf2f(v[i][e]) * rstd * s0[e] + h0[e]);
                    o[e + 4] = f2bf(bf2f(v[i][e + 4]) * rstd * s1[e] + h1[e]);
                }
                yr[c] = o;
            }
        }
    }
}

// ----------------------------------------------------------------------------- qkv preparation
// Work item = (position s, group of 4 heads), its q, k and v rows: 64 lanes x 8 dims = 4 heads x 128 per tensor.
// A head's 128 dims sit on 16 consecutive lanes -> 16-lane segmented reduction for the RMS.
// RoPE pairs (2i, 2i+1) are inside a lane's 8 dims; pair index p = dim/2 in [0,63):
// axis = p / nfreq, freq = p % nfreq, pairs >= 3*nfreq pass through.
struct PrepArgs {
    const bf16* qkv_vid;
    const bf16* qkv_txt;
    const int32_t* src_row;
    const int32_t* pos;
    const float* qn_vid; const float* kn_vid; const float* qn_txt; const float* kn_txt;
    const float* rope_cos; const float* rope_sin;
    bf16* q; bf16* k; bf16* v;
    long long S;
    int H, nfreq, max_pos, rope_mode;
    float eps;
};

__global__ __launch_bounds__(256) void qkv_prep_kernel(const PrepArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hq = (a.H + 3) >> 2;          // head quads per tensor
    const long long items = a.S * hq;       // one item = (position, head quad): its q, k and v rows (3 KiB in flight per wave)
    const int sub = lane >> 4, l16 = lane & 15;
    const long long row_elems = 3ll * a.H * 128;
    for (long long it = (long long)blockIdx.x * 4 + wave; it < items; it += (long long)gridDim.x * 4) {
        const long long s = it / hq;
        const int quad = (int)(it - s * hq);
        const int head = quad * 4 + sub;
        const int src = a.src_row[s];
        const bool is_txt = src < 0;
        const bf16* base = is_txt ? a.qkv_txt + (long long)(-1 - src) * row_elems : a.qkv_vid + (long long)src * row_elems;
        const bool hv = head < a.H;
        bf16x8 val[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            if (hv) val[o] = *(const bf16x8*)(base + ((long long)o * a.H + head) * 128 + l16 * 8);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) val[o][e] = f2bf(0.f);
        }
        const bool rope = a.rope_mode == 1 || (a.rope_mode == 2 && !is_txt);
        float rc[4], rs[4];                 // this lane's four rotation pairs (shared by q and k)
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) { rc[pr] = 1.f; rs[pr] = 0.f; }
        if (rope) {
            const int pt = a.pos[s * 3 + 0], ph = a.pos[s * 3 + 1], pw = a.pos[s * 3 + 2];
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const int p = l16 * 4 + pr;
                const int axis = p / a.nfreq;
                if (axis < 3) {
                    const int fi = p - axis * a.nfreq;
                    if (a.rope_mode == 1) {
                        const int pv = axis == 0 ? pt : (axis == 1 ? ph : pw);
                        rc[pr] = a.rope_cos[pv * a.nfreq + fi];
                        rs[pr] = a.rope_sin[pv * a.nfreq + fi];
                    } else {  // per-position tables [S, 3*nfreq]
                        rc[pr] = a.rope_cos[s * (3 * a.nfreq) + p];
                        rs[pr] = a.rope_sin[s * (3 * a.nfreq) + p];
                    }
                }
            }
        }
        const long long doff = (s * a.H + head) * 128 + l16 * 8;
        if (hv) *(bf16x8*)(a.v + doff) = val[2];   // v: plain gather
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            float f[8];
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f[e] = bf2f(val[o][e]);
                ss += f[e] * f[e];
            }
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) ss += __shfl_xor(ss, m, 64);  // stays inside the 16-lane segment
            const float rstd = 1.0f / sqrtf(ss * (1.0f / 128.f) + a.eps);
            const float* nw = (o == 0) ? (is_txt ? a.qn_txt : a.qn_vid) : (is_txt ? a.kn_txt : a.kn_vid);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = f[e] * rstd * nw[l16 * 8 + e];
            if (rope) {
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const float x1 = f[2 * pr], x2 = f[2 * pr + 1];
                    f[2 * pr] = x1 * rc[pr] - x2 * rs[pr];
                    f[2 * pr + 1] = x2 * rc[pr] + x1 * rs[pr];
                }
            }
            if (hv) {
                bf16x8 o8;
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] = f2bf(f[e]);
                *(bf16x8*)((o == 0 ? a.q : a.k) + doff) = o8;
            }
        }
    }
}

// ----------------------------------------------------------------------------- text mean over windows
__global__ __launch_bounds__(256) void txt_mean_kernel(const bf16* __restrict__ attn, const int32_t* __restrict__ row0,
                                                       int nwin, int Lt, int dim, bf16* __restrict__ out) {
    const int nvec = dim >> 3;
    const long long total = (long long)Lt * nvec;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i / nvec), c = (int)(i - (long long)j * nvec);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // windows are added in order (fixed summation order); eight windows' rows are requested before the first add - with
        // 216 windows at 4K the one-load-per-iteration form spent 164 us per layer waiting on dependent loads
        int w = 0;
        for (; w + 8 <= nwin; w += 8) {
            bf16x8 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const bf16x8*)(attn + ((long long)row0[w + u] + j) * dim + c * 8);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[u][e]);
        }
        for (; w < nwin; ++w) {
            const bf16x8 v = *(const bf16x8*)(attn + ((long long)row0[w] + j) * dim + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
        }
        bf16x8 o;
        const float inv = 1.0f / (float)nwin;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e] * inv);
        *(bf16x8*)(out + (long long)j * dim + c * 8) = o;
    }
}

// ----------------------------------------------------------------------------- patch in / out
// rows[l, (h2 w2 c)] with c over [noise(C) | latent(C) | 1], zero padded to kpad.
__global__ __launch_bounds__(256) void patch_in_kernel(const bf16* __restrict__ noise, const bf16* __restrict__ latent,
                                                       bf16* __restrict__ rows, int T, int H, int W, int C, int kpad) {
    const int Hp = H >> 1, Wp = W >> 1;
    const int cin = 2 * C + 1;
    const long long total = (long long)T * Hp * Wp * kpad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int kcol = (int)(i % kpad);
        const long long l = i / kpad;
        float v = 0.f;
        if (kcol < 4 * cin) {
            const int c = kcol % cin, hw = kcol / cin;
            const int dh = hw >> 1, dw = hw & 1;
            const int wp = (int)(l % Wp);
            const long long t2 = l / Wp;
            const int hp = (int)(t2 % Hp), t = (int)(t2 / Hp);
            const long long pix = ((long long)t * H + (hp * 2 + dh)) * W + (wp * 2 + dw);
            if (c < C) v = bf2f(noise[pix * C + c]);
            else if (c < 2 * C) v = bf2f(latent[pix * C + (c - C)]);
            else v = 1.0f;
        }
        rows[i] = f2bf(v);
    }
}

// x0[t, h, w, c] = noise - pred[l, (dh dw c)]
__global__ __launch_bounds__(256) void patch_out_x0_kernel(const bf16* __restrict__ pred, long long ldp,
                                                           const bf16* __restrict__ noise, bf16* __restrict__ x0, int T,
                                                           int H, int W, int C) {
    const int Hp = H >> 1, Wp = W >> 1;
    const long long total = (long long)T * H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long pix = i / C;
        const int w = (int)(pix % W);
        const long long t2 = pix / W;
        const int h = (int)(t2 % H), t = (int)(t2 / H);
        const long long l = ((long long)t * Hp + (h >> 1)) * Wp + (w >> 1);
        const int col = (((h & 1) * 2) + (w & 1)) * C + c;
        x0[i] = f2bf(bf2f(noise[i]) - bf2f(pred[l * ldp + col]));
    }
}

// ----------------------------------------------------------------------------- explicit im2col (small Cin)
struct Im2colArgs {
    const bf16* x; const bf16* carry; bf16* col;
    int T, Hi, Wi, Cin, kt, kh, kw, st, sh, sw, npad, pad_h0, pad_w0, To, Ho, Wo;
    int crep;       // head frames the carry does not hold (svr2_conv3d_desc.carry_frames): they read carry[0]
    long long kpad;
};
// One thread = 8 consecutive K columns of one output pixel (one 16-byte store; a row of the col matrix is written
// by kpad/8 consecutive lanes).  The (tap, channel) decomposition of the first column is divided out once and then
// stepped; the gathers are 2-byte loads of a few neighbouring pixels (L1 hits).
__global__ __launch_bounds__(256) void im2col3d_kernel(const Im2colArgs a) {
    const long long M = (long long)a.To * a.Ho * a.Wo;
    const int cpr = (int)(a.kpad >> 3);                     // 16-byte chunks per row
    const long long total = M * cpr;
    const int Kreal = a.kt * a.kh * a.kw * a.Cin;
    const long long frame = (long long)a.Hi * a.Wi * a.Cin;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int chunk = (int)(i % cpr);
        const long long m = i / cpr;
        const int x = (int)(m % a.Wo);
        const long long t2 = m / a.Wo;
        const int y = (int)(t2 % a.Ho), t = (int)(t2 / a.Ho);
        const int iy0 = y * a.sh - a.pad_h0, ix0 = x * a.sw - a.pad_w0, it0 = t * a.st;
        int kcol = chunk * 8;
        int c = kcol % a.Cin, tap = kcol / a.Cin;
        int kx = tap % a.kw;
        tap /= a.kw;
        int ky = tap % a.kh, kt = tap / a.kh;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e, ++kcol) {
            float v = 0.f;
            if (kcol < Kreal) {
                const int iy = iy0 + ky, ix = ix0 + kx, it = it0 + kt;
                if ((unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi) {
                    const bf16* fb = it < a.npad ? (a.carry ? a.carry + (long long)max(it - a.crep, 0) * frame : a.x)
                                                 : a.x + (long long)(it - a.npad) * frame;
                    v = bf2f(fb[((long long)iy * a.Wi + ix) * a.Cin + c]);
                }
            }
            o[e] = f2bf(v);
            if (++c == a.Cin) {
                c = 0;
                if (++kx == a.kw) {
                    kx = 0;
                    if (++ky == a.kh) { ky = 0; ++kt; }
                }
            }
        }
        *(bf16x8*)(a.col + m * a.kpad + chunk * 8) = o;
    }
}

inline int grid_for(long long work_items, int per_block) {
    long long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return (int)g;
}

}  // namespace

extern "C" int svr2_rmsnorm_mod(const void* x, void* y, const float* scale, const float* shift, int64_t rows,
                                int32_t dim, float eps, void* stream) {
    SVR2_CHECK(x && y && scale, SVR2_E_ARG, "svr2_rmsnorm_mod: null pointer");
    SVR2_CHECK(rows > 0 && dim > 0 && dim % 8 == 0 && dim <= 8192, SVR2_E_SHAPE,
               "svr2_rmsnorm_mod: need rows > 0, dim %% 8 == 0, dim <= 8192 (dim=%d)", dim);
    SVR2_CHECK(aligned16(x) && aligned16(y) && aligned16(scale) && (!shift || aligned16(shift)), SVR2_E_ALIGN,
               "svr2_rmsnorm_mod: pointers must be 16-byte aligned");
    const int grid = grid_for(rows, 4);
    hipStream_t s = (hipStream_t)stream;
    if (dim <= 1024)
        hipLaunchKernelGGL(rmsnorm_mod_kernel<2>, dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)y, scale, shift,
                           (long long)rows, dim, eps);
    else if (dim <= 3072)
        hipLaunchKernelGGL(rmsnorm_mod_kernel<6>, dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)y, scale, shift,
                           (long long)rows, dim, eps);
    else
        hipLaunchKernelGGL(rmsnorm_mod_kernel<16>, dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)y, scale, shift,
                           (long long)rows, dim, eps);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_qkv_prep(const svr2_qkv_prep_desc* d, void* stream) {
    SVR2_CHECK(d && d->qkv_vid && d->qkv_txt && d->src_row && d->q && d->k && d->v && d->qn_vid && d->kn_vid &&
                   d->qn_txt && d->kn_txt,
               SVR2_E_ARG, "svr2_qkv_prep: null pointer");
    SVR2_CHECK(d->D == 128 && d->H > 0 && d->S > 0, SVR2_E_SHAPE, "svr2_qkv_prep: D must be 128 (D=%d)", d->D);
    SVR2_CHECK(d->rope_mode >= 0 && d->rope_mode <= 2, SVR2_E_ARG, "svr2_qkv_prep: rope_mode");
    if (d->rope_mode != 0) {
        SVR2_CHECK(d->pos && d->rope_cos && d->rope_sin && d->nfreq > 0 && 3 * d->nfreq <= 64, SVR2_E_ARG,
                   "svr2_qkv_prep: rope tables missing or nfreq too large");
    }
    SVR2_CHECK(aligned16(d->qkv_vid) && aligned16(d->qkv_txt) && aligned16(d->q) && aligned16(d->k) && aligned16(d->v),
               SVR2_E_ALIGN, "svr2_qkv_prep: pointers must be 16-byte aligned");
    PrepArgs a;
    a.qkv_vid = (const bf16*)d->qkv_vid; a.qkv_txt = (const bf16*)d->qkv_txt;
    a.src_row = d->src_row; a.pos = d->pos;
    a.qn_vid = d->qn_vid; a.kn_vid = d->kn_vid; a.qn_txt = d->qn_txt; a.kn_txt = d->kn_txt;
    a.rope_cos = d->rope_cos; a.rope_sin = d->rope_sin;
    a.q = (bf16*)d->q; a.k = (bf16*)d->k; a.v = (bf16*)d->v;
    a.S = d->S; a.H = d->H; a.nfreq = d->nfreq; a.max_pos = d->max_pos; a.rope_mode = d->rope_mode; a.eps = d->eps;
    const long long items = a.S * ((a.H + 3) / 4);
    hipLaunchKernelGGL(qkv_prep_kernel, dim3(grid_for(items, 4)), dim3(256), 0, (hipStream_t)stream, a);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_txt_mean(const void* attn, const int32_t* txt_row0, int32_t nwin, int32_t Lt, int32_t dim,
                             void* out, void* stream) {
    SVR2_CHECK(attn && txt_row0 && out, SVR2_E_ARG, "svr2_txt_mean: null pointer");
    SVR2_CHECK(nwin > 0 && Lt > 0 && dim > 0 && dim % 8 == 0, SVR2_E_SHAPE, "svr2_txt_mean: bad sizes");
    SVR2_CHECK(aligned16(attn) && aligned16(out), SVR2_E_ALIGN, "svr2_txt_mean: alignment");
    hipLaunchKernelGGL(txt_mean_kernel, dim3(grid_for((long long)Lt * (dim / 8), 256)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16*)attn, txt_row0, nwin, Lt, dim, (bf16*)out);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_patch_in(const void* noise, const void* latent, void* rows, int32_t T, int32_t H, int32_t W,
                             int32_t C, int32_t kpad, void* stream) {
    SVR2_CHECK(noise && latent && rows, SVR2_E_ARG, "svr2_patch_in: null pointer");
    SVR2_CHECK(T > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && kpad >= 4 * (2 * C + 1), SVR2_E_SHAPE,
               "svr2_patch_in: need even H, W and kpad >= 4*(2C+1)");
    const long long total = (long long)T * (H / 2) * (W / 2) * kpad;
    hipLaunchKernelGGL(patch_in_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)noise, (const bf16*)latent, (bf16*)rows, T, H, W, C, kpad);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_patch_out_x0(const void* pred, int64_t ldp, const void* noise, void* x0, int32_t T, int32_t H,
                                 int32_t W, int32_t C, void* stream) {
    SVR2_CHECK(pred && noise && x0, SVR2_E_ARG, "svr2_patch_out_x0: null pointer");
    SVR2_CHECK(T > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && ldp >= 4 * C, SVR2_E_SHAPE,
               "svr2_patch_out_x0: bad sizes");
    const long long total = (long long)T * H * W * C;
    hipLaunchKernelGGL(patch_out_x0_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)pred, (long long)ldp, (const bf16*)noise, (bf16*)x0, T, H, W, C);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_im2col3d(const svr2_conv3d_desc* d, void* col, int64_t kpad, void* stream) {
    SVR2_CHECK(d && d->x && col, SVR2_E_ARG, "svr2_im2col3d: null pointer");
    SVR2_CHECK(kpad % 64 == 0 && kpad >= (int64_t)d->kt * d->kh * d->kw * d->Cin, SVR2_E_SHAPE,
               "svr2_im2col3d: kpad must be a multiple of 64 and >= kt*kh*kw*Cin");
    SVR2_CHECK(d->To == (d->T + d->npad - d->kt) / d->st + 1 && d->Ho >= 1 && d->Wo >= 1 &&
                   (d->To - 1) * d->st + d->kt - 1 < d->T + d->npad,
               SVR2_E_SHAPE, "svr2_im2col3d: inconsistent output size");
    Im2colArgs a;
    a.x = (const bf16*)d->x; a.carry = (const bf16*)d->carry; a.col = (bf16*)col;
    a.T = d->T; a.Hi = d->Hi; a.Wi = d->Wi; a.Cin = d->Cin; a.kt = d->kt; a.kh = d->kh; a.kw = d->kw;
    a.st = d->st; a.sh = d->sh; a.sw = d->sw; a.npad = d->npad; a.pad_h0 = d->pad_h0; a.pad_w0 = d->pad_w0;
    a.To = d->To; a.Ho = d->Ho; a.Wo = d->Wo; a.kpad = kpad;
    SVR2_CHECK(d->carry_frames >= 0 && d->carry_frames <= d->npad, SVR2_E_SHAPE, "svr2_im2col3d: carry_frames outside [0, npad]");
    a.crep = (d->carry && d->carry_frames > 0) ? d->npad - d->carry_frames : 0;
    const long long total = (long long)d->To * d->Ho * d->Wo * (kpad / 8);
    hipLaunchKernelGGL(im2col3d_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}
