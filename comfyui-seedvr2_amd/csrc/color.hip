// Phase-4 colour correction on the device: wavelet reconstruction, AdaIN and the CIELAB
// histogram transfer of the reference's src/utils/color_fix.py, on channels-last bf16 frames [F, H, W, 3] in [-1, 1].
// All of it is HBM-bound elementwise / stencil / reduction work plus six full sorts per batch (rocPRIM device radix
// sort: the histogram match assigns the i-th smallest reference value to the i-th smallest source pixel).
// The reference runs these functions in the compute dtype (bf16): every op rounds to bf16; the kernels round at the
// same places (stated per kernel) and accumulate in fp32 in between.
#include "common.h"

#include <rocprim/rocprim.hpp>

namespace {

__device__ __forceinline__ float rbf(float v) { return bf2f(f2bf(v)); }   // one bf16 rounding

// ----------------------------------------------------------------------------- wavelet level
// low = bf16(sum_{3x3} k[dy][dx] * img[clamp(y + (dy-1)*r), clamp(x + (dx-1)*r)])   (replicate padding, dilation r)
// high = bf16(bf16(high + img) - low)                                                [ref: color_fix.py:122-184]
__global__ __launch_bounds__(256) void wavelet_level_kernel(const bf16* __restrict__ img, bf16* __restrict__ low,
                                                            bf16* __restrict__ high, int F, int H, int W, int r,
                                                            int first) {
    const long long n = (long long)F * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long long t2 = i / W;
        const int y = (int)(t2 % H);
        const long long f = t2 / H;
        const bf16* fr = img + f * H * W * 3;
        int ys[3], xs[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            ys[d] = min(max(y + (d - 1) * r, 0), H - 1);
            xs[d] = min(max(x + (d - 1) * r, 0), W - 1);
        }
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float k = (dy == 1 ? 0.25f : 0.125f) * (dx == 1 ? 1.f : 0.5f);   // 1/16 2/16 1/16 ...
                const bf16* p = fr + ((long long)ys[dy] * W + xs[dx]) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] += k * bf2f(p[c]);
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float lo = rbf(acc[c]);
            low[i * 3 + c] = f2bf(lo);
            if (high) {
                const float h0 = first ? 0.f : bf2f(high[i * 3 + c]);
                high[i * 3 + c] = f2bf(rbf(h0 + bf2f(img[i * 3 + c])) - lo);
            }
        }
    }
}

// out = clamp(bf16(high + low), -1, 1)                                              [ref: color_fix.py:241-246]
__global__ __launch_bounds__(256) void wavelet_merge_kernel(const bf16* __restrict__ high, const bf16* __restrict__ low,
                                                            bf16* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = f2bf(fminf(fmaxf(rbf(bf2f(high[i]) + bf2f(low[i])), -1.f), 1.f));
}

// ----------------------------------------------------------------------------- per-(frame, channel) moments
// partial[f][blk][c] = (sum, sum of squares) over a strip of pixels; folded in fp64 on the device by adain_apply
__global__ __launch_bounds__(256) void moments_kernel(const bf16* __restrict__ x, double* __restrict__ partial, long long P,
                                                      int nblk) {
    __shared__ double red[256 * 6];
    const int f = blockIdx.y, blk = blockIdx.x;
    const long long span = (P + nblk - 1) / nblk, p0 = blk * span, p1 = min(P, p0 + span);
    double s[3] = {0, 0, 0}, q[3] = {0, 0, 0};
    const bf16* fr = x + (long long)f * P * 3;
    for (long long p = p0 + threadIdx.x; p < p1; p += 256)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double v = bf2f(fr[p * 3 + c]);
            s[c] += v;
            q[c] += v * v;
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        red[threadIdx.x * 6 + c] = s[c];
        red[threadIdx.x * 6 + 3 + c] = q[c];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double a = 0;
        for (int t = 0; t < 256; ++t) a += red[t * 6 + threadIdx.x];
        partial[((long long)f * nblk + blk) * 6 + threadIdx.x] = a;
    }
}

// AdaIN in the reference's bf16 op order [ref: color_fix.py:72-119]: mean, var (unbiased) rounded to bf16;
// std = bf16(sqrt(bf16(var + 1e-5))); y = bf16(bf16(bf16(bf16(x - cm) / cs) * ss) + sm)
__global__ __launch_bounds__(256) void adain_apply_kernel(const bf16* __restrict__ x, const double* __restrict__ pc,
                                                          const double* __restrict__ ps, bf16* __restrict__ y, long long P,
                                                          int nblk) {
    __shared__ float st[12];   // cm[3] cs[3] sm[3] ss[3]
    const int f = blockIdx.y;
    if (threadIdx.x < 6) {
        const int c = threadIdx.x % 3;
        const double* part = (threadIdx.x < 3 ? pc : ps) + (long long)f * nblk * 6;
        double s = 0, q = 0;
        for (int b = 0; b < nblk; ++b) {
            s += part[b * 6 + c];
            q += part[b * 6 + 3 + c];
        }
        const double mean = s / (double)P;
        double var = (q - s * mean) / (double)(P - 1);
        if (var < 0) var = 0;
        const float stdv = rbf(sqrtf(rbf(rbf((float)var) + 1e-5f)));
        st[(threadIdx.x < 3 ? 0 : 6) + c] = rbf((float)mean);
        st[(threadIdx.x < 3 ? 3 : 9) + c] = stdv;
    }
    __syncthreads();
    const long long span = (P + gridDim.x - 1) / gridDim.x, p0 = blockIdx.x * span, p1 = min(P, p0 + span);
    const long long base = (long long)f * P * 3;
    for (long long p = p0 + threadIdx.x; p < p1; p += 256)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = rbf(bf2f(x[base + p * 3 + c]) - st[c]);
            v = rbf(v / st[3 + c]);
            v = rbf(v * st[9 + c]);
            y[base + p * 3 + c] = f2bf(v + st[6 + c]);
        }
}

// ----------------------------------------------------------------------------- CIELAB
__device__ __forceinline__ void rgb2lab(float r, float g, float b, float& L, float& A, float& B) {
    auto lin = [](float v) { return v > 0.04045f ? powf((v + 0.055f) / 1.055f, 2.4f) : v / 12.92f; };
    r = lin(r); g = lin(g); b = lin(b);
    float X = 0.4124564f * r + 0.3575761f * g + 0.1804375f * b;
    float Y = 0.2126729f * r + 0.7151522f * g + 0.0721750f * b;
    float Z = 0.0193339f * r + 0.1191920f * g + 0.9503041f * b;
    X /= 0.95047f;
    Z /= 1.08883f;
    const float eps3 = (6.f / 29.f) * (6.f / 29.f) * (6.f / 29.f), kappa = (29.f / 3.f) * (29.f / 3.f) * (29.f / 3.f);
    auto fw = [&](float t) { return t > eps3 ? powf(t, 1.f / 3.f) : (t * kappa + 16.f) / 116.f; };
    const float fx = fw(X), fy = fw(Y), fz = fw(Z);
    L = fy * 116.f - 16.f;
    A = (fx - fy) * 500.f;
    B = (fy - fz) * 200.f;
}

// x bf16 [N,3] in [-1,1] -> planar fp32 lab[3][N] of clamp((x+1)/2, 0, 1)           [ref: color_fix.py:322-329,368-411]
__global__ __launch_bounds__(256) void rgb_to_lab_kernel(const bf16* __restrict__ x, float* __restrict__ lab, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = fminf(fmaxf((bf2f(x[i * 3 + k]) + 1.f) * 0.5f, 0.f), 1.f);
        float L, A, B;
        rgb2lab(c[0], c[1], c[2], L, A, B);
        lab[i] = L;
        lab[n + i] = A;
        lab[2 * n + i] = B;
    }
}

// out[idx_sorted[i]] = ref_sorted[i]                                                 [ref: color_fix.py:473-513]
__global__ __launch_bounds__(256) void rank_scatter_kernel(const int* __restrict__ idx, const float* __restrict__ ref_sorted,
                                                           float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[idx[i]] = ref_sorted[i];
}

__global__ __launch_bounds__(256) void iota_kernel(int* __restrict__ idx, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        idx[i] = (int)i;
}

// L = Lc*lw + Lm*(1-lw); (L, a, b) -> rgb in [0,1] -> bf16(rgb*2 - 1)               [ref: color_fix.py:335-361,414-470]
__global__ __launch_bounds__(256) void lab_to_rgb_kernel(const float* __restrict__ Lc, const float* __restrict__ Lm,
                                                         const float* __restrict__ am, const float* __restrict__ bm, float lw,
                                                         bf16* __restrict__ y, long long n) {
    const float eps = 6.f / 29.f, kappa = (29.f / 3.f) * (29.f / 3.f) * (29.f / 3.f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float L = Lm ? Lc[i] * lw + Lm[i] * (1.f - lw) : Lc[i];
        const float fy = (L + 16.f) / 116.f, fx = am[i] / 500.f + fy, fz = fy - bm[i] / 200.f;
        auto inv = [&](float t) { return t > eps ? t * t * t : (t * 116.f - 16.f) / kappa; };
        const float X = inv(fx) * 0.95047f, Y = inv(fy), Z = inv(fz) * 1.08883f;
        float c[3] = {3.2404542f * X - 1.5371385f * Y - 0.4985314f * Z, -0.9692660f * X + 1.8760108f * Y + 0.0415560f * Z,
                      0.0556434f * X - 0.2040259f * Y + 1.0572252f * Z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = c[k] > 0.0031308f ? powf(fmaxf(c[k], 0.f), 1.f / 2.4f) * 1.055f - 0.055f : c[k] * 12.92f;
            v = fminf(fmaxf(v, 0.f), 1.f);
            y[i * 3 + k] = f2bf(v * 2.f - 1.f);
        }
    }
}

inline int grid_1d(long long n) {
    long long g = (n + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    return (int)(g < 1 ? 1 : g);
}
inline int strips(long long P) {
    long long nb = (P + 4095) / 4096;
    return (int)(nb < 1 ? 1 : (nb > 512 ? 512 : nb));
}

}  // namespace

extern "C" int svr2_wavelet_level(const void* img, void* low, void* high, int32_t F, int32_t H, int32_t W, int32_t radius,
                                  int32_t first, void* stream) {
    SVR2_CHECK(img && low, SVR2_E_ARG, "svr2_wavelet_level: null pointer");
    SVR2_CHECK(F > 0 && H > 0 && W > 0 && radius >= 1, SVR2_E_SHAPE, "svr2_wavelet_level: bad shape");
    const long long n = (long long)F * H * W;
    hipLaunchKernelGGL(wavelet_level_kernel, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, (const bf16*)img, (bf16*)low,
                       (bf16*)high, F, H, W, radius, first);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_wavelet_merge(const void* high, const void* low, void* out, int64_t n, void* stream) {
    SVR2_CHECK(high && low && out && n > 0, SVR2_E_ARG, "svr2_wavelet_merge: bad argument");
    hipLaunchKernelGGL(wavelet_merge_kernel, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, (const bf16*)high,
                       (const bf16*)low, (bf16*)out, (long long)n);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int64_t svr2_adain_ws_bytes(int32_t F, int64_t P) { return (int64_t)2 * F * strips(P) * 6 * sizeof(double); }

extern "C" int svr2_adain(const void* content, const void* style, void* out, int32_t F, int64_t P, void* workspace,
                          void* stream) {
    SVR2_CHECK(content && style && out && workspace, SVR2_E_ARG, "svr2_adain: null pointer");
    SVR2_CHECK(F > 0 && F <= 65535 && P > 1, SVR2_E_SHAPE, "svr2_adain: bad shape");
    const int nblk = strips(P);
    double* pc = (double*)workspace;
    double* ps = pc + (long long)F * nblk * 6;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(moments_kernel, dim3(nblk, F), dim3(256), 0, s, (const bf16*)content, pc, (long long)P, nblk);
    hipLaunchKernelGGL(moments_kernel, dim3(nblk, F), dim3(256), 0, s, (const bf16*)style, ps, (long long)P, nblk);
    hipLaunchKernelGGL(adain_apply_kernel, dim3(nblk, F), dim3(256), 0, s, (const bf16*)content, pc, ps, (bf16*)out,
                       (long long)P, nblk);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_rgb_to_lab(const void* x, float* lab, int64_t n, void* stream) {
    SVR2_CHECK(x && lab && n > 0, SVR2_E_ARG, "svr2_rgb_to_lab: bad argument");
    hipLaunchKernelGGL(rgb_to_lab_kernel, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, lab,
                       (long long)n);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

/* workspace of svr2_hist_match: sorted keys (2 x n floats), indices (2 x n ints) and rocPRIM's scratch */
extern "C" int64_t svr2_hist_match_ws_bytes(int64_t n) {
    if (n <= 0 || n >= (1ll << 31)) return 0;
    size_t t1 = 0, t2 = 0;
    (void)rocprim::radix_sort_pairs(nullptr, t1, (const float*)nullptr, (float*)nullptr, (const int*)nullptr, (int*)nullptr, (size_t)n);
    (void)rocprim::radix_sort_keys(nullptr, t2, (const float*)nullptr, (float*)nullptr, (size_t)n);
    const size_t t = ((t1 > t2 ? t1 : t2) + 255) & ~(size_t)255;
    return (int64_t)(t + 4 * (((size_t)n * 4 + 255) & ~(size_t)255));
}

extern "C" int svr2_hist_match(const float* source, const float* reference, float* out, int64_t n, void* workspace,
                               int64_t ws_bytes, void* stream) {
    SVR2_CHECK(source && reference && out && workspace, SVR2_E_ARG, "svr2_hist_match: null pointer");
    SVR2_CHECK(n > 0 && n < (1ll << 31), SVR2_E_SHAPE, "svr2_hist_match: n out of range");
    SVR2_CHECK(ws_bytes >= svr2_hist_match_ws_bytes(n), SVR2_E_ARG, "svr2_hist_match: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t slot = ((size_t)n * 4 + 255) & ~(size_t)255;
    char* w = (char*)workspace;
    float* keys_out = (float*)w;
    float* ref_sorted = (float*)(w + slot);
    int* idx_in = (int*)(w + 2 * slot);
    int* idx_out = (int*)(w + 3 * slot);
    void* tmp = w + 4 * slot;
    size_t tbytes = (size_t)ws_bytes - 4 * slot;
    hipLaunchKernelGGL(iota_kernel, dim3(grid_1d(n)), dim3(256), 0, s, idx_in, (long long)n);
    hipError_t e = rocprim::radix_sort_pairs(tmp, tbytes, source, keys_out, (const int*)idx_in, idx_out, (size_t)n, 0, 32, s);
    SVR2_CHECK(e == hipSuccess, SVR2_E_LAUNCH, "svr2_hist_match: radix_sort_pairs failed: %s", hipGetErrorString(e));
    tbytes = (size_t)ws_bytes - 4 * slot;
    e = rocprim::radix_sort_keys(tmp, tbytes, reference, ref_sorted, (size_t)n, 0, 32, s);
    SVR2_CHECK(e == hipSuccess, SVR2_E_LAUNCH, "svr2_hist_match: radix_sort_keys failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(rank_scatter_kernel, dim3(grid_1d(n)), dim3(256), 0, s, (const int*)idx_out, (const float*)ref_sorted,
                       out, (long long)n);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_lab_to_rgb(const float* L_content, const float* L_matched, const float* a, const float* b,
                               float luminance_weight, void* y, int64_t n, void* stream) {
    SVR2_CHECK(L_content && a && b && y && n > 0, SVR2_E_ARG, "svr2_lab_to_rgb: bad argument");
    hipLaunchKernelGGL(lab_to_rgb_kernel, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, L_content, L_matched, a, b,
                       luminance_weight, (bf16*)y, (long long)n);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}
