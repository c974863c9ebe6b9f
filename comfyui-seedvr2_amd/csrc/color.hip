// Phase-4 colour correction on the device: wavelet reconstruction, AdaIN and the CIELAB
// histogram transfer of the reference's src/utils/color_fix.py, on channels-last bf16 frames [F, H, W, 3] in [-1, 1].
// All of it is HBM-bound elementwise / stencil / reduction work plus six full sorts per batch (rocPRIM device radix
// sort: the histogram match assigns the i-th smallest reference value to the i-th smallest source pixel).
// The reference runs these functions in the compute dtype (bf16): every op rounds to bf16; the kernels round at the
// same places (stated per kernel) and accumulate in fp32 in between.
#include "common.h"

#include <rocprim/rocprim.hpp>

namespace {

__device__ __forceinline__ float rbf(float v) { return round_bf16f(v); }   // one bf16 rounding (never folded away)

// ----------------------------------------------------------------------------- wavelet level
// low = bf16(sum_{3x3} k[dy][dx] * img[clamp(y + (dy-1)*r), clamp(x + (dx-1)*r)])   (replicate padding, dilation r)
// high = bf16(bf16(high + img) - low)                                                [ref: color_fix.py:122-184]
template <typename T> __device__ __forceinline__ float ldv(const T* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ float rnd(float v);          // rounding of the working dtype
template <> __device__ __forceinline__ float rnd<bf16>(float v) { return rbf(v); }
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }

template <typename T>
__global__ __launch_bounds__(256) void wavelet_level_kernel(const T* __restrict__ img, T* __restrict__ low,
                                                            T* __restrict__ high, int F, int H, int W, int r,
                                                            int first) {
    const long long n = (long long)F * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long long t2 = i / W;
        const int y = (int)(t2 % H);
        const long long f = t2 / H;
        const T* fr = img + f * H * W * 3;
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
                const T* p = fr + ((long long)ys[dy] * W + xs[dx]) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] += k * ldv(p + c);
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float lo = rnd<T>(acc[c]);
            low[i * 3 + c] = (T)lo;
            if (high) {
                const float h0 = first ? 0.f : ldv(high + i * 3 + c);
                high[i * 3 + c] = (T)(rnd<T>(h0 + ldv(img + i * 3 + c)) - lo);
            }
        }
    }
}

// out = clamp(bf16(high + low), -1, 1)                                              [ref: color_fix.py:241-246]
template <typename T>
__global__ __launch_bounds__(256) void wavelet_merge_kernel(const T* __restrict__ high, const T* __restrict__ low,
                                                            T* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = (T)fminf(fmaxf(rnd<T>(ldv(high + i) + ldv(low + i)), -1.f), 1.f);
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

// ----------------------------------------------------------------------------- HSV saturation match
__device__ __forceinline__ float unit01(float v) { return fminf(fmaxf((v + 1.f) * 0.5f, 0.f), 1.f); }

__device__ __forceinline__ float sat_of(float r, float g, float b) {
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
    return mx > 1e-10f ? (mx - mn) / fmaxf(mx, 1e-10f) : 0.f;
}

// x bf16 [n,3] in [-1,1] -> planar fp32 hsv[3][n] of clamp((x+1)/2, 0, 1)          [ref: color_fix.py:558-563,606-641]
// (equal maxima: the reference's masked assignments run r, g, b in turn, so b wins over g wins over r)
__global__ __launch_bounds__(256) void rgb_to_hsv_kernel(const bf16* __restrict__ x, float* __restrict__ hsv, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float r = unit01(bf2f(x[i * 3])), g = unit01(bf2f(x[i * 3 + 1])), b = unit01(bf2f(x[i * 3 + 2]));
        const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b)), rng = mx - mn;
        float h = 0.f;
        if (rng > 1e-10f) {
            if (mx == r) { const float q = (g - b) / rng; h = q - 6.f * floorf(q / 6.f); }
            if (mx == g) h = (b - r) / rng + 2.f;
            if (mx == b) h = (r - g) / rng + 4.f;
        }
        hsv[i] = h / 6.f;
        hsv[n + i] = mx > 1e-10f ? rng / fmaxf(mx, 1e-10f) : 0.f;
        hsv[2 * n + i] = mx;
    }
}

struct HueBins { float thr[13]; float wrap; int pass; };   // thr[k] = fp32(k/12); pass 0: bins 0..10 (+wrap into 0), 1: bin 11

__device__ __forceinline__ int hue_bin(float h, const HueBins& hb) {
    if (hb.pass == 0) {
        if (h >= hb.wrap) return 0;
#pragma unroll
        for (int k = 0; k < 11; ++k)
            if (h >= hb.thr[k] && h < hb.thr[k + 1]) return k;
        return 15;
    }
    return (h >= hb.thr[11] && h < hb.thr[12]) ? 11 : 15;
}

// key = (hue bin << 32) | bits(s)  (s >= 0: its bit pattern orders like its value); counts per bin
__global__ __launch_bounds__(256) void hue_keys_kernel(const float* __restrict__ h, const float* __restrict__ sv,
                                                       unsigned long long* __restrict__ keys, int* __restrict__ idx,
                                                       unsigned* __restrict__ counts, long long n, HueBins hb) {
    __shared__ unsigned lc[16];
    if (threadIdx.x < 16) lc[threadIdx.x] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int b = hue_bin(h[i], hb);
        keys[i] = ((unsigned long long)b << 32) | __float_as_uint(sv[i]);
        if (idx) idx[i] = (int)i;
        atomicAdd(&lc[b], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 16 && lc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], lc[threadIdx.x]);
}

// rank r of a content pixel inside its bin -> reference quantile: r itself for equal counts, else
// long(linspace(0, 1, n_src)[r] * (n_ref - 1)) with torch's fp32 linspace (two-sided, step = 1/(n-1))
__global__ __launch_bounds__(256) void hue_map_kernel(const unsigned long long* __restrict__ ckeys, const int* __restrict__ cidx,
                                                      const unsigned long long* __restrict__ skeys,
                                                      const unsigned* __restrict__ ccnt, const unsigned* __restrict__ scnt,
                                                      float* __restrict__ out, long long n, int min_pixels) {
    __shared__ long long cstart[16], sstart[16];
    if (threadIdx.x == 0) {
        long long a = 0, b = 0;
        for (int k = 0; k < 16; ++k) { cstart[k] = a; sstart[k] = b; a += ccnt[k]; b += scnt[k]; }
    }
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(ckeys[i] >> 32);
        if (b > 11) continue;
        const long long ns = ccnt[b], nr = scnt[b];
        if (ns <= min_pixels || nr <= min_pixels) continue;
        const long long r = i - cstart[b];
        long long j = r;
        if (ns != nr) {
            const float step = __fdiv_rn(1.f, (float)(ns - 1));
            // torch's linspace: start + step*i below the midpoint, end - step*(n-1-i) above it, the latter evaluated as ONE
            // fused multiply-add by ATen's vectorised CPU kernel (checked bit for bit by test_hue_sat_match_quantiles_exact)
            const float q = r < ns / 2 ? __fmul_rn(step, (float)r) : __builtin_fmaf(-step, (float)(ns - r - 1), 1.f);
            j = (long long)__fmul_rn(q, (float)(nr - 1));
            j = j < 0 ? 0 : (j > nr - 1 ? nr - 1 : j);
        }
        out[cidx[i]] = __uint_as_float((unsigned)(skeys[sstart[b] + j] & 0xffffffffull));
    }
}

// (h, s, v) -> rgb, clamp, *2-1, stored as T                                          [ref: color_fix.py:583-596,644-687]
template <typename T>
__global__ __launch_bounds__(256) void hsv_to_rgb_kernel(const float* __restrict__ h, const float* __restrict__ s_,
                                                         const float* __restrict__ v_, T* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float h6 = h[i] * 6.f, s = s_[i], v = v_[i];
        const float fl = floorf(h6), f = h6 - fl;
        const int k = ((int)fl) % 6;
        const float p = v * (1.f - s), q = v * (1.f - s * f), t = v * (1.f - s * (1.f - f));
        float r, g, b;
        switch (k) {
            case 0: r = v; g = t; b = p; break;
            case 1: r = q; g = v; b = p; break;
            case 2: r = p; g = v; b = t; break;
            case 3: r = p; g = q; b = v; break;
            case 4: r = t; g = p; b = v; break;
            default: r = v; g = p; b = q; break;
        }
        const float c[3] = {r, g, b};
#pragma unroll
        for (int e = 0; e < 3; ++e) y[i * 3 + e] = (T)(fminf(fmaxf(c[e], 0.f), 1.f) * 2.f - 1.f);
    }
}

// wavelet_adaptive blend: w = sigmoid(5((sat(c) - sat(s)) - 0.15)) * [sat(wav) - sat(s) > 0.075];
// out = bf16(wav (1 - w) + hsv w)                                                     [ref: color_fix.py:805-833,846-860]
__global__ __launch_bounds__(256) void sat_blend_kernel(const bf16* __restrict__ c, const bf16* __restrict__ s_,
                                                        const float* __restrict__ wav, const float* __restrict__ hsv,
                                                        bf16* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float cs = sat_of(unit01(bf2f(c[i * 3])), unit01(bf2f(c[i * 3 + 1])), unit01(bf2f(c[i * 3 + 2])));
        const float ss = sat_of(unit01(bf2f(s_[i * 3])), unit01(bf2f(s_[i * 3 + 1])), unit01(bf2f(s_[i * 3 + 2])));
        const float ws = sat_of(unit01(wav[i * 3]), unit01(wav[i * 3 + 1]), unit01(wav[i * 3 + 2]));
        float w = 1.f / (1.f + expf(-(5.f * ((cs - ss) - 0.15f))));
        w = (ws - ss) > 0.075f ? w : 0.f;
        w = fminf(fmaxf(w, 0.f), 1.f);
#pragma unroll
        for (int e = 0; e < 3; ++e) out[i * 3 + e] = f2bf(wav[i * 3 + e] * (1.f - w) + hsv[i * 3 + e] * w);
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
                                  int32_t first, int32_t is_f32, void* stream) {
    SVR2_CHECK(img && low, SVR2_E_ARG, "svr2_wavelet_level: null pointer");
    SVR2_CHECK(F > 0 && H > 0 && W > 0 && radius >= 1, SVR2_E_SHAPE, "svr2_wavelet_level: bad shape");
    const long long n = (long long)F * H * W;
    if (is_f32)
        hipLaunchKernelGGL(wavelet_level_kernel<float>, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, (const float*)img,
                           (float*)low, (float*)high, F, H, W, radius, first);
    else
        hipLaunchKernelGGL(wavelet_level_kernel<bf16>, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, (const bf16*)img,
                           (bf16*)low, (bf16*)high, F, H, W, radius, first);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_wavelet_merge(const void* high, const void* low, void* out, int64_t n, int32_t is_f32, void* stream) {
    SVR2_CHECK(high && low && out && n > 0, SVR2_E_ARG, "svr2_wavelet_merge: bad argument");
    if (is_f32)
        hipLaunchKernelGGL(wavelet_merge_kernel<float>, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, (const float*)high,
                           (const float*)low, (float*)out, (long long)n);
    else
        hipLaunchKernelGGL(wavelet_merge_kernel<bf16>, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, (const bf16*)high,
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

extern "C" int svr2_rgb_to_hsv(const void* x, float* hsv, int64_t n, void* stream) {
    SVR2_CHECK(x && hsv && n > 0, SVR2_E_ARG, "svr2_rgb_to_hsv: bad argument");
    hipLaunchKernelGGL(rgb_to_hsv_kernel, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, hsv,
                       (long long)n);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

namespace {
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
inline size_t hue_tmp_bytes(size_t n) {
    size_t t1 = 0, t2 = 0;
    (void)rocprim::radix_sort_pairs(nullptr, t1, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                    (const int*)nullptr, (int*)nullptr, n, 0, 36);
    (void)rocprim::radix_sort_keys(nullptr, t2, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, n, 0, 36);
    return up256(t1 > t2 ? t1 : t2);
}
}  // namespace

/* workspace of svr2_hue_sat_match: keys in/out for both images, indices in/out, counters, rocPRIM scratch */
extern "C" int64_t svr2_hue_sat_match_ws_bytes(int64_t n) {
    if (n <= 0 || n >= (1ll << 31)) return 0;
    return (int64_t)(4 * up256((size_t)n * 8) + 2 * up256((size_t)n * 4) + 256 + hue_tmp_bytes((size_t)n));
}

/* out = content saturation with, per 30-degree hue bin holding more than min_pixels pixels in both images, the style
 * saturation distribution moved onto the content ranks; bin 0 also takes the hues >= 11/12, which bin 11 then
 * overwrites [ref: color_fix.py:690-757].  hsv_c / hsv_s: planar fp32 [3][n] from svr2_rgb_to_hsv; thresholds:
 * 14 floats = fp32(k/12) for k = 0..12 and fp32(1 - 1/12). */
extern "C" int svr2_hue_sat_match(const float* hsv_c, const float* hsv_s, float* out, int64_t n, const float* thresholds,
                                  int32_t min_pixels, void* workspace, int64_t ws_bytes, void* stream) {
    SVR2_CHECK(hsv_c && hsv_s && out && thresholds && workspace, SVR2_E_ARG, "svr2_hue_sat_match: null pointer");
    SVR2_CHECK(n > 0 && n < (1ll << 31), SVR2_E_SHAPE, "svr2_hue_sat_match: n out of range");
    SVR2_CHECK(ws_bytes >= svr2_hue_sat_match_ws_bytes(n), SVR2_E_ARG, "svr2_hue_sat_match: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t k8 = up256((size_t)n * 8), k4 = up256((size_t)n * 4);
    char* w = (char*)workspace;
    unsigned long long* ck_in = (unsigned long long*)w;
    unsigned long long* ck_out = (unsigned long long*)(w + k8);
    unsigned long long* sk_in = (unsigned long long*)(w + 2 * k8);
    unsigned long long* sk_out = (unsigned long long*)(w + 3 * k8);
    int* ix_in = (int*)(w + 4 * k8);
    int* ix_out = (int*)(w + 4 * k8 + k4);
    unsigned* counts = (unsigned*)(w + 4 * k8 + 2 * k4);          // [2][16]
    void* tmp = w + 4 * k8 + 2 * k4 + 256;
    const size_t tcap = (size_t)ws_bytes - (4 * k8 + 2 * k4 + 256);
    SVR2_CHECK(hipMemcpyAsync(out, hsv_c + n, (size_t)n * 4, hipMemcpyDeviceToDevice, s) == hipSuccess, SVR2_E_LAUNCH,
               "svr2_hue_sat_match: copy failed");
    HueBins hb;
    for (int k = 0; k < 13; ++k) hb.thr[k] = thresholds[k];
    hb.wrap = thresholds[13];
    for (int pass = 0; pass < 2; ++pass) {
        hb.pass = pass;
        SVR2_CHECK(hipMemsetAsync(counts, 0, 128, s) == hipSuccess, SVR2_E_LAUNCH, "svr2_hue_sat_match: memset failed");
        hipLaunchKernelGGL(hue_keys_kernel, dim3(grid_1d(n)), dim3(256), 0, s, hsv_c, hsv_c + n, ck_in, ix_in, counts,
                           (long long)n, hb);
        hipLaunchKernelGGL(hue_keys_kernel, dim3(grid_1d(n)), dim3(256), 0, s, hsv_s, hsv_s + n, sk_in, (int*)nullptr,
                           counts + 16, (long long)n, hb);
        size_t tb = tcap;
        hipError_t e = rocprim::radix_sort_pairs(tmp, tb, (const unsigned long long*)ck_in, ck_out, (const int*)ix_in, ix_out,
                                                 (size_t)n, 0, 36, s);
        SVR2_CHECK(e == hipSuccess, SVR2_E_LAUNCH, "svr2_hue_sat_match: radix_sort_pairs failed: %s", hipGetErrorString(e));
        tb = tcap;
        e = rocprim::radix_sort_keys(tmp, tb, (const unsigned long long*)sk_in, sk_out, (size_t)n, 0, 36, s);
        SVR2_CHECK(e == hipSuccess, SVR2_E_LAUNCH, "svr2_hue_sat_match: radix_sort_keys failed: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(hue_map_kernel, dim3(grid_1d(n)), dim3(256), 0, s, (const unsigned long long*)ck_out,
                           (const int*)ix_out, (const unsigned long long*)sk_out, (const unsigned*)counts,
                           (const unsigned*)(counts + 16), out, (long long)n, min_pixels);
    }
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_hsv_to_rgb(const float* h, const float* s_, const float* v, void* y, int64_t n, int32_t out_f32,
                               void* stream) {
    SVR2_CHECK(h && s_ && v && y && n > 0, SVR2_E_ARG, "svr2_hsv_to_rgb: bad argument");
    if (out_f32)
        hipLaunchKernelGGL(hsv_to_rgb_kernel<float>, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, h, s_, v, (float*)y,
                           (long long)n);
    else
        hipLaunchKernelGGL(hsv_to_rgb_kernel<bf16>, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, h, s_, v, (bf16*)y,
                           (long long)n);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_sat_blend(const void* content, const void* style, const float* wav, const float* hsv, void* out, int64_t n,
                              void* stream) {
    SVR2_CHECK(content && style && wav && hsv && out && n > 0, SVR2_E_ARG, "svr2_sat_blend: bad argument");
    hipLaunchKernelGGL(sat_blend_kernel, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, (const bf16*)content,
                       (const bf16*)style, wav, hsv, (bf16*)out, (long long)n);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}
