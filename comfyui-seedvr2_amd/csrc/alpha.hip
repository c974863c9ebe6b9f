// Edge-guided alpha upscaling for RGBA inputs [ref: src/core/alpha_upscaling.py:124-430].
// The reference runs this on the host through OpenCV (uint8 grey, 3x3 Sobel in float64) and fp32 torch ops; here every
// stage is a device pass over the upscaled frames (channels-last RGB bf16 in, planar fp32 [T,H,W] in between):
//   svr2_alpha_stats   : |{alpha < 0.1}|, |{alpha > 0.9}| and min(rgb) -- the two data-dependent branches of the reference
//   svr2_alpha_guide   : grey guide = mean_c((x+1)/2) and the uint8 grey of OpenCV's RGB2GRAY ((4899 R + 9617 G + 1868 B + 8192) >> 14)
//   svr2_alpha_sobel   : 3x3 Sobel magnitude (BORDER_REFLECT_101) in float64 + per-frame maximum
//   svr2_alpha_edges   : uint8(mag / max * 255) / 255
//   svr2_resize_plane  : antialiased bicubic resize of a fp32 plane (same tap tables as svr2_resize_pad_normalize)
//   svr2_guided_ab / svr2_guided_out : the guided filter (box means with zero padding, divisor (2r+1)^2)
//   svr2_alpha_refine  : 3x3 max-pool transition zone + the binary-mask snapping rules, or just the clamp
// Products that torch materialises as tensors are rounded separately (__fmul_rn: no fma contraction), so the fp32
// results follow the reference op by op.
#include "common.h"

namespace {

inline unsigned grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}

__global__ __launch_bounds__(256) void alpha_stats_kernel(const bf16* __restrict__ alpha, long long na,
                                                          const bf16* __restrict__ rgb, long long nr,
                                                          int* __restrict__ counts, float* __restrict__ part) {
    __shared__ float smin[4];
    __shared__ int sc[2];
    if (threadIdx.x < 2) sc[threadIdx.x] = 0;
    __syncthreads();
    int c0 = 0, c1 = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += (long long)gridDim.x * blockDim.x) {
        const float a = bf2f(alpha[i]);
        c0 += a < 0.1f;
        c1 += a > 0.9f;
    }
    float m = INFINITY;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += (long long)gridDim.x * blockDim.x)
        m = fminf(m, bf2f(rgb[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m = fminf(m, __shfl_xor(m, o, 64));
        c0 += __shfl_xor(c0, o, 64);
        c1 += __shfl_xor(c1, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = m;
        atomicAdd(&sc[0], c0);
        atomicAdd(&sc[1], c1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        atomicAdd(&counts[0], sc[0]);
        atomicAdd(&counts[1], sc[1]);
    }
}

__global__ __launch_bounds__(256) void alpha_min_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ float s[4];
    float m = INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) m = fminf(m, part[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = fminf(fminf(s[0], s[1]), fminf(s[2], s[3]));
}

__global__ __launch_bounds__(256) void alpha_guide_kernel(const bf16* __restrict__ rgb, uint8_t* __restrict__ gray,
                                                          float* __restrict__ guide, long long npix, int norm_guide,
                                                          int norm_edges) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        float c[3];
        int u[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = bf2f(rgb[i * 3 + k]);
            if (norm_guide) v = __fadd_rn(v, 1.0f) / 2.0f;                 // rgb_normalized = (x + 1) / 2   (:346-348)
            c[k] = v;
            float e = v;
            if (norm_edges) e = __fadd_rn(e, 1.0f) / 2.0f;                 // detect_edges_batch normalises again if min < 0 (:147-149)
            e = fminf(fmaxf(__fmul_rn(e, 255.0f), 0.f), 255.f);            // (x * 255).clip(0, 255).astype(uint8)
            u[k] = (int)e;
        }
        guide[i] = __fadd_rn(__fadd_rn(c[0], c[1]), c[2]) / 3.0f;          // guide.mean(dim=1)   (:217-218)
        gray[i] = (uint8_t)((4899 * u[0] + 9617 * u[1] + 1868 * u[2] + (1 << 13)) >> 14);   // cv2.COLOR_RGB2GRAY
    }
}

__device__ __forceinline__ int refl(int p, int n) {   // BORDER_REFLECT_101 for a one-pixel border
    if (n == 1) return 0;
    return p < 0 ? 1 : (p >= n ? n - 2 : p);
}

__global__ __launch_bounds__(256) void alpha_sobel_kernel(const uint8_t* __restrict__ gray, double* __restrict__ mag,
                                                          unsigned long long* __restrict__ fmax_bits, int T, int H, int W) {
    const long long HW = (long long)H * W, total = HW * T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long long r = i / W;
        const int y = (int)(r % H), t = (int)(r / H);
        const uint8_t* g = gray + t * HW;
        int p[3][3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) p[dy][dx] = g[(long long)refl(y + dy - 1, H) * W + refl(x + dx - 1, W)];
        const int sx = (p[0][2] - p[0][0]) + 2 * (p[1][2] - p[1][0]) + (p[2][2] - p[2][0]);
        const int sy = (p[2][0] - p[0][0]) + 2 * (p[2][1] - p[0][1]) + (p[2][2] - p[0][2]);
        const double m = sqrt((double)(sx * sx + sy * sy));
        mag[i] = m;
        if (m > 0.0) atomicMax(&fmax_bits[t], (unsigned long long)__double_as_longlong(m));   // non-negative doubles order like their bits
    }
}

__global__ __launch_bounds__(256) void alpha_edges_kernel(const double* __restrict__ mag,
                                                          const unsigned long long* __restrict__ fmax_bits,
                                                          float* __restrict__ edges, int T, long long HW) {
    const long long total = HW * T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const double m = __longlong_as_double((long long)fmax_bits[i / HW]);
        // (edge / edge.max() * 255).astype(np.uint8), then .float() / 255.0   (:167-177); a flat frame gives 0
        const int q = m > 0.0 ? (int)(mag[i] / m * 255.0) : 0;
        edges[i] = (float)q / 255.0f;
    }
}

struct PlaneResizeArgs {
    const float* in; float* out;
    const float* wy; const int32_t* ymin; const int32_t* ycnt;
    const float* wx; const int32_t* xmin; const int32_t* xcnt;
    int T, Hin, Win, Hout, Wout, KY, KX, clamp01;
};

__global__ __launch_bounds__(256) void resize_plane_kernel(const PlaneResizeArgs a) {
    const long long total = (long long)a.T * a.Hout * a.Wout;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.Wout);
        const long long r = i / a.Wout;
        const int y = (int)(r % a.Hout), t = (int)(r / a.Hout);
        const int y0 = a.ymin[y], ny = a.ycnt[y], x0 = a.xmin[x], nx = a.xcnt[x];
        const float* wy = a.wy + (long long)y * a.KY;
        const float* wx = a.wx + (long long)x * a.KX;
        const float* base = a.in + (long long)t * a.Hin * a.Win;
        float acc = 0.f;
        for (int j = 0; j < ny; ++j) {          // width pass into an fp32 intermediate, then the height pass (ATen's order)
            const float* row = base + (long long)(y0 + j) * a.Win + x0;
            float h = 0.f;
            for (int k = 0; k < nx; ++k) h = __fadd_rn(h, __fmul_rn(wx[k], row[k]));
            acc = __fadd_rn(acc, __fmul_rn(wy[j], h));
        }
        a.out[i] = a.clamp01 ? fminf(fmaxf(acc, 0.f), 1.f) : acc;
    }
}

template <int MODE>   // 0: guide & src -> a, b ; 1: a, b, guide -> output
__global__ __launch_bounds__(256) void guided_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                     const float* __restrict__ guide, float* __restrict__ o0,
                                                     float* __restrict__ o1, int T, int H, int W, int r, float eps) {
    const long long HW = (long long)H * W, total = HW * T;
    const float div = (float)((2 * r + 1) * (2 * r + 1));      // avg_pool2d(padding=r): zeros count, divisor is the full window
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long long rr = i / W;
        const int y = (int)(rr % H), t = (int)(rr / H);
        const long long fb = t * HW;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int yy = y - r; yy <= y + r; ++yy) {
            if (yy < 0 || yy >= H) continue;
            for (int xx = x - r; xx <= x + r; ++xx) {
                if (xx < 0 || xx >= W) continue;
                const long long j = fb + (long long)yy * W + xx;
                const float u = p0[j], v = p1[j];
                s0 = __fadd_rn(s0, u);
                s1 = __fadd_rn(s1, v);
                if (MODE == 0) {
                    s2 = __fadd_rn(s2, __fmul_rn(u, u));
                    s3 = __fadd_rn(s3, __fmul_rn(u, v));
                }
            }
        }
        if (MODE == 0) {
            const float mg = s0 / div, ms = s1 / div, cg = s2 / div, cgs = s3 / div;
            const float var = __fsub_rn(cg, __fmul_rn(mg, mg));
            const float cov = __fsub_rn(cgs, __fmul_rn(mg, ms));
            const float av = cov / __fadd_rn(var, eps);
            o0[i] = av;
            o1[i] = __fsub_rn(ms, __fmul_rn(av, mg));
        } else {
            o0[i] = __fadd_rn(__fmul_rn(s0 / div, guide[i]), s1 / div);      // mean_a * guide + mean_b
        }
    }
}

__global__ __launch_bounds__(256) void alpha_refine_kernel(const float* __restrict__ refined, const float* __restrict__ edges,
                                                           float* __restrict__ out, int T, int H, int W, int binary) {
    const long long HW = (long long)H * W, total = HW * T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float a = refined[i];
        if (binary) {
            const int x = (int)(i % W);
            const long long r = i / W;
            const int y = (int)(r % H), t = (int)(r / H);
            const float* e = edges + t * HW;
            float zone = -INFINITY;                                               // max_pool2d(3, 1, 1) pads with -inf
            for (int yy = max(y - 1, 0); yy <= min(y + 1, H - 1); ++yy)
                for (int xx = max(x - 1, 0); xx <= min(x + 1, W - 1); ++xx) zone = fmaxf(zone, e[(long long)yy * W + xx]);
            const float ed = e[(long long)y * W + x];
            const float bin = a > 0.5f ? 1.f : 0.f;
            const float enh = 1.0f / (1.0f + expf(-__fmul_rn(__fsub_rn(a, 0.5f), 12.0f)));     // sigmoid((a - 0.5) * 12)
            const float st = fminf(fmaxf(ed / 0.25f, 0.f), 1.f);
            const float in_edges = __fadd_rn(__fmul_rn(a, __fsub_rn(1.0f, st)), __fmul_rn(enh, st));
            float v = zone < 0.05f ? bin : in_edges;
            if (zone < 0.03f) v = v > 0.5f ? 1.f : 0.f;
            if (v > 0.3f && v < 0.7f && !(ed > 0.15f)) v = v > 0.5f ? 1.f : 0.f;
            a = v;
        }
        out[i] = fminf(fmaxf(a, 0.f), 1.f);
    }
}

}  // namespace

extern "C" int svr2_alpha_stats(const void* alpha, int64_t n_alpha, const void* rgb, int64_t n_rgb, int32_t* counts,
                                float* rgb_min, void* workspace, void* stream) {
    SVR2_CHECK(alpha && rgb && counts && rgb_min && workspace, SVR2_E_ARG, "svr2_alpha_stats: null pointer");
    SVR2_CHECK(n_alpha > 0 && n_rgb > 0, SVR2_E_SHAPE, "svr2_alpha_stats: empty input");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), s) != hipSuccess) {
        svr2_set_error("svr2_alpha_stats: memset failed");
        return SVR2_E_LAUNCH;
    }
    const unsigned g = grid_for(n_alpha > n_rgb ? n_alpha : n_rgb);
    hipLaunchKernelGGL(alpha_stats_kernel, dim3(g), dim3(256), 0, s, (const bf16*)alpha, (long long)n_alpha, (const bf16*)rgb,
                       (long long)n_rgb, (int*)counts, (float*)workspace);
    hipLaunchKernelGGL(alpha_min_final_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, (int)g, rgb_min);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_alpha_guide(const void* rgb, void* gray, float* guide, int64_t npix, int32_t norm_guide,
                                int32_t norm_edges, void* stream) {
    SVR2_CHECK(rgb && gray && guide, SVR2_E_ARG, "svr2_alpha_guide: null pointer");
    SVR2_CHECK(npix > 0, SVR2_E_SHAPE, "svr2_alpha_guide: empty input");
    hipLaunchKernelGGL(alpha_guide_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, (const bf16*)rgb,
                       (uint8_t*)gray, guide, (long long)npix, norm_guide, norm_edges);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_alpha_sobel(const void* gray, double* mag, void* frame_max, int32_t T, int32_t H, int32_t W,
                                void* stream) {
    SVR2_CHECK(gray && mag && frame_max, SVR2_E_ARG, "svr2_alpha_sobel: null pointer");
    SVR2_CHECK(T > 0 && H > 0 && W > 0, SVR2_E_SHAPE, "svr2_alpha_sobel: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(frame_max, 0, (size_t)T * 8, s) != hipSuccess) {
        svr2_set_error("svr2_alpha_sobel: memset failed");
        return SVR2_E_LAUNCH;
    }
    hipLaunchKernelGGL(alpha_sobel_kernel, dim3(grid_for((long long)T * H * W)), dim3(256), 0, s, (const uint8_t*)gray, mag,
                       (unsigned long long*)frame_max, T, H, W);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_alpha_edges(const double* mag, const void* frame_max, float* edges, int32_t T, int64_t HW,
                                void* stream) {
    SVR2_CHECK(mag && frame_max && edges, SVR2_E_ARG, "svr2_alpha_edges: null pointer");
    SVR2_CHECK(T > 0 && HW > 0, SVR2_E_SHAPE, "svr2_alpha_edges: bad sizes");
    hipLaunchKernelGGL(alpha_edges_kernel, dim3(grid_for(T * HW)), dim3(256), 0, (hipStream_t)stream, mag,
                       (const unsigned long long*)frame_max, edges, T, (long long)HW);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_resize_plane(const float* in, float* out, const float* wy, const int32_t* ymin, const int32_t* ycnt,
                                 const float* wx, const int32_t* xmin, const int32_t* xcnt, int32_t T, int32_t Hin,
                                 int32_t Win, int32_t Hout, int32_t Wout, int32_t KY, int32_t KX, int32_t clamp01,
                                 void* stream) {
    SVR2_CHECK(in && out && wy && ymin && ycnt && wx && xmin && xcnt, SVR2_E_ARG, "svr2_resize_plane: null pointer");
    SVR2_CHECK(T > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && KY > 0 && KX > 0, SVR2_E_SHAPE,
               "svr2_resize_plane: bad sizes");
    PlaneResizeArgs a{in, out, wy, ymin, ycnt, wx, xmin, xcnt, T, Hin, Win, Hout, Wout, KY, KX, clamp01};
    hipLaunchKernelGGL(resize_plane_kernel, dim3(grid_for((long long)T * Hout * Wout)), dim3(256), 0, (hipStream_t)stream, a);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_guided_ab(const float* guide, const float* src, float* a, float* b, int32_t T, int32_t H, int32_t W,
                              int32_t radius, float eps, void* stream) {
    SVR2_CHECK(guide && src && a && b, SVR2_E_ARG, "svr2_guided_ab: null pointer");
    SVR2_CHECK(T > 0 && H > 0 && W > 0 && radius >= 0 && radius <= 16, SVR2_E_SHAPE, "svr2_guided_ab: bad sizes");
    hipLaunchKernelGGL(guided_kernel<0>, dim3(grid_for((long long)T * H * W)), dim3(256), 0, (hipStream_t)stream, guide, src,
                       (const float*)nullptr, a, b, T, H, W, radius, eps);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_guided_out(const float* a, const float* b, const float* guide, float* out, int32_t T, int32_t H,
                               int32_t W, int32_t radius, void* stream) {
    SVR2_CHECK(a && b && guide && out, SVR2_E_ARG, "svr2_guided_out: null pointer");
    SVR2_CHECK(T > 0 && H > 0 && W > 0 && radius >= 0 && radius <= 16, SVR2_E_SHAPE, "svr2_guided_out: bad sizes");
    hipLaunchKernelGGL(guided_kernel<1>, dim3(grid_for((long long)T * H * W)), dim3(256), 0, (hipStream_t)stream, a, b, guide,
                       out, (float*)nullptr, T, H, W, radius, 0.f);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_alpha_refine(const float* refined, const float* edges, float* out, int32_t T, int32_t H, int32_t W,
                                 int32_t binary, void* stream) {
    SVR2_CHECK(refined && out && (edges || !binary), SVR2_E_ARG, "svr2_alpha_refine: null pointer");
    SVR2_CHECK(T > 0 && H > 0 && W > 0, SVR2_E_SHAPE, "svr2_alpha_refine: bad sizes");
    hipLaunchKernelGGL(alpha_refine_kernel, dim3(grid_for((long long)T * H * W)), dim3(256), 0, (hipStream_t)stream, refined,
                       edges, out, T, H, W, binary);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}
