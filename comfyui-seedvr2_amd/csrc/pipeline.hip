// Pixel-space stages either side of the models (all HBM-bound, one pass each):
//   svr2_resize_pad_normalize : antialiased bicubic resize -> clamp[0,1] -> zero-pad to x16 -> (x-0.5)/0.5
//   svr2_blend_overlap        : Hann / linear cross-fade of the temporal overlap
//   svr2_crop_to_unit         : crop the padded decoder output and map [-1,1] -> [0,1]
#include "common.h"

namespace {

struct ResizeArgs {
    const bf16* in;   // [T, Hin, Win, 3]
    bf16* out;        // [T, Hp, Wp, 3]
    const float* wy; const int32_t* ymin; const int32_t* ycnt;  // [Hout][KY]
    const float* wx; const int32_t* xmin; const int32_t* xcnt;  // [Wout][KX]
    int T, Hin, Win, Hout, Wout, Hp, Wp, KY, KX;
};

__device__ __forceinline__ float round_bf16(float v) { return round_bf16f(v); }

// FINISH = true: resize -> bf16 -> clamp -> pad -> normalise (the model input); false: resize -> bf16 only (the first
// of the two passes the reference runs when max_resolution caps the long edge, side_resize.py:61-73; Hp = Hout, Wp = Wout)
template <bool FINISH>
__global__ __launch_bounds__(256) void resize_pad_norm_kernel(const ResizeArgs a) {
    const long long total = (long long)a.T * a.Hp * a.Wp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.Wp);
        const long long t2 = i / a.Wp;
        const int y = (int)(t2 % a.Hp), t = (int)(t2 / a.Hp);
        float r[3] = {0.f, 0.f, 0.f};  // zero padding is applied before Normalize -> becomes -1
        if (y < a.Hout && x < a.Wout) {
            const int y0 = a.ymin[y], ny = a.ycnt[y], x0 = a.xmin[x], nx = a.xcnt[x];
            const float* wy = a.wy + (long long)y * a.KY;
            const float* wx = a.wx + (long long)x * a.KX;
            const bf16* base = a.in + ((long long)t * a.Hin * a.Win) * 3;
            float acc[3] = {0.f, 0.f, 0.f};
            for (int j = 0; j < ny; ++j) {
                const bf16* row = base + ((long long)(y0 + j) * a.Win + x0) * 3;
                float h[3] = {0.f, 0.f, 0.f};
                for (int k = 0; k < nx; ++k) {
                    const float w = wx[k];
                    h[0] += w * bf2f(row[k * 3 + 0]);
                    h[1] += w * bf2f(row[k * 3 + 1]);
                    h[2] += w * bf2f(row[k * 3 + 2]);
                }
                acc[0] += wy[j] * h[0]; acc[1] += wy[j] * h[1]; acc[2] += wy[j] * h[2];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // the reference rounds to bf16 after the resize, then clamps (generation_utils.py:80)
                float v = round_bf16(acc[c]);
                r[c] = FINISH ? fminf(fmaxf(v, 0.f), 1.f) : v;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) a.out[i * 3 + c] = f2bf(FINISH ? round_bf16(r[c] - 0.5f) / 0.5f : r[c]);
    }
}

__global__ __launch_bounds__(256) void blend_kernel(const bf16* __restrict__ prev, const bf16* __restrict__ cur,
                                                    bf16* __restrict__ out, const float* __restrict__ w_prev,
                                                    long long frame_elems, int overlap) {
    const long long total = frame_elems * overlap;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const float wp = w_prev[i / frame_elems];
        // reference computes in the tensor dtype: prev*w_prev + cur*(1-w_prev), each product rounded
        const float a = round_bf16(bf2f(prev[i]) * wp), b = round_bf16(bf2f(cur[i]) * round_bf16(1.0f - wp));
        out[i] = f2bf(a + b);
    }
}

__global__ __launch_bounds__(256) void crop_to_unit_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int T,
                                                           int Hp, int Wp, int H, int W, int to_unit) {
    const long long total = (long long)T * H * W * 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const long long p = i / 3;
        const int x = (int)(p % W);
        const long long t2 = p / W;
        const int y = (int)(t2 % H), t = (int)(t2 / H);
        float v = bf2f(in[(((long long)t * Hp + y) * Wp + x) * 3 + c]);
        if (to_unit) v = round_bf16(round_bf16(fminf(fmaxf(v, -1.f), 1.f) * 0.5f) + 0.5f);
        out[i] = f2bf(v);
    }
}

inline int grid_1d(long long n) {
    long long g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    return (int)g;
}


// Noise injection of the reference's optional knobs, with torch's rounding: bf16 tensors, python-float scalars applied in
// fp32 and rounded to bf16 after every op (modes 0, 1), or fp32 scalar tensors that promote the whole expression to fp32
// (mode 2).   mode 0 (input_noise_scale, generation_phases.py:409-425):  n = y*0.05;  out = x*a + (x + n)*b
//             mode 1 (aug noise, :679):                                   out = x*a + y*b          (a = 0.1, b = 0.05)
//             mode 2 (schedule.forward with an fp32 timestep, :686-693):  out = bf16(fp32(a*x) + fp32(b*y))
__global__ __launch_bounds__(256) void noise_mix_kernel(const bf16* __restrict__ x, const bf16* __restrict__ y,
                                                        bf16* __restrict__ out, long long n, float a, float b, int mode) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float xv = bf2f(x[i]), yv = bf2f(y[i]);
        float r;
        if (mode == 0) {
            const float nz = round_bf16f(__fmul_rn(yv, 0.05f));
            r = round_bf16f(__fadd_rn(round_bf16f(__fmul_rn(xv, a)), round_bf16f(__fmul_rn(round_bf16f(__fadd_rn(xv, nz)), b))));
        } else if (mode == 1) {
            r = round_bf16f(__fadd_rn(round_bf16f(__fmul_rn(xv, a)), round_bf16f(__fmul_rn(yv, b))));
        } else {
            r = __fadd_rn(__fmul_rn(a, xv), __fmul_rn(b, yv));
        }
        out[i] = f2bf(r);
    }
}

}  // namespace

extern "C" int svr2_resize_pad_normalize(const void* in, void* out, const float* wy, const int32_t* ymin,
                                         const int32_t* ycnt, const float* wx, const int32_t* xmin,
                                         const int32_t* xcnt, int32_t T, int32_t Hin, int32_t Win, int32_t Hout,
                                         int32_t Wout, int32_t Hp, int32_t Wp, int32_t KY, int32_t KX, void* stream) {
    SVR2_CHECK(in && out && wy && ymin && ycnt && wx && xmin && xcnt, SVR2_E_ARG,
               "svr2_resize_pad_normalize: null pointer");
    SVR2_CHECK(T > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && Hp >= Hout && Wp >= Wout && KY > 0 && KX > 0,
               SVR2_E_SHAPE, "svr2_resize_pad_normalize: bad sizes");
    ResizeArgs a;
    a.in = (const bf16*)in; a.out = (bf16*)out; a.wy = wy; a.ymin = ymin; a.ycnt = ycnt; a.wx = wx; a.xmin = xmin;
    a.xcnt = xcnt; a.T = T; a.Hin = Hin; a.Win = Win; a.Hout = Hout; a.Wout = Wout; a.Hp = Hp; a.Wp = Wp; a.KY = KY;
    a.KX = KX;
    hipLaunchKernelGGL(resize_pad_norm_kernel<true>, dim3(grid_1d((long long)T * Hp * Wp)), dim3(256), 0, (hipStream_t)stream,
                       a);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_resize_bicubic_aa(const void* in, void* out, const float* wy, const int32_t* ymin, const int32_t* ycnt,
                                      const float* wx, const int32_t* xmin, const int32_t* xcnt, int32_t T, int32_t Hin,
                                      int32_t Win, int32_t Hout, int32_t Wout, int32_t KY, int32_t KX, void* stream) {
    SVR2_CHECK(in && out && wy && ymin && ycnt && wx && xmin && xcnt, SVR2_E_ARG, "svr2_resize_bicubic_aa: null pointer");
    SVR2_CHECK(T > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && KY > 0 && KX > 0, SVR2_E_SHAPE,
               "svr2_resize_bicubic_aa: bad sizes");
    ResizeArgs a;
    a.in = (const bf16*)in; a.out = (bf16*)out; a.wy = wy; a.ymin = ymin; a.ycnt = ycnt; a.wx = wx; a.xmin = xmin;
    a.xcnt = xcnt; a.T = T; a.Hin = Hin; a.Win = Win; a.Hout = Hout; a.Wout = Wout; a.Hp = Hout; a.Wp = Wout; a.KY = KY;
    a.KX = KX;
    hipLaunchKernelGGL(resize_pad_norm_kernel<false>, dim3(grid_1d((long long)T * Hout * Wout)), dim3(256), 0,
                       (hipStream_t)stream, a);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_blend_overlap(const void* prev_tail, const void* cur_head, void* out, const float* w_prev,
                                  int64_t frame_elems, int32_t overlap, void* stream) {
    SVR2_CHECK(prev_tail && cur_head && out && w_prev, SVR2_E_ARG, "svr2_blend_overlap: null pointer");
    SVR2_CHECK(frame_elems > 0 && overlap > 0, SVR2_E_SHAPE, "svr2_blend_overlap: bad sizes");
    hipLaunchKernelGGL(blend_kernel, dim3(grid_1d(frame_elems * overlap)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)prev_tail, (const bf16*)cur_head, (bf16*)out, w_prev, (long long)frame_elems,
                       overlap);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_crop_to_unit(const void* in, void* out, int32_t T, int32_t Hp, int32_t Wp, int32_t H, int32_t W,
                                 int32_t to_unit, void* stream) {
    SVR2_CHECK(in && out, SVR2_E_ARG, "svr2_crop_to_unit: null pointer");
    SVR2_CHECK(T > 0 && H > 0 && W > 0 && H <= Hp && W <= Wp, SVR2_E_SHAPE, "svr2_crop_to_unit: bad sizes");
    hipLaunchKernelGGL(crop_to_unit_kernel, dim3(grid_1d((long long)T * H * W * 3)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)in, (bf16*)out, T, Hp, Wp, H, W, to_unit);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_noise_mix(const void* x, const void* y, void* out, int64_t n, float a, float b, int32_t mode,
                              void* stream) {
    SVR2_CHECK(x && y && out, SVR2_E_ARG, "svr2_noise_mix: null pointer");
    SVR2_CHECK(n > 0 && mode >= 0 && mode <= 2, SVR2_E_SHAPE, "svr2_noise_mix: bad size or mode");
    hipLaunchKernelGGL(noise_mix_kernel, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (const bf16*)y,
                       (bf16*)out, (long long)n, a, b, mode);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}
