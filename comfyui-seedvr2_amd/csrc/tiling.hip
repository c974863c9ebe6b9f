// Spatial VAE tiling: the blend of overlapping tile results
// [ref: src/models/video_vae_v3/modules/attn_video_vae.py:1403-1452 (tiled_encode), :1565-1614 (tiled_decode)].
// The reference multiplies a tile by its two 1-D fade weights, adds it into the result, accumulates the weights and
// divides at the end, all with in-place torch ops on tensors of the VAE dtype (bf16): every product, every sum and the
// quotient round to bf16.  The kernels keep exactly those roundings (channels-last [T,H,W,C]):
//   svr2_tile_accumulate : result += bf16(bf16(tile * wh[y]) * ww[x]);  count = bf16(count + wh[y] * ww[x])
//   svr2_tile_normalize  : result  = bf16(result / max(count, bf16(1e-6)))
// HBM-bound single passes over the tile / the frame.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void tile_accumulate_kernel(const bf16* __restrict__ tile, int T, int th, int tw, int Ct,
                                                              bf16* __restrict__ result, bf16* __restrict__ count, int H,
                                                              int W, int C, int y0, int x0, int eh, int ew,
                                                              const bf16* __restrict__ wh, const bf16* __restrict__ ww) {
    const long long total = (long long)T * eh * ew;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % ew);
        const long long r = i / ew;
        const int y = (int)(r % eh), t = (int)(r / eh);
        const float fh = bf2f(wh[y]), fw = bf2f(ww[x]);
        const bf16* src = tile + (((long long)t * th + y) * tw + x) * Ct;
        bf16* dst = result + (((long long)t * H + (y0 + y)) * W + (x0 + x)) * C;
        for (int c = 0; c < C; ++c) {
            const float v = round_bf16f(round_bf16f(bf2f(src[c]) * fh) * fw);      // tile.mul_(wh).mul_(ww)
            dst[c] = f2bf(round_bf16f(bf2f(dst[c]) + v));                         // result[...] += tile
        }
        if (t == 0) {
            bf16* cp = count + (long long)(y0 + y) * W + (x0 + x);
            *cp = f2bf(round_bf16f(bf2f(*cp) + fh * fw));                                    // count.addcmul_(wh, ww): one rounding
        }
    }
}

__global__ __launch_bounds__(256) void tile_normalize_kernel(bf16* __restrict__ result, const bf16* __restrict__ count,
                                                             int T, long long HW, int C) {
    const float floor_ = bf2f(f2bf(1e-6f));
    const long long total = (long long)T * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const float d = fmaxf(bf2f(count[i % HW]), floor_);
        bf16* p = result + i * C;
        for (int c = 0; c < C; ++c) p[c] = f2bf(round_bf16f(bf2f(p[c]) / d));
    }
}

inline unsigned grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace

extern "C" int svr2_tile_accumulate(const void* tile, int32_t T, int32_t th, int32_t tw, int32_t Ct, void* result,
                                    void* count, int32_t H, int32_t W, int32_t C, int32_t y0, int32_t x0, int32_t eh,
                                    int32_t ew, const void* wh, const void* ww, void* stream) {
    SVR2_CHECK(tile && result && count && wh && ww, SVR2_E_ARG, "svr2_tile_accumulate: null pointer");
    SVR2_CHECK(T > 0 && C > 0 && C <= Ct && eh > 0 && ew > 0 && eh <= th && ew <= tw, SVR2_E_SHAPE,
               "svr2_tile_accumulate: bad tile sizes (T %d C %d/%d eff %dx%d of %dx%d)", T, C, Ct, eh, ew, th, tw);
    SVR2_CHECK(y0 >= 0 && x0 >= 0 && y0 + eh <= H && x0 + ew <= W, SVR2_E_SHAPE,
               "svr2_tile_accumulate: tile [%d,%d)x[%d,%d) leaves the %dx%d result", y0, y0 + eh, x0, x0 + ew, H, W);
    hipLaunchKernelGGL(tile_accumulate_kernel, dim3(grid_for((long long)T * eh * ew)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)tile, T, th, tw, Ct, (bf16*)result, (bf16*)count, H, W, C, y0, x0, eh, ew,
                       (const bf16*)wh, (const bf16*)ww);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_tile_normalize(void* result, const void* count, int32_t T, int32_t H, int32_t W, int32_t C,
                                   void* stream) {
    SVR2_CHECK(result && count, SVR2_E_ARG, "svr2_tile_normalize: null pointer");
    SVR2_CHECK(T > 0 && H > 0 && W > 0 && C > 0, SVR2_E_SHAPE, "svr2_tile_normalize: bad sizes");
    hipLaunchKernelGGL(tile_normalize_kernel, dim3(grid_for((long long)T * H * W)), dim3(256), 0, (hipStream_t)stream,
                       (bf16*)result, (const bf16*)count, T, (long long)H * W, C);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}
