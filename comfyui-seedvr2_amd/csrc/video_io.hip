// Frame I/O conversions at the two ends of the streaming path: the bytes that cross PCIe are the
// container's own 8-bit pixels, never fp16/fp32 frame tensors.
//   svr2_u8_to_frames  : packed RGB/BGR rows (any row stride, top-down or bottom-up) -> bf16 [T,H,W,3] in [0,1] with the
//                        reference's rounding chain uint8 -> /255 (fp32) -> fp16 -> bf16
//   svr2_frames_to_u8  : bf16 [T,H,W,3] in [0,1] -> packed RGB/BGR rows, (x * 255) truncated like numpy's astype(uint8)
//   svr2_frames_to_yuv : the same quantisation followed by the integer BT.601 studio-range matrix, planar 4:4:4 / 4:2:0
//   svr2_yuv_to_frames : planar 4:4:4 / 4:2:0 (chroma replicated) -> RGB -> the uint8 chain of svr2_u8_to_frames
// All HBM-bound single passes.
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

__device__ __forceinline__ bf16 unit_from_u8(int v) {
    // inference_cli.py:543 (float32 / 255), :627 (.to(float16)); generation_phases.py:376-384 (-> compute dtype bf16)
    const float f = (float)v / 255.0f;
    return f2bf(__half2float(__float2half_rn(f)));
}

__device__ __forceinline__ int u8_from_unit(bf16 v) {
    // inference_cli.py:520,691,734: (frames.cpu().numpy() * 255.0).astype(np.uint8) on values already clamped to [0,1]
    const float f = fminf(fmaxf(bf2f(v) * 255.0f, 0.f), 255.f);
    return (int)f;
}

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ __launch_bounds__(256) void u8_to_frames_kernel(const uint8_t* __restrict__ src, bf16* __restrict__ dst, int T,
                                                           int H, int W, long long row_stride, long long frame_stride,
                                                           int swap_rb, int flip, int nc) {
    const long long total = (long long)T * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long long r = i / W;
        const int y = (int)(r % H), t = (int)(r / H);
        const uint8_t* p = src + t * frame_stride + (long long)(flip ? H - 1 - y : y) * row_stride + x * nc;
        const int c0 = p[0], c1 = p[1], c2 = p[2];
        dst[i * nc + 0] = unit_from_u8(swap_rb ? c2 : c0);
        dst[i * nc + 1] = unit_from_u8(c1);
        dst[i * nc + 2] = unit_from_u8(swap_rb ? c0 : c2);
        if (nc == 4) dst[i * 4 + 3] = unit_from_u8(p[3]);          // alpha keeps its place (RGBA / BGRA)
    }
}

__global__ __launch_bounds__(256) void frames_to_u8_kernel(const bf16* __restrict__ src, uint8_t* __restrict__ dst, int T,
                                                           int H, int W, long long row_stride, long long frame_stride,
                                                           int swap_rb, int flip, int nc) {
    // one thread per pixel, plus the threads x in [W, W + pad) that zero the row padding (at most 3 bytes)
    const int pad = (int)(row_stride - (long long)W * nc);
    const int WX = W + (pad > 0 ? 1 : 0);
    const long long total = (long long)T * H * WX;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % WX);
        const long long r = i / WX;
        const int y = (int)(r % H), t = (int)(r / H);
        uint8_t* p = dst + t * frame_stride + (long long)(flip ? H - 1 - y : y) * row_stride + x * nc;
        if (x == W) {
            for (int k = 0; k < pad; ++k) p[k] = 0;
            continue;
        }
        const bf16* s = src + (((long long)t * H + y) * W + x) * nc;
        const int c0 = u8_from_unit(s[0]), c1 = u8_from_unit(s[1]), c2 = u8_from_unit(s[2]);
        p[0] = (uint8_t)(swap_rb ? c2 : c0);
        p[1] = (uint8_t)c1;
        p[2] = (uint8_t)(swap_rb ? c0 : c2);
        if (nc == 4) p[3] = (uint8_t)u8_from_unit(s[3]);
    }
}

// ITU-R BT.601 studio range, 8-bit integer form (what a BGR24 -> yuv420p software scaler feeds an MPEG-4 encoder)
__device__ __forceinline__ int y_of(int r, int g, int b) { return ((66 * r + 129 * g + 25 * b + 128) >> 8) + 16; }
__device__ __forceinline__ int u_of(int r, int g, int b) { return ((-38 * r - 74 * g + 112 * b + 128) >> 8) + 128; }
__device__ __forceinline__ int v_of(int r, int g, int b) { return ((112 * r - 94 * g - 18 * b + 128) >> 8) + 128; }

__global__ __launch_bounds__(256) void frames_to_yuv_kernel(const bf16* __restrict__ src, uint8_t* __restrict__ dst, int T,
                                                            int H, int W, int sub, long long frame_stride) {
    const int CW = W / sub, CH = H / sub;               // one thread per chroma sample (sub x sub luma block)
    const long long total = (long long)T * CH * CW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int cx = (int)(i % CW);
        const long long r = i / CW;
        const int cy = (int)(r % CH), t = (int)(r / CH);
        uint8_t* yp = dst + t * frame_stride;
        uint8_t* up = yp + (long long)H * W;
        uint8_t* vp = up + (long long)CH * CW;
        int sr = 0, sg = 0, sb = 0;
        for (int dy = 0; dy < sub; ++dy)
            for (int dx = 0; dx < sub; ++dx) {
                const int y = cy * sub + dy, x = cx * sub + dx;
                const bf16* s = src + (((long long)t * H + y) * W + x) * 3;
                const int rr = u8_from_unit(s[0]), gg = u8_from_unit(s[1]), bb = u8_from_unit(s[2]);
                yp[(long long)y * W + x] = (uint8_t)clip8(y_of(rr, gg, bb));
                sr += rr; sg += gg; sb += bb;
            }
        if (sub == 2) { sr = (sr + 2) >> 2; sg = (sg + 2) >> 2; sb = (sb + 2) >> 2; }   // mean of the 2 x 2 block, rounded
        up[(long long)cy * CW + cx] = (uint8_t)clip8(u_of(sr, sg, sb));
        vp[(long long)cy * CW + cx] = (uint8_t)clip8(v_of(sr, sg, sb));
    }
}

__global__ __launch_bounds__(256) void yuv_to_frames_kernel(const uint8_t* __restrict__ src, bf16* __restrict__ dst, int T,
                                                            int H, int W, int sub, long long frame_stride) {
    const int CW = (W + sub - 1) / sub, CH = (H + sub - 1) / sub;
    const long long total = (long long)T * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long long r = i / W;
        const int y = (int)(r % H), t = (int)(r / H);
        const uint8_t* yp = src + t * frame_stride;
        const uint8_t* up = yp + (long long)H * W;
        const uint8_t* vp = up + (long long)CH * CW;
        const int c = 298 * ((int)yp[(long long)y * W + x] - 16);
        const int d = (int)up[(long long)(y / sub) * CW + x / sub] - 128;
        const int e = (int)vp[(long long)(y / sub) * CW + x / sub] - 128;
        dst[i * 3 + 0] = unit_from_u8(clip8((c + 409 * e + 128) >> 8));
        dst[i * 3 + 1] = unit_from_u8(clip8((c - 100 * d - 208 * e + 128) >> 8));
        dst[i * 3 + 2] = unit_from_u8(clip8((c + 516 * d + 128) >> 8));
    }
}

inline unsigned grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace

extern "C" int svr2_u8_to_frames(const void* src, void* dst, int32_t T, int32_t H, int32_t W, int64_t row_stride,
                                 int64_t frame_stride, int32_t flags, void* stream) {
    SVR2_CHECK(src && dst, SVR2_E_ARG, "svr2_u8_to_frames: null pointer");
    const int nc = (flags & SVR2_PIX_ALPHA) ? 4 : 3;
    SVR2_CHECK(T > 0 && H > 0 && W > 0 && row_stride >= (int64_t)W * nc && frame_stride >= row_stride * H, SVR2_E_SHAPE,
               "svr2_u8_to_frames: bad sizes (T %d H %d W %d row %lld frame %lld)", T, H, W, (long long)row_stride,
               (long long)frame_stride);
    hipLaunchKernelGGL(u8_to_frames_kernel, dim3(grid_for((long long)T * H * W)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)src, (bf16*)dst, T, H, W, (long long)row_stride, (long long)frame_stride,
                       flags & SVR2_PIX_SWAP_RB, flags & SVR2_PIX_BOTTOM_UP, nc);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_frames_to_u8(const void* src, void* dst, int32_t T, int32_t H, int32_t W, int64_t row_stride,
                                 int64_t frame_stride, int32_t flags, void* stream) {
    SVR2_CHECK(src && dst, SVR2_E_ARG, "svr2_frames_to_u8: null pointer");
    const int nc = (flags & SVR2_PIX_ALPHA) ? 4 : 3;
    SVR2_CHECK(T > 0 && H > 0 && W > 0 && row_stride >= (int64_t)W * nc && row_stride < (int64_t)W * nc + 4 &&
                   frame_stride >= row_stride * H,
               SVR2_E_SHAPE, "svr2_frames_to_u8: bad sizes (T %d H %d W %d row %lld frame %lld)", T, H, W,
               (long long)row_stride, (long long)frame_stride);
    hipLaunchKernelGGL(frames_to_u8_kernel, dim3(grid_for((long long)T * H * (W + 1))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)src, (uint8_t*)dst, T, H, W, (long long)row_stride, (long long)frame_stride,
                       flags & SVR2_PIX_SWAP_RB, flags & SVR2_PIX_BOTTOM_UP, nc);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_frames_to_yuv(const void* src, void* dst, int32_t T, int32_t H, int32_t W, int32_t sub,
                                  int64_t frame_stride, void* stream) {
    SVR2_CHECK(src && dst, SVR2_E_ARG, "svr2_frames_to_yuv: null pointer");
    SVR2_CHECK(sub == 1 || sub == 2, SVR2_E_ARG, "svr2_frames_to_yuv: chroma subsampling must be 1 (4:4:4) or 2 (4:2:0)");
    SVR2_CHECK(T > 0 && H > 0 && W > 0 && H % sub == 0 && W % sub == 0, SVR2_E_SHAPE,
               "svr2_frames_to_yuv: %d x %d is not a multiple of the chroma subsampling %d", W, H, sub);
    SVR2_CHECK(frame_stride >= (int64_t)H * W + 2 * (int64_t)(H / sub) * (W / sub), SVR2_E_SHAPE,
               "svr2_frames_to_yuv: frame stride too small");
    hipLaunchKernelGGL(frames_to_yuv_kernel, dim3(grid_for((long long)T * (H / sub) * (W / sub))), dim3(256), 0,
                       (hipStream_t)stream, (const bf16*)src, (uint8_t*)dst, T, H, W, sub, (long long)frame_stride);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_yuv_to_frames(const void* src, void* dst, int32_t T, int32_t H, int32_t W, int32_t sub,
                                  int64_t frame_stride, void* stream) {
    SVR2_CHECK(src && dst, SVR2_E_ARG, "svr2_yuv_to_frames: null pointer");
    SVR2_CHECK(sub == 1 || sub == 2, SVR2_E_ARG, "svr2_yuv_to_frames: chroma subsampling must be 1 (4:4:4) or 2 (4:2:0)");
    SVR2_CHECK(T > 0 && H > 0 && W > 0, SVR2_E_SHAPE, "svr2_yuv_to_frames: bad sizes");
    SVR2_CHECK(frame_stride >= (int64_t)H * W + 2 * (int64_t)((H + sub - 1) / sub) * ((W + sub - 1) / sub), SVR2_E_SHAPE,
               "svr2_yuv_to_frames: frame stride too small");
    hipLaunchKernelGGL(yuv_to_frames_kernel, dim3(grid_for((long long)T * H * W)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)src, (bf16*)dst, T, H, W, sub, (long long)frame_stride);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}
