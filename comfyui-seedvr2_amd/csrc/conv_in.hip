// Fused causal 3x3x3 conv for Cin = 3, Cout = 128: the encoder's conv_in at full resolution.
//   [ref: models/video_vae_v3/modules/attn_video_vae.py (Encoder3D.conv_in) through
//    causal_inflation_lib.py:250-305 (InflatedCausalConv3d.forward: causal head / carried frames, zero padding 1)]
//
// The layer is write-bound: 3 input channels in, 128 out (2.1 GB of bf16 per 4K frame against 25 MB read), 81 MACs per
// output value.  The im2col + GEMM path it replaces wrote a 256-byte column row per pixel and read it back (17 GB each way
// per 8-frame 4K chunk) before the 17 GB of output.  Here a workgroup keeps the input halo of a 16 x 32 pixel patch
// (3 frames x 18 x 34 pixels, channels padded 3 -> 4: 8 bytes per pixel) in LDS and every lane gathers its MFMA operand
// straight from it:
//   K order  j = (kt*3 + ky) * 12 + kx*4 + c   (c = 3 is the zero pad channel; 108 used of K = 128, four 16x16x32 steps)
//   so the 12 values of one (kt, ky) pair for output pixel (y, x) are the 12 consecutive LDS elements that start at halo
//   pixel (kt, y + ky, x): a lane's 8 consecutive j are two aligned 8-byte LDS reads.
// W ([128][128] bf16 in that K order) lives in registers for the lifetime of the (persistent) workgroup: waves are 2 x 2,
// a wave owns 64 output channels (4 fragments x 4 K-steps x 4 registers) and half of the patch's rows.  Accumulators leave
// through a per-wave LDS transposition so that 8 lanes store 128 contiguous bytes of a pixel, and the GroupNorm partial
// sums of the rounded outputs are reduced per patch exactly like conv_halo.hip's epilogue does (same tile geometry:
// svr2_conv3d_gn_tiles, 16 rows x 32 pixels for Cout = 128), so the following GroupNorm needs no statistics pass.
#include "common.h"
#include "gemm_args.h"

namespace {

using namespace svr2;

constexpr int TW = 32, TH = 16;                       // output patch
constexpr int HW_ = TW + 2, HH = TH + 2;              // halo
constexpr int HALO_BYTES = 3 * HH * HW_ * 8;          // 14 688
constexpr int TR_LD = 144;                            // transposition row: 64 channels x 2 B + 16 (bank spread)
constexpr int TR_BYTES = 4 * 32 * TR_LD;              // per wave 32 pixels
constexpr int RED_BYTES = 256 * 16 * 4;
constexpr int LDS_TOTAL = HALO_BYTES + TR_BYTES + RED_BYTES;

struct CinArgs {
    const bf16* x;
    const bf16* carry;
    const bf16* w;
    const float* bias;
    bf16* y;
    float* stats;
    int T, H, W, npad, crep, tiles_x, tiles_y;
    long long total;
};

__global__ __launch_bounds__(256, 2) void conv_in3_kernel(const CinArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_TOTAL];
    char* halo = smem;
    char* trans = smem + HALO_BYTES;
    float* red = (float*)(smem + HALO_BYTES + TR_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave & 1, wp = wave >> 1;           // channel half, row half
    const int frow = lane & 15, fq = lane >> 4;

    // ---- W fragments (first MFMA operand: row = output channel, 8 consecutive k per lane) and bias, once
    bf16x8 wreg[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            wreg[m][ks] = *(const bf16x8*)(a.w + (long long)(wc * 64 + m * 16 + frow) * 128 + ks * 32 + fq * 8);
    f32x4 bv[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[m][r] = a.bias ? a.bias[wc * 64 + m * 16 + fq * 4 + r] : 0.f;

    // ---- per-lane gather offsets: K-step ks, lane group fq -> j0 = ks*32 + fq*8 -> (pair, e0) and the following 4
    int off[4][2];      // byte offsets relative to the lane's pixel; -1 = beyond the 108 used k (zeros)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = ks * 32 + fq * 8 + h * 4;
            const int pair = j / 12, e = j - pair * 12;
            const int kt = pair / 3, ky = pair - kt * 3;
            off[ks][h] = j < 108 ? (((kt * HH + ky) * HW_) * 4 + e) * 2 : -1;
        }

    const long long frame_px = (long long)a.H * a.W;
    const int per_frame = a.tiles_x * a.tiles_y;
    for (long long tile = blockIdx.x; tile < a.total; tile += gridDim.x) {
        const int t = (int)(tile / per_frame), rem = (int)(tile - (long long)t * per_frame);
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;

        // ---- halo of the three causal taps -> LDS (4th channel and everything outside the image: zero)
        for (int i = tid; i < 3 * HH * HW_; i += 256) {
            const int kt = i / (HH * HW_), r2 = i - kt * (HH * HW_);
            const int hy = r2 / HW_, hx = r2 - hy * HW_;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const int p = t + kt;                                    // padded frame index
            const bf16* src;
            if (p < a.npad)                                          // head: carried frames (the first one also stands
                src = a.carry ? a.carry + (long long)max(p - a.crep, 0) * frame_px * 3 : a.x;   // for missing ones) or frame 0
            else src = a.x + (long long)(p - a.npad) * frame_px * 3;
            bf16x4 v = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
            if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
                const bf16* s = src + ((long long)gy * a.W + gx) * 3;
                v[0] = s[0]; v[1] = s[1]; v[2] = s[2];
            }
            *(bf16x4*)(halo + i * 8) = v;
        }
        __syncthreads();

        float st_s[8], st_q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.f;

        // ---- this wave: rows wp*8 .. wp*8+7, both 16-pixel halves of a row per pass
        for (int rr = 0; rr < 8; ++rr) {
            const int ly = wp * 8 + rr;
            f32x4 acc[2][4];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[p][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 pf[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const char* base = halo + ((ly * HW_) + p * 16 + frow) * 8;
                    bf16x4 lo = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f}, hi = lo;
                    if (off[ks][0] >= 0) lo = *(const bf16x4*)(base + off[ks][0]);
                    if (off[ks][1] >= 0) hi = *(const bf16x4*)(base + off[ks][1]);
                    pf[p] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        acc[p][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[m][ks], pf[p], acc[p][m], 0, 0, 0);
            }
            // (acc + bias) -> bf16 -> this wave's transposition tile [32 pixels][64 channels]
            char* tr = trans + wave * (32 * TR_LD);
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[p][m][r] + bv[m][r]);
                    *(bf16x4*)(tr + (p * 16 + frow) * TR_LD + (m * 16 + fq * 4) * 2) = o;
                }
            // rows of 8 channels: 8 lanes store the 128 contiguous bytes of a pixel's channel half
            const int gy = y0 + ly;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int px = (lane >> 3) + 8 * i, ch = lane & 7;
                const bf16x8 o = *(const bf16x8*)(tr + px * TR_LD + ch * 16);
                const int gx = x0 + px;
                if (gy < a.H && gx < a.W) {
                    *(bf16x8*)(a.y + (((long long)t * a.H + gy) * a.W + gx) * 128 + wc * 64 + ch * 8) = o;
                    if (a.stats) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float r = bf2f(o[e]);
                            st_s[e] += r;
                            st_q[e] += r * r;
                        }
                    }
                }
            }
        }

        if (a.stats) {   // partial[frame][tile][2][128]: 16 contributors per channel (8 lanes x the two row halves)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[tid * 16 + e] = st_s[e];
                red[tid * 16 + 8 + e] = st_q[e];
            }
            __syncthreads();
            {
                const int ch = tid & 127, which = tid >> 7;
                const int c_wc = ch >> 6, c_chunk = (ch & 63) >> 3, c_e = ch & 7;
                float s = 0.f;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int l8 = 0; l8 < 8; ++l8) {
                        const int src_tid = (h * 2 + c_wc) * 64 + l8 * 8 + c_chunk;
                        s += red[src_tid * 16 + which * 8 + c_e];
                    }
                a.stats[(((long long)t * per_frame + rem) * 2 + which) * 128 + ch] = s;
            }
        }
        __syncthreads();    // halo and reduction buffer are rewritten by the next patch
    }
}

}  // namespace

namespace svr2 {

// d has been validated by svr2_conv3d for the SVR2_WL_CIN3 layout.
int conv_in3_launch(const svr2_conv3d_desc* d, hipStream_t s) {
    CinArgs a;
    a.x = (const bf16*)d->x; a.carry = (const bf16*)d->carry; a.w = (const bf16*)d->w; a.bias = d->bias;
    a.y = (bf16*)d->y; a.stats = d->gn_partial;
    a.T = d->T; a.H = d->Hi; a.W = d->Wi; a.npad = d->npad;
    a.crep = (d->carry && d->carry_frames > 0) ? d->npad - d->carry_frames : 0;
    a.tiles_x = (d->Wo + TW - 1) / TW; a.tiles_y = (d->Ho + TH - 1) / TH;
    a.total = (long long)d->To * a.tiles_x * a.tiles_y;
    const long long grid = a.total < 512 ? a.total : 512;     // persistent: two workgroups per CU of a 256-CU part
    hipLaunchKernelGGL(conv_in3_kernel, dim3((unsigned)grid), dim3(256), 0, s, a);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

}  // namespace svr2
