// Kernel-argument structs shared by the two GEMM / implicit-conv kernels (gemm.hip, gemm2.hip).
#pragma once
#include "common.h"

namespace svr2 {

enum AMode { A_DENSE = 0, A_CONV = 1 };

struct ConvGeom {
    const bf16* carry;
    int T, Hi, Wi, Cin, kt, kh, kw, st, sh, sw, npad, pad_h0, pad_w0, To, Ho, Wo;
    int crep;                      // head frames the carry does not hold (npad - carry_frames): they read carry[0]
    int patch, tiles_x, tiles_y;   // patch != 0: an M-tile is a (BM/16) x 16 pixel patch of one frame
};

// Division by a launch constant without the ~40-instruction integer divide (Granlund-Montgomery, round-up form, exact
// for every 32-bit numerator): the pixel-shuffle epilogue splits every output row index three ways per 16-byte store.
struct FastDiv {
    uint32_t mul, sh, d;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f{0, 0, d};
    if (d <= 1) return f;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.sh = l;
    f.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    if (f.d <= 1) return n;
    const uint32_t t = __umulhi(n, f.mul);
    return (t + ((n - t) >> 1)) >> (f.sh - 1);
}

struct KArgs {
    const bf16* A;
    const bf16* W;
    bf16* C;
    const int32_t* a_rows;
    const float* bias;
    const float* gate;
    const bf16* resid;
    const bf16* zero;
    float* stats;                  // conv_halo.hip: per-tile per-channel (sum, sumsq) of the output, or null
    float* partial;                // conv_halo.hip split-K: fp32 accumulator images [tile][split][32][512] x float4
    int ks_log2;                   // ... log2 of the number of K splits (0 = no split)
    int nsum;                      // ... images per tile the finishing kernel still has to add
    size_t partial_bytes;          // ... capacity of `partial` (host side only)
    long long lda, ldc, ldr;
    int M, N, K, nbm, nbn;
    int ps_frames, ps_h, ps_w, ps_c, ps_tz, ps_drop;
    FastDiv ps_wd, ps_hd, ps_cd;   // divisors ps_w, ps_h, ps_c
    int epi_pref;                  // GATE_RES epilogue: request every residual chunk before the first one is used
    ConvGeom cg;
};

constexpr int CONV_SLICE = 64;    // channels per K slice of the conv weight layout [Cout][kt][Cin/64][kh][kw][64]
constexpr int GROUP_M_TILES = 8;  // band height (in tiles) of the L2-friendly tile order

// XCD-aware remap (bijective form): blocks b and b+8 share an XCD; give every XCD a contiguous run of the
// tile order, and order tiles in bands of GROUP_M_TILES row-tiles so that the workgroups resident on one XCD
// share A row-panels and W column-panels in its L2.
__device__ __forceinline__ void tile_coords(const KArgs& a, int bid, int& tm, int& tn) {
    const int total = a.nbm * a.nbn;
    const int xcd = bid & 7, q = total >> 3, r = total & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int per_band = GROUP_M_TILES * a.nbn;
    const int band = lin / per_band, rem = lin - band * per_band;
    const int rows = min(GROUP_M_TILES, a.nbm - band * GROUP_M_TILES);
    tm = band * GROUP_M_TILES + rem % rows;
    tn = rem / rows;
}
__device__ __forceinline__ void tile_coords(const KArgs& a, int& tm, int& tn) { tile_coords(a, blockIdx.x, tm, tn); }

// Output row (pixel) index of row `row` of M-tile `tm`.  Conv tiles are 2-D pixel patches (BM/16 rows x 16
// columns of one frame) so that a workgroup's nine spatial taps, and the ~64 workgroups resident on one
// XCD, re-read a compact halo from L2 instead of three full image rows per frame.
template <int AMODE, int BM>
__device__ __forceinline__ int tile_row_to_m(const KArgs& a, int tm, int row) {
    if (AMODE == A_CONV && a.cg.patch) {
        const int per_frame = a.cg.tiles_x * a.cg.tiles_y;
        const int t = tm / per_frame, rem = tm - t * per_frame;
        const int ty = rem / a.cg.tiles_x, tx = rem - ty * a.cg.tiles_x;
        const int y = ty * (BM / 16) + (row >> 4), x = tx * 16 + (row & 15);
        return (t * a.cg.Ho + y) * a.cg.Wo + x;
    }
    return tm * BM + row;
}


// SVR2_EPI_PIXSHUF3D: element offset of GEMM output (row m = input pixel, column n = (xx, yy, z, channel)) in the
// pixel-shuffled tensor [F*tz (-1), 2H, 2W, C]; false = the dropped head frame  [ref: attn_video_vae.py:140-153]
__device__ __forceinline__ bool pixshuf_dst(const KArgs& a, int m, int n, long long& dst) {
    const int Cc = a.ps_c;
    const int qd = (int)fdiv((uint32_t)n, a.ps_cd), ch = n - qd * Cc;
    const int q2 = a.ps_tz == 2 ? qd >> 1 : qd, z = a.ps_tz == 2 ? qd & 1 : 0;     // qd / tz, qd % tz
    const int yy = q2 & 1, xx = q2 >> 1;
    const int t2 = (int)fdiv((uint32_t)m, a.ps_wd), w = m - t2 * a.ps_w;
    const int f = (int)fdiv((uint32_t)t2, a.ps_hd), h = t2 - f * a.ps_h;
    int fo = f * a.ps_tz + z;
    if (a.ps_drop) {
        if (fo == 1) return false;
        if (fo > 1) fo -= 1;
    }
    dst = (((long long)fo * (2 * a.ps_h) + (2 * h + xx)) * (2 * a.ps_w) + (2 * w + yy)) * Cc + ch;
    return true;
}

// Fused epilogues on a finished bf16 tile in LDS (row pitch EPI_LD bytes, BM rows x BN columns, bias already added):
// 8 consecutive columns per thread, 16-byte stores.  Shared by gemm2.hip and gemm4.hip.
template <int AMODE, int EPI, int BM, int BN, int NT, int EPI_LD>
__device__ __forceinline__ void tile_epilogue(const KArgs& a, const char* smem, int tm, int n0, int tid) {
    if (EPI == SVR2_EPI_SWIGLU) {
        // columns [16p, 16p+8) = gate, [16p+8, 16p+16) = in  ->  output columns [8p, 8p+8)
        for (int c = tid; c < BM * (BN / 16); c += NT) {
            const int row = c / (BN / 16), p = c % (BN / 16);
            const int m = tile_row_to_m<AMODE, BM>(a, tm, row), ncol = (n0 >> 1) + p * 8;
            if (m >= a.M || (n0 + p * 16) >= a.N) continue;
            const bf16x8 gv = *(const bf16x8*)(smem + row * EPI_LD + p * 32);
            const bf16x8 iv = *(const bf16x8*)(smem + row * EPI_LD + p * 32 + 16);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float g = bf2f(gv[e]);
                o[e] = f2bf(g / (1.f + __expf(-g)) * bf2f(iv[e]));
            }
            *(bf16x8*)(a.C + (long long)m * a.ldc + ncol) = o;
        }
        return;
    }

    if (EPI == SVR2_EPI_GATE_RES && a.epi_pref) {
        // resid + gate x tile: a thread's 16-byte residual chunks are all requested first - one exposed HBM round trip per
        // tile instead of one per chunk (the loop below waits for its load in every iteration; this workgroup is alone on
        // its CU, so nothing else covers that wait: -9 % on the K = 2 560 projection at M = 194 400)
        constexpr int CH = BM * (BN / 8) / NT;
        static_assert(BM * (BN / 8) % NT == 0 && NT % (BN / 8) == 0, "whole chunks per thread, one column group per thread");
        const int cc = tid % (BN / 8), n = n0 + cc * 8;
        bf16x8 rres[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int m = tile_row_to_m<AMODE, BM>(a, tm, (tid + k * NT) / (BN / 8));
            rres[k] = (m < a.M && n < a.N) ? *(const bf16x8*)(a.resid + (long long)m * a.ldr + n) : bf16x8{};
        }
        float g8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g8[e] = (a.gate && n + e < a.N) ? a.gate[n + e] : 1.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int row = (tid + k * NT) / (BN / 8);
            const int m = tile_row_to_m<AMODE, BM>(a, tm, row);
            if (m >= a.M || n >= a.N) continue;
            const bf16x8 tv = *(const bf16x8*)(smem + row * EPI_LD + cc * 16);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(rres[k][e]) + g8[e] * bf2f(tv[e]));
            const long long dst = (long long)m * a.ldc + n;
            if (n + 8 <= a.N) {
                *(bf16x8*)(a.C + dst) = o;
            } else {
                for (int e = 0; e < a.N - n; ++e) a.C[dst + e] = o[e];
            }
        }
        return;
    }
    for (int c = tid; c < BM * (BN / 8); c += NT) {
        const int row = c / (BN / 8), cc = c % (BN / 8);
        const int m = tile_row_to_m<AMODE, BM>(a, tm, row), n = n0 + cc * 8;
        if (m >= a.M || n >= a.N) continue;
        const bf16x8 tv = *(const bf16x8*)(smem + row * EPI_LD + cc * 16);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf2f(tv[e]);
        long long dst;
        if (EPI == SVR2_EPI_GATE_RES) {
            const bf16x8 rv = *(const bf16x8*)(a.resid + (long long)m * a.ldr + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bf2f(rv[e]) + (a.gate ? a.gate[n + e] : 1.f) * v[e];
            dst = (long long)m * a.ldc + n;
        } else if (EPI == SVR2_EPI_PIXSHUF3D) {
            if (!pixshuf_dst(a, m, n, dst)) continue;
        } else if (EPI == SVR2_EPI_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = v[e], u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
                v[e] = x / (1.f + __expf(-2.f * u));
            }
            dst = (long long)m * a.ldc + n;
        } else {
            if (a.resid) {
                const bf16x8 rv = *(const bf16x8*)(a.resid + (long long)m * a.ldr + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bf2f(rv[e]);
            }
            dst = (long long)m * a.ldc + n;
        }
        if (n + 8 <= a.N) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
            *(bf16x8*)(a.C + dst) = o;
        } else {
            for (int e = 0; e < a.N - n; ++e) a.C[dst + e] = f2bf(v[e]);
        }
    }
}

// implemented in gemm4.hip: dense 256x256x64 ping-pong kernel (N % 256 == 0, K % 64 == 0)
int gemm4_launch(int epi, KArgs a, hipStream_t s);
// implemented in gemm2.hip: 256-row tiles, 128x64 per wave, BK = 32, 3-deep DMA ring
int gemm2_launch(int amode, int epi, KArgs a, hipStream_t s);
// implemented in conv_halo.hip: halo-resident stride-1 3x3(xkt) conv
int conv_halo_launch(KArgs a, hipStream_t s);
int conv_in3_launch(const svr2_conv3d_desc* d, hipStream_t s);   // conv_in.hip
size_t conv_halo_split_bytes(const KArgs& a);

}  // namespace svr2
