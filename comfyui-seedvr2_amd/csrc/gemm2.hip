// Second-generation bf16 MFMA GEMM / implicit-conv3d kernel for the large problems of the path.
//
// Why it exists (measured on MI355X, tools/kernel_bench.py and timing builds, round 1): the 128x128x64
// kernel of gemm.hip runs at ~1.0-1.1 PFLOP/s; with its global->LDS DMA disabled the same loop reaches
// 1.5-1.7 PFLOP/s, and removing only the W or only the A loads recovers 15 % / 25 %: the LDS-DMA issue rate
// per MFMA is the limiter.  This kernel halves-to-quarters that rate:
//   * tile 256 x BN x 32 (BN = 128 or 256), waves as 2 x (BN/64), each wave a 128 x 64 sub-tile
//     = 8 x 4 MFMA 16x16x32 tiles (128 fp32 accumulators): per wave and K-tile (32 MFMAs) 4+2 (BN=128) or
//     2+2 (BN=256) DMA instructions instead of 8, and 12 ds_read_b128 instead of 16;
//   * 3-deep DMA ring with counted vmcnt: two K-tiles stay in flight across the (raw) barrier;
//   * 64-byte LDS rows; bank swizzle chunk ^= LUT[(row>>2)&3], LUT = {0,2,3,1}, applied on the DMA source
//     address and on the ds_read_b128 address (conflict-free for the 16-lane read groups);
//   * MFMA operands swapped (W is the A operand) so a lane's 4 accumulator registers are 4 consecutive output
//     columns: the epilogue tile is written with ds_write_b64 of packed bf16 (acc + bias rounded once, exactly
//     where the reference rounds a linear/conv output) and read back as 16-byte rows for the fused epilogues.
#include "gemm_args.h"
#include <stdlib.h>

namespace svr2 {
namespace {

constexpr int BM2 = 256;

// BN = 128: K-tile 32 (64-byte LDS rows), 3-deep ring (72 KiB) -> two workgroups per CU.
// BN = 256: K-tile 64 (128-byte rows = whole cache lines per DMA row), 2-deep ring (128 KiB), one 8-wave
//           workgroup per CU.  (With 64-byte rows every DMA instruction touches half lines: measured 10.8 TB/s
//           of L2->LDS fill at BN = 256, the cap of that variant.)
template <int BN> struct Geo2 {
    static constexpr int BK = BN == 256 ? 64 : 32;
    static constexpr int ROWB = BK * 2;                 // bytes per LDS row
    static constexpr int RPI = 1024 / ROWB;             // rows per 1-KiB DMA instruction
    static constexpr int CPR = ROWB / 16;               // 16-byte chunks per row
    static constexpr int STAGES = BN == 256 ? 2 : 3;
    static constexpr int NW = 2 * (BN / 64);            // waves: 2 (M) x BN/64 (N)
    static constexpr int NT = NW * 64;
    static constexpr int NA = (BM2 / RPI) / NW;         // A DMA instructions per wave per K-tile
    static constexpr int NB = (BN / RPI) / NW;          // W DMA instructions per wave per K-tile
    static constexpr int A_BYTES = BM2 * ROWB, B_BYTES = BN * ROWB;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int EPI_LD = BN * 2 + 16;          // padded bf16 epilogue row (bytes)
    static constexpr int EPI_BYTES = BM2 * EPI_LD;
    static constexpr int LDS = (STAGES * STAGE > EPI_BYTES) ? STAGES * STAGE : EPI_BYTES;
};

// bank swizzle of the 16-byte chunk index inside a row (conflict-free ds_read_b128 of 16 rows x same chunk):
//   64-byte rows : chunk ^= LUT[(row>>2)&3], LUT = {0,2,3,1};   128-byte rows : chunk ^= (row>>1)&7
template <int ROWB>
__device__ __forceinline__ int swz(int row) {
    return ROWB == 64 ? ((0x78 >> (2 * ((row >> 2) & 3))) & 3) : ((row >> 1) & 7);
}

template <int AMODE, int EPI, int BN>
__global__ __launch_bounds__(Geo2<BN>::NT, 2) void gemm2_kernel(const KArgs a) {
    using G = Geo2<BN>;
    constexpr int NA = G::NA, NB = G::NB, NT = G::NT, ROWB = G::ROWB, BK2 = G::BK, RPI = G::RPI, CPR = G::CPR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    tile_coords(a, tm, tn);
    const int n0 = tn * BN;

    // ---- loader setup: per K-tile this thread moves NA rows of A and NB rows of W, 16 bytes each
    const int lrow = lane / CPR, lslot = lane % CPR;
    const char* a_ptr[NA];
    const char* w_ptr[NB];
    int cy[NA], cx[NA], ct[NA];
    long long rowoff[NA];
    const char* zsrc = (const char*)a.zero + lslot * 16;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = (wave * NA + i) * RPI + lrow;
        const int chunk = lslot ^ swz<ROWB>(r);
        const int m = tile_row_to_m<AMODE, BM2>(a, tm, r);
        if (AMODE == A_DENSE) {
            long long src = m;
            if (a.a_rows) src = a.a_rows[m < a.M ? m : a.M - 1];
            a_ptr[i] = (m < a.M) ? (const char*)(a.A + src * a.lda) + chunk * 16 : nullptr;
        } else {
            const int mm = m < a.M ? m : 0;
            const int x = mm % a.cg.Wo, t2 = mm / a.cg.Wo;
            const int y = t2 % a.cg.Ho, t = t2 / a.cg.Ho;
            cx[i] = x * a.cg.sw - a.cg.pad_w0;
            cy[i] = (m < a.M) ? y * a.cg.sh - a.cg.pad_h0 : -(1 << 28);
            ct[i] = t * a.cg.st;
            rowoff[i] = (((long long)cy[i] * a.cg.Wi + cx[i]) * a.cg.Cin) * 2 + chunk * 16;
            a_ptr[i] = nullptr;
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int r = (wave * NB + i) * RPI + lrow;
        w_ptr[i] = (const char*)(a.W + (long long)(n0 + r) * a.K) + ((lslot ^ swz<ROWB>(r)) << 4);
    }
    int tap_t = 0, tap_y = 0, tap_x = 0, tap_c = 0, tap_blk = 0;
    const long long frame_bytes = (long long)a.cg.Hi * a.cg.Wi * a.cg.Cin * 2;
    auto set_frame = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int it = ct[i] + tap_t;
            const bool head = it < a.cg.npad;
            const char* base = (head && a.cg.carry) ? (const char*)a.cg.carry : (const char*)a.A;
            const int fidx = head ? (a.cg.carry ? max(it - a.cg.crep, 0) : 0) : it - a.cg.npad;
            a_ptr[i] = base + fidx * frame_bytes + rowoff[i];
        }
    };
    if (AMODE == A_CONV) set_frame();

    auto issue = [&](int stage) {
        char* sA = smem + stage * G::STAGE;
        char* sB = sA + G::A_BYTES;
        if (AMODE == A_DENSE) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                glds16(a_ptr[i] ? a_ptr[i] : zsrc, sA + (wave * NA + i) * 1024);
                if (a_ptr[i]) a_ptr[i] += BK2 * 2;
            }
        } else {
            const long long toff = (((long long)tap_y * a.cg.Wi + tap_x) * a.cg.Cin + tap_blk + tap_c) * 2;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const bool ok = ((unsigned)(cy[i] + tap_y) < (unsigned)a.cg.Hi) & ((unsigned)(cx[i] + tap_x) < (unsigned)a.cg.Wi);
                glds16(ok ? a_ptr[i] + toff : zsrc, sA + (wave * NA + i) * 1024);
            }
            // K order: 64-channel slice innermost, then the kh*kw spatial taps, then the channel slices, then the
            // temporal taps - the 9 spatial taps of one slice re-read the same 128-byte lines within 9 consecutive
            // K-steps, so they hit in L2 (tap-major order over all channels thrashed it: 18x the input in FETCH_SIZE)
            tap_c += BK2;
            if (tap_c == CONV_SLICE) {
                tap_c = 0;
                if (++tap_x == a.cg.kw) {
                    tap_x = 0;
                    if (++tap_y == a.cg.kh) {
                        tap_y = 0;
                        tap_blk += CONV_SLICE;
                        if (tap_blk == a.cg.Cin) {
                            tap_blk = 0;
                            ++tap_t;
                            set_frame();
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            glds16(w_ptr[i], sB + (wave * NB + i) * 1024);
            w_ptr[i] += BK2 * 2;
        }
    };

    // ---- MFMA fragments: lane = (row l&15 of a 16-row tile, 8-element k group l>>4)
    const int wm = wave / (BN / 64), wn = wave % (BN / 64);
    const int frow = lane & 15, fq = lane >> 4;
    const int fswz = swz<ROWB>(frow);                           // tile bases are multiples of 16 rows
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = a.K / BK2;
    issue(0);
    if (G::STAGES == 3 && nk > 1) issue(1);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt landed for every wave and every wave finished reading the stage refilled below.  3-deep ring:
        // the DMA of tile kt+1 stays in flight across the raw barrier (counted vmcnt = this thread's loads per
        // tile); 2-deep ring: tile kt+1 is issued right after the barrier and flies during tile kt's 64 MFMAs.
        if (G::STAGES == 3) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NA + NB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 2 < nk) issue(stage >= 1 ? stage - 1 : 2);   // (kt + 2) % 3
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 1 < nk) issue(stage ^ 1);
        }
        const char* sA = smem + stage * G::STAGE + (wm * 128 + frow) * ROWB;
        const char* sB = smem + stage * G::STAGE + G::A_BYTES + (wn * 64 + frow) * ROWB;
        stage = (G::STAGES == 3) ? (stage == 2 ? 0 : stage + 1) : (stage ^ 1);
#pragma unroll
        for (int ks = 0; ks < BK2 / 32; ++ks) {
            const int coff = ((fq + 4 * ks) ^ fswz) << 4;
            bf16x8 bfr[4], af[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(sB + j * 16 * ROWB + coff);
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *(const bf16x8*)(sA + i * 16 * ROWB + coff);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)  // W fragment is the MFMA A operand: D[row = n][col = m]
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_iglp_opt(1);
    }

    // ---- epilogue: (acc + bias) -> bf16 tile in LDS -> fused epilogues on 8 consecutive columns
    asm volatile("s_barrier" ::: "memory");   // every wave is done reading the ring
    {
        // lane holds, for row m = wm*128 + i*16 + frow, columns n = wn*64 + j*16 + 4*fq + r (r = 0..3)
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int n = n0 + wn * 64 + j * 16 + 4 * fq;
            if (a.bias && EPI != SVR2_EPI_SWIGLU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[j][r] = (n + r < a.N) ? a.bias[n + r] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[i][j][r] + bv[j][r]);
                *(bf16x4*)(smem + (wm * 128 + i * 16 + frow) * G::EPI_LD + (wn * 64 + j * 16 + 4 * fq) * 2) = o;
            }
    }
    __syncthreads();

    tile_epilogue<AMODE, EPI, BM2, BN, NT, G::EPI_LD>(a, smem, tm, n0, tid);
}

template <int AMODE, int EPI, int BN>
int launch2(KArgs a, hipStream_t s) {
    using G = Geo2<BN>;
    SVR2_SET_MAX_LDS((gemm2_kernel<AMODE, EPI, BN>), G::LDS, "gemm2");
    a.nbm = (a.M + BM2 - 1) / BM2;
    a.nbn = (a.N + BN - 1) / BN;
    if (AMODE == A_CONV) {
        a.cg.patch = (a.cg.Ho % (BM2 / 16) == 0 && a.cg.Wo % 16 == 0) ? 1 : 0;
        a.cg.tiles_x = a.cg.Wo / 16;
        a.cg.tiles_y = a.cg.Ho / (BM2 / 16);
    }
    hipLaunchKernelGGL((gemm2_kernel<AMODE, EPI, BN>), dim3(a.nbm * a.nbn), dim3(G::NT), G::LDS, s, a);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

template <int AMODE, int EPI>
int launch2_bn(const KArgs& a, hipStream_t s) {
    // 256-wide N tiles read W rows [n0, n0+256): only when N is a multiple of 256 (W is padded to 128 rows)
    const bool wide = (a.N % 256 == 0) && !svr2::opt(svr2::OPT_GEMM2_NARROW, 0);
    return wide ? launch2<AMODE, EPI, 256>(a, s) : launch2<AMODE, EPI, 128>(a, s);
}

}  // namespace

int gemm2_launch(int amode, int epi, KArgs a, hipStream_t s) {
    if (amode == A_CONV) return launch2_bn<A_CONV, SVR2_EPI_BIAS>(a, s);
    switch (epi) {
        case SVR2_EPI_BIAS: return launch2_bn<A_DENSE, SVR2_EPI_BIAS>(a, s);
        case SVR2_EPI_GATE_RES: return launch2_bn<A_DENSE, SVR2_EPI_GATE_RES>(a, s);
        case SVR2_EPI_SWIGLU: return launch2_bn<A_DENSE, SVR2_EPI_SWIGLU>(a, s);
        case SVR2_EPI_PIXSHUF3D: return launch2_bn<A_DENSE, SVR2_EPI_PIXSHUF3D>(a, s);
        case SVR2_EPI_GELU: return launch2_bn<A_DENSE, SVR2_EPI_GELU>(a, s);
    }
    svr2_set_error("gemm2: unknown epilogue %d", epi);
    return SVR2_E_ARG;
}

}  // namespace svr2
