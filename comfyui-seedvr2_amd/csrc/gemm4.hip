// Ping-pong dense bf16 GEMM (256 x 256 x 64 tile, 8 waves): the two waves of every SIMD alternate between a LOAD phase
// (LDS fragment reads + LDS-DMA issue) and a COMPUTE phase (64 back-to-back MFMAs on registers), half a K-step apart.
//
// Why (measured on MI355X, round 1): in gemm2.hip all eight waves read their fragments, wait, and multiply in lockstep
// (one barrier per K-step), so the LDS traffic and the MFMAs of a CU never overlap: time per MFMA ~ t_MFMA +
// LDS-read bytes / 256 B/clk + LDS-DMA bytes / 45 B/clk (48-58 % MFMA-busy; hipBLASLt reaches 1.36-1.50 PFLOP/s on the
// same shapes, gemm2 0.90-1.17).  Here group A (waves 0-3, one per SIMD, the upper 128 rows of the tile) and group B
// (waves 4-7, the lower 128 rows) run the schedule
//      phase 2k    : A loads fragments of K-step k          | B multiplies K-step k-1
//      phase 2k+1  : A multiplies K-step k                  | B loads fragments of K-step k
// with one workgroup barrier per phase: while one wave of a SIMD streams 64 MFMAs the other one does all its LDS and
// DMA work, so the pipe is fed continuously by construction.
//
// LDS = five 32-KiB sub-stages (the whole 160 KiB): K-step k is two sub-tiles, X_k = A rows 0..127 | W rows 0..127 and
// Y_k = A rows 128..255 | W rows 128..255 (128-byte rows = whole cache lines per DMA row, chunk ^= (row>>1)&7 swizzle),
// sub-tile u = 2k + part lives in sub-stage u % 5.  A sub-tile is read in two consecutive phases (A, then B); with five
// sub-stages X_{k+2} can be requested by group B in phase 2k+1 and Y_{k+2} by group A in phase 2k+2 - every wave issues
// 8 DMA pieces per LOAD phase and every piece has two or three phases to land before its first reader.
// Fragment mapping, W as the MFMA A operand and the fused epilogues are those of gemm2.hip (gemm_args.h: tile_epilogue).
#include "gemm_args.h"
#include <stdlib.h>

// The timing variants of this kernel (no DMA / no fragment reads / register-staged loads / cache-hot source: WRONG
// results, measurement only) are not in this file: tools/timing_variants/gemm4_dbg.patch, built by
// tools/build_timing_variant.py into a library of its own that the product never loads.

namespace svr2 {
namespace {


constexpr int BM4 = 256, BN4 = 256, BK4 = 64, NT4 = 512;
constexpr int SUB4 = 32 * 1024;                           // one sub-stage: 128 A rows + 128 W rows of 128 bytes
constexpr int NSUB4 = 5;
constexpr int LDS4 = NSUB4 * SUB4;                        // 163 840 = all of the CU's LDS
constexpr int EPI_LD4 = BN4 * 2 + 16;
static_assert(BM4 * EPI_LD4 <= LDS4, "epilogue tile must fit");

template <int EPI, bool GATHER>
__global__ __launch_bounds__(NT4, 2) void gemm4_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wg = wave & 3;             // group A = 0 (rows 0..127), B = 1 (rows 128..255)
    const int wm = grp, wn = wg;
    int tm, tn;
    tile_coords(a, tm, tn);
    const int m0 = tm * BM4, n0 = tn * BN4;
    const int nk = a.K / BK4;

    // ---- DMA: group g streams sub-tile part g (B: X, A: Y); 32 pieces of 8 rows x 128 bytes per sub-tile, wave wg moves
    //      pieces wg*8 .. +8: waves 0,1 of the group the A rows, waves 2,3 the W rows.
    const int lrow = lane >> 3, lslot = lane & 7;
    const int part = grp == 0 ? 1 : 0;                    // sub-tile part this wave's group streams
    const bool isA = wg < 2;
    const int prow0 = (wg & 1) * 64 + lrow;               // first local row (of 128) this lane moves
    const char* zsrc = (const char*)a.zero + lslot * 16;
    // Fast path (dense rows, no row of the part past M): piece i reads rows base + i*8 .. +8, so its address is a
    // wave-uniform base (SGPRs) plus one of two per-lane offsets (the swizzle alternates with i) - one scalar add and the
    // DMA instruction per piece.  The TA accepts an LDS-DMA piece every ~45 cycles per CU (64 pieces per K-step):
    // anything on top of that in the issuing waves lengthens the LOAD phase.  Slow path: gathered rows / ragged tile.
    const long long ld = isA ? a.lda : (long long)a.K;
    const char* ubase = isA ? (const char*)(a.A + (long long)(m0 + part * 128 + (wg & 1) * 64) * ld)
                            : (const char*)(a.W + (long long)(n0 + part * 128 + (wg & 1) * 64) * ld);     // wave-uniform
    const long long ustep = 8 * ld * 2;                                                                  // bytes per piece
    const unsigned voff0 = (unsigned)(lrow * ld * 2) + ((lslot ^ ((prow0 >> 1) & 7)) << 4);              // even pieces
    const unsigned voff1 = (unsigned)(lrow * ld * 2) + ((lslot ^ (((prow0 >> 1) + 4) & 7)) << 4);        // odd pieces
    const bool fast = !isA || (!GATHER && m0 + part * 128 + (wg & 1) * 64 + 64 <= a.M);
    // gathered rows (proj_out reads its A rows through the window-reverse permutation): the rows of a lane's eight pieces
    // are the same for every K-step - looked up once (an index load in front of every DMA costs 3x the whole kernel)
    int arow[GATHER ? 8 : 1];
    if (GATHER && isA) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + part * 128 + prow0 + i * 8;
            arow[i] = m < a.M ? a.a_rows[m] : -1;
        }
    }
    auto issue_sub = [&](int k) {                         // sub-tile (k, part) -> sub-stage (2k + part) % 5
        char* dst = smem + ((2 * k + part) % NSUB4) * SUB4 + (isA ? 0 : 16 * 1024) + (wg & 1) * 8 * 1024;
        if (fast) {
            const char* ub = ubase + k * (BK4 * 2);
#pragma unroll
            for (int i = 0; i < 8; ++i) glds16(ub + i * ustep + ((i & 1) ? voff1 : voff0), dst + i * 1024);
            return;
        }
        if (GATHER) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = prow0 + i * 8;              // local row inside the 128-row part
                const char* src = (const char*)(a.A + (long long)arow[i] * a.lda) + k * (BK4 * 2) + ((lslot ^ ((r >> 1) & 7)) << 4);
                glds16(arow[i] >= 0 ? src : zsrc, dst + i * 1024);
            }
        } else {
#pragma unroll 1
            for (int i = 0; i < 8; ++i) {
                const int r = prow0 + i * 8;
                const int m = m0 + part * 128 + r;
                const char* src = (const char*)(a.A + (long long)m * a.lda) + k * (BK4 * 2) + ((lslot ^ ((r >> 1) & 7)) << 4);
                glds16(m < a.M ? src : zsrc, dst + i * 1024);
            }
        }
    };

    // ---- fragments: lane = (row l&15 of a 16-row tile, 8-element k group l>>4); A rows from part wm, W rows from part wn/2
    const int frow = lane & 15, fq = lane >> 4;
    const int fswz = (frow >> 1) & 7;
    const int a_off = frow * 128;                                        // + i*16*128 inside the A half of sub-tile wm
    const int w_off = 16 * 1024 + ((wn & 1) * 64 + frow) * 128;          // + j*16*128 inside the W half of sub-tile wn/2
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[2][8], bfr[2][4];
    auto load_frags = [&](int k) {
        const char* sA = smem + ((2 * k + wm) % NSUB4) * SUB4 + a_off;
        const char* sW = smem + ((2 * k + (wn >> 1)) % NSUB4) * SUB4 + w_off;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((fq + 4 * ks) ^ fswz) << 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[ks][j] = *(const bf16x8*)(sW + j * 16 * 128 + coff);
#pragma unroll
            for (int i = 0; i < 8; ++i) af[ks][i] = *(const bf16x8*)(sA + i * 16 * 128 + coff);
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)   // W fragment is the MFMA A operand: D[row = n][col = m]
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
    };
    // phase boundary: this wave's DMA pieces have landed (they are first read two or three phases after issue, but
    // waiting here keeps the count exact), nothing moves across
    auto boundary = [&](bool drain) {
        __builtin_amdgcn_sched_barrier(0);
        if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: K-steps 0 and 1 (each group its own part), then the schedule above
    issue_sub(0);
    if (nk > 1) issue_sub(1);
    boundary(true);
    if (grp == 0) {
        load_frags(0);                                    // phase 0
        boundary(false);
        for (int k = 0; k < nk; ++k) {
            compute();                                    // phase 2k+1
            boundary(true);
            if (k + 1 < nk) load_frags(k + 1);
            if (k + 2 < nk) issue_sub(k + 2);   // phase 2k+2: Y_{k+2} into the sub-stage X_k just left
            boundary(false);
        }
    } else {
        boundary(false);                                  // phase 0: nothing to do yet
        for (int k = 0; k < nk; ++k) {
            load_frags(k);
            if (k + 2 < nk) issue_sub(k + 2);   // phase 2k+1: X_{k+2} into the sub-stage Y_{k-1} left
            boundary(false);
            compute();                                    // phase 2k+2
            boundary(true);
        }
    }

    // ---- epilogue: (acc + bias) -> bf16 tile in LDS -> fused epilogues on 8 consecutive columns
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
                *(bf16x4*)(smem + (wm * 128 + i * 16 + frow) * EPI_LD4 + (wn * 64 + j * 16 + 4 * fq) * 2) = o;
            }
    }
    __syncthreads();
    tile_epilogue<A_DENSE, EPI, BM4, BN4, NT4, EPI_LD4>(a, smem, tm, n0, tid);
}

template <int EPI, bool GATHER>
int launch4g(KArgs a, hipStream_t s) {
    SVR2_SET_MAX_LDS((gemm4_kernel<EPI, GATHER>), LDS4, "gemm4");
    a.nbm = (a.M + BM4 - 1) / BM4;
    a.nbn = a.N / BN4;
    hipLaunchKernelGGL((gemm4_kernel<EPI, GATHER>), dim3(a.nbm * a.nbn), dim3(NT4), LDS4, s, a);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

template <int EPI>
int launch4(const KArgs& a, hipStream_t s) {
    return a.a_rows ? launch4g<EPI, true>(a, s) : launch4g<EPI, false>(a, s);
}

}  // namespace

// Shape contract (checked by the caller, gemm.hip): dense A (optionally row-gathered), W rows [N][K], N % 256 == 0,
// K % 64 == 0, K / 64 >= 2.
int gemm4_launch(int epi, KArgs a, hipStream_t s) {
    switch (epi) {
        case SVR2_EPI_BIAS: return launch4<SVR2_EPI_BIAS>(a, s);
        case SVR2_EPI_GATE_RES: return launch4<SVR2_EPI_GATE_RES>(a, s);
        case SVR2_EPI_SWIGLU: return launch4<SVR2_EPI_SWIGLU>(a, s);
        case SVR2_EPI_PIXSHUF3D: return launch4<SVR2_EPI_PIXSHUF3D>(a, s);
        case SVR2_EPI_GELU: return launch4<SVR2_EPI_GELU>(a, s);
    }
    svr2_set_error("gemm4: unknown epilogue %d", epi);
    return SVR2_E_ARG;
}

}  // namespace svr2
