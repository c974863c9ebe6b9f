// Halo-resident implicit-GEMM kernel for the stride-1 3x3 (x kt) causal convs of the VAE - the dominant kernel
// of the path (two thirds of a 4K step).
//
// Why it exists (measured on MI355X, round 1): in gemm2.hip every K-step (one filter tap of one 64-channel
// slice) re-stages a 256-pixel A tile through the global->LDS DMA path although the nine spatial taps of a slice
// read the same pixels shifted by one; with the A loads disabled that loop runs 25 % faster.  Here a workgroup
// owns a 16 x PW pixel patch of one output frame and keeps the (16+2) x (PW+2) input halo of the current channel
// slice resident in LDS: the nine taps are nine *shifted reads* of that image, and the next slice's halo is
// streamed into a second buffer while the current one is multiplied.  Per workgroup and K-step the DMA count
// drops from 64 (32 A + 32 W) to 32 + 4.6 (BN = 256, BK 64) and from 24 to 8 + 4.3 (BN = 128, BK 32).
//
//   * LDS image of the halo: row = halo pixel (hy * (PW+2) + hx), BK channels; 16-byte chunk c of that row sits
//     at slot c ^ f(hx) (applied on the DMA source address), f = (hx>>1)&7 for 128-byte rows, (hx>>2)&3 for 64.
//   * Shifted reads stay bank-conflict-free for EVERY shift because the 16 pixels of an MFMA fragment are
//     assigned to lanes through the permutation pi = {0,2,4,6, 1,3,...,15, 8,10,12,14}: the hardware services a
//     ds_read_b128 in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...; with pi the lanes of a group
//     that read chunk q sit on even pixel offsets and those that read chunk q^1 on odd ones, so the 16 lanes hit
//     16 distinct 16-byte bank slots whatever the base row is.  The epilogue undoes pi when it writes the tile.
//   * Two geometries: <PW 16, BN 256, BK 64> (256 pixels x 256 channels, 128-byte rows, 2-deep W ring) and
//     <PW 32, BN 128, BK 32> (512 pixels x 128 channels, 64-byte rows, 3-deep W ring with counted vmcnt, W read
//     from the pre-tiled, pre-swizzled layout SVR2_WL_TILED_128x32 so that every W DMA is one linear KiB).
//     Both: 8 waves, wave tile 128 pixels x 64 channels = 8 x 4 MFMA 16x16x32 (W is the MFMA A operand).
#include "gemm_args.h"
#include <stdlib.h>

namespace svr2 {
namespace {

template <int PW, int BN, int BK> struct GeoH {
    static constexpr int PH = 16, BM = PH * PW, HP = PW + 2, HROWS = (PH + 2) * HP;
    static constexpr int ROWB = BK * 2, RPI = 1024 / ROWB, CPR = ROWB / 16;
    static constexpr int NW = (BM / 128) * (BN / 64), NT = NW * 64;
    static constexpr int WSTAGES = BK == 64 ? 2 : 3;
    // DMA roles: waves 0-3 stream W (NB pieces each per K-step), waves 4-7 stream the next slice's halo (PPJ pieces
    // each on the first HT taps of a slice).  vmcnt is in order per wave, so a wave that streamed both would have to
    // land its halo pieces (strided 64/128-byte gathers that miss L2) before its next W wait, one or two K-steps
    // later; the halo waves only wait at the slice boundary, up to nine K-steps after issue.
    static constexpr int NHI = (HROWS + RPI - 1) / RPI;                   // halo pieces per slice
    static constexpr int PPJ = 2;
    static constexpr int HT = (NHI + 4 * PPJ - 1) / (4 * PPJ);            // taps that carry halo DMAs
    static constexpr int HALO_BYTES = NHI * 1024;
    static constexpr int W_BYTES = BN * ROWB;
    static constexpr int NB = W_BYTES / 1024 / 4;
    static constexpr int RING = 2 * HALO_BYTES + WSTAGES * W_BYTES;
    static constexpr int EPI_LD = BN * 2 + 16;
    static constexpr int EPI_BYTES = BM * EPI_LD;
    static constexpr int LDS = RING > EPI_BYTES ? RING : EPI_BYTES;
    static_assert(NW == 8 && HT <= 9 && LDS <= 160 * 1024, "geometry");
};

// W tile rows: the swizzle of gemm2.hip.
template <int ROWB>
__device__ __forceinline__ int wswz(int row) {
    return ROWB == 64 ? ((0x78 >> (2 * ((row >> 2) & 3))) & 3) : ((row >> 1) & 7);
}
// Halo rows: swizzle by the x position inside the halo line only, so that a shift by whole lines (the dy taps, the
// 8 patch rows of a wave) is a constant byte offset - one base register per dx serves all 8 fragment reads.
// Conflict-freeness needs distinct values on the 8 same-parity pixels of a fragment: hx steps by 2 (128-byte rows,
// two rows per 256-byte bank line) resp. by 4 within a bank class (64-byte rows, four rows per bank line).
template <int ROWB>
__device__ __forceinline__ int hswz(int hx) {
    return ROWB == 64 ? ((hx >> 2) & 3) : ((hx >> 1) & 7);
}

template <int PW, int BN, int BK>
__global__ __launch_bounds__(512, 2) void conv_halo_kernel(const KArgs a) {
    using G = GeoH<PW, BN, BK>;
    constexpr int HP = G::HP, ROWB = G::ROWB, RPI = G::RPI, CPR = G::CPR, NB = G::NB, HT = G::HT, PPJ = G::PPJ;
    constexpr int WST = G::WSTAGES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo0 = smem;
    char* const wring = smem + 2 * G::HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    tile_coords(a, tm, tn);
    const int per_frame = a.cg.tiles_x * a.cg.tiles_y;
    const int t = tm / per_frame, rem = tm - t * per_frame;
    const int ty = rem / a.cg.tiles_x, tx = rem - ty * a.cg.tiles_x;
    const int y0 = ty * G::PH, x0 = tx * PW, n0 = tn * BN;
    const int Cin = a.cg.Cin, Wi = a.cg.Wi, Hi = a.cg.Hi;

    // ---- halo loader (waves 4-7): on tap j (< HT) wave 4+hw moves pieces q = (j*PPJ + u)*4 + hw, u < PPJ;
    //      piece q = halo rows [q*RPI, (q+1)*RPI)
    const int lrow = lane / CPR, lslot = lane % CPR;
    const char* zsrc = (const char*)a.zero + lslot * 16;
    const long long frame_bytes = (long long)Hi * Wi * Cin * 2;
    const long long org = (((long long)(y0 - 1) * Wi + (x0 - 1)) * Cin) * 2;
    const int ncs = Cin / BK;                      // channel slices per temporal tap
    const int nsl = a.cg.kt * ncs;                 // halo slices in the K loop
    auto slice_base = [&](int tap_t, int cs) -> const char* {
        const int it = t + tap_t;                  // stride 1 in time
        const bool head = it < a.cg.npad;
        const char* base = (head && a.cg.carry) ? (const char*)a.cg.carry : (const char*)a.A;
        const int fidx = head ? (a.cg.carry ? it : 0) : it - a.cg.npad;
        return base + fidx * frame_bytes + org + cs * (BK * 2);
    };
    auto issue_halo = [&](int j, const char* sb, char* hbuf) {
#pragma unroll
        for (int u = 0; u < PPJ; ++u) {
            const int q = (j * PPJ + u) * 4 + (wave - 4);   // wave-uniform
            if (q >= G::NHI) continue;
            const int r = q * RPI + lrow;
            const int hy = r / HP, hx = r - hy * HP;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool ok = (r < G::HROWS) & ((unsigned)iy < (unsigned)Hi) & ((unsigned)ix < (unsigned)Wi);
            const int off = ((hy * Wi + hx) * Cin) * 2 + ((lslot ^ hswz<ROWB>(hx)) << 4);
            glds16(ok ? sb + off : zsrc, hbuf + q * 1024);
        }
    };

    // ---- W loader (waves 0-3): NB 1-KiB pieces per wave per K-step; uniform running base + per-lane 32-bit offsets
    const char* wk;               // uniform: start of this step's W columns / tile
    unsigned woff[2];             // per lane: even / odd piece (their swizzles differ in one bit)
    long long wstep, wpiece;      // uniform strides: per K-step, per piece
    if (a.cg.wtiled) {            // [tn][K/BK][BN rows][BK] pre-swizzled image: one linear KiB per piece
        wk = (const char*)a.W + (long long)tn * (a.K / BK) * G::W_BYTES + wave * (NB * 1024);
        woff[0] = woff[1] = lane * 16;
        wstep = G::W_BYTES;
        wpiece = 1024;
    } else {                      // rows [N_pad][K]
        wk = (const char*)(a.W + (long long)(n0 + wave * NB * RPI) * a.K);
        woff[0] = (unsigned)(lrow * a.K * 2) + ((lslot ^ wswz<ROWB>(wave * NB * RPI + lrow)) << 4);
        woff[1] = (unsigned)(lrow * a.K * 2) + ((lslot ^ wswz<ROWB>(wave * NB * RPI + RPI + lrow)) << 4);
        wstep = BK * 2;
        wpiece = (long long)RPI * a.K * 2;
    }
    auto issue_w = [&](int stage) {
        char* sB = wring + stage * G::W_BYTES + wave * (NB * 1024);
#pragma unroll
        for (int i = 0; i < NB; ++i) glds16(wk + i * wpiece + woff[i & 1], sB + i * 1024);
        wk += wstep;
    };

    // ---- MFMA fragment coordinates
    constexpr int WPX = PW / 16;                   // waves along x
    const int wm = wave / (BN / 64), wn = wave % (BN / 64);
    const int wpy0 = (wm / WPX) * 8, wpx0 = (wm % WPX) * 16;
    const int frow = lane & 15, fq = lane >> 4;
    const int pix = frow < 4 ? 2 * frow : (frow < 12 ? 2 * frow - 7 : 2 * frow - 16);   // pi(frow)
    const int hxl = wpx0 + pix;                    // halo x of this lane's pixel under tap dx = 0
    int rbase = wpy0 * HP + hxl;                   // its halo row for patch row wpy0, tap (0,0)
    const int wfs = wswz<ROWB>(frow);
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: whole halo of slice 0, W of step 0 (and 1)
    const int nk = nsl * 9;
    const bool wrole = wave < 4;
    if (wrole) {
        issue_w(0);
        if (WST == 3) issue_w(1);
    } else {
        const char* sb = slice_base(0, 0);
#pragma unroll
        for (int j = 0; j < HT; ++j) issue_halo(j, sb, halo0);
    }
    // ---- K loop: step k = (slice sl, tap j); 9 taps per slice
    int tap_t = 0, cs = 0, hb = 0, sl = 0, j = 0, dy = 0, dx = 0, wst = 0;
    const char* sbn = nullptr;                     // next slice's halo source
    for (int k = 0; k < nk; ++k) {
        const bool last = sl + 1 == nsl;
        if (j == 0 && !last) {
            if (++cs == ncs) { cs = 0; ++tap_t; }
            sbn = slice_base(tap_t, cs);
        }
        // this step's W tile and (j == 0) its halo have landed for every wave, and every wave is done reading
        // what is overwritten below.  W waves, 3-deep ring: the tile issued in the previous step stays in flight
        // across the barrier.  Halo waves wait only when the slice changes.
        if (!(a.cg.dbg & 8)) {
            if (wrole) {
                if (WST == 3 && k + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (j == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if (!(a.cg.dbg & 1)) asm volatile("s_barrier" ::: "memory");
        if (wrole) {
            if (!(a.cg.dbg & 2)) {
                if (WST == 3) {
                    if (k + 2 < nk) issue_w(wst == 0 ? 2 : wst - 1);      // (k + 2) % 3
                } else {
                    if (k + 1 < nk) issue_w(wst ^ 1);
                }
            }
        } else if (j < HT && !last && !(a.cg.dbg & 4)) {
            issue_halo(j, sbn, halo0 + (hb ^ 1) * G::HALO_BYTES);
        }
        char* const hcur = halo0 + hb * G::HALO_BYTES;

        const char* sB = wring + wst * G::W_BYTES + (wn * 64 + frow) * ROWB;
        const char* sA0 = hcur + (rbase + dy * HP + dx) * ROWB;
        const int hs = hswz<ROWB>(hxl + dx);
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 bfr[4], af[8];
            const int wc = ((fq + 4 * ks) ^ wfs) << 4;
            const char* sA = sA0 + (((fq + 4 * ks) ^ hs) << 4);
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) bfr[jn] = *(const bf16x8*)(sB + jn * 16 * ROWB + wc);
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *(const bf16x8*)(sA + i * (HP * ROWB));
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn)   // W fragment is the MFMA A operand: D[row = n][col = pixel]
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[jn], af[i], acc[i][jn], 0, 0, 0);
        }
        __builtin_amdgcn_iglp_opt(1);

        wst = WST == 3 ? (wst == 2 ? 0 : wst + 1) : (wst ^ 1);
        if (++dx == 3) { dx = 0; ++dy; }
        if (++j == 9) { j = 0; dy = 0; ++sl; hb ^= 1; }
    }

    // ---- epilogue: (acc + bias) -> bf16 tile in LDS (tile row = patch pixel py*PW + px) -> rows of 8 channels
    asm volatile("s_barrier" ::: "memory");
    {
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int n = n0 + wn * 64 + j * 16 + 4 * fq;
            if (a.bias) {
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
                const int prow = (wpy0 + i) * PW + wpx0 + pix;
                *(bf16x4*)(smem + prow * G::EPI_LD + (wn * 64 + j * 16 + 4 * fq) * 2) = o;
            }
    }
    __syncthreads();
    for (int c = tid; c < G::BM * (BN / 8); c += G::NT) {
        const int row = c / (BN / 8), cc = c % (BN / 8);
        const int y = y0 + row / PW, x = x0 + row % PW, n = n0 + cc * 8;
        if (y >= a.cg.Ho || x >= a.cg.Wo || n >= a.N) continue;
        const long long m = ((long long)t * a.cg.Ho + y) * a.cg.Wo + x;
        const bf16x8 tv = *(const bf16x8*)(smem + row * G::EPI_LD + cc * 16);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf2f(tv[e]);
        if (a.resid) {
            const bf16x8 rv = *(const bf16x8*)(a.resid + m * a.ldr + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bf2f(rv[e]);
        }
        const long long dst = m * a.ldc + n;
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

template <int PW, int BN, int BK>
int launch_halo(KArgs a, hipStream_t s) {
    using G = GeoH<PW, BN, BK>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_halo_kernel<PW, BN, BK>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        SVR2_CHECK(e == hipSuccess, SVR2_E_LAUNCH, "conv_halo: cannot reserve %d bytes of LDS", G::LDS);
        attr_set = true;
    }
    a.cg.patch = 2;
    { const char* e = getenv("SVR2_HALO_DBG"); a.cg.dbg = e ? atoi(e) : 0; }
    a.cg.tiles_x = (a.cg.Wo + PW - 1) / PW;
    a.cg.tiles_y = (a.cg.Ho + G::PH - 1) / G::PH;
    a.nbm = a.cg.To * a.cg.tiles_x * a.cg.tiles_y;
    a.nbn = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL((conv_halo_kernel<PW, BN, BK>), dim3(a.nbm * a.nbn), dim3(G::NT), G::LDS, s, a);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

}  // namespace

// Shape contract (checked by the caller, gemm.hip): kh = kw = 3, stride 1, zero padding 1 on every spatial side
// (Ho = Hi, Wo = Wi), any kt.  wtiled: Cin % 32 == 0, W in SVR2_WL_TILED_128x32; else Cin % 64 == 0, N % 256 == 0.
int conv_halo_launch(KArgs a, hipStream_t s) {
    if (a.cg.wtiled) return launch_halo<32, 128, 32>(a, s);
    return launch_halo<16, 256, 64>(a, s);
}

}  // namespace svr2
