// Halo-resident implicit-GEMM kernel for the stride-1 3x3 (x kt) causal convs of the VAE - the dominant kernel
// of the path (two thirds of a 4K step).
//
// Why it exists (measured on MI355X, round 1): the tap-staged kernels (gemm.hip, gemm2.hip) re-stage a 256-pixel
// A tile through the global->LDS DMA path for every filter tap although the nine spatial taps of a channel slice
// read the same pixels shifted by one, and they synchronise the workgroup once per tap.  Here a workgroup owns a
// 16 x PW pixel patch of one output frame and keeps the (16+2) x (PW+2) input halo of the current 32-channel
// slice resident in LDS:
//   * the nine taps are nine *shifted reads* of that image; the next slice's halo is streamed into a second
//     buffer while the current one is multiplied;
//   * one K super-step = the three ky taps of one kx column: a wave loads its ten activation fragments (patch
//     rows 0..9 of its 8-row band) once and uses each for up to three taps, and the workgroup meets at ONE barrier
//     per 96 MFMAs per wave (the tap-staged kernels: one per 32 or 64);
//   * W comes from the pre-tiled, pre-swizzled layout SVR2_WL_TILED_128x32 (include/seedvr2_hip.h): the three ky
//     tiles of a super-step are 24 contiguous KiB per 128 output channels, every DMA piece is one linear KiB.
//
// LDS image of the halo: row = halo pixel (hy * (PW+2) + hx), 32 channels = 64 bytes; the 16-byte chunk c of a
// row sits at slot c ^ ((hx >> 2) & 3) (applied on the DMA source address).  Swizzling by hx only makes a shift by
// whole halo lines (ky taps, the rows of a wave's band) a constant byte offset, so one base register serves all
// ten fragment reads of a super-step.  Shifted reads stay bank-conflict-free for EVERY shift because the 16
// pixels of an MFMA fragment are assigned to lanes through pi = {0,2,4,6, 1,3,...,15, 8,10,12,14}: the hardware
// services a ds_read_b128 in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...; with pi the lanes of a
// group that read chunk q sit on even pixel offsets and those that read chunk q^1 on odd ones, so the 16 lanes hit
// 16 distinct 16-byte bank slots whatever the base row is.  The epilogue undoes pi when it writes the tile.
//
// DMA roles: waves 0-3 stream W, waves 4-7 stream the halo.  vmcnt is in order per wave, so a wave that streamed
// both would have to land its halo pieces (strided 64-byte gathers that often miss L2) before its next W wait; the
// halo waves only wait when the slice changes.
//
// Two geometries, both 8 waves with a 128-pixel x 64-channel wave tile (8 x 4 MFMA 16x16x32, W is the MFMA A
// operand): <PW 16, BN 256> (256 pixels x 256 channels) and <PW 32, BN 128> (512 pixels x 128 channels).
#include "gemm_args.h"
#include <stdlib.h>

// The timing variant without epilogue (WRONG results, measurement only) lives in tools/timing_variants/conv_halo_dbg.patch
// (tools/build_timing_variant.py builds it into a library of its own that the product never loads).

namespace svr2 {
namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PW, int BN> struct GeoH {
    static constexpr int PH = 16, BM = PH * PW, HP = PW + 2, HROWS = (PH + 2) * HP;
    static constexpr int BK = 32, ROWB = 64, RPI = 16;                  // 64-byte LDS rows, 16 rows per DMA piece
    static constexpr int NT = 512;
    static constexpr int NHI = (HROWS + RPI - 1) / RPI;                 // halo pieces per slice
    static constexpr int HALO_BYTES = NHI * 1024;
    static constexpr int HPW = (NHI + 3) / 4;                           // pieces per halo wave per slice
    static constexpr int HPS = (HPW + 1) / 2;                           // ... issued on each of the super-steps 0, 1
    static constexpr int TAPB = BN * ROWB;                              // W bytes per tap
    static constexpr int WS_BYTES = 3 * TAPB;                           // W bytes per super-step (three ky taps)
    static constexpr int JN = BN >= 64 ? 4 : BN / 16;                   // 16-channel MFMA tiles per wave
    static constexpr int NWN = BN >= 64 ? BN / 64 : 1;                  // waves along the channels
    static constexpr int RING = 2 * HALO_BYTES + 2 * WS_BYTES;
    static constexpr int EPI_LD = BN * 2 + 16;
    static constexpr int EPI_BYTES = BM * EPI_LD;
    static constexpr int LDS = RING > EPI_BYTES ? RING : EPI_BYTES;
    static_assert((BM / 128) * NWN == 8 && LDS <= 160 * 1024, "geometry");
};

// Epilogue shared by both kernels: (acc + bias) -> bf16 tile in LDS (tile row = patch pixel py*PW + px) -> rows of 8
// channels (+ residual, + GroupNorm partial sums).  Call with every wave past its last LDS read of the K loop.
template <int PW, int BN, int MI, int JN>
__device__ __forceinline__ void halo_epilogue(const KArgs& a, char* smem, f32x4 (&acc)[MI][JN], int tid, int t,
                                              int per_frame, int rem, int y0, int x0, int n0, int cb, int wpy0,
                                              int wpx0, int pix, int fq) {
    using G = GeoH<PW, BN>;
    //      The residual chunks this thread will add are requested first, so that their HBM latency runs under the
    //      accumulator -> LDS round trip instead of once per store (measured: resid convs were 6 % slower).
    constexpr int EIT = G::BM * (BN / 8) / G::NT;      // 16-byte chunks per thread (16, or 4 for BN = 16)
    bf16x8 rres[EIT];
    if (a.resid) {
#pragma unroll
        for (int k = 0; k < EIT; ++k) {
            const int c = tid + k * G::NT;
            const int row = c / (BN / 8), cc = c % (BN / 8);
            const int y = y0 + row / PW, x = x0 + row % PW, n = n0 + cc * 8;
            const bool ok = y < a.cg.Ho && x < a.cg.Wo && n < a.N;
            const long long m = ((long long)t * a.cg.Ho + y) * a.cg.Wo + x;
            rres[k] = ok ? *(const bf16x8*)(a.resid + m * a.ldr + n) : bf16x8{};
        }
    }
    asm volatile("s_barrier" ::: "memory");
    {
        f32x4 bv[JN];
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int n = n0 + cb + j * 16 + 4 * fq;
            if (a.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[j][r] = (n + r < a.N) ? a.bias[n + r] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < JN; ++j) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[i][j][r] + bv[j][r]);
                const int prow = (wpy0 + i) * PW + wpx0 + pix;
                *(bf16x4*)(smem + prow * G::EPI_LD + (cb + j * 16 + 4 * fq) * 2) = o;
            }
    }
    __syncthreads();
    // optional GroupNorm statistics of the tile (of the rounded bf16 outputs, the values the next norm reads): a
    // thread always handles the same 8 channels (BN/8 divides the block size), so it keeps 8 (sum, sumsq) pairs
    float st_s[8], st_q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.f;
#pragma unroll
    for (int k = 0; k < EIT; ++k) {
        const int c = tid + k * G::NT;
        const int row = c / (BN / 8), cc = c % (BN / 8);
        const int y = y0 + row / PW, x = x0 + row % PW, n = n0 + cc * 8;
        if (y >= a.cg.Ho || x >= a.cg.Wo || n >= a.N) continue;
        const long long m = ((long long)t * a.cg.Ho + y) * a.cg.Wo + x;
        const bf16x8 tv = *(const bf16x8*)(smem + row * G::EPI_LD + cc * 16);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf2f(tv[e]);
        if (a.resid) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bf2f(rres[k][e]);
        }
        const long long dst = m * a.ldc + n;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
        if (n + 8 <= a.N) {
            *(bf16x8*)(a.C + dst) = o;
        } else {
            for (int e = 0; e < a.N - n; ++e) a.C[dst + e] = o[e];
        }
        if (a.stats) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float r = bf2f(o[e]);
                st_s[e] += r;
                st_q[e] += r * r;
            }
        }
    }
    if (a.stats) {   // partial[frame][tile][2][N], folded over tiles by svr2_groupnorm_finalize (fixed order: deterministic)
        constexpr int NV = BN / 8, RPT = G::NT / NV;
        __syncthreads();                                   // the epilogue tile has been read
        float* red = (float*)smem;                         // [NT][16]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[tid * 16 + e] = st_s[e];
            red[tid * 16 + 8 + e] = st_q[e];
        }
        __syncthreads();
        for (int i = tid; i < NV * 16; i += G::NT) {
            const int v = i >> 4, e = i & 15;
            float acc2 = 0.f;
            for (int r = 0; r < RPT; ++r) acc2 += red[(r * NV + v) * 16 + e];
            const int n = n0 + v * 8 + (e & 7);
            if (n < a.N) a.stats[(((long long)t * per_frame + rem) * 2 + (e >> 3)) * a.N + n] = acc2;
        }
    }
}

template <int PW, int BN>
__global__ __launch_bounds__(512, 2) void conv_halo_kernel(const KArgs a) {
    using G = GeoH<PW, BN>;
    constexpr int HP = G::HP, ROWB = G::ROWB, RPI = G::RPI, BK = G::BK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo0 = smem;
    char* const wring = smem + 2 * G::HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool wrole = wave < 4;
    int tm, tn;
    tile_coords(a, tm, tn);
    const int per_frame = a.cg.tiles_x * a.cg.tiles_y;
    // patch == 3: frame-fastest tile order (the To tiles of one patch are neighbours in an XCD's run, so the temporal
    // taps they share are re-read from that XCD's L2 instead of HBM)
    const int t = a.cg.patch == 3 ? tm % a.cg.To : tm / per_frame;
    const int rem = a.cg.patch == 3 ? tm / a.cg.To : tm - t * per_frame;
    const int ty = rem / a.cg.tiles_x, tx = rem - ty * a.cg.tiles_x;
    const int y0 = ty * G::PH, x0 = tx * PW, n0 = tn * BN;
    const int Cin = a.cg.Cin, Wi = a.cg.Wi, Hi = a.cg.Hi;
    const int ncs = Cin / BK;                      // channel slices per temporal tap
    const int nsl = a.cg.kt * ncs;                 // halo slices in the K loop
    const int nss = nsl * 3;                       // super-steps

    // ---- halo loader (waves 4-7): wave 4+hw moves pieces q = u*4 + hw, u < HPW; piece q = halo rows [16q, 16q+16)
    const int lrow = lane >> 2, lslot = lane & 3;
    const char* zsrc = (const char*)a.zero + lslot * 16;
    const long long frame_bytes = (long long)Hi * Wi * Cin * 2;
    const long long org = (((long long)(y0 - 1) * Wi + (x0 - 1)) * Cin) * 2;
    auto slice_base = [&](int tap_t, int cs) -> const char* {
        const int it = t + tap_t;                  // stride 1 in time
        const bool head = it < a.cg.npad;
        const char* base = (head && a.cg.carry) ? (const char*)a.cg.carry : (const char*)a.A;
        const int fidx = head ? (a.cg.carry ? max(it - a.cg.crep, 0) : 0) : it - a.cg.npad;
        return base + fidx * frame_bytes + org + cs * (BK * 2);
    };
    auto issue_halo = [&](int u0, int u1, const char* sb, char* hbuf) {
        for (int u = u0; u < u1; ++u) {
            const int q = u * 4 + (wave - 4);      // wave-uniform
            if (q >= G::NHI) break;
            const int r = q * RPI + lrow;
            const int hy = r / HP, hx = r - hy * HP;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool ok = (r < G::HROWS) & ((unsigned)iy < (unsigned)Hi) & ((unsigned)ix < (unsigned)Wi);
            const int off = ((hy * Wi + hx) * Cin) * 2 + ((lslot ^ ((hx >> 2) & 3)) << 4);
            glds16(ok ? sb + off : zsrc, hbuf + q * 1024);
        }
    };

    // ---- W loader (waves 0-3): linear 1-KiB pieces.  A 128-channel tile column is 24 contiguous KiB per super-step
    //      (ky-major, eight 16-row KiB blocks per ky); wave w streams blocks [wsub*BPW, (wsub+1)*BPW) of every ky of
    //      column wcol  ->  LDS stage + ky*TAPB + wcol*8 KiB + block KiB.  BN = 16 reads only the first block (rows
    //      0..15) of every ky: three pieces, one each on waves 0..2.
    constexpr int WPC = BN >= 128 ? 4 / (BN / 128) : 4, BPW = BN >= 128 ? 8 / WPC : 1;
    const int wcol = BN >= 128 ? wave / WPC : 0, wsub = BN >= 128 ? wave % WPC : 0;
    const int tcol = BN >= 128 ? tn * (BN / 128) + wcol : (tn * BN) / 128;       // 128-channel tile column read
    const char* wsrc = (const char*)a.W + (long long)tcol * nss * (24 * 1024) + wsub * (BPW * 1024) +
                       (BN >= 128 ? 0 : ((tn * BN) % 128) * ROWB);
    const unsigned l16 = lane * 16;
    char* const wdst = wring + wcol * 8192 + wsub * (BPW * 1024);
    auto issue_w = [&](int stage) {
        if (BN >= 128) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int b = 0; b < BPW; ++b)
                    glds16(wsrc + (ky * 8192 + b * 1024) + l16, wdst + stage * G::WS_BYTES + ky * G::TAPB + b * 1024);
        } else if (wave < 3) {
            glds16(wsrc + wave * 8192 + l16, wring + stage * G::WS_BYTES + wave * G::TAPB);
        }
        wsrc += 24 * 1024;
    };

    // ---- MFMA fragment coordinates
    constexpr int WPX = PW / 16;                   // waves along x
    constexpr int JN = G::JN;
    const int wm = wave / G::NWN, wn = wave % G::NWN;
    const int wpy0 = (wm / WPX) * 8, wpx0 = (wm % WPX) * 16;
    const int frow = lane & 15, fq = lane >> 4;
    const int pix = frow < 4 ? 2 * frow : (frow < 12 ? 2 * frow - 7 : 2 * frow - 16);   // pi(frow)
    const int hxl = wpx0 + pix;                    // halo x of this lane's pixel under kx = 0
    const int rbase = wpy0 * HP + hxl;             // its halo row for the first row of the wave's band, tap (0,0)
    const int wfo = (wn * 64 + frow) * ROWB + ((fq ^ ((0x78 >> (2 * ((frow >> 2) & 3))) & 3)) << 4);
    f32x4 acc[8][JN];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: whole halo of slice 0, W of super-step 0
    if (wrole) issue_w(0);
    else issue_halo(0, G::HPW, slice_base(0, 0), halo0);

    // ---- K loop over super-steps (slice sl, column kx)
    int tap_t = 0, cs = 0, hb = 0, sl = 0, kx = 0, wst = 0;
    const char* sbn = nullptr;                     // next slice's halo source
    for (int ss = 0; ss < nss; ++ss) {
        const bool last = sl + 1 == nsl;
        if (kx == 0 && !last) {
            if (++cs == ncs) { cs = 0; ++tap_t; }
            sbn = slice_base(tap_t, cs);
        }
        // this super-step's W tiles and (kx == 0) its halo have landed for every wave, and every wave is done
        // reading what is overwritten below
        if (wrole || kx == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (wrole) {
            if (ss + 1 < nss) issue_w(wst ^ 1);
        } else if (kx < 2 && !last) {
            issue_halo(kx * G::HPS, kx ? G::HPW : G::HPS, sbn, halo0 + (hb ^ 1) * G::HALO_BYTES);
        }

        const char* sA = halo0 + hb * G::HALO_BYTES + (rbase + kx) * ROWB + ((fq ^ (((hxl + kx) >> 2) & 3)) << 4);
        const char* sB = wring + wst * G::WS_BYTES + wfo;
        bf16x8 af[10];
#pragma unroll
        for (int r = 0; r < 10; ++r) af[r] = *(const bf16x8*)(sA + r * (HP * ROWB));
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            bf16x8 bfr[JN];
#pragma unroll
            for (int jn = 0; jn < JN; ++jn) bfr[jn] = *(const bf16x8*)(sB + ky * G::TAPB + jn * 16 * ROWB);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jn = 0; jn < JN; ++jn)   // W fragment is the MFMA A operand: D[row = n][col = pixel]
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[jn], af[i + ky], acc[i][jn], 0, 0, 0);
        }
        __builtin_amdgcn_iglp_opt(1);

        wst ^= 1;
        if (++kx == 3) { kx = 0; ++sl; hb ^= 1; }
    }

    halo_epilogue<PW, BN, 8, JN>(a, smem, acc, tid, t, per_frame, rem, y0, x0, n0, wn * 64, wpy0, wpx0, pix, fq);
}

// ---------------------------------------------------------------------------------------------------------------------
// W-through-registers kernel.  The W fragments never touch LDS: every wave loads them straight from the tiled global
// layout into registers (one fully coalesced KiB per fragment), three taps ahead of their use.  Only the halo goes
// through LDS, so the workgroup synchronises once per channel slice (nine taps, 288 MFMAs per wave) and the LDS-DMA
// stream that the MFMA pipe was measured to wait for shrinks from 37-55 KiB to 7-13 KiB per super-step.
//   * Wave tile = one 16 x 16 pixel block x 32 channels (16 x 2 MFMA 16x16x32): waves are laid out (PW/16) x (BN/32),
//     so the 256-channel geometry loads every W fragment exactly once per workgroup (the texture-data return path was
//     93 % busy with 128 x 64 wave tiles, which load each fragment twice) and a wave holds the 18 activation fragments
//     of a kx column (all halo lines of its block; LDS reads are cheap now that W is gone from LDS).
//   * A halo wave's W loads queue behind its halo DMAs (vmcnt retires in order); its SIMD partner issues no halo DMA
//     and meets it only at the slice barrier, so it keeps the MFMA pipe fed meanwhile.
//   * The nine taps of a slice are straight-line code so that the compiler's counted vmcnt waits stay exact (a branch
//     between a load and its use makes it fall back to vmcnt(0)); sched_barrier pins each load two taps before its use.
//   * SPLIT (small grids, see conv_halo_split): blockIdx.x = tile * S + ks; the workgroup multiplies only the channel
//     slices [ks * nsl / S, (ks + 1) * nsl / S) and stores its fp32 accumulators as a register image; conv_halo_finish_kernel
//     adds the S images of a tile in the order ks = 0..S-1 (deterministic) and runs the shared epilogue.
template <int PW, int BN, bool SPLIT, bool PREF = false>
__global__ __launch_bounds__(512, 2) void conv_halo_wr_kernel(const KArgs a) {
    using G = GeoH<PW, BN>;
    constexpr int HP = G::HP, ROWB = G::ROWB, RPI = G::RPI, BK = G::BK;
    constexpr int WPX = PW / 16, NWN = 8 / WPX, CPW = BN / NWN, JN = CPW / 16, MI = 16;   // CPW = 32 channels per wave
    static_assert(CPW == 32 && JN == 2, "wave tile is 256 pixels x 32 channels");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo0 = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool hrole = wave >= 4;
    int tm, tn;
    const int ks = SPLIT ? (int)(blockIdx.x & ((1u << a.ks_log2) - 1)) : 0;
    tile_coords(a, SPLIT ? (int)(blockIdx.x >> a.ks_log2) : (int)blockIdx.x, tm, tn);
    const int per_frame = a.cg.tiles_x * a.cg.tiles_y;
    // patch == 3: frame-fastest tile order (the To tiles of one patch are neighbours in an XCD's run, so the temporal
    // taps they share are re-read from that XCD's L2 instead of HBM)
    const int t = a.cg.patch == 3 ? tm % a.cg.To : tm / per_frame;
    const int rem = a.cg.patch == 3 ? tm / a.cg.To : tm - t * per_frame;
    const int ty = rem / a.cg.tiles_x, tx = rem - ty * a.cg.tiles_x;
    const int y0 = ty * G::PH, x0 = tx * PW, n0 = tn * BN;
    const int Cin = a.cg.Cin, Wi = a.cg.Wi, Hi = a.cg.Hi;
    const int ncs = Cin / BK, nsl = a.cg.kt * ncs, ntap = nsl * 9;

    const int lrow = lane >> 2, lslot = lane & 3;
    const char* zsrc = (const char*)a.zero + lslot * 16;
    const long long frame_bytes = (long long)Hi * Wi * Cin * 2;
    const long long org = (((long long)(y0 - 1) * Wi + (x0 - 1)) * Cin) * 2;
    auto slice_base = [&](int tap_t, int cs) -> const char* {
        const int it = t + tap_t;
        const bool head = it < a.cg.npad;
        const char* base = (head && a.cg.carry) ? (const char*)a.cg.carry : (const char*)a.A;
        const int fidx = head ? (a.cg.carry ? max(it - a.cg.crep, 0) : 0) : it - a.cg.npad;
        return base + fidx * frame_bytes + org + cs * (BK * 2);
    };
    auto issue_halo = [&](const char* sb, char* hbuf) {      // waves 4-7: pieces q = u*4 + (wave - 4)
        int lr = lrow;
        asm volatile("" : "+v"(lr));   // recompute the piece addresses per call: hoisted, they cost 2 VGPRs per piece
        for (int u = 0; u < G::HPW; ++u) {
            const int q = u * 4 + (wave - 4);
            if (q >= G::NHI) break;
            const int r = q * RPI + lr;
            const int hy = r / HP, hx = r - hy * HP;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool ok = (r < G::HROWS) & ((unsigned)iy < (unsigned)Hi) & ((unsigned)ix < (unsigned)Wi);
            const int off = ((hy * Wi + hx) * Cin) * 2 + ((lslot ^ ((hx >> 2) & 3)) << 4);
            glds16(ok ? sb + off : zsrc, hbuf + q * 1024);
        }
    };

    const int wm = wave / NWN, wn = wave % NWN;    // x block of the patch, 32-channel column
    const int wpx0 = wm * 16;
    const int frow = lane & 15, fq = lane >> 4;
    const int pix = frow < 4 ? 2 * frow : (frow < 12 ? 2 * frow - 7 : 2 * frow - 16);   // pi(frow)
    const int hxl = wpx0 + pix;                    // halo x of this lane's pixel under kx = 0; halo row of patch row 0
    // this wave's fragments of tap u are the KiB blocks [cch%128/16 + jn] of the u-th 8-KiB tap tile of 128-channel
    // tile column cch/128 (cch = the wave's first channel); lane (frow, fq) reads row frow, chunk fq, which the
    // LDS-image order of the tile keeps at slot fq ^ f(frow)
    const int cch = n0 + wn * CPW;
    const char* wfk = (const char*)a.W + (long long)(cch / 128) * ntap * 8192 + (cch % 128) * ROWB;        // uniform
    const unsigned wfl = frow * ROWB + ((fq ^ ((0x78 >> (2 * ((frow >> 2) & 3))) & 3)) << 4);              // per lane
    f32x4 acc[MI][JN];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 bw[3][JN];
    // buffer loads: resource (this wave's tile column) and tap offset in SGPRs, one VGPR of per-lane offset
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wfk, 0, 0x7fffffff, 0x00020000);
    auto load_w = [&](int slot, int u) {
        const int so = (u < ntap ? u : ntap - 1) * 8192;                      // past the end: a harmless reload
#pragma unroll
        for (int jn = 0; jn < JN; ++jn) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, wfl, so + jn * 1024, 0);
            bw[slot][jn] = __builtin_bit_cast(bf16x8, v);
        }
    };

    const int sl0 = SPLIT ? (ks * nsl) >> a.ks_log2 : 0;
    const int sl1 = SPLIT ? ((ks + 1) * nsl) >> a.ks_log2 : nsl;
    int tap_t = SPLIT ? sl0 / ncs : 0, cs = SPLIT ? sl0 - tap_t * ncs : 0, hb = 0;
    if (hrole) issue_halo(slice_base(tap_t, cs), halo0);
    load_w(0, sl0 * 9);
    load_w(1, sl0 * 9 + 1);
    load_w(2, sl0 * 9 + 2);

    for (int sl = sl0; sl < sl1; ++sl) {
        // the halo of this slice has landed for every wave and every wave is done reading the other buffer
        if (hrole) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (hrole && sl + 1 < sl1) {
            if (++cs == ncs) { cs = 0; ++tap_t; }
            issue_halo(slice_base(tap_t, cs), halo0 + (hb ^ 1) * G::HALO_BYTES);
        }
        const char* hcur = halo0 + hb * G::HALO_BYTES;
        const int u0 = sl * 9;
        if constexpr (PREF) {
            // Fragment prefetch: the 18 activation fragments of column kx + 1 are requested DURING the last tap (ky = 2) of
            // column kx, one per pair of MFMAs, each into the register its predecessor has just freed (fragment i + 2 of the
            // current column dies with row i of that tap), so a column never starts by waiting out an LDS round trip with the
            // MFMA pipe idle; only a slice's first column (its halo is not valid before the slice barrier) still does.
            // Measured +0.8 .. 2 % (profiles/r03_conv/halopref_ab.log), bit-identical.  The cross-slice form (three halo
            // buffers, slice barrier in front of a slice's last tap so that this tap can request the next slice's first
            // column) was written too: the DMA issue then sits where all 18 fragments are live and spills 11-12 registers
            // into the loop; not kept (commit "conv_halo_wr fragment prefetch measured").
            auto col_ptr = [&](int kx) { return hcur + (hxl + kx) * ROWB + ((fq ^ (((hxl + kx) >> 2) & 3)) << 4); };
            bf16x8 af[MI + 2];
            {
                const char* sA = col_ptr(0);
#pragma unroll
                for (int r = 0; r < MI + 2; ++r) af[r] = *(const bf16x8*)(sA + r * (HP * ROWB));
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int ky = 0; ky < 2; ++ky) {
                    const int tapi = kx * 3 + ky;
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int jn = 0; jn < JN; ++jn)
                            acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[ky][jn], af[i + ky], acc[i][jn], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    load_w(ky, u0 + tapi + 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // fragment i of the next column goes straight into slot i: slots 0 .. i + 1 are dead by row i of this tap
                const char* sN = col_ptr(kx < 2 ? kx + 1 : 2);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
#pragma unroll
                    for (int jn = 0; jn < JN; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[2][jn], af[i + 2], acc[i][jn], 0, 0, 0);
                    if (kx < 2) af[i] = *(const bf16x8*)(sN + i * (HP * ROWB));
                }
                if (kx < 2) {
                    af[MI] = *(const bf16x8*)(sN + MI * (HP * ROWB));
                    af[MI + 1] = *(const bf16x8*)(sN + (MI + 1) * (HP * ROWB));
#pragma unroll
                    for (int i = 0; i < MI; ++i) {            // keep the interleave: two MFMAs, one fragment read
                        __builtin_amdgcn_sched_group_barrier(0x008, JN, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                load_w(2, u0 + kx * 3 + 2 + 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const char* sA = hcur + (hxl + kx) * ROWB + ((fq ^ (((hxl + kx) >> 2) & 3)) << 4);
            bf16x8 af[MI + 2];                                  // the 18 halo lines of the block, shifted by kx
#pragma unroll
            for (int r = 0; r < MI + 2; ++r) af[r] = *(const bf16x8*)(sA + r * (HP * ROWB));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int tapi = kx * 3 + ky;                  // tap index inside the slice; ring slot = tapi % 3 = ky
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jn = 0; jn < JN; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[ky][jn], af[i + ky], acc[i][jn], 0, 0, 0);
                // the tap is done: refill its ring slot with the tap three ahead.  Issued AFTER the tap's MFMAs so that
                // the (conservative, post-barrier) vmcnt wait in front of them never covers a load issued a moment
                // ago; sched_barrier keeps the scheduler from sinking the load next to its use.
                __builtin_amdgcn_sched_barrier(0);
                load_w(ky, u0 + tapi + 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        hb ^= 1;
    }
    if constexpr (SPLIT) {
        f32x4* img = (f32x4*)a.partial + ((((long long)(tm * a.nbn + tn)) << a.ks_log2) + ks) * (MI * JN * G::NT) + tid;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < JN; ++j) img[(i * JN + j) * G::NT] = acc[i][j];
        return;
    }
    // (An epilogue that stores straight from the accumulator layout - four channels of one pixel per lane, 8-byte stores and
    // residual loads, GroupNorm partial sums by lane shuffles, no LDS tile and two barriers fewer - was built and measured
    // 0.3-5 % SLOWER, most with a residual: profiles/r03_conv/halodirect_ab.log.  Not kept.)
    asm volatile("s_barrier" ::: "memory");
    halo_epilogue<PW, BN, MI, JN>(a, smem, acc, tid, t, per_frame, rem, y0, x0, n0, wn * CPW, 0, wpx0, pix, fq);
}

// Split convs with many images per tile (S >= 4): the images are added by the whole chip first (one float4 per thread,
// split order, into image 0) - 32 finishing workgroups reading 8 x 256 KiB each took as long as the multiplication.
__global__ __launch_bounds__(256) void conv_split_reduce_kernel(f32x4* img, long long n4, int per_tile4, int ks_log2) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n4) return;
    const long long tile = e / per_tile4;
    f32x4* p = img + ((tile << ks_log2) * per_tile4) + (e - tile * per_tile4);
    f32x4 s = p[0];
    for (int k = 1; k < (1 << ks_log2); ++k) {
        const f32x4 v = p[(long long)k * per_tile4];
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += v[r];
    }
    p[0] = s;
}

// Last launch of a split conv: one workgroup per tile adds the accumulator images that are still separate (a.nsum of
// them, split order) and finishes the tile exactly like the unsplit kernel does (bias, bf16 rounding, residual,
// GroupNorm partial sums).
template <int PW, int BN>
__global__ __launch_bounds__(512, 2) void conv_halo_finish_kernel(const KArgs a) {
    using G = GeoH<PW, BN>;
    constexpr int WPX = PW / 16, NWN = 8 / WPX, CPW = BN / NWN, JN = CPW / 16, MI = 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    tile_coords(a, tm, tn);
    const int per_frame = a.cg.tiles_x * a.cg.tiles_y;
    const int t = a.cg.patch == 3 ? tm % a.cg.To : tm / per_frame;
    const int rem = a.cg.patch == 3 ? tm / a.cg.To : tm - t * per_frame;
    const int ty = rem / a.cg.tiles_x, tx = rem - ty * a.cg.tiles_x;
    const int y0 = ty * G::PH, x0 = tx * PW, n0 = tn * BN;
    const int wm = wave / NWN, wn = wave % NWN;
    const int wpx0 = wm * 16;
    const int frow = lane & 15, fq = lane >> 4;
    const int pix = frow < 4 ? 2 * frow : (frow < 12 ? 2 * frow - 7 : 2 * frow - 16);
    const int S = a.nsum;
    const f32x4* img = (const f32x4*)a.partial + (((long long)(tm * a.nbn + tn)) << a.ks_log2) * (MI * JN * G::NT) + tid;
    f32x4 acc[MI][JN];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = img[(i * JN + j) * G::NT];
    for (int k = 1; k < S; ++k) {
        img += MI * JN * G::NT;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < JN; ++j) {
                const f32x4 v = img[(i * JN + j) * G::NT];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += v[r];
            }
    }
    halo_epilogue<PW, BN, MI, JN>(a, smem, acc, tid, t, per_frame, rem, y0, x0, n0, wn * CPW, 0, wpx0, pix, fq);
}

// K splits of a W-through-registers conv: the latent-resolution 512-channel convs of a small image are 16-128 tiles for
// 256 CUs with a K loop of 48 slices; splitting the slices S ways fills the chip.  Power of two, >= 6 slices per split.
// Second case, grids a little over ONE tile per CU (the latent-resolution convs of a 1080p frame: 270 tiles): every CU
// works through one tile and a handful of CUs through two; halves of tiles pack tighter - ceil(2 tiles / 256) / 2 = 1.5
// up to 384 tiles - at the price of the finishing launch (about 7 slices' worth of time): 270 tiles 503 -> 439 us.
// Past two tiles per CU the same arithmetic predicts a gain that does not come (540 tiles: 760 -> 783 us), 480 tiles
// split in two lose (509 -> 605 us) and so do 120 tiles split four ways (157 -> 177 us)
// (tools/kernel_bench.py convsplit, profiles/r03_conv_split_mid.log).
static int conv_halo_split(long long tiles, int nsl) {
    if (!svr2::opt(svr2::OPT_CONV_SPLIT, 1)) return 0;
    int lg = 0;
    while ((tiles << lg) < 192 && lg < 3 && (nsl >> (lg + 1)) >= 6) ++lg;
    if (lg == 0 && tiles > 256 && nsl >= 12) {
        const long long c0 = (tiles + 255) / 256, c1 = (2 * tiles + 255) / 256;
        if (c0 == 2 && c1 * nsl + 14 < 2 * c0 * nsl) lg = 1;
    }
    return lg;
}

template <int PW, int BN, bool WR>
int launch_halo(KArgs a, hipStream_t s) {
    using G = GeoH<PW, BN>;
    const void* fn;
    if constexpr (WR) fn = (const void*)conv_halo_wr_kernel<PW, BN, false>;
    else fn = (const void*)conv_halo_kernel<PW, BN>;
    SVR2_SET_MAX_LDS(fn, G::LDS, "conv_halo");
    // SVR2_HALO_PREF (default 1): the W-through-registers kernel with the next column's activation fragments requested
    // under the current column's last tap; 0 = every column reads its fragments first (A/B; same bits)
    const bool pref = WR && svr2::opt(svr2::OPT_HALO_PREF, 1) != 0;
    a.cg.patch = svr2::opt(svr2::OPT_CONV_TFAST, 0) ? 3 : 2;
    a.cg.tiles_x = (a.cg.Wo + PW - 1) / PW;
    a.cg.tiles_y = (a.cg.Ho + G::PH - 1) / G::PH;
    a.nbm = a.cg.To * a.cg.tiles_x * a.cg.tiles_y;
    a.nbn = (a.N + BN - 1) / BN;
    if constexpr (WR) {
        const long long tiles = (long long)a.nbm * a.nbn;
        const int lg = a.partial ? conv_halo_split(tiles, a.cg.kt * (a.cg.Cin / G::BK)) : 0;
        if (lg > 0 && (size_t)(tiles << lg) * G::BM * BN * 4 <= a.partial_bytes) {
            SVR2_SET_MAX_LDS((const void*)(conv_halo_wr_kernel<PW, BN, true>), G::LDS, "conv_halo (split)");
            SVR2_SET_MAX_LDS((const void*)(conv_halo_finish_kernel<PW, BN>), G::EPI_BYTES, "conv_halo (finish)");
            a.ks_log2 = lg;
            hipLaunchKernelGGL((conv_halo_wr_kernel<PW, BN, true>), dim3((unsigned)(tiles << lg)), dim3(G::NT), G::LDS, s, a);
            SVR2_LAUNCH_CHECK();
            a.nsum = 1 << lg;
            if (lg >= 2) {
                const int per_tile4 = G::BM * BN / 4;
                const long long n4 = tiles * per_tile4;
                hipLaunchKernelGGL(conv_split_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s,
                                   (f32x4*)a.partial, n4, per_tile4, lg);
                SVR2_LAUNCH_CHECK();
                a.nsum = 1;
            }
            hipLaunchKernelGGL((conv_halo_finish_kernel<PW, BN>), dim3((unsigned)tiles), dim3(G::NT), G::EPI_BYTES, s, a);
            SVR2_LAUNCH_CHECK();
            return SVR2_OK;
        }
        a.ks_log2 = 0;
        if (pref) {
            SVR2_SET_MAX_LDS((const void*)(conv_halo_wr_kernel<PW, BN, false, true>), G::LDS, "conv_halo (prefetch)");
            hipLaunchKernelGGL((conv_halo_wr_kernel<PW, BN, false, true>), dim3(a.nbm * a.nbn), dim3(G::NT), G::LDS, s, a);
        } else {
            hipLaunchKernelGGL((conv_halo_wr_kernel<PW, BN, false>), dim3(a.nbm * a.nbn), dim3(G::NT), G::LDS, s, a);
        }
    } else {
        hipLaunchKernelGGL((conv_halo_kernel<PW, BN>), dim3(a.nbm * a.nbn), dim3(G::NT), G::LDS, s, a);
    }
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

// Workspace a split conv of this shape would use (0: the shape is not split).
template <int PW, int BN>
size_t split_bytes(const KArgs& a) {
    using G = GeoH<PW, BN>;
    const long long tiles = (long long)a.cg.To * ((a.cg.Wo + PW - 1) / PW) * ((a.cg.Ho + G::PH - 1) / G::PH) * ((a.N + BN - 1) / BN);
    const int lg = conv_halo_split(tiles, a.cg.kt * (a.cg.Cin / G::BK));
    return lg > 0 ? (size_t)(tiles << lg) * G::BM * BN * 4 : 0;
}

}  // namespace

// Shape contract (checked by the caller, gemm.hip): kh = kw = 3, stride 1, zero padding 1 on every spatial side
// (Ho = Hi, Wo = Wi), any kt, Cin % 32 == 0, W in SVR2_WL_TILED_128x32 (cout_pad = round_up(Cout, 128) rows).
int conv_halo_launch(KArgs a, hipStream_t s) {
    // SVR2_HALO_WR: bit 0 = W through registers (conv_halo_wr_kernel) for the 256-channel geometry, bit 1 = for the
    // 128-channel geometry (default 3; 0 = both through LDS).  Measured, same box (profiles/r01_kernel_bench_halo_wr.log):
    // 1438 -> 1603, 1510 -> 1656 TFLOP/s (256-channel blocks), 1393 -> 1448 (128-channel blocks).
    const int wr = svr2::opt(svr2::OPT_HALO_WR, 3);
    // 256-channel column blocks read two 128-row W tile columns: only when Cout is a multiple of 256
    if (a.N % 256 == 0) return (wr & 1) ? launch_halo<16, 256, true>(a, s) : launch_halo<16, 256, false>(a, s);
    // conv_out (Cout = 3): 1024-pixel x 16-channel tiles, 4x fewer MFMAs per pixel than a 128-channel tile
    if (a.N <= 16) return launch_halo<64, 16, false>(a, s);
    return (wr & 2) ? launch_halo<32, 128, true>(a, s) : launch_halo<32, 128, false>(a, s);
}


size_t conv_halo_split_bytes(const KArgs& a) {
    const int wr = svr2::opt(svr2::OPT_HALO_WR, 3);
    if (a.N % 256 == 0) return (wr & 1) ? split_bytes<16, 256>(a) : 0;
    if (a.N <= 16) return 0;
    return (wr & 2) ? split_bytes<32, 128>(a) : 0;
}

}  // namespace svr2
