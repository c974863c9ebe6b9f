// bf16 MFMA GEMM core for gfx950: dense / row-gathered / implicit-conv3d A operand, fused epilogues.
//
// Tile: 128 x 128 x 64 per 256-thread workgroup (4 waves as 2 x 2, each 64 x 64 = 4 x 4 MFMA
// 16x16x32 tiles, fp32 accumulators).  Both operands are K-contiguous and are staged with direct
// global->LDS 16-byte loads (global_load_lds_dwordx4) into a 2-deep ring; the next K-tile's loads
// are in flight while the current one is multiplied, one barrier per K-tile.  The LDS image is
// lane-linear (that is what the DMA writes), so the bank swizzle chunk ^= (row >> 1) & 7 is applied
// on the per-lane SOURCE address and again on the ds_read_b128 address (conflict-free).
// For the implicit conv the A row of an output pixel is the 128-byte channel run of the input pixel
// under the current filter tap, so the same loader serves it with a different per-row base pointer;
// zero padding and rows past M read a zero page.  Accumulators leave through an fp32 LDS tile so
// every epilogue works on 8 consecutive columns and stores 16 bytes per lane, whole 256-byte rows.
#include "common.h"
#include "gemm_args.h"
#include <stdarg.h>
#include <stdlib.h>
#include <atomic>

namespace {

constexpr int BN = 128, BK = 64;
constexpr int B_TILE_BYTES = BN * BK * 2;         // 16 KiB
// Two geometries share one kernel body (template parameters BM, ST = depth of the LDS ring):
//   BM = 128, ST 2: 4 waves (2 x 2), 2-deep LDS ring (64 KiB), 2 workgroups per CU  - small problems / tails
//   BM = 256, ST 3: 8 waves (4 x 2), 3-deep LDS ring (144 KiB), counted vmcnt: two K-tiles of DMA stay in flight
//             across the barrier, W traffic per FLOP halves                          - the large GEMMs / convs
//   BM = 128, ST 3: the 128-row tile with a 3-deep ring (96 KiB, one workgroup per CU) - tile-starved dense GEMMs whose
//             weights come from HBM (round 2 measured a deeper ring at no gain, but on a replayed, cache-hot weight;
//             round 3 with weights rotated through 300 MB: -19...-24 %, see launch())
template <int BM, int ST> struct Geo {
    static constexpr int NT = BM * 2;                       // threads
    static constexpr int NW = NT / 64;                      // waves
    static constexpr int STAGES = ST;
    static constexpr int A_TILE_BYTES = BM * BK * 2;
    static constexpr int STAGE_BYTES = A_TILE_BYTES + B_TILE_BYTES;
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;  // >= BM*BN*4 (fp32 epilogue tile)
    static constexpr int NBI = 16 / NW;                     // W DMA instructions per wave per K-tile
};

using namespace svr2;

template <int AMODE, int EPI, int BM, int ST>
__global__ __launch_bounds__(BM * 2) void gemm_kernel(const KArgs a) {
    using G = Geo<BM, ST>;
    constexpr int NTHREADS = G::NT, STAGE_BYTES = G::STAGE_BYTES, TILE_BYTES = G::A_TILE_BYTES, NBI = G::NBI;
    static_assert(G::LDS_BYTES >= BM * BN * 4, "epilogue tile must fit");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> LDS-DMA base stays scalar
    int tm, tn;
    tile_coords(a, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- loader setup: this thread moves 4 A rows and 4 W rows (16 bytes each) per K-tile.
    const int lrow = lane >> 3, lpos = lane & 7;
    const char* a_ptr[4];   // dense: running source pointer; conv: frame base + pixel offset of the current tap_t
    const char* w_ptr[NBI]; // running weight pointer
    int cy[4], cx[4], ct[4];        // conv: top-left input coordinate / first padded frame of each output row
    long long rowoff[4];            // conv: byte offset of (cy, cx) inside a frame + this lane's chunk
    const char* zsrc = (const char*)a.zero + lpos * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + lrow;
        const int chunk = lpos ^ ((r >> 1) & 7);
        const int m = tile_row_to_m<AMODE, BM>(a, tm, r);
        if (AMODE == A_DENSE) {
            long long src = m;
            if (a.a_rows) src = a.a_rows[m < a.M ? m : a.M - 1];
            a_ptr[i] = (m < a.M) ? (const char*)(a.A + src * a.lda) + chunk * 16 : nullptr;
        } else {
            const int mm = m < a.M ? m : 0;
            const int x = mm % a.cg.Wo, t2 = mm / a.cg.Wo;
            const int y = t2 % a.cg.Ho, t = t2 / a.cg.Ho;
            cx[i] = x * a.cg.sw - a.cg.pad_w0;
            cy[i] = (m < a.M) ? y * a.cg.sh - a.cg.pad_h0 : -(1 << 28);  // invalid row: never in range
            ct[i] = t * a.cg.st;
            rowoff[i] = (((long long)cy[i] * a.cg.Wi + cx[i]) * a.cg.Cin) * 2 + chunk * 16;
            a_ptr[i] = nullptr;
        }
    }
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
        const int r = (wave * NBI + i) * 8 + lrow;
        w_ptr[i] = (const char*)(a.W + (long long)(n0 + r) * a.K) + ((lpos ^ ((r >> 1) & 7)) << 4);
    }
    // conv tap state (wave-uniform scalars); K order is ((kt*kh + ky)*kw + kx)*Cin + c
    int tap_t = 0, tap_y = 0, tap_x = 0, tap_c = 0, tap_blk = 0;
    const long long frame_bytes = (long long)a.cg.Hi * a.cg.Wi * a.cg.Cin * 2;
    auto set_frame = [&]() {  // a_ptr[i] = base of padded frame (ct[i] + tap_t) + rowoff[i]; runs when tap_t changes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int it = ct[i] + tap_t;
            const bool head = it < a.cg.npad;
            const char* base = (head && a.cg.carry) ? (const char*)a.cg.carry : (const char*)a.A;
            const int fidx = head ? (a.cg.carry ? max(it - a.cg.crep, 0) : 0) : it - a.cg.npad;
            a_ptr[i] = base + fidx * frame_bytes + rowoff[i];
        }
    };
    if (AMODE == A_CONV) set_frame();

    auto issue = [&](int stage) {
        char* sA = smem + stage * STAGE_BYTES;
        char* sB = sA + TILE_BYTES;
        if (AMODE == A_DENSE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                glds16(a_ptr[i] ? a_ptr[i] : zsrc, sA + (wave * 4 + i) * 1024);
                if (a_ptr[i]) a_ptr[i] += BK * 2;
            }
        } else {
            const long long toff = (((long long)tap_y * a.cg.Wi + tap_x) * a.cg.Cin + tap_blk + tap_c) * 2;  // scalar
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = ((unsigned)(cy[i] + tap_y) < (unsigned)a.cg.Hi) & ((unsigned)(cx[i] + tap_x) < (unsigned)a.cg.Wi);
                const char* src = a_ptr[i] + toff;
                glds16(ok ? src : zsrc, sA + (wave * 4 + i) * 1024);
            }
            // K order: 64-channel slice innermost, then the kh*kw spatial taps, then the channel slices, then the
            // temporal taps - the 9 spatial taps of one slice re-read the same 128-byte lines within 9 consecutive
            // K-steps, so they hit in L2 (tap-major order over all channels thrashed it: 18x the input in FETCH_SIZE)
            tap_c += BK;
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
        for (int i = 0; i < NBI; ++i) {
            glds16(w_ptr[i], sB + (wave * NBI + i) * 1024);
            w_ptr[i] += BK * 2;
        }
    };

    // ---- MFMA fragment addressing
    const int wm = wave >> 1, wn = wave & 1;   // waves are (BM/64) x 2
    const int frow = lane & 15, fq = lane >> 4;
    const int swz = (frow >> 1) & 7;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = a.K / BK;
    constexpr int S = G::STAGES, PT = 4 + NBI;   // PT: DMA instructions per thread and K-tile
    issue(0);
#pragma unroll
    for (int i = 1; i < S - 1; ++i)
        if (i < nk) issue(i);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // Tile kt must have landed for every wave, and every wave must be done reading the stage that is
        // about to be refilled.  Rings of 3 and more stages leave the DMA of the next S-2 tiles in flight across the
        // barrier (counted vmcnt = this thread's loads per tile x tiles); raw s_barrier, never __syncthreads (it
        // drains vmcnt).
        if (S >= 3) {
            const int ahead = min(S - 2, nk - 1 - kt);
            if (S >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * PT) : "memory");
            else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + S - 1 < nk) issue(stage == 0 ? S - 1 : stage - 1);   // (kt + S - 1) % S
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk) issue(stage ^ 1);
        }
        const char* sA = smem + stage * STAGE_BYTES;
        const char* sB = sA + TILE_BYTES;
        stage = (stage == S - 1) ? 0 : stage + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // (hipcc's own interleave of these reads and MFMAs measured 7 % faster than pinning
            //  "16 reads, then 32 MFMAs" with sched_group_barrier)
            bf16x8 af[4], bfr[4];
            const int chunk = ((fq + 4 * ks) ^ swz) << 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = *(const bf16x8*)(sA + (wm * 64 + i * 16 + frow) * 128 + chunk);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bfr[j] = *(const bf16x8*)(sB + (wn * 64 + j * 16 + frow) * 128 + chunk);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        // LLVM's MFMA/DS interleave strategy for this scheduling region: measured +1.3 % (dense GEMM) to +5 %
        // (512-channel conv) over the default schedule; s_setprio around the MFMA cluster measured 0 %.
        __builtin_amdgcn_iglp_opt(1);
    }

    // ---- epilogue through an fp32 LDS tile [128][128]
    __syncthreads();
    float* tile = (float*)smem;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                tile[(wm * 64 + i * 16 + fq * 4 + r) * BN + wn * 64 + j * 16 + frow] = acc[i][j][r];
    __syncthreads();

    if (EPI == SVR2_EPI_SWIGLU) {
        // columns [16p, 16p+8) = gate, [16p+8, 16p+16) = in  ->  output columns [8p, 8p+8)
        for (int c = tid; c < BM * 8; c += NTHREADS) {
            const int row = c >> 3, p = c & 7;
            const int m = tile_row_to_m<AMODE, BM>(a, tm, row), ncol = (n0 >> 1) + p * 8;
            if (m >= a.M || (n0 + p * 16) >= a.N) continue;
            const float* g = tile + row * BN + p * 16;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float gv = g[e], iv = g[8 + e];
                o[e] = f2bf(gv / (1.f + __expf(-gv)) * iv);
            }
            *(bf16x8*)(a.C + (long long)m * a.ldc + ncol) = o;
        }
        return;
    }

    for (int c = tid; c < BM * 16; c += NTHREADS) {
        const int row = c >> 4, cc = c & 15;
        const int m = tile_row_to_m<AMODE, BM>(a, tm, row), n = n0 + cc * 8;
        if (m >= a.M || n >= a.N) continue;
        const float* src = tile + row * BN + cc * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[e];
        if (a.bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (n + e < a.N) ? a.bias[n + e] : 0.f;
        }
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
            for (int e = 0; e < 8; ++e) {  // nn.GELU("tanh"): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
                const float x = v[e], u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
                v[e] = x / (1.f + __expf(-2.f * u));   // 0.5 x (1 + tanh u) == x * sigmoid(2u)
            }
            dst = (long long)m * a.ldc + n;
        } else {
            if (a.resid) {  // conv residual add
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

template <int AMODE, int EPI, int BM, int ST>
int launch_bm(KArgs a, hipStream_t s) {
    using G = Geo<BM, ST>;
    SVR2_SET_MAX_LDS((gemm_kernel<AMODE, EPI, BM, ST>), G::LDS_BYTES, "gemm");
    a.nbm = (a.M + BM - 1) / BM;
    a.nbn = (a.N + BN - 1) / BN;
    if (AMODE == A_CONV) {
        a.cg.patch = (a.cg.Ho % (BM / 16) == 0 && a.cg.Wo % 16 == 0 && !svr2::opt(svr2::OPT_CONV_LINEAR_TILES, 0)) ? 1 : 0;
        a.cg.tiles_x = a.cg.Wo / 16;
        a.cg.tiles_y = a.cg.Ho / (BM / 16);
    }
    hipLaunchKernelGGL((gemm_kernel<AMODE, EPI, BM, ST>), dim3(a.nbm * a.nbn), dim3(G::NT), G::LDS_BYTES, s, a);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

inline int g_force_bm() { return svr2::opt(svr2::OPT_GEMM_BM, 0); }
template <int AMODE, int EPI>
int launch(const KArgs& a, hipStream_t s) {
    // measured on MI355X (profiles/r01_kernel_bench_*.log): two 128-row workgroups per CU beat one 256-row,
    // 3-stage workgroup by ~5 % on every shape of the path, so 128 is the default; SVR2_GEMM_BM=256 selects
    // the other geometry for A/B runs.
    // Large problems go to the second-generation kernel (gemm2.hip: 256-row tiles, 128x64 per wave, BK 32,
    // 3-deep ring); SVR2_GEMM_V=1 keeps everything on this kernel for A/B runs.
    const int ver = svr2::opt(svr2::OPT_GEMM_V, 2);         // tests switch kernels with svr2_set_option
    const long long tiles2 = (long long)((a.M + 255) / 256) * ((a.N + 127) / 128);
    // measured (profiles/r01_*): the 256x256x64 geometry wins by 16-20 % wherever N is a multiple of 256; for
    // N = 128 the 256x128x32 geometry is +3 % on 128-channel inputs and -18 % on 256-channel inputs.
    // dense GEMMs with N % 256 == 0: the ping-pong kernel (gemm4.hip); SVR2_GEMM4=0 keeps them on gemm2 for A/B runs
    // (long K loops only: its prologue fills two K-steps and the phases need a few steps to pay off; measured slower on
    // the K = 256..512 GEMMs of the VAE)
    // the row-gathered proj_out GEMM stays on gemm2: scattered rows make its DMA pieces cost the two A-streaming waves of a
    // group more than a LOAD phase has - measured 535 vs 850 TFLOP/s)
    // from where on: one round of the 128-row kernel is 512 workgroups (two per CU); past that a second round starts and one
    // round of 256 x 256 ping-pong tiles is faster although it fills fewer CUs (M = 3 328, N = K = 2 560: 90 -> 82 us; NaDiT-7B
    // qkv at M = 1 024: 98 -> 67 us; below 512 tiles the 128-row kernel wins by 30-45 %: profiles/r03_gemm_small_m.log)
    const long long tiles1 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (AMODE == A_DENSE && a.N % 256 == 0 && a.K % 64 == 0 &&
        ((a.K >= 1024 && tiles1 > 512 && !a.a_rows) || (ver == 3 && a.K >= 128))) {
        if (svr2::opt(svr2::OPT_GEMM4, 1) != 0) return gemm4_launch(EPI, a, s);
    }
    // (N = 128 with more than 128 input channels: slower on the 3x3 convs, see above - but the 1x1x1 shortcut 256 -> 128 at
    //  full resolution is HBM-bound with a four-step K loop, where the 3-deep ring of the 256-row geometry streams at
    //  5.0 TB/s against 2.6: 9.7 -> 5.1 ms on four 4K frames, tools/kernel_bench.py shortcut4k)
    const bool pointwise = AMODE == A_CONV && a.cg.kt * a.cg.kh * a.cg.kw == 1;
    const bool v2_shape = (a.N % 256 == 0) || (AMODE == A_CONV && a.N == 128 && (a.cg.Cin <= 128 || pointwise));
    if (((ver == 2 && tiles2 >= 384 && v2_shape) || ver == 3) && a.K % 32 == 0)
        return gemm2_launch(AMODE, EPI, a, s);  // ver 3: always (tests)
    const bool big = g_force_bm() == 256;
    if (big) return launch_bm<AMODE, EPI, 256, 3>(a, s);
    // Tile-starved dense GEMMs with a long K loop (the NaDiT projections of a small image: M ~ 1 000 tokens, and the
    // text lane's M = 58): at most one workgroup per CU, so the LDS a second workgroup would have used buys a third ring
    // stage.  With weights that come from HBM (a layer's weights are never cache-hot in the pipeline) the 2-deep ring
    // waits a memory round trip per K-step: measured 44.4 -> 35.9 us (N = K = 2 560) and 98 -> 74.5 us (K = 6 912) at
    // M = 1 024; a 4th and 5th stage add nothing (profiles/r03_gemm_small_m.log).  Same K order: bit-identical results.
    // Where two 128-row workgroups per CU would share a CU but 256-row tiles fit one per CU, and the weight is the big
    // operand (N >= 4 M: qkv at M = 1 024, 480 vs 240 tiles), one 256-row workgroup moves 25 % fewer bytes per CU:
    // 55.4 -> 51.5 us; with M comparable to N it is no faster (M = 2 560 ... 3 072, N = 2 560: 58 vs 54-56 us).
    // SVR2_GEMM_BM=128 keeps the 2-deep geometry for A/B runs.
    if constexpr (AMODE == A_DENSE) {
        if (g_force_bm() != 128 && a.K >= 1024) {
            if (tiles1 <= 256) return launch_bm<AMODE, EPI, 128, 3>(a, s);
            if (tiles2 <= 256 && a.N >= 4 * a.M) return launch_bm<AMODE, EPI, 256, 3>(a, s);   // W-dominated only
        }
    }
    return launch_bm<AMODE, EPI, 128, 2>(a, s);
}

}  // namespace

// ---- tuning switches and per-device bookkeeping (common.h)
namespace svr2 {
static const char* const kOptNames[OPT_COUNT] = {"SVR2_HALO_WR", "SVR2_GEMM_V", "SVR2_GEMM4", "SVR2_GEMM_BM",
                                                 "SVR2_GEMM2_NARROW", "SVR2_CONV_LINEAR_TILES", "SVR2_ATTN_V",
                                                 "SVR2_CONV_TFAST", "SVR2_CONV_SPLIT", "SVR2_ATTN5_V", "SVR2_HALO_PREF",
                                                 "SVR2_EPI_PREF"};
static const int kUnset = INT32_MIN;
static std::atomic<int> g_opt[OPT_COUNT] = {{kUnset}, {kUnset}, {kUnset}, {kUnset}, {kUnset}, {kUnset}, {kUnset}, {kUnset}, {kUnset}, {kUnset}, {kUnset}, {kUnset}};
static const int kDefault = INT32_MIN + 1;   // "resolved: neither set nor in the environment"

int opt(Opt o, int dflt) {
    int v = g_opt[o].load(std::memory_order_relaxed);
    if (v == kUnset) {                         // first use: the environment decides, once
        const char* e = getenv(kOptNames[o]);
        int want = e ? atoi(e) : kDefault;
        // publish only over "unset": a concurrent svr2_set_option wins and its value is what we return
        v = g_opt[o].compare_exchange_strong(v, want, std::memory_order_relaxed) ? want : g_opt[o].load(std::memory_order_relaxed);
    }
    return v == kDefault ? dflt : v;
}

// 0: the attribute is already set on the current device; otherwise the device's bit, which the caller publishes with
// attr_done() AFTER hipFuncSetAttribute succeeded - a second host thread that races here sets the attribute again
// (harmless) instead of launching with > 64 KiB of dynamic LDS before it is allowed.
unsigned long long attr_pending(const unsigned long long* mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return ~0ull;   // unknown: set the attribute every time
    const unsigned long long bit = 1ull << dev;
    return (__atomic_load_n(mask, __ATOMIC_ACQUIRE) & bit) ? 0ull : bit;
}
void attr_done(unsigned long long* mask, unsigned long long bit) {
    if (bit != ~0ull) __atomic_fetch_or(mask, bit, __ATOMIC_RELEASE);
}
}  // namespace svr2

extern "C" int svr2_set_option(const char* name, int value, int unset) {
    SVR2_CHECK(name, SVR2_E_ARG, "svr2_set_option: null name");
    for (int i = 0; i < svr2::OPT_COUNT; ++i)
        if (!strcmp(name, svr2::kOptNames[i])) {
            svr2::g_opt[i].store(unset ? svr2::kDefault : value, std::memory_order_relaxed);
            return SVR2_OK;
        }
    svr2_set_error("svr2_set_option: unknown option '%s'", name);
    return SVR2_E_ARG;
}

static thread_local char g_err[512] = "";
void svr2_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* svr2_last_error(void) { return g_err; }
extern "C" int svr2_abi_version(void) { return SVR2_ABI_VERSION; }

extern "C" int svr2_device_check(int dev) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
        svr2_set_error("hipGetDeviceProperties(%d) failed", dev);
        return SVR2_E_ARCH;
    }
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        svr2_set_error("device %d is %s, libseedvr2_hip is built for gfx950 only", dev, p.gcnArchName);
        return SVR2_E_ARCH;
    }
    return p.multiProcessorCount;
}

extern "C" int svr2_gemm(const svr2_gemm_desc* d, void* stream) {
    SVR2_CHECK(d && d->A && d->W && d->C && d->zero_page, SVR2_E_ARG, "svr2_gemm: null pointer");
    SVR2_CHECK(d->M > 0 && d->N > 0 && d->K > 0 && d->K % BK == 0, SVR2_E_SHAPE,
               "svr2_gemm: need M,N,K > 0 and K %% 64 == 0 (M=%lld N=%lld K=%lld)", (long long)d->M, (long long)d->N,
               (long long)d->K);
    SVR2_CHECK(d->M < (1ll << 31) - 256 && d->N < (1ll << 30), SVR2_E_SHAPE, "svr2_gemm: M/N too large");
    SVR2_CHECK(d->lda % 8 == 0 && d->ldc % 8 == 0 && aligned16(d->A) && aligned16(d->W) && aligned16(d->C),
               SVR2_E_ALIGN, "svr2_gemm: A/W/C must be 16-byte aligned with lda, ldc %% 8 == 0");
    SVR2_CHECK(d->lda >= d->K, SVR2_E_SHAPE, "svr2_gemm: lda < K");
    KArgs a;
    memset(&a, 0, sizeof(a));
    a.A = (const bf16*)d->A; a.W = (const bf16*)d->W; a.C = (bf16*)d->C;
    a.a_rows = d->a_rows; a.bias = d->bias; a.gate = d->gate; a.resid = (const bf16*)d->resid;
    a.zero = (const bf16*)d->zero_page;
    a.lda = d->lda; a.ldc = d->ldc; a.ldr = d->ldr;
    a.M = (int)d->M; a.N = (int)d->N; a.K = (int)d->K;
    hipStream_t s = (hipStream_t)stream;
    switch (d->epilogue) {
        case SVR2_EPI_BIAS:
            SVR2_CHECK(d->resid == nullptr || (d->ldr % 8 == 0 && aligned16(d->resid)), SVR2_E_ALIGN,
                       "svr2_gemm: resid alignment");
            return launch<A_DENSE, SVR2_EPI_BIAS>(a, s);
        case SVR2_EPI_GATE_RES:
            SVR2_CHECK(d->resid && d->ldr % 8 == 0 && aligned16(d->resid) && d->N % 8 == 0, SVR2_E_ARG,
                       "svr2_gemm: GATE_RES needs resid (16B aligned, ldr %% 8 == 0) and N %% 8 == 0");
            a.epi_pref = svr2::opt(svr2::OPT_EPI_PREF, 1);      // SVR2_EPI_PREF=0: residual loaded chunk by chunk (A/B; same bits)
            return launch<A_DENSE, SVR2_EPI_GATE_RES>(a, s);
        case SVR2_EPI_SWIGLU:
            SVR2_CHECK(d->N % 16 == 0 && d->bias == nullptr, SVR2_E_SHAPE,
                       "svr2_gemm: SWIGLU needs N %% 16 == 0 (interleaved gate/in rows) and no bias");
            return launch<A_DENSE, SVR2_EPI_SWIGLU>(a, s);
        case SVR2_EPI_PIXSHUF3D: {
            a.ps_frames = d->ps_frames; a.ps_h = d->ps_h; a.ps_w = d->ps_w; a.ps_c = d->ps_c; a.ps_tz = d->ps_tz;
            a.ps_drop = d->ps_drop_head;
            a.ps_wd = make_fastdiv((uint32_t)d->ps_w); a.ps_hd = make_fastdiv((uint32_t)d->ps_h);
            a.ps_cd = make_fastdiv((uint32_t)d->ps_c);
            SVR2_CHECK(d->ps_c > 0 && d->ps_c % 8 == 0 && (d->ps_tz == 1 || d->ps_tz == 2) &&
                           d->N == (int64_t)d->ps_c * 4 * d->ps_tz &&
                           d->M == (int64_t)d->ps_frames * d->ps_h * d->ps_w,
                       SVR2_E_SHAPE, "svr2_gemm: PIXSHUF3D geometry mismatch");
            SVR2_CHECK(!(d->ps_drop_head && d->ps_tz != 2), SVR2_E_SHAPE, "svr2_gemm: drop_head needs tz == 2");
            return launch<A_DENSE, SVR2_EPI_PIXSHUF3D>(a, s);
        }
        case SVR2_EPI_GELU:
            return launch<A_DENSE, SVR2_EPI_GELU>(a, s);
        default:
            svr2_set_error("svr2_gemm: unknown epilogue %d", d->epilogue);
            return SVR2_E_ARG;
    }
}

// Pixel-patch tiling of the halo-resident kernel for a conv (conv_halo.hip picks the same geometry from Cout).
extern "C" int32_t svr2_conv3d_gn_tiles(int32_t Ho, int32_t Wo, int32_t Cout) {
    if (Ho <= 0 || Wo <= 0 || Cout <= 0) return 0;
    const int pw = Cout % 256 == 0 ? 16 : (Cout <= 16 ? 64 : 32);
    return ((Ho + 15) / 16) * ((Wo + pw - 1) / pw);
}

static void conv_kargs(const svr2_conv3d_desc* d, KArgs& a) {
    memset(&a, 0, sizeof(a));
    a.A = (const bf16*)d->x; a.W = (const bf16*)d->w; a.C = (bf16*)d->y;
    a.bias = d->bias; a.resid = (const bf16*)d->resid; a.zero = (const bf16*)d->zero_page;
    a.ldc = d->Cout; a.ldr = d->Cout;
    a.M = (int)((long long)d->To * d->Ho * d->Wo); a.N = d->Cout; a.K = d->kt * d->kh * d->kw * d->Cin;
    a.cg.carry = (const bf16*)d->carry;
    a.cg.crep = (d->carry && d->carry_frames > 0) ? d->npad - d->carry_frames : 0;
    a.cg.T = d->T; a.cg.Hi = d->Hi; a.cg.Wi = d->Wi; a.cg.Cin = d->Cin;
    a.cg.kt = d->kt; a.cg.kh = d->kh; a.cg.kw = d->kw; a.cg.st = d->st; a.cg.sh = d->sh; a.cg.sw = d->sw;
    a.cg.npad = d->npad; a.cg.pad_h0 = d->pad_h0; a.cg.pad_w0 = d->pad_w0;
    a.cg.To = d->To; a.cg.Ho = d->Ho; a.cg.Wo = d->Wo;
}

static bool conv_halo_shape(const svr2_conv3d_desc* d) {
    return d->kh == 3 && d->kw == 3 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pad_h0 == 1 && d->pad_w0 == 1 &&
           d->Ho == d->Hi && d->Wo == d->Wi;
}

extern "C" int64_t svr2_conv3d_split_ws_bytes(const svr2_conv3d_desc* d) {
    if (!d || d->w_layout != SVR2_WL_TILED_128x32 || !conv_halo_shape(d) || d->Cin <= 0 || d->Cin % 32 || d->Cout <= 0 ||
        d->To <= 0 || d->Ho <= 0 || d->Wo <= 0 || (long long)d->To * d->Ho * d->Wo >= (1ll << 31) - 256)
        return 0;
    KArgs a;
    conv_kargs(d, a);
    return (int64_t)conv_halo_split_bytes(a);
}

extern "C" int svr2_conv3d(const svr2_conv3d_desc* d, void* stream) {
    SVR2_CHECK(d && d->x && d->w && d->y && d->zero_page, SVR2_E_ARG, "svr2_conv3d: null pointer");
    SVR2_CHECK(d->carry_frames >= 0 && d->carry_frames <= d->npad, SVR2_E_SHAPE,
               "svr2_conv3d: carry_frames=%d outside [0, npad=%d]", d->carry_frames, d->npad);
    if (d->w_layout == SVR2_WL_CIN3) {   // fused conv_in (conv_in.hip)
        SVR2_CHECK(d->Cin == 3 && d->Cout == 128 && d->kt == 3 && conv_halo_shape(d) && d->npad == 2 && d->To == d->T &&
                       d->T >= 1 && d->Hi >= 1 && d->Wi >= 1,
                   SVR2_E_SHAPE, "svr2_conv3d: SVR2_WL_CIN3 needs Cin 3, Cout 128, 3x3x3, stride 1, pad 1, npad 2");
        SVR2_CHECK(!d->resid, SVR2_E_ARG, "svr2_conv3d: SVR2_WL_CIN3 has no residual input");
        SVR2_CHECK(aligned16(d->w) && aligned16(d->y) && (((uintptr_t)d->x) & 1) == 0 &&
                       (!d->carry || (((uintptr_t)d->carry) & 1) == 0),
                   SVR2_E_ALIGN, "svr2_conv3d: SVR2_WL_CIN3 needs 16-byte aligned w and y");
        SVR2_CHECK((long long)d->To * d->Ho * d->Wo < (1ll << 31) - 256, SVR2_E_SHAPE,
                   "svr2_conv3d: too many output pixels for one call");
        return conv_in3_launch(d, (hipStream_t)stream);
    }
    const bool tiled = d->w_layout == SVR2_WL_TILED_128x32;
    SVR2_CHECK(d->w_layout == SVR2_WL_ROWS || tiled, SVR2_E_ARG, "svr2_conv3d: unknown w_layout %d", d->w_layout);
    SVR2_CHECK(d->Cin > 0 && d->Cin % (tiled ? 32 : BK) == 0, SVR2_E_SHAPE,
               "svr2_conv3d: Cin %% %d != 0 (Cin=%d); use svr2_im2col3d", tiled ? 32 : BK, d->Cin);
    SVR2_CHECK(d->Cout > 0 && (d->Cout % 8 == 0 || d->resid == nullptr), SVR2_E_SHAPE,
               "svr2_conv3d: Cout %% 8 != 0 needs resid == NULL (Cout=%d)", d->Cout);
    SVR2_CHECK(d->kt >= 1 && d->kh >= 1 && d->kw >= 1 && d->st >= 1 && d->sh >= 1 && d->sw >= 1 && d->npad >= 0,
               SVR2_E_SHAPE, "svr2_conv3d: bad kernel/stride");
    SVR2_CHECK(d->T >= 1 && (d->T + d->npad - d->kt) >= 0 && d->To == (d->T + d->npad - d->kt) / d->st + 1,
               SVR2_E_SHAPE, "svr2_conv3d: To=%d inconsistent with T=%d npad=%d kt=%d st=%d", d->To, d->T, d->npad,
               d->kt, d->st);
    // every input coordinate the kernel forms must be either in range or rejected by its bounds test:
    SVR2_CHECK(d->Ho >= 1 && d->Wo >= 1 && (d->Ho - 1) * d->sh - d->pad_h0 < d->Hi &&
                   (d->Wo - 1) * d->sw - d->pad_w0 < d->Wi,
               SVR2_E_SHAPE, "svr2_conv3d: Ho/Wo inconsistent with input size");
    SVR2_CHECK((d->To - 1) * d->st + d->kt - 1 < d->T + d->npad, SVR2_E_SHAPE, "svr2_conv3d: temporal overrun");
    SVR2_CHECK(aligned16(d->x) && aligned16(d->w) && aligned16(d->y) && (!d->carry || aligned16(d->carry)) &&
                   (!d->resid || aligned16(d->resid)),
               SVR2_E_ALIGN, "svr2_conv3d: pointers must be 16-byte aligned");
    const long long M = (long long)d->To * d->Ho * d->Wo;
    SVR2_CHECK(M < (1ll << 31) - 256, SVR2_E_SHAPE, "svr2_conv3d: too many output pixels for one call");
    KArgs a;
    conv_kargs(d, a);
    if (tiled) {   // halo-resident kernel (conv_halo.hip): 3x3 spatial taps, stride 1, zero padding 1 on every side
        SVR2_CHECK(conv_halo_shape(d), SVR2_E_SHAPE, "svr2_conv3d: SVR2_WL_TILED_128x32 needs a 3x3 stride-1 pad-1 conv");
        SVR2_CHECK(!d->split_ws || aligned16(d->split_ws), SVR2_E_ALIGN, "svr2_conv3d: split_ws must be 16-byte aligned");
        a.stats = d->gn_partial;
        a.partial = (float*)d->split_ws;
        a.partial_bytes = d->split_ws ? (size_t)d->split_ws_bytes : 0;
        return conv_halo_launch(a, (hipStream_t)stream);
    }
    SVR2_CHECK(d->gn_partial == nullptr, SVR2_E_ARG, "svr2_conv3d: gn_partial needs w_layout SVR2_WL_TILED_128x32");
    return launch<A_CONV, SVR2_EPI_BIAS>(a, (hipStream_t)stream);
}
