// Variable-length flash attention for gfx950, bf16 in/out, fp32 softmax/accumulate: the NaDiT window
// attention (20/24 heads x 128, sequences = [window tokens ; text tokens]).  (The VAE's d=512 attention
// lives in vae.hip: at 512 dims a 32-query wave needs 384+ live registers and spills, measured 8x slower
// than the 16-query 16x16x32 form used there.)
//
// One 256-thread workgroup = 128 queries of one (sequence, head); each of the 4 waves owns 32 queries and
// walks the sequence's keys in tiles of KT.  K and V tiles are staged by direct global->LDS loads
// (2-deep ring, next tile in flight during the current one).  Matrix work is v_mfma_f32_32x32x16_bf16 in
// the "swapped" orientation:
//     S^T[key][query] = K . Q^T      (A = K fragment from LDS, B = Q fragment held in registers)
//     O^T[d][query]  += V^T . P^T    (A = V^T via ds_read_b64_tr_b16, B = P^T straight from the S^T
//                                     accumulators: no LDS round trip, no cross-lane traffic for P)
// so a lane owns one query column: running max / sum / rescale are per-lane scalars; the two lanes
// (l, l^32) that share a query exchange only their tile maxima.  VALU diet per tile: the softmax scale is
// folded into one FMA feeding v_exp, key masking runs only on a sequence's last tile, and the O^T rescale is
// deferred until some query's maximum grows by more than 2^RESCALE_THR (wave-uniform branch).
// LDS images are lane-linear (DMA), swizzled through the per-lane source address:
//     K: 16-byte chunk ^= key & 15          (row reads by ds_read_b128: conflict-free)
//     V: 16-byte chunk ^= (key & 3) << 2    (4-key x 16-col transposed reads: conflict-free)
#include "common.h"

namespace {

constexpr int QT = 128, NTH = 256;
constexpr float RESCALE_THR = 4.0f;  // log2 units: P stays <= 16 between rescales

struct AttnArgs {
    const bf16* q; const bf16* k; const bf16* v; bf16* out;
    const int32_t* cu; const int32_t* tile_seq; const int32_t* tile_q0;
    const int32_t* out_rows;   // optional: output row of every position (scatter, e.g. back to natural token order)
    long long ld, ldo;   // elements between consecutive positions (q/k/v and out)
    int fixed_len;       // > 0: sequences are frames of this length (cu/tile tables unused)
    float scale_log2;
    // XCD-aware tile order (ntiles > 0): a 1-D grid of 8 * per_xcd blocks; block b works on the logical tile
    // (b % 8) * per_xcd + b / 8 of the [head][tile] list, so every XCD (blocks are dealt round-robin over the 8 XCDs)
    // walks a contiguous run of it: the query tiles of one (window, head) run side by side on ONE XCD and share the
    // window's K/V through that XCD's L2 instead of each fetching them from HBM.
    int ntiles, total, per_xcd;
};

template <int HD, int KT>
__global__ __launch_bounds__(NTH, (HD <= 128 ? (KT <= 32 ? 3 : 2) : 1)) void flash_attn_kernel(const AttnArgs a) {
    constexpr int RB = HD * 2;            // bytes per K/V row
    constexpr int CPR = RB / 16;          // 16-byte chunks per row
    constexpr int KPI = 1024 / RB > 0 ? 1024 / RB : 1;  // keys per 1-KiB DMA instruction
    constexpr int TILE_BYTES = KT * RB;
    constexpr int NI = TILE_BYTES / 1024 / 4;  // DMA instructions per wave per tile (K or V)
    constexpr int NKS = HD / 16, NDT = HD / 32, NK2 = KT / 32;
    static_assert(RB <= 1024 && TILE_BYTES % 4096 == 0, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int head = blockIdx.y, tile = blockIdx.x;
    if (a.ntiles > 0) {
        const int logical = (int)(blockIdx.x & 7) * a.per_xcd + (int)(blockIdx.x >> 3);
        if (logical >= a.total) return;   // whole workgroup: before any barrier
        head = logical / a.ntiles;
        tile = logical - head * a.ntiles;
    }
    int q0, len;
    long long s0;
    if (a.fixed_len > 0) {
        s0 = (long long)blockIdx.z * a.fixed_len; len = a.fixed_len; q0 = tile * QT;
    } else {
        const int seq = a.tile_seq[tile];
        q0 = a.tile_q0[tile];
        s0 = a.cu[seq]; len = a.cu[seq + 1] - (int)s0;
    }
    const int l31 = lane & 31, hh = lane >> 5;
    const long long hoff = (long long)head * HD;

    // ---- Q fragments (B operand): lane = (query l31, k-half hh), NKS k-steps of 16 dims
    bf16x8 qf[NKS];
    {
        int qr = q0 + wave * 32 + l31;
        if (qr > len - 1) qr = len - 1;
        const bf16* qp = a.q + (s0 + qr) * a.ld + hoff + hh * 8;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }

    // ---- K/V tile loader: DMA instruction g = wave*NI + i moves KPI keys (1 KiB)
    const int lkey = lane / CPR, lslot = lane % CPR;
    auto issue = [&](int t, int stage) {
        char* sK = smem + stage * TILE_BYTES;
        char* sV = smem + (2 + stage) * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int key = (wave * NI + i) * KPI + lkey;
            int kr = t * KT + key;
            if (kr > len - 1) kr = len - 1;
            const long long row = (s0 + kr) * a.ld + hoff;
            glds16(a.k + row + ((lslot ^ (key & 15)) << 3), sK + (wave * NI + i) * 1024);
            glds16(a.v + row + ((lslot ^ ((key & 3) << 2)) << 3), sV + (wave * NI + i) * 1024);
        }
    };

    f32x16 oacc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;   // m_run in raw score units (before the softmax scale)
    const float c = a.scale_log2;

    // transposed-read lane constants: 16-lane group g -> d block, lane i -> (key row i>>2, cols 4*(i&3))
    const int tg = (lane >> 4) & 1, ti = lane & 15;
    const int tq = ti >> 2, tp = ti & 3;
    // Swizzled LDS offsets split into a lane-dependent table indexed by the low bits of the k-step / d-tile
    // and a compile-time part that folds into the ds_read offset field (keeps address VGPRs to 8 + 4):
    //   K chunk (2ks+hh) ^ (key&15)      = (((ks&7) ^ (key>>1&7)) << 1 | (hh ^ (key&1)))  + 16*(ks>>3)
    //   V chunk (4dt+2tg+(tp>>1)) ^ ((key&3)<<2) = (((dt&3) ^ (key&3)) << 2 | 2tg | tp>>1) + 16*(dt>>2)
    int koff[8], voff[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) koff[j] = l31 * RB + (((((j ^ (l31 >> 1)) & 7) << 1) | (hh ^ (l31 & 1))) << 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        voff[j] = (4 * hh + tq) * RB + (((((j ^ tq) & 3) << 2) | (2 * tg) | (tp >> 1)) << 4) + (tp & 1) * 8;

    const int nt = (len + KT - 1) / KT;
    issue(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
        const char* sK = smem + (t & 1) * TILE_BYTES;
        const char* sV = smem + (2 + (t & 1)) * TILE_BYTES;

        // ---- S^T = K . Q^T  (32-key halves)
        f32x16 sacc[NK2];
#pragma unroll
        for (int k2 = 0; k2 < NK2; ++k2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[k2][r] = 0.f;
            const char* kp = sK + k2 * 32 * RB;   // key = k2*32 + l31: (key & 15) == (l31 & 15)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kp + koff[ks & 7] + (ks >> 3) * 256);
                sacc[k2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[k2], 0, 0, 0);
                if (HD > 128 && (ks & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // bound the live K fragments
            }
        }

        // ---- online softmax; this lane holds keys (r&3) + 8*(r>>2) + 4*hh (+32*k2) of its query
        if (t == nt - 1 && (len % KT) != 0) {  // only a sequence's last tile can hold padding keys
            const int kbase = t * KT + 4 * hh;
#pragma unroll
            for (int k2 = 0; k2 < NK2; ++k2)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + k2 * 32 + (r & 3) + 8 * (r >> 2) >= len) sacc[k2][r] = -INFINITY;
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int k2 = 0; k2 < NK2; ++k2)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[k2][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // deferred rescale: P = 2^(c*(s - m_run)) is allowed to reach 2^RESCALE_THR before m_run moves
        if (__builtin_amdgcn_ballot_w64((mx - m_run) * c > RESCALE_THR) != 0) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        }
        const float mc = m_run * c;
        float psum = 0.f;
        bf16x8 pf[NK2][2];
#pragma unroll
        for (int k2 = 0; k2 < NK2; ++k2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[k2][r], c, -mc));
                psum += p;
                pf[k2][r >> 3][r & 7] = f2bf(p);
            }
        l_run += psum;

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int k2 = 0; k2 < NK2; ++k2)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // key0 = k2*32 + 16*s + 4*hh + tq is the first 4-key block (key0 & 3 == tq); the second is +8
                const char* vrow = sV + (k2 * 32 + 16 * s) * RB;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const char* p0 = vrow + voff[dt & 3] + (dt >> 2) * 256;
                    const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)LDS_PTR(p0));
                    const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)LDS_PTR(p0 + 8 * RB));  // key0+8: same key&3
                    const bf16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[k2][s], oacc[dt], 0, 0, 0);
                    if (HD > 128 && (dt & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the live V fragments
                }
            }
    }

    // ---- normalise and store: lane holds d = dt*32 + (r&3) + 8*(r>>2) + 4*hh of query l31
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qr = q0 + wave * 32 + l31;
    if (qr < len) {
        const long long orow = a.out_rows ? (long long)a.out_rows[s0 + qr] : (long long)(s0 + qr);
        bf16* op = a.out + orow * a.ldo + hoff + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(oacc[dt][g * 4 + e] * inv);
                *(bf16x4*)(op + dt * 32 + g * 8) = o;
            }
    }
}

// ---- second-generation window-attention kernel (HD = 128), the default: same algorithm, same tile geometry and the same
// summation order as flash_attn_kernel (bit-identical results for equal KT) with lane-constant DMA piece offsets (one
// 64-bit add per piece; only a sequence's last partial tile clamps rows).  The variant that requested the LDS fragment
// groups one group ahead in a second register set was measured slower (721-737 vs 762-780 TFLOP/s at 4K: with both waves
// of a SIMD in MFMA bursts the bursts collide; profiles/r02_attn/) and is gone.
template <int KT>
__global__ __launch_bounds__(NTH, (KT <= 32 ? 3 : 2)) void flash_attn2_kernel(const AttnArgs a) {
    constexpr int HD = 128, RB = HD * 2, CPR = RB / 16, KPI = 1024 / RB;
    constexpr int TILE_BYTES = KT * RB;
    constexpr int NI = TILE_BYTES / 1024 / 4;
    constexpr int NKS = HD / 16, NDT = HD / 32, NK2 = KT / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int logical = (int)(blockIdx.x & 7) * a.per_xcd + (int)(blockIdx.x >> 3);
    if (logical >= a.total) return;
    const int head = logical / a.ntiles;
    const int tile = logical - head * a.ntiles;
    const int seq = a.tile_seq[tile];
    const int q0 = a.tile_q0[tile];
    const long long s0 = a.cu[seq];
    const int len = a.cu[seq + 1] - (int)s0;
    const int l31 = lane & 31, hh = lane >> 5;
    const long long hoff = (long long)head * HD;

    bf16x8 qf[NKS];
    {
        int qr = q0 + wave * 32 + l31;
        if (qr > len - 1) qr = len - 1;
        const bf16* qp = a.q + (s0 + qr) * a.ld + hoff + hh * 8;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }
    const int lkey = lane / CPR, lslot = lane % CPR;
    // K/V tile loader.  Full tiles: every lane keeps the byte offset of its NI pieces from the tile's first row (lane
    // constants: key row and swizzled 16-byte chunk) and adds the tile's base, so a piece costs one 64-bit add; only a
    // sequence's last, partial tile clamps its rows to the sequence (the address arithmetic of the clamped form is
    // ~70 vector instructions per tile and wave, a quarter of the loop's VALU issue time when paid on every tile).
    long long koffs[NI], voffs[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int key = (wave * NI + i) * KPI + lkey;
        koffs[i] = ((long long)key * a.ld + ((lslot ^ (key & 15)) << 3)) * 2;
        voffs[i] = ((long long)key * a.ld + ((lslot ^ ((key & 3) << 2)) << 3)) * 2;
    }
    const char* kseq = (const char*)(a.k + s0 * a.ld + hoff);
    const char* vseq = (const char*)(a.v + s0 * a.ld + hoff);
    const long long tile_step = (long long)KT * a.ld * 2;
    auto issue = [&](int t, int stage) {
        char* sK = smem + stage * TILE_BYTES;
        char* sV = smem + (2 + stage) * TILE_BYTES;
        if ((t + 1) * KT <= len) {
            const char* kb = kseq + t * tile_step;
            const char* vb = vseq + t * tile_step;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                glds16(kb + koffs[i], sK + (wave * NI + i) * 1024);
                glds16(vb + voffs[i], sV + (wave * NI + i) * 1024);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int key = (wave * NI + i) * KPI + lkey;
            int kr = t * KT + key;
            if (kr > len - 1) kr = len - 1;
            const long long row = (s0 + kr) * a.ld + hoff;
            glds16(a.k + row + ((lslot ^ (key & 15)) << 3), sK + (wave * NI + i) * 1024);
            glds16(a.v + row + ((lslot ^ ((key & 3) << 2)) << 3), sV + (wave * NI + i) * 1024);
        }
    };

    f32x16 oacc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c = a.scale_log2;
    const int tg = (lane >> 4) & 1, ti = lane & 15;
    const int tq = ti >> 2, tp = ti & 3;
    int koff[8], voff[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) koff[j] = l31 * RB + (((((j ^ (l31 >> 1)) & 7) << 1) | (hh ^ (l31 & 1))) << 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        voff[j] = (4 * hh + tq) * RB + (((((j ^ tq) & 3) << 2) | (2 * tg) | (tp >> 1)) << 4) + (tp & 1) * 8;

    auto load_v = [&](const char* sV, int g, bf16x8* vf) {      // g = k2*2 + s: the 16-key step
        const char* vrow = sV + g * 16 * RB;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const char* p0 = vrow + voff[dt];
            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)LDS_PTR(p0));
            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)LDS_PTR(p0 + 8 * RB));
            vf[dt] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };

    const int nt = (len + KT - 1) / KT;
    issue(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
        const char* sK = smem + (t & 1) * TILE_BYTES;
        const char* sV = smem + (2 + (t & 1)) * TILE_BYTES;

        // ---- S^T = K . Q^T
        f32x16 sacc[NK2];
        bf16x8 vfa[NDT];
#pragma unroll
        for (int k2 = 0; k2 < NK2; ++k2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[k2][r] = 0.f;
            const char* kp = sK + k2 * 32 * RB;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kp + koff[ks]);
                sacc[k2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[k2], 0, 0, 0);
            }
        }
        if (t == nt - 1 && (len % KT) != 0) {
            const int kbase = t * KT + 4 * hh;
#pragma unroll
            for (int k2 = 0; k2 < NK2; ++k2)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + k2 * 32 + (r & 3) + 8 * (r >> 2) >= len) sacc[k2][r] = -INFINITY;
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int k2 = 0; k2 < NK2; ++k2)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[k2][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (__builtin_amdgcn_ballot_w64((mx - m_run) * c > RESCALE_THR) != 0) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        }
        const float mc = m_run * c;
        float psum = 0.f;
        bf16x8 pf[NK2][2];
#pragma unroll
        for (int k2 = 0; k2 < NK2; ++k2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[k2][r], c, -mc));
                psum += p;
                pf[k2][r >> 3][r & 7] = f2bf(p);
            }
        l_run += psum;

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int g = 0; g < 2 * NK2; ++g) {
            load_v(sV, g, vfa);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfa[dt], pf[g >> 1][g & 1], oacc[dt], 0, 0, 0);
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qr = q0 + wave * 32 + l31;
    if (qr < len) {
        const long long orow = a.out_rows ? (long long)a.out_rows[s0 + qr] : (long long)(s0 + qr);
        bf16* op = a.out + orow * a.ldo + hoff + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(oacc[dt][g * 4 + e] * inv);
                *(bf16x4*)(op + dt * 32 + g * 8) = o;
            }
    }
}

template <int KT>
int launch_attn2(const AttnArgs& a, dim3 grid, hipStream_t s) {
    constexpr int LDS = 4 * KT * 128 * 2;
    SVR2_SET_MAX_LDS((flash_attn2_kernel<KT>), LDS, "flash_attn2");
    hipLaunchKernelGGL((flash_attn2_kernel<KT>), grid, dim3(NTH), LDS, s, a);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

template <int HD, int KT>
int launch_attn(const AttnArgs& a, dim3 grid, hipStream_t s) {
    constexpr int LDS = 4 * KT * HD * 2;
    SVR2_SET_MAX_LDS((flash_attn_kernel<HD, KT>), LDS, "flash_attn");
    hipLaunchKernelGGL((flash_attn_kernel<HD, KT>), grid, dim3(NTH), LDS, s, a);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

}  // namespace

static int window_attn_impl(const void* q, const void* k, const void* v, void* out, const int32_t* out_rows, const int32_t* cu_seqlens,
                                const int32_t* tile_seq, const int32_t* tile_q0, int32_t ntiles, int32_t H, int32_t D,
                                float scale, void* stream) {
    SVR2_CHECK(q && k && v && out && cu_seqlens && tile_seq && tile_q0, SVR2_E_ARG, "svr2_window_attn: null pointer");
    SVR2_CHECK(D == 128, SVR2_E_SHAPE, "svr2_window_attn: head_dim must be 128 (got %d)", D);
    SVR2_CHECK(ntiles > 0 && H > 0 && H <= 65535, SVR2_E_SHAPE, "svr2_window_attn: bad ntiles/H");
    SVR2_CHECK(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out), SVR2_E_ALIGN,
               "svr2_window_attn: pointers must be 16-byte aligned");
    AttnArgs a;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.out = (bf16*)out;
    a.cu = cu_seqlens; a.tile_seq = tile_seq; a.tile_q0 = tile_q0; a.out_rows = out_rows;
    a.ld = (long long)H * 128; a.ldo = a.ld; a.fixed_len = 0;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.ntiles = a.total = a.per_xcd = 0;
    // Both kernels: XCD-aware 1-D grid, 32-key tiles, three workgroups per CU.  Default = flash_attn2_kernel (one-add DMA
    // addressing; the round-2 A/B of six variants is in profiles/r02_attn/, the slower ones are gone); SVR2_ATTN_V=1 =
    // the plain flash_attn_kernel of the same tile geometry and summation order: results must be BIT-IDENTICAL
    // (tests/test_gpu_kernels.py), which is what the switch is for.
    const long long total = (long long)ntiles * H;
    SVR2_CHECK(total < (1ll << 30), SVR2_E_SHAPE, "svr2_window_attn: too many tiles");
    a.ntiles = ntiles; a.total = (int)total; a.per_xcd = (int)((total + 7) / 8);
    if (svr2::opt(svr2::OPT_ATTN_V, 5) == 1) return launch_attn<128, 32>(a, dim3(a.per_xcd * 8, 1, 1), (hipStream_t)stream);
    return launch_attn2<32>(a, dim3(a.per_xcd * 8, 1, 1), (hipStream_t)stream);
}

extern "C" int svr2_window_attn(const void* q, const void* k, const void* v, void* out, const int32_t* cu_seqlens,
                                const int32_t* tile_seq, const int32_t* tile_q0, int32_t ntiles, int32_t H, int32_t D,
                                float scale, void* stream) {
    return window_attn_impl(q, k, v, out, nullptr, cu_seqlens, tile_seq, tile_q0, ntiles, H, D, scale, stream);
}

extern "C" int svr2_window_attn_scatter(const void* q, const void* k, const void* v, void* out, const int32_t* out_rows,
                                        const int32_t* cu_seqlens, const int32_t* tile_seq, const int32_t* tile_q0,
                                        int32_t ntiles, int32_t H, int32_t D, float scale, void* stream) {
    SVR2_CHECK(out_rows, SVR2_E_ARG, "svr2_window_attn_scatter: null out_rows");
    return window_attn_impl(q, k, v, out, out_rows, cu_seqlens, tile_seq, tile_q0, ntiles, H, D, scale, stream);
}
