// VAE-side kernels: per-frame GroupNorm (+SiLU), HBM-bound (2 reads + 1 write of the activation).
// (The mid-block d=512 spatial attention is the <512,32> instance of csrc/attn.hip.)
#include "common.h"

namespace {

// ----------------------------------------------------------------------------- GroupNorm statistics
// x [F][P][C] bf16 (channels-last).  Pass 1: every workgroup reduces a strip of pixels of one frame to
// per-channel (sum, sumsq) fp32 partials.  Pass 2: one wave per (frame, group) folds the partials in
// fp64 and writes per-(frame, channel) affine coefficients a = rstd*gamma, b = beta - mean*rstd*gamma.
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16* __restrict__ x, float* __restrict__ partial,
                                                         long long P, int C, int nblk) {
    __shared__ float red[256 * 16];
    const int nvec = C >> 3, ppb = 256 / nvec;
    const int vec = threadIdx.x % nvec, prow = threadIdx.x / nvec;
    const int f = blockIdx.y, blk = blockIdx.x;
    const long long span = (P + nblk - 1) / nblk;
    const long long p0 = blk * span, p1 = min(P, p0 + span);
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
    if (prow < ppb) {
        const bf16* base = x + ((long long)f * P) * C + vec * 8;
        for (long long p = p0 + prow; p < p1; p += ppb) {
            const bf16x8 v = *(const bf16x8*)(base + p * C);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = bf2f(v[e]);
                s[e] += t;
                ss[e] += t * t;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[threadIdx.x * 16 + e] = s[e];
        red[threadIdx.x * 16 + 8 + e] = ss[e];
    }
    __syncthreads();
    // threads [0, nvec*16): (vec, e16) pairs sum over prow
    for (int i = threadIdx.x; i < nvec * 16; i += 256) {
        const int v = i >> 4, e = i & 15;
        float acc = 0.f;
        for (int r = 0; r < ppb; ++r) acc += red[(r * nvec + v) * 16 + e];
        // partial layout [F][nblk][2][C]
        const int c = v * 8 + (e & 7);
        partial[(((long long)f * nblk + blk) * 2 + (e >> 3)) * C + c] = acc;
    }
}

// (sum, sumsq) of CPG channels of four tiles 256 apart (partial rows are [2][C] floats), added in tile then channel order
template <int CPG>
__device__ __forceinline__ void gn_fold4(const float* __restrict__ ps, int C, double& s, double& ss) {
    f32x4 v[4][2][CPG / 4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < CPG / 4; ++q) {
            v[u][0][q] = *(const f32x4*)(ps + (long long)u * 512 * C + q * 4);
            v[u][1][q] = *(const f32x4*)(ps + (long long)u * 512 * C + C + q * 4);
        }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < CPG / 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s += (double)v[u][0][q][e];
                ss += (double)v[u][1][q][e];
            }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ coef,
                                                          long long P, int C, int G, int nblk, float eps) {
    __shared__ double red[2][4];
    const int f = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpg = C / G;
    double s = 0.0, ss = 0.0;
    // each thread folds every 256th tile in order (fixed summation order: the sums are the same bits as those of the plain
    // loop below); gn_fold4 issues the loads of four tiles before the first add - at 4K a frame has 16 200 tiles and the
    // dependent-load form of this loop took 76 us per norm (0.4 % of a step)
    int b = tid;
    const float* base = partial + ((long long)f * nblk * 2) * C + g * cpg;
    const bool vec = (((uintptr_t)partial) & 15) == 0;         // 16-byte loads (C = G * cpg is a multiple of 4 below)
    if (!vec) {}
    else if (cpg == 4) for (; b + 3 * 256 < nblk; b += 4 * 256) gn_fold4<4>(base + (long long)b * 2 * C, C, s, ss);
    else if (cpg == 8) for (; b + 3 * 256 < nblk; b += 4 * 256) gn_fold4<8>(base + (long long)b * 2 * C, C, s, ss);
    else if (cpg == 16) for (; b + 3 * 256 < nblk; b += 4 * 256) gn_fold4<16>(base + (long long)b * 2 * C, C, s, ss);
    for (; b < nblk; b += 256) {
        const float* ps = partial + (((long long)f * nblk + b) * 2) * C + g * cpg;
        for (int c = 0; c < cpg; ++c) {
            s += (double)ps[c];
            ss += (double)ps[C + c];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        ss += __shfl_xor(ss, o, 64);
    }
    if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; }
    __syncthreads();
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    ss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double n = (double)P * cpg;
    const double mean = s / n;
    double var = ss / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = tid; c < cpg; c += 256) {
        const int ch = g * cpg + c;
        const float a = rstd * gamma[ch];
        coef[((long long)f * C + ch) * 2 + 0] = a;
        coef[((long long)f * C + ch) * 2 + 1] = beta[ch] - (float)mean * a;
    }
}

template <bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16* __restrict__ x, const float* __restrict__ coef,
                                                       bf16* __restrict__ y, long long P, int C, int nblk,
                                                       bf16* __restrict__ carry, int f0) {
    const int nvec = C >> 3, ppb = 256 / nvec;
    const int vec = threadIdx.x % nvec, prow = threadIdx.x / nvec;
    if (prow >= ppb) return;
    const int f = blockIdx.y, blk = blockIdx.x;
    const long long span = (P + nblk - 1) / nblk;
    const long long p0 = blk * span, p1 = min(P, p0 + span);
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = coef[((long long)f * C + vec * 8 + e) * 2];
        b[e] = coef[((long long)f * C + vec * 8 + e) * 2 + 1];
    }
    const long long base = ((long long)f * P) * C + vec * 8;
    // frames >= f0 are also written to the consumer conv's carry buffer (its last kt - st input frames)
    bf16* const cdst = (carry && f >= f0) ? carry + ((long long)(f - f0) * P) * C + vec * 8 : nullptr;
    // four independent 16-byte loads in flight per thread (the pass is HBM-bound: latency x bandwidth needs ~10 MB in
    // flight chip-wide); streaming data: non-temporal loads and stores keep it from displacing the conv weights in L2
    auto one = [&](long long p, const bf16x8 v) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = bf2f(v[e]) * a[e] + b[e];
            if (SILU) t = t / (1.f + __expf(-t));
            o[e] = f2bf(t);
        }
        __builtin_nontemporal_store(o, (bf16x8*)(y + base + p * C));
        if (cdst) *(bf16x8*)(cdst + p * C) = o;
    };
    long long p = p0 + prow;
    for (; p + 3 * ppb < p1; p += 4 * ppb) {
        bf16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load((const bf16x8*)(x + base + (p + u * ppb) * C));
#pragma unroll
        for (int u = 0; u < 4; ++u) one(p + u * ppb, v[u]);
    }
    for (; p < p1; p += ppb) one(p, __builtin_nontemporal_load((const bf16x8*)(x + base + p * C)));
}

// ----------------------------------------------------------------------------- attention, head dim 512
constexpr int D5 = 512, QB5 = 128, KB5 = 32, NTH5 = 512;
constexpr int KV5_BYTES = KB5 * D5 * 2;   // 32 KiB per K or V tile
constexpr int ATT5_LDS = 4 * KV5_BYTES;   // K[2] + V[2] = 128 KiB

struct Attn5Args {
    const bf16* q; const bf16* k; const bf16* v; bf16* out;
    long long ld, ldo;
    int P;  // tokens per frame
    float scale_log2;
};

template <int PIPE>
__global__ __launch_bounds__(NTH5) void attn_d512_kernel(const Attn5Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const long long s0 = (long long)blockIdx.y * a.P;
    const int q0 = blockIdx.x * QB5;
    const int len = a.P;

    bf16x8 qf[16];
    {
        int qr = q0 + wave * 16 + l15;
        if (qr > len - 1) qr = len - 1;
        const bf16* qp = a.q + (s0 + qr) * a.ld + kq * 8;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 32);
    }

    auto issue = [&](int t, int stage) {
        char* sK = smem + stage * KV5_BYTES;
        char* sV = smem + (2 + stage) * KV5_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = wave * 4 + i;                         // wave-uniform: the row base below lives in SGPRs,
            int kr = t * KB5 + key;                               // the lane adds only its swizzled 16-byte chunk
            if (kr > len - 1) kr = len - 1;
            const long long row = (s0 + kr) * a.ld;
            // the lane's swizzled chunk offset is recomputed per piece (one v_xor + shift): hoisted out of the loop the
            // eight per-lane 64-bit row pointers cost 16 VGPRs, which the pipelined loops need for their second
            // fragment group
            int l = lane;
            asm volatile("" : "+v"(l));
            glds16(a.k + row + ((l ^ (key & 15)) << 3), sK + key * 1024);
            glds16(a.v + row + ((l ^ ((key & 7) << 1)) << 3), sV + key * 1024);
        }
    };

    f32x4 oacc[32];
#pragma unroll
    for (int dt = 0; dt < 32; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const int tq = l15 >> 2, tp = l15 & 3;
    // lane-dependent swizzled offsets.  The swizzle is an XOR of a 2- / 3-bit field with the fragment index j, so
    // offset(j) = offset(0) ^ (j << 6) for K and ^ (j << 5) for V: ONE register each survives the loop; the tables of 4 / 8
    // are rebuilt per tile from the tile's stage base (K table live during QK^T only, V table during PV only) and the
    // compile-time parts are folded into the ds_read offsets
    const int koff0 = l15 * 1024 + (((((l15 >> 2) & 3) << 2) | (kq ^ (l15 & 3))) << 4);
    const int voff0 = (4 * kq + tq) * 1024 + (((((4 * kq + tq) & 7) << 1) | (tp >> 1)) << 4) + (tp & 1) * 8;
    const int nt = (len + KB5 - 1) / KB5;
    issue(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
        const char* sK = smem + (t & 1) * KV5_BYTES;
        const char* sV = smem + (2 + (t & 1)) * KV5_BYTES;
        int koff[4], voff[8];
        {
            int kb = koff0, vb = voff0;
            asm volatile("" : "+v"(kb), "+v"(vb));     // per tile, not hoisted: 12 registers that the loop cannot spare
#pragma unroll
            for (int j = 0; j < 4; ++j) koff[j] = kb ^ (j << 6);
#pragma unroll
            for (int j = 0; j < 8; ++j) voff[j] = vb ^ (j << 5);
        }

        f32x4 sacc[2];
        bf16x8 fa[4], fb[4];                      // PIPE: two rotating fragment groups (K fragments, then V fragments)
        if constexpr (PIPE) {
            // software pipeline: the four fragments of group g+1 are requested BEFORE the four MFMAs of group g, so a
            // group's LDS latency runs under the previous group's 64 matrix cycles (hipcc's own schedule of the plain
            // loop is read -> s_waitcnt lgkmcnt(0) -> MFMA with two fragment registers: every round trip exposed,
            // MFMA busy 40.6 % / LDS array 41.8 %, alternating: profiles/r02_attn512)
            auto load_k = [&](int g, bf16x8* f) {     // group g = (k2 = g >> 2, ks = 4 (g & 3) .. +4)
                const char* kp = sK + (g >> 2) * 16 * 1024 + (g & 3) * 256;
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = *(const bf16x8*)(kp + koff[j]);
            };
            load_k(0, fa);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                bf16x8* cur = (g & 1) ? fb : fa;
                bf16x8* nxt = (g & 1) ? fa : fb;
                if (g + 1 < 8) load_k(g + 1, nxt);
                __builtin_amdgcn_sched_barrier(0);
                if ((g & 3) == 0) sacc[g >> 2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    sacc[g >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[j], qf[(g & 3) * 4 + j], sacc[g >> 2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            sacc[k2] = f32x4{0.f, 0.f, 0.f, 0.f};
            const char* kp = sK + k2 * 16 * 1024;   // key = k2*16 + l15: (key & 15) == l15
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kp + koff[ks & 3] + (ks >> 2) * 256);
                sacc[k2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], sacc[k2], 0, 0, 0);
                if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        // lane holds keys 16*k2 + 4*kq + r of query l15
        if (t == nt - 1 && (len % KB5) != 0) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (t * KB5 + k2 * 16 + 4 * kq + r >= len) sacc[k2][r] = -INFINITY;
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[k2][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float c = a.scale_log2;
        if (__builtin_amdgcn_ballot_w64((mx - m_run) * c > 4.0f) != 0) {  // deferred rescale (see attn.hip)
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 32; ++dt) oacc[dt] *= alpha;
        }
        const float mc = m_run * c;
        float psum = 0.f;
        bf16x8 pf;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[k2][r], c, -mc));
                psum += p;
                pf[k2 * 4 + r] = f2bf(p);
            }
        l_run += psum;

        auto load_v = [&](int g, bf16x8* f) {         // group g = dt 4g .. 4g+3
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int dt = g * 4 + j;
                const char* p0 = sV + voff[dt & 7] + (dt >> 3) * 256;
                const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)LDS_PTR(p0));
                const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)LDS_PTR(p0 + 16 * 1024));
                f[j] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        if constexpr (PIPE) {
            // (the first V group could travel under the softmax arithmetic; requested here it already overlaps the
            // exp / convert tail, which the compiler schedules after these reads)
            load_v(0, fa);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                bf16x8* cur = (g & 1) ? fb : fa;
                bf16x8* nxt = (g & 1) ? fa : fb;
                if (g + 1 < 8) load_v(g + 1, nxt);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    oacc[g * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[j], pf, oacc[g * 4 + j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int dt = 0; dt < 32; ++dt) {
            const char* p0 = sV + voff[dt & 7] + (dt >> 3) * 256;
            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)LDS_PTR(p0));
            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)LDS_PTR(p0 + 16 * 1024));
            const bf16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[dt], 0, 0, 0);
            if ((dt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        }
    }

    float l_tot = l_run + __shfl_xor(l_run, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qr = q0 + wave * 16 + l15;
    if (qr < len) {
        bf16* op = a.out + (s0 + qr) * a.ldo + 4 * kq;
#pragma unroll
        for (int dt = 0; dt < 32; ++dt) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(oacc[dt][e] * inv);
            *(bf16x4*)(op + dt * 16) = o;
        }
    }
}

inline int gn_blocks(long long P, int C) {
    const int ppb = 256 / (C >> 3);
    long long nb = (P + (long long)ppb * 8 - 1) / ((long long)ppb * 8);
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    return (int)nb;
}

}  // namespace

/* workspace floats needed by svr2_groupnorm_stats */
extern "C" int64_t svr2_groupnorm_ws_floats(int32_t F, int64_t P, int32_t C) {
    if (F <= 0 || P <= 0 || C <= 0 || C % 8) return 0;
    return (int64_t)F * gn_blocks(P, C) * 2 * C;
}

extern "C" int svr2_groupnorm_stats(const void* x, int32_t F, int64_t P, int32_t C, int32_t G, float eps,
                                    const float* gamma, const float* beta, float* workspace, float* coef,
                                    void* stream) {
    SVR2_CHECK(x && gamma && beta && workspace && coef, SVR2_E_ARG, "svr2_groupnorm_stats: null pointer");
    SVR2_CHECK(F > 0 && F <= 65535 && P > 0 && C >= 8 && C <= 2048 && C % 8 == 0 && 256 % (C / 8) == 0 && G > 0 &&
                   C % G == 0,
               SVR2_E_SHAPE, "svr2_groupnorm_stats: unsupported shape F=%d C=%d G=%d", F, C, G);
    SVR2_CHECK(aligned16(x), SVR2_E_ALIGN, "svr2_groupnorm_stats: x must be 16-byte aligned");
    const int nblk = gn_blocks(P, C);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nblk, F), dim3(256), 0, s, (const bf16*)x, workspace, (long long)P, C,
                       nblk);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, F), dim3(256), 0, s, workspace, gamma, beta, coef, (long long)P, C,
                       G, nblk, eps);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_groupnorm_finalize(const float* partial, int32_t nblk, int32_t F, int64_t P, int32_t C, int32_t G,
                                       float eps, const float* gamma, const float* beta, float* coef, void* stream) {
    SVR2_CHECK(partial && gamma && beta && coef, SVR2_E_ARG, "svr2_groupnorm_finalize: null pointer");
    SVR2_CHECK(F > 0 && F <= 65535 && P > 0 && C > 0 && G > 0 && C % G == 0 && nblk > 0, SVR2_E_SHAPE,
               "svr2_groupnorm_finalize: unsupported shape F=%d C=%d G=%d nblk=%d", F, C, G, nblk);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, F), dim3(256), 0, (hipStream_t)stream, partial, gamma, beta, coef,
                       (long long)P, C, G, nblk, eps);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_groupnorm_apply(const void* x, const float* coef, void* y, int32_t F, int64_t P, int32_t C,
                                    int32_t silu, void* carry_out, int32_t carry_frames, void* stream) {
    SVR2_CHECK(x && coef && y, SVR2_E_ARG, "svr2_groupnorm_apply: null pointer");
    SVR2_CHECK(F > 0 && F <= 65535 && P > 0 && C >= 8 && C <= 2048 && C % 8 == 0 && 256 % (C / 8) == 0, SVR2_E_SHAPE,
               "svr2_groupnorm_apply: unsupported shape");
    SVR2_CHECK(aligned16(x) && aligned16(y) && (!carry_out || aligned16(carry_out)), SVR2_E_ALIGN,
               "svr2_groupnorm_apply: alignment");
    SVR2_CHECK(!carry_out || (carry_frames > 0 && carry_frames <= F), SVR2_E_SHAPE,
               "svr2_groupnorm_apply: carry_frames must be in [1, F]");
    const int nblk = gn_blocks(P, C);
    const int f0 = carry_out ? F - carry_frames : F;
    hipStream_t s = (hipStream_t)stream;
    if (silu)
        hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(nblk, F), dim3(256), 0, s, (const bf16*)x, coef, (bf16*)y,
                           (long long)P, C, nblk, (bf16*)carry_out, f0);
    else
        hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(nblk, F), dim3(256), 0, s, (const bf16*)x, coef, (bf16*)y,
                           (long long)P, C, nblk, (bf16*)carry_out, f0);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}

extern "C" int svr2_attn_d512(const void* q, const void* k, const void* v, int64_t ld, void* out, int64_t ldo,
                                  int32_t frames, int32_t P, float scale, void* stream) {
    SVR2_CHECK(q && k && v && out, SVR2_E_ARG, "svr2_attn_d512: null pointer");
    SVR2_CHECK(frames > 0 && frames <= 65535 && P > 0 && ld >= D5 && ldo >= D5 && ld % 8 == 0 && ldo % 8 == 0,
               SVR2_E_SHAPE, "svr2_attn_d512: bad sizes");
    SVR2_CHECK(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out), SVR2_E_ALIGN,
               "svr2_attn_d512: pointers must be 16-byte aligned");
    // SVR2_ATTN5_V: 1 (default) = the software-pipelined fragment schedule; 0 = the plain loops (A/B; same bits).
    // A third form - pairs of waves that split the head dimension (32 queries per pair on 32x32x16 MFMAs, S^T halves
    // exchanged through LDS: 40 instead of 64 KiB of LDS reads per wave and tile) - was built, parity-green and SLOWER
    // (923 vs 1 063 TFLOP/s at 129 600 tokens, profiles/r03_attn512): the kernel is bound by the K/V delivery into LDS
    // plus reads plus MFMAs in series, not by the LDS reads alone, and the exchange barrier and the duplicated softmax
    // cost more than the reads they save.  Removed; the code is in the history (commit 8fe3fb8).
    // A fourth - waves 4-7 one PV phase behind waves 0-3 (three V stages = all of the LDS), so that a SIMD's DMA issue and
    // softmax phases sit under its partner's MFMAs - was built too, bit-identical, and 2 % SLOWER (1 035 vs 1 054-1 061
    // TFLOP/s): the phases were not what the pipe waits for either.  Removed as well (history: the commit before this one).
    const int ver = svr2::opt(svr2::OPT_ATTN5_V, 1);
    if (ver == 1) SVR2_SET_MAX_LDS(attn_d512_kernel<1>, ATT5_LDS, "svr2_attn_d512");
    else SVR2_SET_MAX_LDS(attn_d512_kernel<0>, ATT5_LDS, "svr2_attn_d512");
    Attn5Args a;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.out = (bf16*)out;
    a.ld = ld; a.ldo = ldo; a.P = P; a.scale_log2 = scale * 1.4426950408889634f;
    if (ver == 1)
        hipLaunchKernelGGL(attn_d512_kernel<1>, dim3((P + QB5 - 1) / QB5, frames), dim3(NTH5), ATT5_LDS, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(attn_d512_kernel<0>, dim3((P + QB5 - 1) / QB5, frames), dim3(NTH5), ATT5_LDS, (hipStream_t)stream, a);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}
