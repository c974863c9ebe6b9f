// Variable-length (per-window) flash attention, head_dim 128, bf16 in/out, fp32 softmax/accumulate.
//
// One 256-thread workgroup = 128 queries of one (window, head); each of the 4 waves owns 32 queries and
// walks the window's keys in tiles of 64.  K and V tiles are staged by direct global->LDS loads
// (2-deep ring, next tile in flight during the current one).  Matrix work uses v_mfma_f32_32x32x16_bf16
// in the "swapped" orientation:
//     S^T[key][query] = K . Q^T      (A = K fragment from LDS, B = Q fragment held in registers)
//     O^T[d][query]  += V^T . P^T    (A = V^T via ds_read_b64_tr_b16, B = P^T straight from the S^T
//                                     accumulators: no LDS round trip, no cross-lane traffic for P)
// so a lane owns one query column: running max / sum / rescale are per-lane scalars; the two lanes
// (l, l^32) that share a query exchange only their tile maxima.
// LDS images are lane-linear (DMA), swizzled through the per-lane source address:
//     K: 16-byte chunk ^= key & 15          (row reads by ds_read_b128: conflict-free)
//     V: 16-byte chunk ^= (key & 3) << 2    (4-key x 16-col transposed reads: conflict-free)
#include "common.h"

namespace {

constexpr int HD = 128, QT = 128, KT = 64, NTH = 256;
constexpr int KV_TILE_BYTES = KT * HD * 2;  // 16 KiB
constexpr int ATT_LDS = 4 * KV_TILE_BYTES;  // K[2] + V[2] = 64 KiB

struct AttnArgs {
    const bf16* q; const bf16* k; const bf16* v; bf16* out;
    const int32_t* cu; const int32_t* tile_seq; const int32_t* tile_q0;
    int H;
    float scale_log2;
};

__global__ __launch_bounds__(NTH) void window_attn_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.y;
    const int seq = a.tile_seq[blockIdx.x], q0 = a.tile_q0[blockIdx.x];
    const int s0 = a.cu[seq], len = a.cu[seq + 1] - s0;
    const long long rstride = (long long)a.H * HD;  // elements between consecutive positions
    const int l31 = lane & 31, hh = lane >> 5;

    // ---- Q fragments (B operand): lane = (query l31, k-half hh), 8 k-steps of 16 dims
    bf16x8 qf[8];
    {
        int qr = q0 + wave * 32 + l31;
        if (qr > len - 1) qr = len - 1;
        const bf16* qp = a.q + ((long long)(s0 + qr) * a.H + head) * HD + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }

    // ---- K/V tile loader: wave w, instruction i moves keys (w*4+i)*4 .. +3 (4 x 256 B)
    const int lkey = lane >> 4, lpos = lane & 15;
    auto issue = [&](int t, int stage) {
        char* sK = smem + stage * KV_TILE_BYTES;
        char* sV = smem + (2 + stage) * KV_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = (wave * 4 + i) * 4 + lkey;
            int kr = t * KT + key;
            if (kr > len - 1) kr = len - 1;
            const long long row = ((long long)(s0 + kr) * a.H + head) * HD;
            glds16(a.k + row + ((lpos ^ (key & 15)) << 3), sK + (wave * 4 + i) * 1024);
            glds16(a.v + row + ((lpos ^ ((key & 3) << 2)) << 3), sV + (wave * 4 + i) * 1024);
        }
    };

    f32x16 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // transposed-read lane constants: 16-lane group g -> d block, lane i -> (key row i>>2, cols 4*(i&3))
    const int tg = (lane >> 4) & 1, ti = lane & 15;
    const int tq = ti >> 2, tp = ti & 3;

    const int nt = (len + KT - 1) / KT;
    issue(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
        const char* sK = smem + (t & 1) * KV_TILE_BYTES;
        const char* sV = smem + (2 + (t & 1)) * KV_TILE_BYTES;

        // ---- S^T = K . Q^T  (two 32-key halves)
        f32x16 sacc[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[k2][r] = 0.f;
            const int key = k2 * 32 + l31;
            const char* kp = sK + key * 256;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kp + (((2 * ks + hh) ^ (key & 15)) << 4));
                sacc[k2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[k2], 0, 0, 0);
            }
        }

        // ---- online softmax; this lane holds keys (r&3) + 8*(r>>2) + 4*hh (+32*k2) of its query
        const int kbase = t * KT + 4 * hh;
        float mx = -INFINITY;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kidx = kbase + k2 * 32 + (r & 3) + 8 * (r >> 2);
                float sv = sacc[k2][r] * a.scale_log2;
                sv = kidx < len ? sv : -INFINITY;
                sacc[k2][r] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        bf16x8 pf[2][2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(sacc[k2][r] - m_new);
                psum += p;
                pf[k2][r >> 3][r & 7] = f2bf(p);
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int key0 = k2 * 32 + 16 * s + 4 * hh + tq;  // first 4-key block; second is +8
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int chunk = dt * 4 + 2 * tg + (tp >> 1);
                    const char* p0 = sV + key0 * 256 + ((chunk ^ ((key0 & 3) << 2)) << 4) + (tp & 1) * 8;
                    const char* p1 = p0 + 8 * 256;  // key0 + 8 has the same (key & 3)
                    const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)LDS_PTR(p0));
                    const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)LDS_PTR(p1));
                    bf16x8 vf;
                    vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
                    vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[k2][s], oacc[dt], 0, 0, 0);
                }
            }
    }

    // ---- normalise and store: lane holds d = dt*32 + (r&3) + 8*(r>>2) + 4*hh of query l31
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qr = q0 + wave * 32 + l31;
    if (qr < len) {
        bf16* op = a.out + ((long long)(s0 + qr) * a.H + head) * HD + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(oacc[dt][g * 4 + e] * inv);
                *(bf16x4*)(op + dt * 32 + g * 8) = o;
            }
    }
}

}  // namespace

extern "C" int svr2_window_attn(const void* q, const void* k, const void* v, void* out, const int32_t* cu_seqlens,
                                const int32_t* tile_seq, const int32_t* tile_q0, int32_t ntiles, int32_t H, int32_t D,
                                float scale, void* stream) {
    SVR2_CHECK(q && k && v && out && cu_seqlens && tile_seq && tile_q0, SVR2_E_ARG, "svr2_window_attn: null pointer");
    SVR2_CHECK(D == HD, SVR2_E_SHAPE, "svr2_window_attn: head_dim must be 128 (got %d)", D);
    SVR2_CHECK(ntiles > 0 && H > 0 && H <= 65535, SVR2_E_SHAPE, "svr2_window_attn: bad ntiles/H");
    SVR2_CHECK(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out), SVR2_E_ALIGN,
               "svr2_window_attn: pointers must be 16-byte aligned");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)window_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS);
        attr_set = true;
    }
    AttnArgs a;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.out = (bf16*)out;
    a.cu = cu_seqlens; a.tile_seq = tile_seq; a.tile_q0 = tile_q0; a.H = H;
    a.scale_log2 = scale * 1.4426950408889634f;
    hipLaunchKernelGGL(window_attn_kernel, dim3(ntiles, H), dim3(NTH), ATT_LDS, (hipStream_t)stream, a);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}
