// Baseline JPEG encoder on the device for Motion-JPEG output ("device -> encoder writer"; see
// comfyui-seedvr2_amd/jpeg.py for the format notes and oracle/seedvr2_ref/jpeg.py for the arithmetic this file must
// reproduce bit for bit).  Three integer / byte kernels over the finished bf16 frames [T,H,W,3] in [0,1]:
//   jpeg_coef_kernel    : one thread per 8x8 block: trunc(x*255) -> JFIF YCbCr (16-bit fixed point; chroma from the rounded
//                         2x2 RGB mean) -> level shift -> LL&M integer forward DCT (13-bit constants, output x8) -> rounding
//                         division by 8*Q -> zig-zag, int16 [T][rows][cols][6][64]   (HBM-bound: 6 B in, 3 B out per pixel)
//   jpeg_entropy_kernel : one thread per MCU row (= one restart interval, so rows are independent): DC differences,
//                         run/size Huffman codes (Annex K tables), 0xFF byte stuffing, 1-padding to the byte boundary
//   jpeg_pack_kernel    : one workgroup per frame: scan of the row lengths, rows + RSTn markers copied into one stream
#include "common.h"

namespace {

constexpr int C_BITS = 13, P1_BITS = 2;
constexpr int F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633;
constexpr int F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// one 1-D LL&M pass over d[0..7*S] (stride S); FIRST: even outputs shifted up by P1_BITS, odd ones descaled by
// C_BITS - P1_BITS; second pass: P1_BITS / C_BITS + P1_BITS
template <int S, bool FIRST>
__device__ __forceinline__ void dct_pass(int* d) {
    const int t0 = d[0] + d[7 * S], t7 = d[0] - d[7 * S], t1 = d[S] + d[6 * S], t6 = d[S] - d[6 * S];
    const int t2 = d[2 * S] + d[5 * S], t5 = d[2 * S] - d[5 * S], t3 = d[3 * S] + d[4 * S], t4 = d[3 * S] - d[4 * S];
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    constexpr int NO = FIRST ? C_BITS - P1_BITS : C_BITS + P1_BITS;
    d[0] = FIRST ? (t10 + t11) << P1_BITS : descale(t10 + t11, P1_BITS);
    d[4 * S] = FIRST ? (t10 - t11) << P1_BITS : descale(t10 - t11, P1_BITS);
    int z1 = (t12 + t13) * F_0_541;
    d[2 * S] = descale(z1 + t13 * F_0_765, NO);
    d[6 * S] = descale(z1 - t12 * F_1_847, NO);
    z1 = t4 + t7;
    int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const int z5 = (z3 + z4) * F_1_175;
    const int a4 = t4 * F_0_298, a5 = t5 * F_2_053, a6 = t6 * F_3_072, a7 = t7 * F_1_501;
    z1 = -z1 * F_0_899; z2 = -z2 * F_2_562; z3 = -z3 * F_1_961 + z5; z4 = -z4 * F_0_390 + z5;
    d[7 * S] = descale(a4 + z1 + z3, NO);
    d[5 * S] = descale(a5 + z2 + z4, NO);
    d[3 * S] = descale(a6 + z2 + z3, NO);
    d[S] = descale(a7 + z1 + z4, NO);
}

__constant__ unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                          41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                          30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

__device__ __forceinline__ int u8_of(bf16 v) {   // the quantisation of svr2_frames_to_u8
    return (int)fminf(fmaxf(bf2f(v) * 255.0f, 0.f), 255.f);
}

__global__ __launch_bounds__(128) void jpeg_coef_kernel(const bf16* __restrict__ src, short* __restrict__ coef, int T, int H, int W,
                                                        int R, int Cn, const int* __restrict__ div) {
    const long long total = (long long)T * R * Cn * 6;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % 6);
    long long m = i / 6;
    const int c = (int)(m % Cn);
    m /= Cn;
    const int r = (int)(m % R), t = (int)(m / R);
    const bf16* f = src + (long long)t * H * W * 3;
    int d[64];
    if (k < 4) {
        const int y0 = r * 16 + (k >> 1) * 8, x0 = c * 16 + (k & 1) * 8;
#pragma unroll
        for (int yy = 0; yy < 8; ++yy) {
            const int y = min(y0 + yy, H - 1);
#pragma unroll
            for (int xx = 0; xx < 8; ++xx) {
                const bf16* p = f + ((long long)y * W + min(x0 + xx, W - 1)) * 3;
                d[yy * 8 + xx] = ((19595 * u8_of(p[0]) + 38470 * u8_of(p[1]) + 7471 * u8_of(p[2]) + 32768) >> 16) - 128;
            }
        }
    } else {
#pragma unroll
        for (int yy = 0; yy < 8; ++yy)
#pragma unroll
            for (int xx = 0; xx < 8; ++xx) {
                int sr = 0, sg = 0, sb = 0;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const bf16* p = f + ((long long)min(r * 16 + yy * 2 + dy, H - 1) * W + min(c * 16 + xx * 2 + dx, W - 1)) * 3;
                        sr += u8_of(p[0]); sg += u8_of(p[1]); sb += u8_of(p[2]);
                    }
                sr = (sr + 2) >> 2; sg = (sg + 2) >> 2; sb = (sb + 2) >> 2;
                const int v = k == 4 ? (-11059 * sr - 21709 * sg + 32768 * sb + (128 << 16) + 32767) >> 16
                                     : (32768 * sr - 27439 * sg - 5329 * sb + (128 << 16) + 32767) >> 16;
                d[yy * 8 + xx] = v - 128;
            }
    }
#pragma unroll
    for (int row = 0; row < 8; ++row) dct_pass<1, true>(d + row * 8);
#pragma unroll
    for (int col = 0; col < 8; ++col) dct_pass<8, false>(d + col);
    const int* dv = div + (k < 4 ? 0 : 64);
    short* o = coef + i * 64;
#pragma unroll
    for (int z = 0; z < 64; ++z) {
        const int v = d[kZigzag[z]], q = dv[z];
        const int mag = (abs(v) + (q >> 1)) / q;
        o[z] = (short)(v < 0 ? -mag : mag);
    }
}

struct BitWriter {
    unsigned char* p;
    unsigned char* end;
    unsigned long long acc;
    int n;
    bool overflow;
    __device__ __forceinline__ void put(unsigned code, int len) {
        acc = (acc << len) | (code & ((1u << len) - 1u));
        n += len;
        while (n >= 8) {
            const unsigned char b = (unsigned char)(acc >> (n - 8));
            n -= 8;
            if (p + 2 > end) { overflow = true; continue; }
            *p++ = b;
            if (b == 0xFF) *p++ = 0;
        }
    }
};

__device__ __forceinline__ int bit_length(int v) { return 32 - __clz(v); }   // v >= 0

__global__ __launch_bounds__(64) void jpeg_entropy_kernel(const short* __restrict__ coef, unsigned char* __restrict__ slots,
                                                          int* __restrict__ row_len, int rows_total, int Cn, long long slot_bytes,
                                                          const unsigned* __restrict__ huff) {
    // one row per wave, coded by lane 0: the loop is data-dependent branching, 64 rows in one wave would serialise
    const int row = blockIdx.x;
    if (row >= rows_total || threadIdx.x != 0) return;
    BitWriter bw;
    bw.p = slots + (long long)row * slot_bytes;
    bw.end = bw.p + slot_bytes;
    bw.acc = 0; bw.n = 0; bw.overflow = false;
    unsigned char* const start = bw.p;
    int last[3] = {0, 0, 0};
    const short* cp = coef + (long long)row * Cn * 6 * 64;
    for (int m = 0; m < Cn; ++m)
        for (int k = 0; k < 6; ++k, cp += 64) {
            const int comp = k < 4 ? 0 : k - 3;
            const unsigned* dc = huff + (k < 4 ? 0 : 268);
            const unsigned* ac = dc + 12;
            const int diff = cp[0] - last[comp];
            last[comp] = cp[0];
            int s = bit_length(abs(diff));
            bw.put(dc[s] & 0xffffu, dc[s] >> 16);
            if (s) bw.put((unsigned)(diff > 0 ? diff : diff - 1), s);
            int run = 0;
            for (int z = 1; z < 64; ++z) {
                const int v = cp[z];
                if (v == 0) { ++run; continue; }
                while (run >= 16) { bw.put(ac[0xF0] & 0xffffu, ac[0xF0] >> 16); run -= 16; }
                s = bit_length(abs(v));
                const unsigned h = ac[(run << 4) | s];
                bw.put(h & 0xffffu, h >> 16);
                bw.put((unsigned)(v > 0 ? v : v - 1), s);
                run = 0;
            }
            if (run) bw.put(ac[0] & 0xffffu, ac[0] >> 16);
        }
    if (bw.n) bw.put((1u << (8 - bw.n)) - 1u, 8 - bw.n);       // pad the last byte of the restart interval with ones
    row_len[row] = bw.overflow ? -1 : (int)(bw.p - start);
}

__global__ __launch_bounds__(256) void jpeg_pack_kernel(const unsigned char* __restrict__ slots, const int* __restrict__ row_len,
                                                        unsigned char* __restrict__ packed, int* __restrict__ frame_len, int R,
                                                        long long slot_bytes, long long frame_cap) {
    extern __shared__ long long off[];          // R + 1 offsets
    const int t = blockIdx.x;
    const int* len = row_len + (long long)t * R;
    if (threadIdx.x == 0) {
        long long o = 0;
        bool bad = false;
        for (int r = 0; r < R; ++r) {
            off[r] = o;
            bad |= len[r] < 0;
            o += (len[r] < 0 ? 0 : len[r]) + (r + 1 < R ? 2 : 0);
        }
        off[R] = o;
        frame_len[t] = (bad || o > frame_cap) ? -1 : (int)o;
        if (bad || o > frame_cap) off[R] = -1;
    }
    __syncthreads();
    if (off[R] < 0) return;
    unsigned char* dst = packed + (long long)t * frame_cap;
    for (int r = 0; r < R; ++r) {
        const unsigned char* s = slots + ((long long)t * R + r) * slot_bytes;
        unsigned char* d = dst + off[r];
        for (int j = threadIdx.x; j < len[r]; j += blockDim.x) d[j] = s[j];
        if (r + 1 < R && threadIdx.x == 0) {
            d[len[r]] = 0xFF;
            d[len[r] + 1] = (unsigned char)(0xD0 + (r & 7));
        }
    }
}

}  // namespace

extern "C" int64_t svr2_jpeg_ws_bytes(int32_t T, int32_t H, int32_t W) {
    if (T <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t R = (H + 15) / 16, Cn = (W + 15) / 16;
    // coefficients (int16) + one byte slot per MCU row (8 bytes per coefficient bounds any baseline code) + row lengths
    return T * R * Cn * 384 * 2 + T * R * (Cn * 384 * 8) + T * R * 4 + 256;
}

extern "C" int svr2_jpeg_encode(const void* frames, int32_t T, int32_t H, int32_t W, const int32_t* div, const uint32_t* huff,
                                void* packed, int64_t frame_cap, int32_t* frame_len, void* workspace, int64_t ws_bytes,
                                void* stream) {
    SVR2_CHECK(frames && div && huff && packed && frame_len && workspace, SVR2_E_ARG, "svr2_jpeg_encode: null pointer");
    SVR2_CHECK(T > 0 && H > 0 && W > 0 && H <= 65535 && W <= 65535 && frame_cap > 0, SVR2_E_SHAPE, "svr2_jpeg_encode: bad sizes");
    SVR2_CHECK(ws_bytes >= svr2_jpeg_ws_bytes(T, H, W), SVR2_E_SHAPE, "svr2_jpeg_encode: workspace too small");
    const int R = (H + 15) / 16, Cn = (W + 15) / 16;
    SVR2_CHECK(R <= 4096, SVR2_E_SHAPE, "svr2_jpeg_encode: too many MCU rows");
    const long long slot_bytes = (long long)Cn * 384 * 8;
    short* coef = (short*)workspace;
    unsigned char* slots = (unsigned char*)workspace + (long long)T * R * Cn * 384 * 2;
    int* row_len = (int*)(((uintptr_t)(slots + (long long)T * R * slot_bytes) + 15) & ~(uintptr_t)15);
    hipStream_t s = (hipStream_t)stream;
    const long long nblk = (long long)T * R * Cn * 6;
    hipLaunchKernelGGL(jpeg_coef_kernel, dim3((unsigned)((nblk + 127) / 128)), dim3(128), 0, s, (const bf16*)frames, coef, T, H, W,
                       R, Cn, div);
    hipLaunchKernelGGL(jpeg_entropy_kernel, dim3(T * R), dim3(64), 0, s, coef, slots, row_len, T * R, Cn, slot_bytes,
                       huff);
    hipLaunchKernelGGL(jpeg_pack_kernel, dim3(T), dim3(256), (R + 1) * sizeof(long long), s, slots, row_len,
                       (unsigned char*)packed, frame_len, R, slot_bytes, (long long)frame_cap);
    SVR2_LAUNCH_CHECK();
    return SVR2_OK;
}
