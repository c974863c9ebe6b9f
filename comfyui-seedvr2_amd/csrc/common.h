// Shared device/host helpers for libseedvr2_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/seedvr2_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 16-byte direct global -> LDS copy: LDS destination is wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

// Round to bf16 precision (nearest even), result kept in fp32.  Integer arithmetic on purpose: written as
// (float)(__bf16)v the round trip is folded away by the compiler when the value feeds an fma (seen in the ISA of
// tile_accumulate_kernel: mul, cvt, fmac -- the second rounding was gone), which silently changes results that must
// follow the reference's op-by-op bf16 rounding.
__device__ __forceinline__ float round_bf16f(float v) {
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return v;   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xffff0000u);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- host side error plumbing (thread-local message, see include/seedvr2_hip.h)
void svr2_set_error(const char* fmt, ...);
#define SVR2_CHECK(cond, code, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            svr2_set_error(__VA_ARGS__);     \
            return (code);                   \
        }                                    \
    } while (0)
#define SVR2_LAUNCH_CHECK()                                                        \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            svr2_set_error("launch failed: %s", hipGetErrorString(e_));            \
            return SVR2_E_LAUNCH;                                                  \
        }                                                                          \
    } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- process-wide tuning switches (A/B runs and tests only; see svr2_set_option in include/seedvr2_hip.h).
// Every switch is resolved ONCE: from svr2_set_option if it was called, else from the environment variable of the same
// name at first use, else the built-in default.  No getenv on the launch path.
namespace svr2 {
enum Opt { OPT_HALO_WR, OPT_GEMM_V, OPT_GEMM4, OPT_GEMM_BM, OPT_GEMM2_NARROW, OPT_CONV_LINEAR_TILES, OPT_ATTN_V, OPT_CONV_TFAST, OPT_CONV_SPLIT, OPT_ATTN5_V, OPT_HALO_PREF, OPT_EPI_PREF, OPT_COUNT };
int opt(Opt o, int dflt);
// hipFuncSetAttribute is a per-device setting, done once per (call site, device): attr_pending returns the device's bit
// while the attribute is not set yet (0 afterwards); attr_done publishes it once the call has succeeded
unsigned long long attr_pending(const unsigned long long* mask);
void attr_done(unsigned long long* mask, unsigned long long bit);
}  // namespace svr2
#define SVR2_SET_MAX_LDS(fn, bytes, what)                                                              \
    do {                                                                                                 \
        static unsigned long long mask_ = 0;                                                             \
        if (const unsigned long long bit_ = svr2::attr_pending(&mask_)) {                                \
            hipError_t e_ = hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); \
            SVR2_CHECK(e_ == hipSuccess, SVR2_E_LAUNCH, "%s: cannot reserve %d bytes of LDS", what, (int)(bytes));         \
            svr2::attr_done(&mask_, bit_);                                                               \
        }                                                                                                \
    } while (0)
