// Probe: how fast can a CU move global memory into LDS?  (a) LDS-DMA (global_load_lds_dwordx4), (b) global_load_dwordx4
// into VGPRs followed by ds_write_b128.  Every workgroup streams 64-KiB "tiles" (64 rows of 1 KiB) from a buffer of `span`
// bytes (small span = L2 / Infinity-Cache resident, like GEMM operands re-read by neighbouring tiles) into a 2 x 64 KiB LDS
// ring.  Prints GB/s per CU-clock terms:  hipcc --offload-arch=gfx950 -O3 tools/probes/stage_rate.hip -o stage_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int NW, bool DMA>
__global__ __launch_bounds__(NW * 64) void stage_kernel(const char* src, long long span, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PPW = 64 / NW;                       // 1-KiB pieces per wave per tile
    const long long tiles = span / 65536;
    long long tile = (blockIdx.x * 7919ll) % tiles;
    unsigned acc = 0;
    u32x4 r[DMA ? 1 : PPW];
    for (int it = 0; it < iters; ++it) {
        const char* base = src + tile * 65536 + (wave * PPW) * 1024 + lane * 16;
        char* dst = smem + (it & 1) * 65536 + (wave * PPW) * 1024;
        if (DMA) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) glds16(base + i * 1024, dst + i * 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int i = 0; i < PPW; ++i) r[i] = *(const u32x4*)(base + i * 1024);
#pragma unroll
            for (int i = 0; i < PPW; ++i) *(u32x4*)(dst + i * 1024 + lane * 16) = r[i];
        }
        __syncthreads();
        acc += *(const unsigned*)(smem + (it & 1) * 65536 + ((threadIdx.x * 52) & 65532));
        tile += 1;
        if (tile >= tiles) tile -= tiles;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int NW, bool DMA>
void run(const char* name, const char* buf, long long span, int iters) {
    unsigned* sink;
    hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)stage_kernel<NW, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stage_kernel<NW, DMA>), dim3(256), dim3(NW * 64), 131072, 0, buf, span, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * iters * 65536;
    printf("%-34s span %6lld MiB: %7.2f TB/s total, %6.1f GB/s per CU, %5.1f cycles per KiB per CU at 2.4 GHz\n", name,
           span >> 20, bytes / ms / 1e9, bytes / ms / 1e6 / 256, 2.4e9 * (ms * 1e-3) / (iters * 64.0));
    hipFree(sink);
}

int main() {
    const long long big = 2048ll << 20;
    char* buf;
    hipMalloc(&buf, big);
    hipMemset(buf, 1, big);
    for (long long span : {1ll << 20, 2ll << 20, 4ll << 20, 16ll << 20}) {
        run<8, true>("LDS-DMA, 8 waves", buf, span, 2000);
        run<8, false>("VGPR + ds_write_b128, 8 waves", buf, span, 2000);
        run<4, true>("LDS-DMA, 4 waves", buf, span, 2000);
        run<4, false>("VGPR + ds_write_b128, 4 waves", buf, span, 2000);
    }
    return 0;
}
