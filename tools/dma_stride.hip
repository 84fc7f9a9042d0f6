// dev probe: LDS-DMA throughput per CU when a stage is 256 rows x 256 B of a row-major matrix (the GEMM's operand pattern: 16-row x 64-B
// pieces) as a function of the row stride — L2 channel conflicts.  One 512-thread workgroup per CU, every CU its own 256-row band,
// 24 stages of 256 B along the row, repeated.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 1) void probe(const char* src, long long stride, int stages, int rounds, int share, unsigned long long* ticks, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int band = share ? (blockIdx.x >> 3) % 4 + 4 * (blockIdx.x & 7) : blockIdx.x;       // share: the 4-32 CUs of an XCD read 4 bands
    const char* base = src + (long long)band * 256 * stride;
    const long long lane_off = (long long)(wave * 16 + (lane >> 2)) * stride + (lane & 3) * 16;
    f4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        const int s = r % stages;
#pragma unroll
        for (int k = 0; k < 8; ++k)       // piece k: rows (k >> 2) * 128 + wave * 16 .., 64-B column block k & 3 of the stage's 256 B
            __builtin_amdgcn_global_load_lds((gbl_void*)(base + lane_off + (long long)(k >> 2) * 128 * stride + s * 256 + (k & 3) * 64),
                                             (lds_void*)(smem + (r & 1) * 65536 + k * 8192 + wave * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += *(const f4*)(smem + (r & 1) * 65536 + ((threadIdx.x * 16 + r * 64) & 65535 & ~15));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) sink[0] = 1.f;
}

int main() {
    const size_t cap = 600u << 20;
    char* src; if (hipMalloc(&src, cap) != hipSuccess) return 1;
    (void)hipMemset(src, 1, cap);
    unsigned long long* dt; float* sink;
    (void)hipMalloc(&dt, 8); (void)hipMalloc(&sink, 4);
    (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int rounds = 24 * 40;
    for (int share = 0; share < 2; ++share)
        for (long long stride : {6144LL, 6144LL + 64, 6144LL + 128, 6144LL + 256, 6144LL + 512, 1536LL, 1536LL + 64, 4096LL, 8192LL, 8192LL + 128}) {
            const int grid = 256;
            if ((long long)grid * 256 * stride + 8192 > (long long)cap) continue;
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            probe<<<grid, 512, 131072>>>(src, stride, 24, 24, share, dt, sink);
            (void)hipEventRecord(e0);
            probe<<<grid, 512, 131072>>>(src, stride, 24, rounds, share, dt, sink);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t; (void)hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
            const double bytes = 65536.0 * rounds;
            printf("share %d stride %5lld: %6.1f cycles per KB per CU, %5.1f B/clk/CU, %8.1f GB/s whole launch\n", share, stride, t / (bytes / 1024), bytes / t, bytes * grid / ms / 1e6);
        }
    return 0;
}
