// dev probe: per-CU throughput of the two global -> LDS paths on an L2-resident source, one 512-thread workgroup per CU:
//   mode 0: LDS-DMA (global_load_lds_dwordx4, 1 KB per wave-instruction), 8 pieces per wave per round, vmcnt(0) + barrier per round
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128, 8 x 16 B per thread per round
//   mode 2: both at once, 4 + 4
// prints bytes per clock per CU (s_memtime ticks of workgroup 0) and GB/s of the whole launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int NP>
__global__ __launch_bounds__(512, 1) void probe(const char* src, size_t span, int rounds, unsigned long long* ticks, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* base = src + ((size_t)blockIdx.x * 65536) % span;       // every CU its own 64 KB window (re-read each round: L2 hits)
    f4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        const char* rb = base + (r & 1) * 0;
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int k = 0; k < (MODE == 2 ? NP / 2 : NP); ++k)
                __builtin_amdgcn_global_load_lds((gbl_void*)(rb + k * 8192 + threadIdx.x * 16), (lds_void*)(smem + (r & 1) * 65536 + k * 8192 + wave * 1024), 16, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
            f4 v[NP];
#pragma unroll
            for (int k = (MODE == 2 ? NP / 2 : 0); k < NP; ++k) v[k] = *(const f4*)(rb + k * 8192 + threadIdx.x * 16);
#pragma unroll
            for (int k = (MODE == 2 ? NP / 2 : 0); k < NP; ++k) *(f4*)(smem + (r & 1) * 65536 + k * 8192 + threadIdx.x * 16) = v[k];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += *(const f4*)(smem + (r & 1) * 65536 + ((threadIdx.x * 16 + r * 64) & 65535 & ~15));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) sink[0] = 1.f;
}

template <int MODE, int NP>
void run(const char* name, const char* src, size_t span, int grid) {
    unsigned long long* dt; float* sink;
    hipMalloc(&dt, 8); hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)probe<MODE, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int rounds = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, NP><<<grid, 512, 131072>>>(src, span, 10, dt, sink);
    hipEventRecord(e0);
    probe<MODE, NP><<<grid, 512, 131072>>>(src, span, rounds, dt, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
    const double bytes = (double)NP * 8192 * rounds;
    printf("%-34s grid %3d: %6.1f cycles per KB per CU, %5.1f B/clk/CU (s_memtime), %7.1f GB/s whole launch, %.3f ms\n", name, grid, t / (bytes / 1024), bytes / t,
           bytes * grid / ms / 1e6, ms);
}

int main() {
    const size_t span = 64u << 20;
    char* src; hipMalloc(&src, span + 65536); hipMemset(src, 1, span + 65536);
    for (int grid : {1, 32, 256}) {
        run<0, 8>("LDS-DMA 8 pieces/wave/round", src, span, grid);
        run<1, 8>("load + ds_write_b128 8/thread", src, span, grid);
        run<2, 8>("4 DMA + 4 load/ds_write", src, span, grid);
        run<0, 4>("LDS-DMA 4 pieces/wave/round", src, span, grid);
        run<1, 4>("load + ds_write_b128 4/thread", src, span, grid);
    }
    return 0;
}
