// dev probe: cost of a grid-wide barrier between 256 co-resident 512-thread workgroups (one per CU), two forms:
//   flat: every workgroup adds to ONE device-scope counter and spins on it;
//   hier: workgroups of an XCD (blockIdx % 8) meet on a per-XCD counter, the last arrival of each XCD adds to the global one, all spin on the global one.
// Each barrier includes the release / acquire fences a data-carrying barrier needs.  Spins are bounded (no hang: prints "TIMEOUT").
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned ld_acq(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(unsigned* ctr, int iters, unsigned long long* ticks, int* err) {
    const unsigned nwg = gridDim.x;
    unsigned long long t0 = 0;
    for (int it = 0; it < iters + 10; ++it) {
        if (it == 10) t0 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();                                   // release: this workgroup's writes are visible device-wide
            const unsigned target = (unsigned)(it + 1);
            if (MODE == 0) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (ld_acq(ctr) < target * nwg) { if (++spins > (1 << 22)) { *err = 1; break; } __builtin_amdgcn_s_sleep(1); }
            } else {
                const unsigned x = blockIdx.x & 7, per = (nwg + 7 - x) / 8;       // workgroups with this blockIdx % 8
                const unsigned old = __hip_atomic_fetch_add(ctr + 16 + x * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == target * per) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (ld_acq(ctr) < target * 8) { if (++spins > (1 << 22)) { *err = 1; break; } __builtin_amdgcn_s_sleep(1); }
            }
            __threadfence();                                   // acquire
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = __builtin_amdgcn_s_memtime() - t0;
}

template <int MODE>
void run(const char* name, int grid) {
    unsigned* ctr; unsigned long long* dt; int* err;
    (void)hipMalloc(&ctr, 4096); (void)hipMemset(ctr, 0, 4096);
    (void)hipMalloc(&dt, 8); (void)hipMalloc(&err, 4); (void)hipMemset(err, 0, 4);
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    probe<MODE><<<grid, 512>>>(ctr, iters, dt, err);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; int he; (void)hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost);
    printf("%-6s grid %3d: %.2f us per barrier (%.0f cycles)%s\n", name, grid, ms * 1e3 / (iters + 10), (double)t / iters, he ? "  TIMEOUT" : "");
}
int main() {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    for (int grid : {8, 64, cus}) { run<0>("flat", grid); run<1>("hier", grid); }
    return 0;
}
