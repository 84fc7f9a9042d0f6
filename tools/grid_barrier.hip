// dev probe: cost of a grid-wide barrier between co-resident workgroups (one per CU), four forms:
//   flat-acq : ONE device-scope counter, polled with ACQUIRE loads, __threadfence() on both sides   (round 3's probe)
//   hier-acq : per-(blockIdx % 8) counters in front of the global one, same fences and polling       (round 3's probe)
//   counter  : MI355X_MICROARCH.md row barrier-counter — one monotonic counter, lane-0 release fence before the arrive,
//              RELAXED sc1 polling + s_sleep, ONE acquire fence after the match
//   xcd      : row barrier-xcd — arrivals on a per-XCC counter (the REAL XCC id, s_getreg; counts from a census round, so the
//              protocol does not depend on the block -> XCD map); the last arriver of an XCD is its leader: ONE release fence
//              (it writes back the whole XCD's L2, covering every workgroup of that XCD whose stores have reached L2) ->
//              top counter -> relaxed poll -> acquire -> per-XCC generation word; every other workgroup polls its XCC's
//              generation word relaxed, then one acquire fence
// Each barrier carries the fences a data-carrying barrier needs.  `payload`: every workgroup also writes 128 B before the barrier
// and reads another workgroup's record after it (checked).  Spins are bounded (no hang: prints "TIMEOUT").
#include <hip/hip_runtime.h>
#include <stdio.h>

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ unsigned ld_acq(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_rlx(const unsigned* p) { return __hip_atomic_load(p, RLX_AGENT); }
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}
// state words (each on its own 128-byte line): [0] top counter, [32 + 32 x] XCC x arrivals, [512 + 32 x] XCC x generation,
// [1024 + x] census of XCC x, [1100] number of XCCs present
template <int MODE, int NT>
__global__ __launch_bounds__(NT, 1) void probe(unsigned* st, int iters, unsigned long long* ticks, int* err, unsigned* rec, int payload) {
    const unsigned nwg = gridDim.x;
    unsigned x = 0, n_x = 0, n_present = 0;
    if (MODE == 3) {
        // census round (flat barrier): how many workgroups sit on each XCC
        x = xcc_id();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(st + 1024 + x, 1u, RLX_AGENT);
            __hip_atomic_fetch_add(st + 1200, 1u, RLX_AGENT);
            int spins = 0;
            while (ld_rlx(st + 1200) < nwg) { if (++spins > (1 << 22)) { *err = 1; break; } __builtin_amdgcn_s_sleep(2); }
        }
        __syncthreads();
        n_x = ld_rlx(st + 1024 + x);
        for (int i = 0; i < 16; ++i) n_present += ld_rlx(st + 1024 + i) ? 1 : 0;
    }
    unsigned long long t0 = 0;
    int bad = 0;
    for (int it = 0; it < iters + 10; ++it) {
        if (it == 10) t0 = __builtin_amdgcn_s_memtime();
        const unsigned e = (unsigned)(it + 1);
        unsigned* recb = rec + (size_t)(e & 1) * nwg * 32;      // records double-buffered by epoch parity: epoch e + 2's write needs barrier e + 1,
                                                                // which every reader of epoch e has already reached
        if (payload && threadIdx.x < 32) recb[(size_t)blockIdx.x * 32 + threadIdx.x] = e * 1000u + blockIdx.x;
        __syncthreads();
        if (threadIdx.x == 0) {
            if (MODE <= 1) {
                __threadfence();                                   // release + acquire, both sides (round 3)
                if (MODE == 0) {
                    __hip_atomic_fetch_add(st, 1u, RLX_AGENT);
                    int spins = 0;
                    while (ld_acq(st) < e * nwg) { if (++spins > (1 << 22)) { *err = 1; break; } __builtin_amdgcn_s_sleep(1); }
                } else {
                    const unsigned xb = blockIdx.x & 7, per = (nwg + 7 - xb) / 8;
                    const unsigned old = __hip_atomic_fetch_add(st + 32 + xb * 32, 1u, RLX_AGENT);
                    if (old + 1 == e * per) __hip_atomic_fetch_add(st, 1u, RLX_AGENT);
                    int spins = 0;
                    while (ld_acq(st) < e * (nwg < 8 ? nwg : 8)) { if (++spins > (1 << 22)) { *err = 1; break; } __builtin_amdgcn_s_sleep(1); }
                }
                __threadfence();
            } else if (MODE == 2) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(st, 1u, RLX_AGENT);
                int spins = 0;
                while (ld_rlx(st) < e * nwg) { if (++spins > (1 << 22)) { *err = 1; break; } __builtin_amdgcn_s_sleep(2); }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this workgroup's stores have reached its XCD's L2
                const unsigned old = __hip_atomic_fetch_add(st + 32 + x * 32, 1u, RLX_AGENT);
                if (old + 1 == e * n_x) {                              // last arriver of this XCD: its leader for this barrier
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // buffer_wbl2 sc1: the XCD's dirty lines, every workgroup's
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_fetch_add(st, 1u, RLX_AGENT);
                    int spins = 0;
                    while (ld_rlx(st) < e * n_present) { if (++spins > (1 << 22)) { *err = 1; break; } __builtin_amdgcn_s_sleep(1); }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    __hip_atomic_store(st + 512 + x * 32, e, RLX_AGENT);
                } else {
                    int spins = 0;
                    while (ld_rlx(st + 512 + x * 32) < e) { if (++spins > (1 << 22)) { *err = 1; break; } __builtin_amdgcn_s_sleep(1); }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
            }
        }
        __syncthreads();
        if (payload && threadIdx.x < 32) {
            const unsigned other = (blockIdx.x * 37u + 11u + (unsigned)it) % nwg;
            if (recb[(size_t)other * 32 + threadIdx.x] != e * 1000u + other) bad = 1;
        }
    }
    if (bad) atomicAdd(err + 1, 1);
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = __builtin_amdgcn_s_memtime() - t0;
}

template <int MODE, int NT>
void run(const char* name, int grid, int payload) {
    unsigned *st, *rec; unsigned long long* dt; int* err;
    (void)hipMalloc(&st, 8192); (void)hipMemset(st, 0, 8192);
    (void)hipMalloc(&rec, (size_t)grid * 256); (void)hipMemset(rec, 0, (size_t)grid * 256);
    (void)hipMalloc(&dt, 8); (void)hipMalloc(&err, 8); (void)hipMemset(err, 0, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, NT>), dim3(grid), dim3(NT), 0, 0, st, iters, dt, err, rec, payload);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; int he[2];
    (void)hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(he, err, 8, hipMemcpyDeviceToHost);
    printf("%-9s %3d-thread WGs, grid %3d%s: %6.2f us per barrier (%6.0f clocks)%s%s\n", name, NT, grid, payload ? ", 128-B record per WG" : "",
           ms * 1e3 / (iters + 10), (double)t / iters, he[0] ? "  TIMEOUT" : "", he[1] ? "  STALE RECORDS" : "");
    (void)hipFree(st); (void)hipFree(rec); (void)hipFree(dt); (void)hipFree(err);
}
int main() {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    for (int grid : {8, 64, cus}) {
        run<0, 512>("flat-acq", grid, 0); run<1, 512>("hier-acq", grid, 0); run<2, 512>("counter", grid, 0); run<3, 512>("xcd", grid, 0);
    }
    run<2, 256>("counter", cus, 0); run<3, 256>("xcd", cus, 0);
    run<2, 512>("counter", cus, 1); run<3, 512>("xcd", cus, 1);
    run<3, 768>("xcd", cus, 1);
    return 0;
}
