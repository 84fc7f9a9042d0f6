// inflight_probe.hip — HBM read throughput as a function of (waves per CU, 16-byte loads in flight per thread).
// Each wave streams its own contiguous range of a 4 GiB buffer: K loads are issued back to back, then consumed (xor-summed),
// repeatedly.  Answers: is memory-level parallelism limited per wave (deep register prefetch does not help beyond some depth)
// or per CU (fewer, fatter waves are as good as many thin ones)?
//   hipcc --offload-arch=gfx950 -O3 tools/inflight_probe.hip -o tools/_bin/inflight_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int K>
__global__ __launch_bounds__(256) void probe(const u4* __restrict__ x, unsigned* out, long long n_vec, int pad_lds) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const long long wid = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long per = n_vec / n_waves / (64 * K) * (64 * K);          // vectors per wave, whole K-batches
    const u4* p = x + wid * per + lane;
    u4 acc = {0, 0, 0, 0};
    for (long long i = 0; i < per; i += 64 * K) {
        u4 r[K];
#pragma unroll
        for (int k = 0; k < K; ++k) r[k] = p[i + 64 * k];
#pragma unroll
        for (int k = 0; k < K; ++k) acc ^= r[k];
    }
    if (pad_lds < 0) smem[0] = 1;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

template <int K>
void run(const u4* x, unsigned* out, long long n_vec, int blocks_per_cu, int lds_bytes) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    (void)hipFuncSetAttribute((const void*)probe<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    probe<K><<<grid, 256, lds_bytes>>>(x, out, n_vec, 0);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        probe<K><<<grid, 256, lds_bytes>>>(x, out, n_vec, 0);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const long long n_waves = (long long)grid * 4, per = n_vec / n_waves / (64 * K) * (64 * K);
    const double bytes = (double)per * n_waves * 16;
    printf("waves/CU %2d  loads in flight/thread %2d  (%3d KB/CU nominal)  %7.3f ms  %7.1f GB/s\n", blocks_per_cu * 4, K,
           blocks_per_cu * 4 * K, best, bytes / best * 1e-6);
}

int main() {
    const long long bytes = 4LL << 30, n_vec = bytes / 16;
    u4* x; unsigned* out;
    (void)hipMalloc(&x, bytes); (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    (void)hipMemset(x, 1, bytes);
    // occupancy is pinned by the dynamic LDS request: 160 KB / blocks_per_cu
    const int bpc[4] = {1, 2, 4, 8};
    for (int i = 0; i < 4; ++i) {
        const int lds = 160 * 1024 / bpc[i] - 1024;
        run<2>(x, out, n_vec, bpc[i], lds);
        run<4>(x, out, n_vec, bpc[i], lds);
        run<8>(x, out, n_vec, bpc[i], lds);
        run<16>(x, out, n_vec, bpc[i], lds);
        run<32>(x, out, n_vec, bpc[i], lds);
        run<48>(x, out, n_vec, bpc[i], lds);
    }
    return 0;
}
