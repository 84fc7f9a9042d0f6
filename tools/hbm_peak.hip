// hbm_peak.hip — what this box's HBM sustains for the access mixes of the fitness pass: read-only, write-only,
// copy (1R:1W, the high-resolution conv layers), 2R:1W.  Standalone:
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_peak.hip -o tools/_bin/hbm_peak && tools/_bin/hbm_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 read, 1 write, 2 copy, 3 two reads + one write
__global__ __launch_bounds__(256) void stream(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, long long n,
                                              float* sink) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (MODE == 0) acc += a[i];
        if (MODE == 1) c[i] = f4{1.f, 2.f, 3.f, (float)i};
        if (MODE == 2) c[i] = a[i];
        if (MODE == 3) c[i] = a[i] + b[i];
    }
    if (MODE == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <int MODE>
void run(const char* name, int bytes_per_elem, int grid) {
    const long long n = (1LL << 30) / 16;   // 1 GiB per array
    f4 *a, *b, *c;
    float* sink;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16); hipMalloc(&sink, 4);
    hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    stream<MODE><<<grid, 256>>>(a, b, c, n, sink);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        stream<MODE><<<grid, 256>>>(a, b, c, n, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-22s grid=%6d  %7.3f ms  %7.1f GB/s\n", name, grid, best, (double)n * bytes_per_elem / best * 1e-6);
    hipFree(a); hipFree(b); hipFree(c); hipFree(sink);
}

int main() {
    for (int grid : {2048, 8192, 65536}) {
        run<0>("read", 16, grid);
        run<1>("write", 16, grid);
        run<2>("copy (1R:1W)", 32, grid);
        run<3>("add (2R:1W)", 48, grid);
    }
    return 0;
}
