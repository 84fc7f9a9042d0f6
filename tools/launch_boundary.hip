// dev probe: what a DEPENDENT KERNEL BOUNDARY costs on one stream (MI355X_MICROARCH.md price list, row "boundary": 1.45 us between
// trivial 256-workgroup kernels, eager == hipGraph) — and which property of a launch moves it.  DESIGN.md section 5 priced a launch of
// the GPT-2 decode graph at ~4.2 us from rocprofv3's kernel trace; this probe measures the same thing WITHOUT a profiler attached:
// N launches back to back, hipEvent pair around all of them, (t / N) = boundary + body of a kernel that does (almost) nothing.
// Arms: grid / block size, kernarg bytes, opt-in dynamic LDS, bytes left dirty by the predecessor, a consumer that reads them,
// two alternating functions, and the same chains captured into a hipGraph.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

struct Big { unsigned long long v[60]; };     // 480 bytes of kernel arguments

__global__ void k_empty(unsigned* p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p == nullptr) p[0] = 1; }
__global__ void k_empty2(unsigned* p, int z) { if (threadIdx.x == 0 && blockIdx.x == 0 && z == 12345) p[0] = 1; }
__global__ void k_bigarg(unsigned* p, Big b) { if (threadIdx.x == 0 && blockIdx.x == 0 && b.v[59] == 77) p[0] = 1; }
__global__ void k_lds(unsigned* p) {
    extern __shared__ unsigned sm[];
    if (threadIdx.x == 0 && blockIdx.x == 0 && p == nullptr) p[0] = sm[5];
}
// every workgroup writes `bytes_per_wg` bytes (16 B per thread per round): the lines stay dirty in the XCD's L2 at the boundary
__global__ void k_write(uint4* buf, int rounds) {
    uint4 v = {blockIdx.x, threadIdx.x, 1u, 2u};
    for (int r = 0; r < rounds; ++r) buf[((size_t)blockIdx.x * rounds + r) * blockDim.x + threadIdx.x] = v;
}
// reads what the predecessor wrote (another workgroup's part) and writes its own part again: a dependent producer/consumer chain
__global__ void k_rw(uint4* buf, int rounds) {
    const unsigned other = (blockIdx.x * 37u + 11u) % gridDim.x;
    uint4 a = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        const uint4 t = buf[((size_t)other * rounds + r) * blockDim.x + threadIdx.x];
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    for (int r = 0; r < rounds; ++r) buf[((size_t)blockIdx.x * rounds + r) * blockDim.x + threadIdx.x] = a;
}

template <class F>
static void chain(const char* name, int n, F launch) {
    hipStream_t st; (void)hipStreamCreate(&st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) launch(st, i);
    (void)hipStreamSynchronize(st);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < n; ++i) launch(st, i);
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // the same chain as ONE captured graph of 200 nodes, replayed
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 200; ++i) launch(st, i);
    (void)hipStreamEndCapture(st, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
    float gbest = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < 10; ++i) (void)hipGraphLaunch(ge, st);
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < gbest) gbest = ms;
    }
    printf("%-58s eager %6.2f us / launch   graph %6.2f us / launch\n", name, best * 1e3 / n, gbest * 1e3 / 2000);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    (void)hipStreamDestroy(st);
}

int main() {
    unsigned* p; (void)hipMalloc(&p, 4096);
    uint4* buf; (void)hipMalloc(&buf, 256u << 20);
    (void)hipMemset(buf, 0, 256u << 20);
    (void)hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    Big b; for (int i = 0; i < 60; ++i) b.v[i] = i;
    const int N = 2000;
    chain("empty <<<256, 256>>>", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, p); });
    chain("empty <<<1, 64>>>", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p); });
    chain("empty <<<64, 256>>>", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s, p); });
    chain("empty <<<256, 768>>>", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(768), 0, s, p); });
    chain("empty <<<2048, 256>>>", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, s, p); });
    chain("two functions alternating <<<256, 256>>>", N, [&](hipStream_t s, int i) {
        if (i & 1) hipLaunchKernelGGL(k_empty2, dim3(256), dim3(256), 0, s, p, i); else hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, p); });
    chain("480-byte kernarg <<<256, 256>>>", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_bigarg, dim3(256), dim3(256), 0, s, p, b); });
    chain("96 KB dynamic LDS (opt-in) <<<256, 256>>>", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 96 * 1024, s, p); });
    chain("32 KB dynamic LDS <<<256, 256>>>", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 32 * 1024, s, p); });
    for (int rounds : {1, 4, 16, 64}) {
        char nm[96];
        snprintf(nm, sizeof nm, "writes %5.2f MB (dirty at the boundary) <<<256, 256>>>", 256.0 * rounds * 256 * 16 / 1e6);
        chain(nm, N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_write, dim3(256), dim3(256), 0, s, buf, rounds); });
        snprintf(nm, sizeof nm, "reads + writes %5.2f MB of the predecessor <<<256, 256>>>", 256.0 * rounds * 256 * 16 / 1e6);
        chain(nm, N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_rw, dim3(256), dim3(256), 0, s, buf, rounds); });
    }
    return 0;
}
