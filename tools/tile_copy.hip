// tile_copy.hip — how fast the memory system serves the conv kernels' ACCESS PATTERN with no compute: every block
// reads a (TH+2) x (TW+2) pixel patch (64-byte pixels, halo included, like conv_tiled's staging) of a 16 x 1024 x 1024 x 32ch
// fp16 map and writes the TH x TW tile of a second map.  Compared with tools/hbm_peak's flat copy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int TH, int TW, int HALO>
__global__ __launch_bounds__(256) void tile_copy(const _Float16* __restrict__ x, _Float16* __restrict__ y, int H, int W) {
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO, NV = PH * PW * 4, NA = (NV + 255) / 256;
    __shared__ h8 lds[PH * PW * 4];
    const int tiles_x = W / TW, tiles_y = H / TH;
    int id = blockIdx.x;
    const int tx = id % tiles_x; id /= tiles_x;
    const int ty = id % tiles_y; const int b = id / tiles_y;
    const _Float16* xb = x + (long long)b * H * W * 32;
    h8 r[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int v = threadIdx.x + 256 * k, pix = v >> 2, pr = pix / PW, pc = pix % PW;
        const int iy = ty * TH - HALO + pr, ix = tx * TW - HALO + pc;
        for (int j = 0; j < 8; ++j) r[k][j] = 0;
        if (v < NV && iy >= 0 && iy < H && ix >= 0 && ix < W) r[k] = *(const h8*)(xb + ((long long)iy * W + ix) * 32 + (v & 3) * 8);
    }
#pragma unroll
    for (int k = 0; k < NA; ++k) if (threadIdx.x + 256 * k < NV) lds[threadIdx.x + 256 * k] = r[k];
    __syncthreads();
    _Float16* yb = y + (long long)b * H * W * 32;
    for (int v = threadIdx.x; v < TH * TW * 4; v += 256) {
        const int pix = v >> 2, pr = pix / TW, pc = pix % TW;
        yb_store:
        *(h8*)(yb + ((long long)(ty * TH + pr) * W + tx * TW + pc) * 32 + (v & 3) * 8) = lds[((pr + HALO) * PW + pc + HALO) * 4 + (v & 3)];
    }
}

template <int TH, int TW, int HALO>
void run(const char* name) {
    const int B = 16, H = 1024, W = 1024;
    const size_t bytes = (size_t)B * H * W * 64;
    _Float16 *x, *y;
    (void)hipMalloc(&x, bytes); (void)hipMalloc(&y, bytes);
    (void)hipMemset(x, 0, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = B * (H / TH) * (W / TW);
    tile_copy<TH, TW, HALO><<<grid, 256>>>(x, y, H, W);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        tile_copy<TH, TW, HALO><<<grid, 256>>>(x, y, H, W);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-34s grid=%6d  %7.3f ms  %7.1f GB/s (in + out maps)\n", name, grid, best, 2.0 * bytes / best * 1e-6);
    (void)hipFree(x); (void)hipFree(y);
}

int main() {
    run<8, 32, 1>("tile 8x32 + halo (conv_tiled)");
    run<8, 32, 0>("tile 8x32, no halo");
    run<4, 64, 1>("tile 4x64 + halo");
    run<2, 128, 1>("tile 2x128 + halo");
    run<16, 32, 1>("tile 16x32 + halo");
    run<8, 64, 1>("tile 8x64 + halo");
    return 0;
}
