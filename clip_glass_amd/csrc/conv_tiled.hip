// conv_tiled.hip — LDS-tiled MFMA implicit-GEMM convolution / GEMM (fast path).
#include "common.h"
#include "kernels.h"

bool launch_conv_tiled(const ConvParams& p, hipStream_t st) { (void)p; (void)st; return false; }
bool launch_gemm_tiled(const GemmParams& p, hipStream_t st) { (void)p; (void)st; return false; }
