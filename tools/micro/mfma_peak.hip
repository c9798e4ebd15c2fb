// What the bf16 matrix cores deliver on this chip with NOTHING else in the loop: 8 waves per
// CU (two per SIMD), four independent accumulators per wave, operands in registers (random
// bits, or zeros with argv[1] = 0).  Prints TFLOP/s.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ __launch_bounds__(512, 2) void peak(const unsigned *seed, float *out, int iters) {
    u32x4 a[4], b[4];
    for (int i = 0; i < 4; i++)
        for (int e = 0; e < 4; e++) {
            a[i][e] = seed[(threadIdx.x * 37 + i * 4 + e) & 1023];
            b[i][e] = seed[(threadIdx.x * 11 + i * 4 + e + 500) & 1023];
        }
    f32x16 acc[4];
    for (int i = 0; i < 4; i++)
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 6; rep++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8, a[(i + rep) & 3]), __builtin_bit_cast(bf16x8, b[i]), acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++)
        for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main(int argc, char **argv) {
    const bool random = argc < 2 || atoi(argv[1]) != 0;
    unsigned h[1024];
    for (int i = 0; i < 1024; i++) {
        // bf16 pairs with exponents near 1.0 so that nothing overflows: 0x3F80 +- mantissa bits
        unsigned lo = 0x3F00u | (rand() & 0xFF) | ((rand() & 1) << 15);
        unsigned hi = 0x3F00u | (rand() & 0xFF) | ((rand() & 1) << 15);
        h[i] = random ? (hi << 16 | lo) : 0u;
    }
    unsigned *seed; float *out;
    hipMalloc(&seed, sizeof(h)); hipMalloc(&out, 256 * 512 * 4 * 4);
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int warm = 0; warm < 3; warm++) hipLaunchKernelGGL(peak, dim3(blocks), dim3(512), 0, 0, seed, out, iters);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(peak, dim3(blocks), dim3(512), 0, 0, seed, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)reps * blocks * 8 * iters * 24 * 32768.0;
    printf("%s operands: %.1f TFLOP/s (%.3f ms per launch, %d MFMAs per wave)\n", random ? "random" : "zero",
           flops / (ms * 1e-3) / 1e12, ms / reps, iters * 24);
    return 0;
}
