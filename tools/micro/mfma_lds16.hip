// mfma_lds.hip's k-loop once more with v_mfma_f32_16x16x32_bf16: per "k-step" (32 k) and wave
// 48 MFMAs on 16 accumulators of a 64 x 64 tile (chains of 3, as gg_mma3), fed by the same 16
// ds_read_b128 (row = lane & 15 of a 16-row group, chunk = lane >> 4: conflict-free under the GEMM's
// swizzle), same modes: reads (1), barrier (2), pipelined (4), LDS-DMA (8).  Same flops per step.
// Build: hipcc -O3 --offload-arch=gfx950.  Prints TFLOP/s per mode, to be read beside mfma_lds.hip's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int MODE>
__global__ __launch_bounds__(512, 2) void loop(const unsigned *seed, const char *gsrc, float *out, int iters) {
    __shared__ __attribute__((aligned(1024))) unsigned lds[2][12288];       // 2 x 48 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 2 * 12288; i += 512) (&lds[0][0])[i] = seed[i & 1023];
    __syncthreads();
    // a[g][part], b[g][part]: 16-row group g = 0 .. 3, part = hi / lo
    u32x4 fa[4][2], fb[4][2];
    for (int g = 0; g < 4; g++)
        for (int p = 0; p < 2; p++)
            for (int e = 0; e < 4; e++) {
                fa[g][p][e] = seed[(threadIdx.x * 37 + g * 8 + p * 4 + e) & 1023];
                fb[g][p][e] = seed[(threadIdx.x * 11 + g * 8 + p * 4 + e + 500) & 1023];
            }
    f32x4 acc[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) acc[i][j] = {0.f, 0.f, 0.f, 0.f};
    const char *base = reinterpret_cast<const char *>(&lds[0][0]);
    const int row = (wave >> 1) * 64 + (lane & 15), kq = lane >> 4;
    auto rd = [&](int st, int tile_row, int c) {
        const int r = tile_row + row;
        return *reinterpret_cast<const u32x4 *>(base + st * 49152 + (r & 255) * 128 + ((c ^ ((r >> 1) & 7)) << 4));
    };
    auto read_a = [&](int st, int g) { fa[g][0] = rd(st, 16 * g, kq); fa[g][1] = rd(st, 16 * g, 4 + kq); };
    auto read_b = [&](int st, int g) { fb[g][0] = rd(st, 256 + 16 * g, kq); fb[g][1] = rd(st, 256 + 16 * g, 4 + kq); };
    auto mmas = [&](int i) {            // row group i against the four column groups
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f32x4 c = acc[i][j];
            c = mma(fa[i][1], fb[j][0], c);
            c = mma(fa[i][0], fb[j][1], c);
            c = mma(fa[i][0], fb[j][0], c);
            acc[i][j] = c;
        }
    };
    const char *g = gsrc + ((size_t)blockIdx.x * 65536 + lane * 16);
    for (int it = 0; it < iters; it++) {
        const int st = it & 1;
        if (MODE & 8) {
#pragma unroll
            for (int j = 0; j < 6; j++)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(g + ((it * 6 + j) & 63) * 1024),
                    (__attribute__((address_space(3))) void *)(&lds[st ^ 1][0] + (wave + 8 * j) * 256), 16, 0, 0);
        }
        if ((MODE & 1) && !(MODE & 4)) {
            for (int q = 0; q < 4; q++) { read_a(st, q); read_b(st, q); }
            for (int i = 0; i < 4; i++) mmas(i);
        } else if (MODE & 1) {
            // first half: all of B and two row groups of A; second half: the other two row groups
            for (int q = 0; q < 4; q++) read_b(st, q);
            read_a(st, 0); read_a(st, 1);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            read_a(st, 2); read_a(st, 3);
            __builtin_amdgcn_sched_barrier(0);
            mmas(0); mmas(1);
            __builtin_amdgcn_sched_barrier(0);
            mmas(2); mmas(3);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            for (int i = 0; i < 4; i++) mmas(i);
        }
        if (MODE & 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            for (int r = 0; r < 4; r++) s += acc[i][j][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE>
static void run(const unsigned *seed, const char *gsrc, float *out) {
    const int iters = 2000, blocks = 256, reps = 5;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(loop<MODE>, dim3(blocks), dim3(512), 0, 0, seed, gsrc, out, iters);
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(loop<MODE>, dim3(blocks), dim3(512), 0, 0, seed, gsrc, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)reps * blocks * 8 * iters * 48 * 16384.0;
    printf("16x16x32 mode %2d [%s%s%s%s]: %7.1f TFLOP/s  %.0f cycles/step at 2.4 GHz\n", MODE, MODE & 1 ? "reads " : "", MODE & 2 ? "barrier " : "",
           MODE & 4 ? "pipelined " : "", MODE & 8 ? "dma " : "", flops / (ms * 1e-3) / 1e12, ms * 1e-3 / reps / iters * 2.4e9);
}
int main() {
    unsigned h[1024];
    for (int i = 0; i < 1024; i++) {
        unsigned lo = 0x3F00u | (rand() & 0xFF) | ((rand() & 1) << 15), hi = 0x3F00u | (rand() & 0xFF) | ((rand() & 1) << 15);
        h[i] = hi << 16 | lo;
    }
    unsigned *seed; float *out; char *gsrc;
    hipMalloc(&seed, sizeof(h)); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&gsrc, 256 * 65536 + 4096);
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    for (int i = 0; i < 256 * 64 + 4; i++) hipMemcpy(gsrc + (size_t)i * 1024, h, 1024, hipMemcpyHostToDevice);
    run<0>(seed, gsrc, out); run<2>(seed, gsrc, out); run<1>(seed, gsrc, out); run<3>(seed, gsrc, out);
    run<5>(seed, gsrc, out); run<7>(seed, gsrc, out); run<11>(seed, gsrc, out); run<15>(seed, gsrc, out);
    return 0;
}
