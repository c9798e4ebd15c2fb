// Which ingredient of the grouped GEMM's k-loop costs the matrix pipe its time?  8 waves per CU
// (one 512-thread workgroup), per "k-step" and wave: 24 MFMAs 32x32x16 bf16 on 4 accumulators
// (chains of 3 on one accumulator, as gg_mma3), optionally fed by 16 ds_read_b128 from an LDS
// tile (mode bit 0), with a workgroup barrier per step (bit 1), with the reads pipelined half a
// step ahead (bit 2), with 6 LDS-DMA instructions per wave and step from a 16 MB buffer (bit 3).
// Prints TFLOP/s per mode.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ f32x16 mma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int MODE>
__global__ __launch_bounds__(512, 2) void loop(const unsigned *seed, const char *gsrc, float *out, int iters) {
    __shared__ __attribute__((aligned(1024))) unsigned lds[2][12288];       // 2 x 48 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 2 * 12288; i += 512) (&lds[0][0])[i] = seed[i & 1023];
    __syncthreads();
    u32x4 fa[2][4], fb[2][4];
    for (int h = 0; h < 2; h++)
        for (int i = 0; i < 4; i++)
            for (int e = 0; e < 4; e++) {
                fa[h][i][e] = seed[(threadIdx.x * 37 + i * 4 + e + 64 * h) & 1023];
                fb[h][i][e] = seed[(threadIdx.x * 11 + i * 4 + e + 500 + 64 * h) & 1023];
            }
    f32x16 acc[4];
    for (int i = 0; i < 4; i++)
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    const char *base = reinterpret_cast<const char *>(&lds[0][0]);
    // conflict-free: row = lane & 31 of a [rows][128 B] tile, chunk swizzled as in the GEMM
    const int row = (wave >> 1) * 64 + (lane & 31), fh = lane >> 5;
    auto rd = [&](int st, int tile_row, int c) {
        const int r = tile_row + row;
        return *reinterpret_cast<const u32x4 *>(base + st * 49152 + (r & 255) * 128 + ((c ^ ((r >> 1) & 7)) << 4));
    };
    auto reads = [&](int st, int q2, u32x4 (&a)[4], u32x4 (&b)[4]) {
        a[0] = rd(st, 0, 2 * q2 + fh); a[1] = rd(st, 0, 4 + 2 * q2 + fh);
        a[2] = rd(st, 32, 2 * q2 + fh); a[3] = rd(st, 32, 4 + 2 * q2 + fh);
        b[0] = rd(st, 256, 2 * q2 + fh); b[1] = rd(st, 256, 4 + 2 * q2 + fh);
        b[2] = rd(st, 288, 2 * q2 + fh); b[3] = rd(st, 288, 4 + 2 * q2 + fh);
    };
    auto mmas = [&](u32x4 (&a)[4], u32x4 (&b)[4]) {
        // (a[0], a[1]) = hi, lo of row block 0; (a[2], a[3]) of row block 1; b likewise
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                f32x16 c = acc[2 * i + j];
                c = mma(a[2 * i + 1], b[2 * j], c);
                c = mma(a[2 * i], b[2 * j + 1], c);
                c = mma(a[2 * i], b[2 * j], c);
                acc[2 * i + j] = c;
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
            reads(st, 0, fa[0], fb[0]);
            mmas(fa[0], fb[0]);
            reads(st, 1, fa[1], fb[1]);
            mmas(fa[1], fb[1]);
        } else if (MODE & 1) {
            reads(st, 0, fa[0], fb[0]);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            reads(st, 1, fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
            mmas(fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            mmas(fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            mmas(fa[0], fb[0]);
            mmas(fa[1], fb[1]);
        }
        if (MODE & 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++)
        for (int r = 0; r < 16; r++) s += acc[i][r];
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
    const double flops = (double)reps * blocks * 8 * iters * 24 * 32768.0;
    printf("mode %2d [%s%s%s%s]: %7.1f TFLOP/s  %.0f cycles/step at 2.4 GHz\n", MODE, MODE & 1 ? "reads " : "", MODE & 2 ? "barrier " : "",
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
