// spt_common.h -- shared device helpers for libspt_hip.so (gfx950 / CDNA4 only).
//
// wave = 64 lanes everywhere in this library; a "group" is a power-of-two run of
// consecutive lanes inside one wave that co-operates on one CSR entry / one row.
#ifndef SPT_COMMON_H
#define SPT_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/spt_hip.h"

#define SPT_WAVE 64

#define SPT_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

#define SPT_HIP_TRY(expr)                          \
    do {                                           \
        hipError_t e__ = (expr);                   \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

// element type of the dense attention operands (mfma_attention.hip)
enum { SPT_F32 = 0, SPT_BF16 = 1 };

namespace spt {

// Zero `n_words` 32-bit words, as a KERNEL.  Not hipMemsetAsync: in a captured HIP graph a memset
// node did not keep its place between the kernel nodes around it when the replay started on an idle
// GPU (round 3: the near-the-kink queue's counters held the bytes of whatever had owned the block
// before, relu_fix_kernel walked a queue of garbage entries -> `Memory access fault`); a kernel node
// does.  One launch of a few waves instead of a blit: the same cost.
static __global__ __launch_bounds__(256) void zero_words_kernel(unsigned *p, int n_words) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_words; i += gridDim.x * 256) p[i] = 0u;
}
#define SPT_ZERO_WORDS(ptr, n_words, stream)                                                    \
    hipLaunchKernelGGL(spt::zero_words_kernel, dim3(((n_words) + 255) / 256 > 64 ? 64 : ((n_words) + 255) / 256), \
                       dim3(256), 0, (stream), reinterpret_cast<unsigned *>(ptr), (int)(n_words))

__device__ __forceinline__ int lane_id() { return threadIdx.x & (SPT_WAVE - 1); }

// ---- DPP helpers: lane movement inside a row of 16 lanes, no LDS traffic ----
// row_ror:n rotates the 16-lane row right by n lanes.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

// One butterfly step of a sum all-reduce.  Steps must be applied in the order
// 1,2,4,8,16,32: the 4- and 8-lane steps use DPP mirrors, which equal an xor
// exchange only when the value is already uniform over the smaller sub-group.
// Every lane of a group ends with the bitwise-identical sum.
template <int STEP>
__device__ __forceinline__ float butterfly_partner(float v) {
    if constexpr (STEP == 1) {
        return dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    } else if constexpr (STEP == 2) {
        return dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    } else if constexpr (STEP == 4) {
        return dpp_mov<0x141>(v);  // row_half_mirror: lane i <-> 7-i
    } else if constexpr (STEP == 8) {
        return dpp_mov<0x140>(v);  // row_mirror: lane i <-> 15-i
    } else if constexpr (STEP == 16) {
        // ds_swizzle bit mode (and=0x1f, or=0, xor=16): crossbar only, no LDS memory
        return __builtin_bit_cast(
            float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (16 << 10) | 0x1F));
    } else {
        return __shfl_xor(v, 32, SPT_WAVE);
    }
}

// all-reduce (sum) over groups of G consecutive lanes, G a power of two <= 64.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (G >= 2) v += butterfly_partner<1>(v);
    if constexpr (G >= 4) v += butterfly_partner<2>(v);
    if constexpr (G >= 8) v += butterfly_partner<4>(v);
    if constexpr (G >= 16) v += butterfly_partner<8>(v);
    if constexpr (G >= 32) v += butterfly_partner<16>(v);
    if constexpr (G >= 64) v += butterfly_partner<32>(v);
    return v;
}

// Blocks that work on the same batch should share an XCD (its 4 MiB L2 then holds
// that batch's K / V / CSR slice).  Workgroups are dealt round-robin over the 8
// XCDs, so ids with equal (id % 8) share one: give each residue class a contiguous
// run of logical ids.  Bijective for any grid size (cdna guide 5, T1).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + j;
}

// Copy n4 float4 from global memory into an LDS tile.  Four independent loads per
// thread are issued before the first ds_write so their HBM latencies overlap.
__device__ __forceinline__ void stage_tile(float *__restrict__ lds_dst,
                                           const float *__restrict__ src, int n4, int tid,
                                           int nthreads) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(lds_dst);
    int i = tid;
    for (; i + 3 * nthreads < n4; i += 4 * nthreads) {
        const float4 a = s4[i], b = s4[i + nthreads], c = s4[i + 2 * nthreads],
                     d = s4[i + 3 * nthreads];
        d4[i] = a;
        d4[i + nthreads] = b;
        d4[i + 2 * nthreads] = c;
        d4[i + 3 * nthreads] = d;
    }
    for (; i < n4; i += nthreads) d4[i] = s4[i];
}

// Dense per-batch operands are either [B, S, E] (heads == 0) or live inside an
// [N, S, heads, E] tensor with batch b = n * heads + h (the attention layers' layout):
// row r of batch b starts at  base + r * ld.
struct DenseView {
    size_t base;
    int ld;
};
__device__ __forceinline__ DenseView dense_view(int b, int S, int E, int heads) {
    DenseView v;
    if (heads > 0) {
        const int n = b / heads, h = b - n * heads;
        v.base = ((size_t)n * S * heads + h) * E;
        v.ld = heads * E;
    } else {
        v.base = (size_t)b * S * E;
        v.ld = E;
    }
    return v;
}

// Copy S rows of E floats (row stride ld) into a packed [S][E] LDS tile; four independent
// 16-byte loads per thread in flight.
__device__ __forceinline__ void stage_rows(float *__restrict__ lds_dst,
                                           const float *__restrict__ src, int ld, int S, int E,
                                           int tid, int nthreads) {
    const int e4 = E >> 2;
    const int n4 = S * e4;
    float4 *d4 = reinterpret_cast<float4 *>(lds_dst);
    auto ld4 = [&](int i) {
        const int r = i / e4, c = i - r * e4;
        return *reinterpret_cast<const float4 *>(src + (size_t)r * ld + 4 * c);
    };
    int i = tid;
    for (; i + 3 * nthreads < n4; i += 4 * nthreads) {
        const float4 a = ld4(i), b = ld4(i + nthreads), c = ld4(i + 2 * nthreads),
                     d = ld4(i + 3 * nthreads);
        d4[i] = a;
        d4[i + nthreads] = b;
        d4[i + 2 * nthreads] = c;
        d4[i + 3 * nthreads] = d;
    }
    for (; i < n4; i += nthreads) d4[i] = ld4(i);
}

__host__ __device__ __forceinline__ int pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace spt

#endif  // SPT_COMMON_H
