// lora_side.hip -- the rank-r "down" product of a LoRA adapter fused with the activation's other
// by-products, as one streaming pass.
//
// Reference: naive_gpt/layers/tuning/lora.py:70-80 (y += (x L) R^T) and the per-block forms of
// naive_gpt/layers/tuning/lora_ffn.py:87-111.  Every adapter multiplies a tall activation
// [rows, K] (rows = tokens, 8192-16384; K = 1024-4096) by a table of 16 columns.  As a library
// GEMM that is a 10-14 us pass over the 33-67 MB activation; the grouped GEMM's image path makes
// a second pass over the same activation (spt_split_bf16) and the GEMM in front of a ReLU a
// third (the rows' 2-norms).  All three are HBM-bound reads of the same bytes:
//
//   spt_lora_down   u[rows, n] = x[rows, K] . l[K, n]            n = 16, 32, 48 or 64
//                   and optionally the pre-split bf16 image of x (spt_split_bf16's layout) and
//                   the rows' 2-norms, from ONE read of x.  Several adapters on one input (the
//                   q / k / v projections) are one call with their tables side by side.
//
// Measured alone (rocprofv3, 16384 x 1024): u 13.8 us (the library GEMM: 13.8), u + image + norms
// 22.8 us against 20.1 (split) + 13.8 + a norm pass.  The transposed table-gradient products
// (x^T du) were written the same way and were NOT faster than the library's split-K form
// (13.5 + reduction against 12.0 + 5.2 us): they stay torch matmuls (layers/tuning/lora.py: tall_tn).
//
// Arithmetic as everywhere else in this library: fp32 operands split in two bf16 parts, three
// MFMAs per product (v_mfma_f32_16x16x32_bf16: its 16-wide output is the adapter's rank), fp32
// accumulation; <= 2^-16 relative error per product.
#include "spt_common.h"

namespace spt {

typedef __attribute__((ext_vector_type(8))) __bf16 ls_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 ls_bf16x2;
typedef __attribute__((ext_vector_type(2))) float ls_f32x2;
typedef __attribute__((ext_vector_type(4))) float ls_f32x4;

// two floats -> packed bf16 pairs (first value in the low half): hi = RNE, lo = RNE(x - hi)
__device__ __forceinline__ void ls_split2(float a, float b, unsigned &hi, unsigned &lo) {
    const ls_f32x2 x = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, ls_bf16x2));
    const ls_f32x2 hf = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(x - hf, ls_bf16x2));
}
struct LsFrag { uint4 hi, lo; };
__device__ __forceinline__ LsFrag ls_split8(const float (&v)[8]) {
    LsFrag f;
    ls_split2(v[0], v[1], f.hi.x, f.lo.x);
    ls_split2(v[2], v[3], f.hi.y, f.lo.y);
    ls_split2(v[4], v[5], f.hi.z, f.lo.z);
    ls_split2(v[6], v[7], f.hi.w, f.lo.w);
    return f;
}
__device__ __forceinline__ ls_f32x4 ls_mma(const uint4 &a, const uint4 &b, ls_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ls_bf16x8, a),
                                                   __builtin_bit_cast(ls_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ ls_f32x4 ls_mma3(const LsFrag &a, const LsFrag &b, ls_f32x4 c) {
    c = ls_mma(a.lo, b.hi, c);
    c = ls_mma(a.hi, b.lo, c);
    return ls_mma(a.hi, b.hi, c);
}

// ---- u = x . l -------------------------------------------------------------------------------
// A workgroup = 16 rows, its four waves = the four quarters of K; lane (r, g) = row r, elements
// 8g .. 8g + 7 of every 32-wide k-step: the A operand of the 16x16x32 MFMA as it lies in memory
// (32 bytes per lane, 128 contiguous bytes per row and k-step), and one 128-byte block of the
// split image.  All of a wave's loads of x for 256 k (16 x 16 bytes per lane) are issued before
// the first is used; with eight such waves per CU that is 128 KiB in flight per CU.  The table's
// fragments (lane (c, g): l[k + 8g + i][c]) come from L2 one k-step ahead.
constexpr int LS_CH = 8;            // k-steps (of 32) per chunk of loads
template <int NB, bool IMAGE, bool NORMS>
__global__ __launch_bounds__(256) void lora_down_kernel(
    const float *__restrict__ x, long long ldx, long long rows, int K, const float *__restrict__ l,
    int n, float *__restrict__ u, long long u_ld, long long u_block, char *__restrict__ image,
    float *__restrict__ norms) {
    __shared__ float red[4][NB][256];
    __shared__ float nred[4][16];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const long long row0 = (long long)blockIdx.x * 16;
    const long long row = min(row0 + r, rows - 1);          // (clamped rows are never stored)
    const int KQ = K >> 2, kbeg = w * KQ;
    const float *xp = x + row * ldx + kbeg + 8 * g;
    const float *lp = l + (size_t)(kbeg + 8 * g) * n + r;
    char *ip = IMAGE ? image + ((size_t)row * (K >> 5) + (kbeg >> 5)) * 128 + 16 * g : nullptr;
    ls_f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = {0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;
    auto load_l = [&](int kstep, float (&bv)[NB][8]) {
        const float *p = lp + (size_t)kstep * 32 * n;
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int i = 0; i < 8; i++) bv[b][i] = p[(size_t)i * n + 16 * b];
    };
    for (int s0 = 0; s0 < (KQ >> 5); s0 += LS_CH) {
        float4 a[LS_CH][2];
#pragma unroll
        for (int c = 0; c < LS_CH; c++) {
            a[c][0] = *reinterpret_cast<const float4 *>(xp + (s0 + c) * 32);
            a[c][1] = *reinterpret_cast<const float4 *>(xp + (s0 + c) * 32 + 4);
        }
        float bnext[NB][8];
        load_l(s0, bnext);
#pragma unroll
        for (int c = 0; c < LS_CH; c++) {
            float bv[NB][8];
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int i = 0; i < 8; i++) bv[b][i] = bnext[b][i];
            if (c + 1 < LS_CH) load_l(s0 + c + 1, bnext);
            const float av[8] = {a[c][0].x, a[c][0].y, a[c][0].z, a[c][0].w,
                                 a[c][1].x, a[c][1].y, a[c][1].z, a[c][1].w};
            const LsFrag af = ls_split8(av);
#pragma unroll
            for (int b = 0; b < NB; b++) acc[b] = ls_mma3(af, ls_split8(bv[b]), acc[b]);
            if (IMAGE && row0 + r < rows) {
                char *dst = ip + (size_t)(s0 + c) * 128;
                *reinterpret_cast<uint4 *>(dst) = af.hi;
                *reinterpret_cast<uint4 *>(dst + 64) = af.lo;
            }
            if (NORMS) {
#pragma unroll
                for (int i = 0; i < 8; i++) ss = fmaf(av[i], av[i], ss);
            }
        }
    }
    // the four k-quarters of the 16 x (16 NB) result through LDS; lane (c, g) holds rows 4g + j
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int j = 0; j < 4; j++) red[w][b][4 * lane + j] = acc[b][j];
    if (NORMS) {
        ss += __shfl_xor(ss, 16, SPT_WAVE);
        ss += __shfl_xor(ss, 32, SPT_WAVE);
        if (g == 0) nred[w][r] = ss;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NB * 256; idx += 256) {
        const int b = idx >> 8, e = idx & 255, ln = e >> 2, j = e & 3;
        const float v = (red[0][b][e] + red[1][b][e]) + (red[2][b][e] + red[3][b][e]);
        const long long orow = row0 + 4 * (ln >> 4) + j;
        if (orow < rows) u[b * u_block + orow * u_ld + (ln & 15)] = v;
    }
    if (NORMS && threadIdx.x < 16 && row0 + threadIdx.x < rows)
        norms[row0 + threadIdx.x] = sqrtf((nred[0][threadIdx.x] + nred[1][threadIdx.x]) +
                                          (nred[2][threadIdx.x] + nred[3][threadIdx.x]));
}

}  // namespace spt

using namespace spt;

static bool ls_aligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int spt_lora_down(const float *x, long long ldx, long long rows, int k, const float *l,
                             int n, float *u, int u_block_major, void *image, float *norms,
                             void *stream) {
    if (!x || !l || !u) return SPT_EINVAL;
    if (rows <= 0 || k <= 0 || n <= 0 || ldx < k) return SPT_EINVAL;
    if (k % (4 * 32 * LS_CH) != 0 || n % 16 != 0 || n > 64) return SPT_EUNSUP;
    if (ldx % 4 != 0 || !ls_aligned(x) || (image && !ls_aligned(image))) return SPT_ESHAPE;
    const long long nblk = (rows + 15) / 16;
    if (nblk > 0x7FFFFFFFll) return SPT_EUNSUP;
    hipStream_t s = (hipStream_t)stream;
    char *img = static_cast<char *>(image);
    // u [rows, n], or (u_block_major) n / 16 contiguous matrices [rows, 16]: one per adapter
    const long long u_ld = u_block_major ? 16 : n, u_block = u_block_major ? rows * 16 : 16;
#define SPT_LD(NB, IM, NO)                                                                      \
    hipLaunchKernelGGL((lora_down_kernel<NB, IM, NO>), dim3((unsigned)nblk), dim3(256), 0, s, x, \
                       ldx, rows, k, l, n, u, u_ld, u_block, img, norms)
#define SPT_LD_NB(NB)                                                   \
    do {                                                                \
        if (image && norms) SPT_LD(NB, true, true);                     \
        else if (image) SPT_LD(NB, true, false);                        \
        else if (norms) SPT_LD(NB, false, true);                        \
        else SPT_LD(NB, false, false);                                  \
    } while (0)
    switch (n / 16) {
        case 1: SPT_LD_NB(1); break;
        case 2: SPT_LD_NB(2); break;
        case 3: SPT_LD_NB(3); break;
        default: SPT_LD_NB(4); break;
    }
#undef SPT_LD_NB
#undef SPT_LD
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
