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
// (x^T du) are spt_tall_tn further down: plain fp32 FMAs, as fast as the library's split-K form
// (11.8 + 3.9 us against 11.5 + 5.1) at a quarter of its host time.
//
// Arithmetic, two forms (`exact` of the C ABI):
//   exact = 0  as everywhere else in this library: fp32 operands split in two bf16 parts, three
//              MFMAs per product (v_mfma_f32_16x16x32_bf16: its 16-wide output is the adapter's
//              rank), fp32 accumulation; <= 2^-16 relative error per product.
//   exact = 1  EXACT fp32 on the matrix cores (v_mfma_f32_16x16x4_f32: fp32 in, fp32 FMA chain).
//              For the one u that is an INPUT of the GEMM in front of the routed FFN's ReLU: an
//              error of 2^-16 |x| |l| there moves a pre-activation by ~1e-4 |r1 row|, enough to
//              flip the sign of the few elements per million within that distance of zero -- each
//              flip an O(1) change of its token's gradient, which no later fp32 recomputation can
//              see (round 3: 18 of 8192 tokens at BERT-large dimensions against the fp64 layer;
//              the fp32 per-block loop has none).  The fp32 MFMA runs at the vector rate (1/16 of
//              bf16): 9.0 us for 8192 x 1024 x 16, under the pass's memory time; at 48-64 columns
//              it would be the bound (21-42 us), which is why it is an option.
// The split image (hi / lo bf16 of x) is written by the same pass in both forms.
#include "spt_common.h"
#include <type_traits>

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
// exact fp32: lane (r, g) holds a[r][8 g + i] and b[8 g + i][c = r], i = 0 .. 7; MFMA i contracts
// the four k's {i, 8 + i, 16 + i, 24 + i} of the 32-wide k-step (one per lane group g)
__device__ __forceinline__ ls_f32x4 ls_mma_f32x8(const float (&a)[8], const float (&b)[8], ls_f32x4 c) {
#pragma unroll
    for (int i = 0; i < 8; i++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], c, 0, 0, 0);
    return c;
}

// ---- u = x . l -------------------------------------------------------------------------------
// A workgroup = 16 rows, its LS_NW waves = contiguous shares of K (K % 32 == 0); lane (r, g) = row r, elements
// 8g .. 8g + 7 of every 32-wide k-step (32 bytes per lane, 128 contiguous bytes per row and
// k-step): eight A operands of the 16x16x4 fp32 MFMA, and one 128-byte block of the split image.  All of a wave's loads of x for 256 k (16 x 16 bytes per lane) are issued before
// the first is used; with eight such waves per CU that is 128 KiB in flight per CU.  The table's
// fragments (lane (c, g): l[k + 8g + i][c]) come from L2 one k-step ahead.
constexpr int LS_CH = 8;            // k-steps (of 32) per chunk of loads
constexpr int LS_NW = 8;            // waves per workgroup = shares of K (8192 rows are only 512
                                    // workgroups: with four waves each a SIMD held two, and the fp32
                                    // MFMA chain of one was not covered by the loads of the other)
// Grouped form (offsets != null): rows offsets[q] .. offsets[q + 1] - 1 use the table l + q *
// l_gstride (the routed FFN's per-block tables; rows are sorted by block).  A workgroup's 16 rows
// then lie inside ONE group -- the MFMA's B operand is common to the tile -- so the tiles are
// counted per group (at most one short tile each: rows / 16 + n_groups workgroups, the surplus
// ones find no group and leave).
// Second table (l2 != null): the LAST 16-column block of the result is x . l2^T for a row-major
// l2 [n2 <= 16, K] -- an nn.Linear weight as it lies in memory (the routed FFN's router,
// sparse/feedforward.py:22-25, riding the pass that forms x . L1); columns n2 .. 15 of that block are 0.
// (L2 is a template parameter: as a run-time branch in the table loads it took the single-table
// kernel from 98 to 230 registers and from 11 to 20 us)
template <int NB, bool IMAGE, bool NORMS, bool EXACT, bool L2 = false>
__global__ __launch_bounds__(64 * LS_NW) void lora_down_kernel(
    const float *__restrict__ x, long long ldx, long long rows, int K, const float *__restrict__ l,
    int n, float *__restrict__ u, long long u_ld, long long u_block, char *__restrict__ image,
    float *__restrict__ norms, const int32_t *__restrict__ offsets, int n_groups, long long l_gstride,
    const float *__restrict__ l2, int n2, long long l_bstride, int group_cols) {
    // (group_cols, grouped form: row r of group q is stored at u[(r - offsets[q]) * u_ld + q * n + c]
    // -- equal groups side by side as column blocks of ONE matrix: dU_q | dU_k | dU_v of a joint
    // projection's backward from one launch over [dQ; dK; dV])
    // (l_bstride: floats from one 16-column block of the table to the next -- 16 inside one [K, n]
    // table; the distance between SEPARATE [K, 16] tables, e.g. the adapters of q, k, v as they lie
    // in a tuner's flat parameter buffer: no concatenated copy of them is made)
    __shared__ float red[LS_NW][NB][256];
    __shared__ float nred[LS_NW][16];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    long long row0 = (long long)blockIdx.x * 16;
    if (offsets) {                                           // tile -> (group, first row, end)
        int base = 0, grp = -1;
        long long first = 0, end = 0;
        for (int q = 0; q < n_groups; q++) {
            const int lo = offsets[q], hi = offsets[q + 1];
            const int nt = (hi - lo + 15) >> 4;
            if (grp < 0 && (int)blockIdx.x < base + nt) {
                grp = q; first = lo + 16ll * ((int)blockIdx.x - base); end = hi;
            }
            base += nt;
        }
        if (grp < 0) return;
        row0 = first; rows = end;                            // (rows: one past the tile's last usable row)
        l += grp * l_gstride;
        if (group_cols) u += (long long)grp * n - (long long)offsets[grp] * u_ld;
    }
    const long long row = min(row0 + r, rows - 1);          // (clamped rows are never stored)
    // the K / 32 k-steps dealt to the four waves as evenly as they go (K = 2752, the LLaMA-7B FFN
    // block: 22, 22, 21, 21); a wave walks its share in chunks of LS_CH steps, the last one ragged
    const int nsteps = K >> 5, per = nsteps / LS_NW, extra = nsteps % LS_NW;
    const int sbeg = w * per + min(w, extra), cnt = per + (w < extra ? 1 : 0);
    const float *xp = x + row * ldx + 32 * sbeg + 8 * g;
    const float *lp = l + (size_t)(32 * sbeg + 8 * g) * n + r;
    char *ip = IMAGE ? image + ((size_t)row * nsteps + sbeg) * 128 + 16 * g : nullptr;
    ls_f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = {0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;
    // (second table: lane (c = r, g) reads l2[c][k + 8 g + 0 .. 7], 32 contiguous bytes)
    const float *l2p = L2 ? l2 + (size_t)min(r, n2 - 1) * K + 32 * sbeg + 8 * g : nullptr;
    auto load_l = [&](int kstep, float (&bv)[NB][8]) {
        const float *p = lp + (size_t)kstep * 32 * n;
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if (L2 && b == NB - 1) {
                const float4 t0 = *reinterpret_cast<const float4 *>(l2p + (size_t)kstep * 32);
                const float4 t1 = *reinterpret_cast<const float4 *>(l2p + (size_t)kstep * 32 + 4);
                const bool keep = r < n2;
                bv[b][0] = keep ? t0.x : 0.f; bv[b][1] = keep ? t0.y : 0.f; bv[b][2] = keep ? t0.z : 0.f;
                bv[b][3] = keep ? t0.w : 0.f; bv[b][4] = keep ? t1.x : 0.f; bv[b][5] = keep ? t1.y : 0.f;
                bv[b][6] = keep ? t1.z : 0.f; bv[b][7] = keep ? t1.w : 0.f;
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) bv[b][i] = p[(size_t)i * n + b * l_bstride];
            }
        }
    };
    // one chunk; RAGGED: steps s0 + c >= cnt load a clamped (valid) address and are skipped
    // (CH: K = 1024 gives a wave 4 k-steps -- as one ragged chunk of 8 that was 8 clamped duplicates among
    // its 16 load instructions)
    auto chunk = [&](int s0, auto ragged, auto steps) {
        constexpr bool RAGGED = decltype(ragged)::value;
        constexpr int CH = decltype(steps)::value;
        float4 a[CH][2];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int st = RAGGED ? min(s0 + c, cnt - 1) : s0 + c;
            a[c][0] = *reinterpret_cast<const float4 *>(xp + st * 32);
            a[c][1] = *reinterpret_cast<const float4 *>(xp + st * 32 + 4);
        }
        float bnext[NB][8];
        load_l(s0, bnext);
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (RAGGED && s0 + c >= cnt) break;              // (wave-uniform)
            float bv[NB][8];
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int i = 0; i < 8; i++) bv[b][i] = bnext[b][i];
            if (c + 1 < CH) load_l(RAGGED ? min(s0 + c + 1, cnt - 1) : s0 + c + 1, bnext);
            const float av[8] = {a[c][0].x, a[c][0].y, a[c][0].z, a[c][0].w,
                                 a[c][1].x, a[c][1].y, a[c][1].z, a[c][1].w};
            LsFrag af;
            if (IMAGE || !EXACT) af = ls_split8(av);
#pragma unroll
            for (int b = 0; b < NB; b++)
                acc[b] = EXACT ? ls_mma_f32x8(av, bv[b], acc[b]) : ls_mma3(af, ls_split8(bv[b]), acc[b]);
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
    };
    int s0 = 0;
    for (; s0 + LS_CH <= cnt; s0 += LS_CH) chunk(s0, std::false_type{}, std::integral_constant<int, LS_CH>{});
    if constexpr (IMAGE) {
        // (measured, 8192 / 16384 rows x 1024, three tables + image: 22.8 -> 18.6 and 45.6 -> 43.4 us warm,
        // 31.5 -> 28.3 and 58 -> 52.5 cold; the plain kernel LOST with the second instantiation
        // -- 98 -> 128 registers, 8.4 -> 12.3 us -- and keeps the one ragged chunk)
        for (; s0 + LS_CH / 2 <= cnt; s0 += LS_CH / 2)
            chunk(s0, std::false_type{}, std::integral_constant<int, LS_CH / 2>{});
        if (s0 < cnt) chunk(s0, std::true_type{}, std::integral_constant<int, LS_CH / 2>{});
    } else {
        // (half chunks alone, one instantiation: 9.6 / 19.0 us warm against 8.8 / 14.4 -- not this either)
        if (s0 < cnt) chunk(s0, std::true_type{}, std::integral_constant<int, LS_CH>{});
    }
    // the waves' shares of the 16 x (16 NB) result through LDS; lane (c, g) holds rows 4g + j
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
    for (int idx = threadIdx.x; idx < NB * 256; idx += 64 * LS_NW) {
        const int b = idx >> 8, e = idx & 255, ln = e >> 2, j = e & 3;
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < LS_NW; q += 2) v += red[q][b][e] + red[q + 1][b][e];
        const long long orow = row0 + 4 * (ln >> 4) + j;
        if (orow < rows) u[b * u_block + orow * u_ld + (ln & 15)] = v;
    }
    if (NORMS && threadIdx.x < 16 && row0 + threadIdx.x < rows)
        norms[row0 + threadIdx.x] = sqrtf(((nred[0][threadIdx.x] + nred[1][threadIdx.x]) +
                                           (nred[2][threadIdx.x] + nred[3][threadIdx.x])) +
                                          ((nred[4][threadIdx.x] + nred[5][threadIdx.x]) +
                                           (nred[6][threadIdx.x] + nred[7][threadIdx.x])));
}

}  // namespace spt

using namespace spt;

static bool ls_aligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int lora_down_any(const float *x, long long ldx, long long rows, int k, const float *l,
                         int n, float *u, long long ldu, int u_block_major, void *image,
                         float *norms, int exact, const int32_t *offsets, int n_groups,
                         long long l_gstride, void *stream, const float *l2 = nullptr, int n2 = 0,
                         long long table_stride = 0, int group_cols = 0) {
    if (!x || !l || !u) return SPT_EINVAL;
    if (rows <= 0 || k <= 0 || n <= 0 || ldx < k) return SPT_EINVAL;
    if (l2 && (n2 <= 0 || n2 > 16 || offsets || (reinterpret_cast<uintptr_t>(l2) & 15) != 0)) return SPT_EINVAL;
    // (with a second table `n` counts its block of 16 too; the first table is [k, n - 16])
    // (table_stride != 0: n / 16 separate tables [k, 16], table_stride floats apart)
    if (table_stride != 0 && (l2 || offsets || table_stride % 4 != 0)) return SPT_EINVAL;
    const int n_l = table_stride ? 16 : (l2 ? n - 16 : n);
    const long long l_bstride = table_stride ? table_stride : 16;
    if (k % 32 != 0 || n % 16 != 0 || n > 64 || n_l <= 0) return SPT_EUNSUP;
    if (ldx % 4 != 0 || !ls_aligned(x) || (image && !ls_aligned(image))) return SPT_ESHAPE;
    if (offsets && (n_groups <= 0 || n_groups > 64)) return SPT_EINVAL;
    const long long nblk = (rows + 15) / 16 + (offsets ? n_groups : 0);
    if (nblk > 0x7FFFFFFFll) return SPT_EUNSUP;
    hipStream_t s = (hipStream_t)stream;
    char *img = static_cast<char *>(image);
    // u [rows, n], or (u_block_major) n / 16 contiguous matrices [rows, 16]: one per adapter
    if (ldu != 0 && (ldu < n || u_block_major)) return SPT_EINVAL;
    const long long u_ld = u_block_major ? 16 : (ldu ? ldu : n), u_block = u_block_major ? rows * 16 : 16;
#define SPT_LD5(NB, IM, NO, EX, L2)                                                              \
    hipLaunchKernelGGL((lora_down_kernel<NB, IM, NO, EX, L2>), dim3((unsigned)nblk), dim3(64 * LS_NW), 0, s, \
                       x, ldx, rows, k, l, n_l, u, u_ld, u_block, img, norms, offsets, n_groups, l_gstride, \
                       l2, n2, l_bstride, group_cols)
    // (a second table only in the exact form and behind at least one block of the first table: the
    // routed FFN's router riding x . L1)
    if (l2 && (!exact || n < 32)) return SPT_EUNSUP;
#define SPT_LD(NB, IM, NO)                                              \
    do {                                                                \
        if (l2) { if constexpr (NB >= 2) SPT_LD5(NB, IM, NO, true, true); } \
        else if (exact) SPT_LD5(NB, IM, NO, true, false);               \
        else SPT_LD5(NB, IM, NO, false, false);                         \
    } while (0)
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
#undef SPT_LD5
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_lora_down(const float *x, long long ldx, long long rows, int k, const float *l,
                             int n, float *u, long long ldu, int u_block_major, void *image,
                             float *norms, int exact, void *stream) {
    return lora_down_any(x, ldx, rows, k, l, n, u, ldu, u_block_major, image, norms, exact, nullptr, 1, 0,
                         stream);
}

extern "C" int spt_lora_down2(const float *x, long long ldx, long long rows, int k, const float *l,
                              int n, const float *l2, int n2, float *u, long long ldu, int u_block_major,
                              void *image, float *norms, int exact, void *stream) {
    if (!l2) return SPT_EINVAL;
    return lora_down_any(x, ldx, rows, k, l, n + 16, u, ldu, u_block_major, image, norms, exact, nullptr, 1,
                         0, stream, l2, n2);
}

extern "C" int spt_lora_down_grouped_cols(const float *x, long long ldx, long long rows, int k,
                                          const float *l, long long l_group_stride, int n,
                                          const int32_t *offsets, int n_groups, float *u, long long ldu,
                                          void *stream) {
    if (!offsets || ldu < (long long)n * n_groups) return SPT_EINVAL;
    return lora_down_any(x, ldx, rows, k, l, n, u, ldu, 0, nullptr, nullptr, 0, offsets, n_groups,
                         l_group_stride, stream, nullptr, 0, 0, 1);
}

extern "C" int spt_lora_down_tables(const float *x, long long ldx, long long rows, int k, const float *l,
                                    int n_tables, long long table_stride, float *u, void *image,
                                    float *norms, int exact, void *stream) {
    if (n_tables <= 0 || n_tables > 4 || table_stride < (long long)k * 16) return SPT_EINVAL;
    return lora_down_any(x, ldx, rows, k, l, 16 * n_tables, u, 0, 1, image, norms, exact, nullptr, 1, 0,
                         stream, nullptr, 0, table_stride);
}

extern "C" int spt_lora_down_grouped(const float *x, long long ldx, long long rows, int k,
                                     const float *l, long long l_group_stride, int n,
                                     const int32_t *offsets, int n_groups, float *u, long long ldu,
                                     void *image, float *norms, void *stream) {
    if (!offsets) return SPT_EINVAL;
    return lora_down_any(x, ldx, rows, k, l, n, u, ldu, 0, image, norms, 0, offsets, n_groups,
                         l_group_stride, stream);
}

// ---- out = wide^T . narrow (the LoRA table gradients) ----------------------------------------
// Reference: autograd of lora.py:70-80 / lora_ffn.py:87-111 -- every table gradient is
//   out[w, j] = sum_r wide[r, w] * narrow[r, j]     wide [R, W] an activation or its gradient
//                                                    (W = 1024 ...), narrow [R, n] the rank-r side
// 2 R W n flops on R W 4 bytes: HBM-bound for n <= 48 on plain fp32 FMAs (exact fp32 products, no
// split).  A lane owns two adjacent columns of `wide` and streams rows of them (coalesced rows,
// UNROLL loads in flight); narrow[r, :] is the same for the whole wave, so it arrives through the
// scalar cache and enters the FMAs as an SGPR operand: no cross-lane traffic.  The problem is small
// (8 M elements = 8 K per SIMD), so what matters is latency: a workgroup is 8 waves on the same
// 128 columns, 16 rows each (two workgroups per CU = 4 waves per SIMD, each with one or two
// batches of loads), added through LDS in wave order; the chunks' sums (TN_RC = 128 rows) go to
// `workspace` and a second launch adds them in chunk order (deterministic throughout).
// History: one wave per SIMD with 64-row chunks per wave: 26 us (16384 x 1024 x 16: 34) where
// the library's batched product takes 11-13.
//   grouped form (offsets != null): rows offsets[g] .. offsets[g + 1] - 1 contribute to out[g]
//   (the routed FFN's per-block tables: rows are sorted by block) -- the torch composition
//   scatters `narrow` into a [R, G n] matrix of mostly zeros first.
//   gather != null: narrow row = gather[r] (rows of a per-token matrix picked per (token, block) row).
namespace spt {

constexpr int TN_WAVES = 8;
constexpr int TN_RW = 16;                      // rows per wave
constexpr int TN_RC = TN_WAVES * TN_RW;        // rows per chunk
constexpr int TN_COLS = 128;                   // columns per workgroup (two per lane)

// chunk c -> (group, first row, rows): groups' chunk ranges are consecutive
__device__ __forceinline__ void tn_chunk(const int32_t *offsets, int G, long long R, int c, int &g,
                                         long long &row0, int &rows) {
    if (!offsets) { g = 0; row0 = (long long)c * TN_RC; rows = (int)min((long long)TN_RC, R - row0); return; }
    int base = 0;
    g = -1; row0 = 0; rows = 0;
    for (int i = 0; i < G; i++) {
        const int lo = offsets[i], hi = offsets[i + 1];
        const int nc = (hi - lo + TN_RC - 1) / TN_RC;
        if (g < 0 && c < base + nc) {
            g = i; row0 = lo + (long long)(c - base) * TN_RC;
            rows = (int)min((long long)TN_RC, hi - row0);
        }
        base += nc;
    }
}

// Up to TN_BATCH products of one shape in one launch (blockIdx.z picks the problem): the two table
// gradients of a LoRA linear, the three `right` gradients of q / k / v -- each alone is 512
// workgroups for ~6 us of HBM time, half of what the launch takes.
constexpr int TN_BATCH = 4;
struct TnBatch {
    const float *wide[TN_BATCH];
    const float *narrow[TN_BATCH];
    float *out[TN_BATCH];
};

template <int N, int UNROLL, bool GATHER>
__global__ __launch_bounds__(64 * TN_WAVES, 2) void tall_tn_partial_kernel(
    TnBatch batch, long long ldw, long long ldn,
    const int32_t *__restrict__ gather, const int32_t *__restrict__ offsets, int G, long long R, int W,
    float *__restrict__ partial) {
    constexpr int JB = N < 16 ? N : 16;                      // table columns per reduction pass
    __shared__ float2 red[TN_WAVES][JB][64];
    const float *__restrict__ wide = batch.wide[blockIdx.z];
    const float *__restrict__ narrow = batch.narrow[blockIdx.z];
    partial += (size_t)blockIdx.z * gridDim.x * N * W;
    int g, rows; long long row0;
    tn_chunk(offsets, G, R, blockIdx.x, g, row0, rows);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col0 = blockIdx.y * TN_COLS;
    const int col = min(col0 + 2 * lane, W - 2);             // (clamped lanes are never stored)
    float acc[2][N];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int j = 0; j < N; j++) acc[c][j] = 0.0f;
    const int wr0 = wave * TN_RW;                            // this wave's rows of the chunk
    const int wrows = min(TN_RW, rows - wr0);                // (<= 0: nothing, g < 0: rows = 0)
    // the next batch of rows is requested before the current one is used -- and the first one
    // before the chunk's `narrow` rows are staged: a wave has only two batches, and every
    // workgroup of the launch is resident at once, so nothing else hides its prologue
    float2 vn[UNROLL];
    const float *wp = wide + (row0 + wr0) * ldw + col;
    auto request = [&](int r0) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)                     // rows past the end: clamped, weight 0
            vn[u] = *reinterpret_cast<const float2 *>(wp + (long long)min(r0 + u, wrows - 1) * ldw);
    };
    if (wrows > 0) request(0);
    // the chunk's rows of `narrow` -> LDS (the memory of `red`, free until the sums are formed):
    // read from there as wave-wide broadcasts.  (As scalar loads every row was a miss of the
    // scalar cache -- a 64-byte line used once -- in the middle of the FMA chain: 12 us for 33 MB.)
    float *nar = reinterpret_cast<float *>(&red[0][0][0]);   // [rows][N]
    static_assert(sizeof(red) >= (size_t)TN_RC * N * sizeof(float), "narrow chunk fits in red");
    for (int i = threadIdx.x; i < rows * (N / 4); i += 64 * TN_WAVES) {
        const int r = i / (N / 4), q = i - r * (N / 4);
        const long long src = GATHER ? (long long)gather[row0 + r] : row0 + r;
        reinterpret_cast<float4 *>(nar)[i] = *reinterpret_cast<const float4 *>(narrow + src * ldn + 4 * q);
    }
    __syncthreads();
    if (wrows > 0) {
        for (int r0 = 0; r0 < wrows; r0 += UNROLL) {
            float2 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) v[u] = vn[u];
            if (r0 + UNROLL < wrows) request(r0 + UNROLL);
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                if (r0 + u >= wrows) v[u] = make_float2(0.0f, 0.0f);
                const float4 *np = reinterpret_cast<const float4 *>(nar + (wr0 + min(r0 + u, wrows - 1)) * N);
#pragma unroll
                for (int j4 = 0; j4 < N / 4; j4++) {
                    const float4 b = np[j4];                 // the same address in every lane
                    acc[0][4 * j4 + 0] = fmaf(v[u].x, b.x, acc[0][4 * j4 + 0]);
                    acc[1][4 * j4 + 0] = fmaf(v[u].y, b.x, acc[1][4 * j4 + 0]);
                    acc[0][4 * j4 + 1] = fmaf(v[u].x, b.y, acc[0][4 * j4 + 1]);
                    acc[1][4 * j4 + 1] = fmaf(v[u].y, b.y, acc[1][4 * j4 + 1]);
                    acc[0][4 * j4 + 2] = fmaf(v[u].x, b.z, acc[0][4 * j4 + 2]);
                    acc[1][4 * j4 + 2] = fmaf(v[u].y, b.z, acc[1][4 * j4 + 2]);
                    acc[0][4 * j4 + 3] = fmaf(v[u].x, b.w, acc[0][4 * j4 + 3]);
                    acc[1][4 * j4 + 3] = fmaf(v[u].y, b.w, acc[1][4 * j4 + 3]);
                }
            }
        }
    }
    __syncthreads();                                         // `nar` is read: `red` may be written
    // the eight waves' sums through LDS, JB table columns at a time; partial [chunk][N][W]
    float *pp = partial + (size_t)blockIdx.x * N * W;
#pragma unroll
    for (int jb = 0; jb < N; jb += JB) {
        if (jb) __syncthreads();
#pragma unroll
        for (int jj = 0; jj < JB; jj++) red[wave][jj][lane] = make_float2(acc[0][jb + jj], acc[1][jb + jj]);
        __syncthreads();
        for (int o = threadIdx.x; o < JB * 64; o += 64 * TN_WAVES) {
            const int jj = o >> 6, pair = o & 63;
            float2 t = red[0][jj][pair];
#pragma unroll
            for (int w = 1; w < TN_WAVES; w++) { t.x += red[w][jj][pair].x; t.y += red[w][jj][pair].y; }
            const int c = col0 + 2 * pair;
            if (c < W) *reinterpret_cast<float2 *>(pp + (size_t)(jb + jj) * W + c) = t;
        }
    }
}

// out[g][w][j] (or transposed: out[g][j][w]) = sum over the chunks of group g.  A workgroup = 64
// elements x 4 slices of the chunk range (consecutive quarters), the slices added in order: fixed
// summation tree.  (One thread per element: 16 K threads on 64 CUs, 8 us for 4 MB.)
__global__ __launch_bounds__(256) void tall_tn_reduce_kernel(
    const float *__restrict__ partial, const int32_t *__restrict__ offsets, int G, int W, int n,
    int nchunks, TnBatch batch, int transposed) {
    __shared__ float slice_sum[4][64];
    float *__restrict__ out = batch.out[blockIdx.z];
    partial += (size_t)blockIdx.z * nchunks * n * W;
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;                    // element (j, w) of the group's table
    int first = 0, count;
    if (!offsets) {
        count = nchunks;
    } else {
        for (int i = 0; i < g; i++) first += (offsets[i + 1] - offsets[i] + TN_RC - 1) / TN_RC;
        count = (offsets[g + 1] - offsets[g] + TN_RC - 1) / TN_RC;
    }
    const int per = (count + 3) / 4;
    const int c0 = min(slice * per, count), c1 = min(c0 + per, count);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (e < W * n) {
        const size_t step = (size_t)W * n;
        const float *p = partial + (size_t)first * step + e;
        int c = c0;
        for (; c + 8 <= c1; c += 8) {                        // eight loads in flight
            float t[8];
#pragma unroll
            for (int i = 0; i < 8; i++) t[i] = p[(size_t)(c + i) * step];
#pragma unroll
            for (int i = 0; i < 8; i++) a[i & 3] += t[i];
        }
        for (; c < c1; c++) a[0] += p[(size_t)c * step];
    }
    slice_sum[slice][lane] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (slice == 0 && e < W * n) {
        const float v = (slice_sum[0][lane] + slice_sum[1][lane]) + (slice_sum[2][lane] + slice_sum[3][lane]);
        const int j = e / W, w = e - j * W;
        // layout 0: [w][j]; 1: [j][w]; 2: [j / 16][w][16] (rank-16 tables side by side: each its own matrix)
        const size_t at = transposed == 1 ? (size_t)e
                        : transposed == 2 ? ((size_t)(j >> 4) * W + w) * 16 + (j & 15)
                                          : (size_t)w * n + j;
        out[(size_t)g * W * n + at] = v;
    }
}

}  // namespace spt

extern "C" long long spt_tall_tn_workspace_bytes(long long rows, int n_groups, int width, int n) {
    if (rows <= 0 || n_groups <= 0 || width <= 0 || n <= 0) return 0;
    const long long nchunks = (rows + TN_RC - 1) / TN_RC + n_groups;
    return nchunks * width * n * (long long)sizeof(float);
}

static int tall_tn_any(int count, const float *const *wides, long long ldw, const float *const *narrows,
                       long long ldn, const int32_t *gather, const int32_t *offsets, int n_groups,
                       long long rows, int width, int n, float *const *outs, int transposed,
                       void *workspace, void *stream) {
    if (count <= 0 || count > TN_BATCH || !wides || !narrows || !outs || !workspace) return SPT_EINVAL;
    if (rows <= 0 || width <= 0 || n <= 0 || n_groups <= 0 || ldw < width || ldn < n) return SPT_EINVAL;
    if (!offsets && n_groups != 1) return SPT_EINVAL;
    if (n != 4 && n != 16 && n != 48) return SPT_EUNSUP;
    if (transposed < 0 || transposed > 2 || (transposed == 2 && n % 16 != 0)) return SPT_EINVAL;
    if (n_groups > 64 || rows > 0x7FFFFFFFll) return SPT_EUNSUP;
    if (width % 2 != 0 || ldw % 2 != 0 || ldn % 4 != 0) return SPT_ESHAPE;
    TnBatch batch = {};
    for (int i = 0; i < count; i++) {
        if (!wides[i] || !narrows[i] || !outs[i]) return SPT_EINVAL;
        if ((reinterpret_cast<uintptr_t>(wides[i]) & 7) != 0 || !ls_aligned(narrows[i])) return SPT_ESHAPE;
        batch.wide[i] = wides[i]; batch.narrow[i] = narrows[i]; batch.out[i] = outs[i];
    }
    // (upper bound of the chunk count: each group ends with at most one short chunk; chunks past a
    // group's rows find no group and write zeros that the reduction never reads)
    const int nchunks = (int)((rows + TN_RC - 1) / TN_RC) + (offsets ? n_groups : 0);
    hipStream_t s = (hipStream_t)stream;
    float *partial = static_cast<float *>(workspace);
    const dim3 grid(nchunks, (width + TN_COLS - 1) / TN_COLS, count);
#define SPT_TN(N, U)                                                                              \
    do {                                                                                          \
        if (gather)                                                                               \
            hipLaunchKernelGGL((tall_tn_partial_kernel<N, U, true>), grid, dim3(64 * TN_WAVES), 0, s, \
                               batch, ldw, ldn, gather, offsets, n_groups, rows, width, partial); \
        else                                                                                      \
            hipLaunchKernelGGL((tall_tn_partial_kernel<N, U, false>), grid, dim3(64 * TN_WAVES), 0, s, \
                               batch, ldw, ldn, gather, offsets, n_groups, rows, width, partial); \
    } while (0)
    if (n == 4) SPT_TN(4, 8);
    else if (n == 16) SPT_TN(16, 8);
    else SPT_TN(48, 4);
#undef SPT_TN
    SPT_LAUNCH_CHECK();
    hipLaunchKernelGGL(tall_tn_reduce_kernel, dim3((width * n + 63) / 64, n_groups, count), dim3(256), 0, s,
                       partial, offsets, n_groups, width, n, nchunks, batch, transposed);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_tall_tn(const float *wide, long long ldw, const float *narrow, long long ldn,
                           const int32_t *gather, const int32_t *offsets, int n_groups,
                           long long rows, int width, int n, float *out, int transposed,
                           void *workspace, void *stream) {
    return tall_tn_any(1, &wide, ldw, &narrow, ldn, gather, offsets, n_groups, rows, width, n, &out,
                       transposed, workspace, stream);
}

extern "C" int spt_tall_tn_batch(int count, const float *const *wides, long long ldw,
                                 const float *const *narrows, long long ldn, const int32_t *gather,
                                 const int32_t *offsets, int n_groups, long long rows, int width,
                                 int n, float *const *outs, int transposed, void *workspace,
                                 void *stream) {
    return tall_tn_any(count, wides, ldw, narrows, ldn, gather, offsets, n_groups, rows, width, n, outs,
                       transposed, workspace, stream);
}
