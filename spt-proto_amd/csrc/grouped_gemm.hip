// grouped_gemm.hip -- token-bucketed grouped GEMM for the routed FFN, fp32 on MFMA.
//
// The reference evaluates a routed FFN with a Python loop over blocks, boolean-mask
// gathers (one device->host sync per block) and cuBLAS calls on the gathered rows
// (naive_gpt/layers/tuning/lora_ffn.py:87-111, layers/sparse/feedforward.py:66-85).
// Here the (token, block) pairs are sorted by block on the device; bucket g is the row
// range [offsets[g], offsets[g+1]) of the sorted order and ONE launch multiplies every
// bucket by its own weight block:
//
//     out[p, n] = rowscale[p] * ( sum_k A[src(p), k] * W_g(n, k)  +  bias[g, n] )  (+ A2 . B2_g^T)
//     src(p) = gather ? gather[p] : p          (fuses the token gather)
//     W_g(n, k) = w[g * gstride + n * ldn + k * ldk]
//         ldk == 1 : "BT" weights, k contiguous  (forward: x.W1_g^T, h.W2_g^T)
//         ldn == 1 : "BN" weights, n contiguous  (backward: dY.W2_g, dH.W1_g)
//
// Bucket sizes never visit the host: the grid is sized for the worst case
// (ceil(P/128) + G row tiles) and every workgroup finds its bucket from `offsets`.
//
// This is the one place of the hot path where the contraction is dense, so it runs on
// the matrix cores: three v_mfma_f32_32x32x16_bf16 per 16 k's on fp32 operands split in two
// (see GgFrag below; 205-212 TFLOP/s fp32-equivalent measured), or -- for the forward GEMM in
// front of a ReLU -- six on a three-way split (fp32-level accuracy).  The fp32 MFMA
// v_mfma_f32_32x32x2_f32 (157 TF peak, 98-116 measured) remains behind -DGG_EXACT_FP32.  128 x 128 output tile per 256-thread
// workgroup, each wave a 64 x 64 quadrant (2 x 2 MFMA tiles, 64 accumulator registers);
// K is consumed in steps of 32 through LDS.  LDS image of an operand tile: [row][k] in
// natural order; an MFMA contracts two k's, one from each half of the wave, and any
// pairing is a valid summation order, so lane l takes k = 8 q + 4 (l >> 5) + e for step e
// of group q: one ds_read_b128 feeds four consecutive MFMAs and the staging store is one
// ds_write_b128 of the float4 that came from global memory (no register shuffling).  Rows
// are padded by 16 bytes, which makes the b128 reads of 16 consecutive rows conflict-free.
#include "spt_common.h"
#include <type_traits>

namespace spt {

// Workgroup shape: 256 threads own a 128 x 128 tile (waves 2 x 2, 64 x 64 each); -DGG_WAVES_M=4:
// 512 threads own 256 x 128 (waves 4 x 2), the B tile staged once per 256 rows -- 48 KiB of
// operands per k-step for twice the products, one workgroup per CU.
#ifndef GG_WAVES_M
#define GG_WAVES_M 2
#endif
constexpr int GG_THREADS = 128 * GG_WAVES_M;
constexpr int GG_BM = 64 * GG_WAVES_M;
constexpr int GG_BN = 128;
#ifndef GG_BK_VALUE
#define GG_BK_VALUE 32
#endif
constexpr int GG_BK = GG_BK_VALUE;
// Register sets of prefetched k-steps in flight per thread (each 32 VGPRs: 164 / 196 / 228), and
// the workgroups per CU the round arithmetic of grouped_gemm_kernel counts on.  Measured at the
// FFN shape (1040 tiles; tools/time_gemm_variants.sh): PF 2 and 3 = +-0 % (160 / 172 / 168 us),
// so a k-step is not waiting for its loads; 2 slots per CU (more full tiles, fewer 64-row
// halves -- a half stages 75 % of a full tile's bytes for half its products) 152 us against 162
// with 3, the upgraded block 2.68 against 2.77 ms.
#ifndef GG_PF_VALUE
#define GG_PF_VALUE 1
#endif
#ifndef GG_SLOTS_PER_CU
#define GG_SLOTS_PER_CU 2
#endif
// Two LDS stages (80 KiB per workgroup, two workgroups per CU): the next tile's images are
// written while the current one is contracted, one barrier per k-step.  Not for the three-part
// images of the activation epilogue (120 KiB: one workgroup per CU).
#ifndef GG_DB_VALUE
#define GG_DB_VALUE 0
#endif
template <int EPI> struct GgStages { static constexpr int value = (GG_DB_VALUE && EPI != 1) ? 2 : 1; };
constexpr int GG_PF = GG_PF_VALUE;
constexpr int GG_KQ = GG_BK / 4;               // float4 per tile row
constexpr int GG_RPP = GG_THREADS / GG_KQ;      // tile rows staged per pass
constexpr int GG_NU = GG_BN / GG_RPP;           // float4 of B per thread per k-step
constexpr int GG_BNK = GG_THREADS / 32;        // k rows of an n-contiguous B tile staged per pass
// LDS images of a tile: bf16, one image per part of the split (hi, lo and -- GEMMs in front of a
// ReLU -- mid), written ONCE when the tile is staged (each element is an operand of two waves'
// MFMAs: splitting the fragments in every wave doubled the conversion work and had the matrix
// pipe at 28 %).  k-contiguous tiles: rows of 32 k (64 B + 16 pad): a fragment (8 k of one row)
// is one conflict-free ds_read_b128 per part.  n-contiguous weight tiles stay [k][n] (rows of
// 128 n, 256 B + 64 pad) and are read with the transposing ds_read_b64_tr_b16.
constexpr int GG_ROWB = GG_BK * 2 + 16;          // bytes per row of a k-contiguous image: 80
constexpr int GG_BNROWB = GG_BN * 2 + 64;        // bytes per k-row of an n-contiguous image: 320
constexpr int GG_AIMG = GG_BM * GG_ROWB;         // one part of the A tile: 10240 B
constexpr int GG_BIMG = GG_BN * GG_ROWB;         // one part of a B tile (either orientation)
static_assert(GG_BK * GG_BNROWB == GG_BIMG, "both B orientations fit the same slot");

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Contraction on the bf16 matrix cores with fp32 operands split in two (x = hi + lo, hi = RNE
// bf16 of x, lo = RNE bf16 of x - hi; lo*hi + hi*lo + hi*hi, fp32 accumulation): <= 2^-16
// relative error per product at three v_mfma_f32_32x32x16_bf16 per 16 k's -- 1/5 of the time
// of the eight v_mfma_f32_32x32x2_f32 they replace.  Same scheme as mfma_attention.hip.
// -DGG_EXACT_FP32 keeps the fp32 MFMA (a fixed-order fmaf chain, bit-exact fp32).
typedef __attribute__((ext_vector_type(8))) __bf16 gg_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 gg_bf16x2;
typedef __attribute__((ext_vector_type(2))) float gg_f32x2;
struct GgFrag { uint4 hi, lo; };
__device__ __forceinline__ void gg_split2(float a, float b, unsigned &hi, unsigned &lo) {
#ifdef GG_EXP_NOSPLIT   // timing experiment only (wrong numbers): what the split's VALU costs
    hi = (__builtin_bit_cast(unsigned, a) >> 16) | (__builtin_bit_cast(unsigned, b) & 0xffff0000u);
    lo = hi;
    return;
#endif
    const gg_f32x2 x = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, gg_bf16x2));
    const gg_f32x2 hf = {__builtin_bit_cast(float, hi << 16),
                         __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(x - hf, gg_bf16x2));
}
__device__ __forceinline__ GgFrag gg_split8(const float (&x)[8]) {
    GgFrag f;
    gg_split2(x[0], x[1], f.hi.x, f.lo.x);
    gg_split2(x[2], x[3], f.hi.y, f.lo.y);
    gg_split2(x[4], x[5], f.hi.z, f.lo.z);
    gg_split2(x[6], x[7], f.hi.w, f.lo.w);
    return f;
}
typedef short gg_v4s16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 gg_tr_b64(const char *p) {      // ds_read_b64_tr_b16
    const gg_v4s16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) gg_v4s16 *)(p));
    return __builtin_bit_cast(uint2, r);
}
__device__ __forceinline__ f32x16 gg_mma(const uint4 &a, const uint4 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gg_bf16x8, a),
                                                   __builtin_bit_cast(gg_bf16x8, b), c, 0, 0, 0);
}
// Three-way split x = hi + mid + lo (3 x 8 mantissa bits: exact to ~2^-24) and the six products
// of order >= 2^-16: fp32-level accuracy (error ~2^-23 per product) at 6 x 32 cycles per 16 k
// against 8 x 64 for the fp32 MFMA.  Used where a 1e-5 error is not acceptable: the
// pre-activation of a ReLU (see exact_fp32 in gemm_tile).
struct GgFrag3 { uint4 hi, mid, lo; };
__device__ __forceinline__ void gg_split2x3(float a, float b, unsigned &hi, unsigned &mid,
                                            unsigned &lo) {
    gg_f32x2 x = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, gg_bf16x2));
    x -= gg_f32x2{__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(x, gg_bf16x2));
    x -= gg_f32x2{__builtin_bit_cast(float, mid << 16), __builtin_bit_cast(float, mid & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(x, gg_bf16x2));
}
__device__ __forceinline__ GgFrag3 gg_split8x3(const float (&x)[8]) {
    GgFrag3 f;
    gg_split2x3(x[0], x[1], f.hi.x, f.mid.x, f.lo.x);
    gg_split2x3(x[2], x[3], f.hi.y, f.mid.y, f.lo.y);
    gg_split2x3(x[4], x[5], f.hi.z, f.mid.z, f.lo.z);
    gg_split2x3(x[6], x[7], f.hi.w, f.mid.w, f.lo.w);
    return f;
}
__device__ __forceinline__ f32x16 gg_mma6(const GgFrag3 &a, const GgFrag3 &b, f32x16 c) {
    c = gg_mma(a.lo, b.hi, c);      // small terms first
    c = gg_mma(a.hi, b.lo, c);
    c = gg_mma(a.mid, b.mid, c);
    c = gg_mma(a.mid, b.hi, c);
    c = gg_mma(a.hi, b.mid, c);
    return gg_mma(a.hi, b.hi, c);
}
__device__ __forceinline__ f32x16 gg_mma3(const GgFrag &a, const GgFrag &b, f32x16 c) {
    c = gg_mma(a.lo, b.hi, c);
    c = gg_mma(a.hi, b.lo, c);
    return gg_mma(a.hi, b.hi, c);
}

enum { EPI_PLAIN = 0, EPI_ACT = 1, EPI_DACT = 2 };
enum { ACT_RELU = 0, ACT_GELU = 1, ACT_SILU = 2 };

struct GroupedArgs {
    const float *a;         // [*, K] row-major, leading dimension lda
    const int32_t *gather;  // [P] or null
    const float *w;
    const float *bias;      // [G, N] or null
    const float *rowscale;  // [P] or null
    const int32_t *offsets; // [G + 1]
    float *out;             // [P, N] row-major
    int P, K, N, G;
    int lda;
    long long gstride;
    int ldn, ldk;
    // K extension (the LoRA side product), added AFTER rowscale / bias:
    //   v[p, n] += sum_{j < R} a2[src2(p), j] * b2[g * b2_gstride + n * b2_ldn + j]
    const float *a2;
    const int32_t *gather2;
    const float *b2;
    int lda2, R;
    long long b2_gstride;
    int b2_ldn;
    // epilogue
    int act;
    float *out2;            // EPI_ACT: pre-activation (null: not kept, e.g. ReLU)
    const float *h_in;      // EPI_DACT: activated values [P, N] (used when s_in is null: ReLU)
    const float *s_in;      // EPI_DACT: pre-activation values [P, N] or null
    float *pdot_main;       // EPI_DACT: [P, pdot_ld]: sum_n v[p, n] * h[p, n], v = the value before act'
    float *pdot_act;        // EPI_DACT: [P, pdot_ld]: sum_n out[p, n] * s[p, n] per half tile
    int pdot_ld;
    int slots;              // workgroups resident at a time (CUs x occupancy)
};

__device__ __forceinline__ float act_forward(int act, float s) {
    if (act == ACT_RELU) return fmaxf(s, 0.0f);
    if (act == ACT_GELU) return 0.5f * s * (1.0f + erff(s * 0.70710678118654752f));
    return s / (1.0f + __expf(-s));
}

__device__ __forceinline__ float act_derivative(int act, float s) {
    if (act == ACT_RELU) return s > 0.0f ? 1.0f : 0.0f;
    if (act == ACT_GELU) {
        const float cdf = 0.5f * (1.0f + erff(s * 0.70710678118654752f));
        return cdf + s * 0.3989422804014327f * __expf(-0.5f * s * s);
    }
    const float sg = 1.0f / (1.0f + __expf(-s));
    return sg * (1.0f + s * (1.0f - sg));
}

// One output tile of BM x 128: BM = 128 (each wave a 64 x 64 quadrant) or BM = 64 (each
// wave 32 x 64), same B tile, same LDS image, same epilogue.
template <int BM, bool BN_LAYOUT, int EPI, bool EXT, bool KTAIL>
__device__ __forceinline__ void gemm_tile(const GroupedArgs &g, float *smem, int bucket,
                                          int row_lo, int row_hi, int col_tile) {
    constexpr int NI = BM / (32 * GG_WAVES_M);   // 32-row sub-blocks per wave
    constexpr int NUA = BM / GG_RPP;          // float4 of A per thread per k-step
    constexpr int NPART = (EPI == EPI_ACT) ? 3 : 2;      // hi, lo (, mid)
    constexpr int STAGES = GgStages<EPI>::value;
    constexpr int STAGE_BYTES = NPART * (GG_AIMG + GG_BIMG);
    char *const lds0 = reinterpret_cast<char *>(smem);
    char *As = lds0;                                      // [NPART][GG_AIMG]: hi | lo | mid
    char *Bs = As + NPART * GG_AIMG;                       // [NPART][GG_BIMG]
    const int n0 = col_tile * GG_BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = (wave >> 1) * (BM / GG_WAVES_M);    // the wave's origin inside the tile
    const int wn = (wave & 1) * 64;
    const float *wg = g.w + (size_t)bucket * g.gstride;

    // ---- staging assignment: tile = rows x GG_KQ float4 along k ----
    // thread -> row (tid / GG_KQ) + GG_RPP u, k-quad tid % GG_KQ
    // (8 lanes per row; the two rows of a 16-lane ds_write_b64 group are 4 apart: 320 bytes = 64
    // mod 128, so their 64-byte pieces fall on disjoint banks -- adjacent rows, 80 bytes apart,
    // share four: SQ_LDS_BANK_CONFLICT was a third of the LDS-array cycles)
#ifdef GG_ROWS_IN_ORDER
    const int s_row = tid / GG_KQ, s_kq = tid % GG_KQ;
#else
    static_assert(GG_KQ == 8, "row interleave below assumes 8 lanes per tile row");
    const int s_slot = (tid >> 3) & 7;
    const int s_row = (tid >> 6) * 8 + (((s_slot & 1) << 2) | (s_slot >> 1)), s_kq = tid & 7;
#endif
    // Rows past the bucket end and columns past N are computed but never stored, so their
    // operands only have to be readable: clamp them to the last valid row / column instead
    // of predicating the loads.
    const float *a_src[NUA];
#pragma unroll
    for (int u = 0; u < NUA; u++) {
        const int p = min(row_lo + s_row + GG_RPP * u, row_hi - 1);
        const int src = g.gather ? g.gather[p] : p;
        a_src[u] = g.a + (size_t)src * g.lda;
    }
    const float *b_src[GG_NU];
#pragma unroll
    for (int u = 0; u < GG_NU; u++)
        b_src[u] = wg + (size_t)min(n0 + s_row + GG_RPP * u, g.N - 1) * g.ldn;
    const float *bn_src = wg + min(n0 + 4 * (tid & 31), g.N - 4);

    f32x16 acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    const int frow = lane & 31, fh = lane >> 5;
    // ---- fragments: lane l holds A[row = l & 31][k = 16 q2 + 8 (l >> 5) + 0..7] (16 bytes of a
    // k-contiguous image) and B likewise; `bt_image`: B tile stored [n][k] (k-contiguous weights
    // and the K extension), else [k][n], read with the transposing ds_read_b64_tr_b16 (lane 4q+p
    // of a 16-lane group supplies row k0 + q, columns 4p.. of the group's 16 n; lane i receives
    // column i of the four rows: two blocks = the fragment's 8 k's) ----
    auto frag_rows = [&](const char *img, int row, int q2) {
        return *reinterpret_cast<const uint4 *>(img + row * GG_ROWB + 32 * q2 + 16 * fh);
    };
    auto frag_cols = [&](const char *img, int ncol0, int q2) {
        const int gl = lane & 15;
        const char *p = img + (16 * q2 + 8 * fh + (gl >> 2)) * GG_BNROWB +
                        (ncol0 + 16 * ((lane >> 4) & 1) + 4 * (gl & 3)) * 2;
        const uint2 lo = gg_tr_b64(p), hi = gg_tr_b64(p + 4 * GG_BNROWB);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    auto b_frag = [&](int part, int j, int q2, bool bt_image) {
        const char *img = Bs + part * GG_BIMG;
        return bt_image ? frag_rows(img, wn + 32 * j + frow, q2) : frag_cols(img, wn + 32 * j, q2);
    };
    // 16 k's, two-way split: three products
    auto mfma_group16 = [&](int q2, bool bt_image) {
        GgFrag af[NI], bf[2];
#pragma unroll
        for (int i = 0; i < NI; i++) {
            af[i].hi = frag_rows(As, wm + 32 * i + frow, q2);
            af[i].lo = frag_rows(As + GG_AIMG, wm + 32 * i + frow, q2);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            bf[j].hi = b_frag(0, j, q2, bt_image);
            bf[j].lo = b_frag(1, j, q2, bt_image);
        }
#pragma unroll
        for (int i = 0; i < NI; i++) {
            acc[i][0] = gg_mma3(af[i], bf[0], acc[i][0]);
            acc[i][1] = gg_mma3(af[i], bf[1], acc[i][1]);
        }
    };
    // the same 16 k's at fp32-level accuracy: three-way split, six products
    auto mfma_group16x6 = [&](int q2, bool bt_image) {
        GgFrag3 af[NI], bf[2];
#pragma unroll
        for (int i = 0; i < NI; i++) {
            af[i].hi = frag_rows(As, wm + 32 * i + frow, q2);
            af[i].lo = frag_rows(As + GG_AIMG, wm + 32 * i + frow, q2);
            af[i].mid = frag_rows(As + 2 * GG_AIMG, wm + 32 * i + frow, q2);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            bf[j].hi = b_frag(0, j, q2, bt_image);
            bf[j].lo = b_frag(1, j, q2, bt_image);
            bf[j].mid = b_frag(2, j, q2, bt_image);
        }
#pragma unroll
        for (int i = 0; i < NI; i++) {
            acc[i][0] = gg_mma6(af[i], bf[0], acc[i][0]);
            acc[i][1] = gg_mma6(af[i], bf[1], acc[i][1]);
        }
    };
    // ---- staging: four fp32 values -> the 8-byte pieces of every part of an image ----
    // (lo of the two-way split = RNE(x - hi); of the three-way split the image order is
    // hi | lo(last part) | mid, so that parts 0 and 1 are what mfma_group16 reads either way:
    // a two-way tile just has a coarser `lo`)
    auto put4 = [&](char *img, int part_stride, int off, const float4 &v, bool three) {
        if (three) {
            unsigned h0, m0, l0, h1, m1, l1;
            gg_split2x3(v.x, v.y, h0, m0, l0);
            gg_split2x3(v.z, v.w, h1, m1, l1);
            *reinterpret_cast<uint2 *>(img + off) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(img + part_stride + off) = make_uint2(l0, l1);
            *reinterpret_cast<uint2 *>(img + 2 * part_stride + off) = make_uint2(m0, m1);
        } else {
            unsigned h0, l0, h1, l1;
            gg_split2(v.x, v.y, h0, l0);
            gg_split2(v.z, v.w, h1, l1);
            *reinterpret_cast<uint2 *>(img + off) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(img + part_stride + off) = make_uint2(l0, l1);
        }
    };
    // The one contraction that needs fp32-level accuracy: the forward GEMM whose result goes
    // through ReLU.  A 1e-5 perturbation of a pre-activation that sits on the kink flips its
    // derivative (measured with the two-way split: ~6 of 614 k elements, each an O(1) error in
    // one token's gradients); every other product feeds smooth functions.  It takes the
    // three-way split (this GEMM took 352 us on the fp32 MFMA at the block-bench shape).
    const bool fp32_level = (EPI == EPI_ACT) && g.act == ACT_RELU;
    auto contract = [&](int kmax, bool bt_image) {        // k = 0 .. kmax of the staged tiles
        if (EPI == EPI_ACT && fp32_level) {
            for (int q2 = 0; q2 < (kmax + 15) / 16; q2++) mfma_group16x6(q2, bt_image);
        } else {
            for (int q2 = 0; q2 < (kmax + 15) / 16; q2++) mfma_group16(q2, bt_image);
        }
    };

    // ---- software pipeline: the global loads of k-step t+1 are in flight while the MFMAs
    // of step t run; registers -> LDS happens at the top of the next step.  (A second
    // register set, two steps in flight, changed nothing for the half tiles of the last
    // round: +-0 % measured.) ----
    constexpr int PF = GG_PF;
    float4 av[PF][NUA], bv[PF][GG_NU];
    // KTAIL == false (K % GG_BK == 0, every shape of the FFN): no predicate anywhere in the
    // loads.  The predicated form compiles into branches around the loads, 8 per k-step.
    auto load_tile = [&](float4 (&a)[NUA], float4 (&b)[GG_NU], int k0) {
        const int k = k0 + 4 * s_kq;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < NUA; u++) {
            if constexpr (KTAIL)
                a[u] = k < g.K ? *reinterpret_cast<const float4 *>(a_src[u] + k) : zero;
            else
                a[u] = *reinterpret_cast<const float4 *>(a_src[u] + k);
        }
        if constexpr (!BN_LAYOUT) {
            // W_g(n, k), k contiguous: same shape as A
#pragma unroll
            for (int u = 0; u < GG_NU; u++) {
                if constexpr (KTAIL)
                    b[u] = k < g.K ? *reinterpret_cast<const float4 *>(b_src[u] + k) : zero;
                else
                    b[u] = *reinterpret_cast<const float4 *>(b_src[u] + k);
            }
        } else {
            // W_g(n, k), n contiguous: thread -> k row (tid >> 5) + 8 u, n-quad tid & 31
#pragma unroll
            for (int u = 0; u < GG_NU; u++) {
                const int kk = k0 + (tid >> 5) + GG_BNK * u;
                if constexpr (KTAIL)
                    b[u] = kk < g.K ? *reinterpret_cast<const float4 *>(bn_src + (size_t)kk * g.ldk)
                                    : zero;
                else
                    b[u] = *reinterpret_cast<const float4 *>(bn_src + (size_t)kk * g.ldk);
            }
        }
    };
#ifdef GG_STAMP
    // diagnostic build only: shader-clock time of wave 0 in each phase of the k-loop
    // (barrier 1 | wait for the prefetched tile | split + LDS stores | barrier 2 | reads + MFMAs)
    unsigned long long ph[5] = {0, 0, 0, 0, 0};
#define GG_PHASE(i)                                                        \
    do {                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                 \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
        ph[i] += now_ - ph_t;                                              \
        ph_t = now_;                                                       \
        __builtin_amdgcn_sched_barrier(0);                                 \
    } while (0)
#else
#define GG_PHASE(i)
#endif
    // ---- registers -> LDS images at (a_img, b_img) ----
    // (component-wise: a struct copy of a[u] keeps the whole array in scratch memory)
    auto stage_store = [&](float4 (&a)[NUA], float4 (&b)[GG_NU], char *a_img, char *b_img) {
        const bool three = EPI == EPI_ACT && fp32_level;
#pragma unroll
        for (int u = 0; u < NUA; u++)
            put4(a_img, GG_AIMG, (s_row + GG_RPP * u) * GG_ROWB + 8 * s_kq,
                 make_float4(a[u].x, a[u].y, a[u].z, a[u].w), three);
        if (!BN_LAYOUT) {
#pragma unroll
            for (int u = 0; u < GG_NU; u++)
                put4(b_img, GG_BIMG, (s_row + GG_RPP * u) * GG_ROWB + 8 * s_kq,
                     make_float4(b[u].x, b[u].y, b[u].z, b[u].w), three);
        } else {
            // n-contiguous weights keep their orientation in LDS: Bs[k][n], rows of GG_BN + 4
            // floats (transposing them into the [n][k] image needs 4-byte writes 4 rows apart:
            // 16-way bank conflicts)
#pragma unroll
            for (int u = 0; u < GG_NU; u++)
                put4(b_img, GG_BIMG, ((tid >> 5) + GG_BNK * u) * GG_BNROWB + 8 * (tid & 31),
                     make_float4(b[u].x, b[u].y, b[u].z, b[u].w), three);
        }
    };
    // one register of the next tile (no k tail here: only used by the interleaved steady state)
    auto load_a1 = [&](int u, int k0) {
        return *reinterpret_cast<const float4 *>(a_src[u] + k0 + 4 * s_kq);
    };
    auto load_b1 = [&](int u, int k0) {
        if constexpr (!BN_LAYOUT)
            return *reinterpret_cast<const float4 *>(b_src[u] + k0 + 4 * s_kq);
        else
            return *reinterpret_cast<const float4 *>(bn_src + (size_t)(k0 + (tid >> 5) + GG_BNK * u) * g.ldk);
    };
    auto k_step = [&](float4 (&a)[NUA], float4 (&b)[GG_NU], int k_next) {
#ifdef GG_STAMP
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long ph_t = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();  // previous tile fully consumed
        GG_PHASE(0);
#ifdef GG_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GG_PHASE(1);
#endif
        stage_store(a, b, As, Bs);
        GG_PHASE(2);
        __syncthreads();
        GG_PHASE(3);
#ifdef GG_EXP_NOLOAD     // timing experiment only: no global loads after the second k-step
        if (k_next < 2 * GG_BK) load_tile(a, b, k_next);
#else
        if (k_next < g.K) load_tile(a, b, k_next);   // this register set is free again
#endif
#ifdef GG_EXP_NOMFMA     // timing experiment only: staging without the contraction
        if (g.K > 0) return;
#endif
#ifdef GG_PRIO_MFMA
        __builtin_amdgcn_s_setprio(GG_PRIO_MFMA);
#endif
        if (EPI == EPI_ACT && fp32_level) {
#pragma unroll
            for (int q2 = 0; q2 < GG_BK / 16; q2++) mfma_group16x6(q2, !BN_LAYOUT);
        } else {
#pragma unroll
            for (int q2 = 0; q2 < GG_BK / 16; q2++) mfma_group16(q2, !BN_LAYOUT);
        }
#ifdef GG_PRIO_MFMA
        __builtin_amdgcn_s_setprio(GG_PRIO_STAGE);
#endif
        GG_PHASE(4);
    };
    if constexpr (STAGES == 2) {
        // k-step t: write tile t+1 (in registers since step t-1) into the other stage, fetch
        // tile t+2, contract stage t & 1, ONE barrier: it closes both the writes of stage
        // (t+1) & 1 and the reads of stage t & 1.
        // `full`: neither condition can fail (steady state) -- one basic block, so that the
        // split's VALU and the LDS stores can be spread over the MFMA gaps of the same wave
        // (all waves of a CU otherwise fall into lockstep: every one staging, then every one
        // waiting for the matrix pipe)
        auto db_step = [&](auto cur, auto full, int k0) {
            constexpr int CUR = decltype(cur)::value;
            constexpr bool FULL = decltype(full)::value;
            char *a_nxt = lds0 + (CUR ^ 1) * STAGE_BYTES;
            if (FULL || k0 + GG_BK < g.K) stage_store(av[0], bv[0], a_nxt, a_nxt + NPART * GG_AIMG);
            if (FULL || k0 + 2 * GG_BK < g.K) load_tile(av[0], bv[0], k0 + 2 * GG_BK);
            As = lds0 + CUR * STAGE_BYTES;
            Bs = As + NPART * GG_AIMG;
#pragma unroll
            for (int q2 = 0; q2 < GG_BK / 16; q2++) mfma_group16(q2, !BN_LAYOUT);
            __syncthreads();
        };
        // The same step written out for the steady state of a full tile: the wave's 24 MFMAs in
        // eight groups of three (consecutive ones on different accumulators), and after each
        // group one register of the next tile: wait for it, split, two LDS stores, and the
        // global load that refills it for the tile after -- pinned in this order, so that every
        // wave issues matrix work all along the step instead of staging first.
        auto db_step_woven = [&](auto cur, int k0) {
            constexpr int CUR = decltype(cur)::value;
            char *a_nxt = lds0 + (CUR ^ 1) * STAGE_BYTES, *b_nxt = a_nxt + NPART * GG_AIMG;
            As = lds0 + CUR * STAGE_BYTES;
            Bs = As + NPART * GG_AIMG;
            auto chunk = [&](int c) {
                if (c < NUA) {
                    const int u = c;
                    put4(a_nxt, GG_AIMG, (s_row + GG_RPP * u) * GG_ROWB + 8 * s_kq,
                         make_float4(av[0][u].x, av[0][u].y, av[0][u].z, av[0][u].w), false);
                    av[0][u] = load_a1(u, k0 + 2 * GG_BK);
                } else {
                    const int u = c - NUA;
                    const int off = BN_LAYOUT ? ((tid >> 5) + GG_BNK * u) * GG_BNROWB + 8 * (tid & 31)
                                              : (s_row + GG_RPP * u) * GG_ROWB + 8 * s_kq;
                    put4(b_nxt, GG_BIMG, off,
                         make_float4(bv[0][u].x, bv[0][u].y, bv[0][u].z, bv[0][u].w), false);
                    bv[0][u] = load_b1(u, k0 + 2 * GG_BK);
                }
            };
            GgFrag af[2][NI], bf[2][2];
            auto read_frags = [&](int q2) {
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    af[q2][i].hi = frag_rows(As, wm + 32 * i + frow, q2);
                    af[q2][i].lo = frag_rows(As + GG_AIMG, wm + 32 * i + frow, q2);
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    bf[q2][j].hi = b_frag(0, j, q2, !BN_LAYOUT);
                    bf[q2][j].lo = b_frag(1, j, q2, !BN_LAYOUT);
                }
            };
            // product p of accumulator tile t: lo.hi, hi.lo, hi.hi (small terms first)
            auto one = [&](int q2, int idx) {
                const int pth = idx / (2 * NI), t = idx % (2 * NI), i = t >> 1, j = t & 1;
                acc[i][j] = gg_mma(pth == 0 ? af[q2][i].lo : af[q2][i].hi,
                                   pth == 1 ? bf[q2][j].lo : bf[q2][j].hi, acc[i][j]);
            };
            read_frags(0);
#pragma unroll
            for (int q2 = 0; q2 < 2; q2++) {
#pragma unroll
                for (int grp = 0; grp < 4; grp++) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < 3; m++) one(q2, 3 * grp + m);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q2 == 0 && grp == 2) read_frags(1);
                    chunk(4 * q2 + grp);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        };
        using C0 = std::integral_constant<int, 0>;
        using C1 = std::integral_constant<int, 1>;
        load_tile(av[0], bv[0], 0);
        __syncthreads();          // the K extension's reads of stage 0 are done
        stage_store(av[0], bv[0], lds0, lds0 + NPART * GG_AIMG);
        if (GG_BK < g.K) load_tile(av[0], bv[0], GG_BK);
        __syncthreads();
        int k0 = 0;
        for (; k0 + 3 * GG_BK < g.K; k0 += 2 * GG_BK) {     // both steps store and load
            if constexpr (GG_DB_VALUE == 2 && NI == 2 && NUA == 4 && GG_NU == 4 && GG_BK == 32 &&
                          !KTAIL) {
                db_step_woven(C0{}, k0);
                db_step_woven(C1{}, k0 + GG_BK);
            } else {
                db_step(C0{}, std::true_type{}, k0);
                db_step(C1{}, std::true_type{}, k0 + GG_BK);
            }
        }
        for (; k0 < g.K; k0 += 2 * GG_BK) {                 // the last two or three steps
            db_step(C0{}, std::false_type{}, k0);
            if (k0 + GG_BK < g.K) db_step(C1{}, std::false_type{}, k0 + GG_BK);
        }
        As = lds0;
        Bs = As + NPART * GG_AIMG;
    } else {
#pragma unroll
        for (int st = 0; st < PF; st++)
            if (st * GG_BK < g.K) load_tile(av[st], bv[st], st * GG_BK);
        for (int k0 = 0; k0 < g.K; k0 += PF * GG_BK) {
#pragma unroll
            for (int st = 0; st < PF; st++)
                if (k0 + st * GG_BK < g.K) k_step(av[st], bv[st], k0 + (st + PF) * GG_BK);
        }
    }

    // ---- rowscale, then the K extension: acc = rowscale * (A W^T), acc += A2 . B2_g^T; the
    // epilogue adds rowscale * bias.  (Until ABI 14 the extension ran FIRST on A2 / rowscale so
    // that one multiplication in the epilogue served both: a router coefficient of 0 made that
    // 0 * inf, and a tiny one cost the base product its mantissa.) ----
    if (g.rowscale) {
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int p = row_lo + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const float rs = g.rowscale[min(p, row_hi - 1)];
                acc[i][0][r] *= rs;
                acc[i][1][r] *= rs;
            }
    }
    if (EXT) {
        __syncthreads();                  // the last k-step's tiles are consumed
        const int k = 4 * s_kq;
#pragma unroll
        for (int u = 0; u < NUA; u++) {
            const int r = s_row + GG_RPP * u;
            const int p = row_lo + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < row_hi && k < g.R) {
                const int src = g.gather2 ? g.gather2[p] : p;
                v = *reinterpret_cast<const float4 *>(g.a2 + (size_t)src * g.lda2 + k);
            }
            put4(As, GG_AIMG, r * GG_ROWB + 8 * s_kq, v, EPI == EPI_ACT && fp32_level);
        }
#pragma unroll
        for (int u = 0; u < GG_NU; u++) {
            const int r = s_row + GG_RPP * u;
            const int n = n0 + r;
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < g.N && k < g.R)
                b = *reinterpret_cast<const float4 *>(g.b2 + (size_t)bucket * g.b2_gstride +
                                                      (size_t)n * g.b2_ldn + k);
            put4(Bs, GG_BIMG, r * GG_ROWB + 8 * s_kq, b, EPI == EPI_ACT && fp32_level);
        }
        __syncthreads();
        contract(g.R, true);              // (the tiles are zero beyond R, up to GG_BK)
    }

#ifdef GG_STAMP
    if (EPI == EPI_PLAIN && g.pdot_main && tid == 0) {
        unsigned long long *st = reinterpret_cast<unsigned long long *>(g.pdot_main);
        for (int i = 0; i < 5; i++) st[12 * blockIdx.x + 4 + i] = ph[i];
    }
#endif
    // ---- epilogue ----
    // MFMA C layout: acc[i][j][r] = C[32 i + (r & 3) + 8 (r >> 2) + 4 (l >> 5)][32 j + (l & 31)]
    // of the wave's (32 NI) x 64 part.  Each wave transposes it through LDS, 32 rows at a
    // time, into a row layout -- 16 lanes x float4 = one 64-column row segment -- so that
    // every global access of the epilogue (the stores, the h / s tiles of EPI_DACT, the
    // bias) is 16 bytes per lane and 256 contiguous bytes per row, and a row dot is a
    // 16-lane DPP reduction.
    __syncthreads();   // all waves are done with the last k-step's tiles
    constexpr int CS_ROW = 64 + 4;
    float *cs = smem + wave * (32 * CS_ROW);
    const int ccol = lane & 31, chalf = lane >> 5;
    const int rrow = lane >> 4, rcol = 4 * (lane & 15);
    const int pslot = 2 * col_tile + (wave & 1);   // this wave's half tile of columns
    const float *sh = (EPI == EPI_DACT) ? (g.s_in ? g.s_in : g.h_in) : nullptr;
    const int n = n0 + wn + rcol;
    const bool vec = (g.N & 3) == 0;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias) {
        const float *bp = g.bias + (size_t)bucket * g.N + n;
        if (vec) {
            if (n < g.N) bias4 = *reinterpret_cast<const float4 *>(bp);
        } else {
            bias4.x = n + 0 < g.N ? bp[0] : 0.f; bias4.y = n + 1 < g.N ? bp[1] : 0.f;
            bias4.z = n + 2 < g.N ? bp[2] : 0.f; bias4.w = n + 3 < g.N ? bp[3] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < NI; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * chalf;
            cs[row * CS_ROW + ccol] = acc[i][0][r];
            cs[row * CS_ROW + 32 + ccol] = acc[i][1][r];
        }
        // (a wave only reads what it wrote: no workgroup barrier)
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int row = rrow + 4 * t;
            const int p = row_lo + wm + 32 * i + row;
            const bool live = p < row_hi;
            const float4 c4 = *reinterpret_cast<const float4 *>(&cs[row * CS_ROW + rcol]);
            float c[4] = {c4.x, c4.y, c4.z, c4.w};
            const float b[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
            const float rs = (live && g.rowscale) ? g.rowscale[p] : 1.0f;
            const size_t at = (size_t)p * g.N + n;
            float sv[4] = {0.f, 0.f, 0.f, 0.f};
            if (EPI == EPI_DACT && live) {
                if (vec) {
                    if (n < g.N) {
                        const float4 t4 = *reinterpret_cast<const float4 *>(sh + at);
                        sv[0] = t4.x; sv[1] = t4.y; sv[2] = t4.z; sv[3] = t4.w;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (n + e < g.N) sv[e] = sh[at + e];
                }
            }
            float dot_h = 0.0f, dot_s = 0.0f;
            float pre[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float v = fmaf(rs, b[e], c[e]);
                if (EPI == EPI_DACT) {
                    const float hv = g.s_in ? act_forward(g.act, sv[e]) : sv[e];
                    dot_h = fmaf(v, hv, dot_h);
                }
                pre[e] = v;
                if (EPI == EPI_ACT) v = act_forward(g.act, v);
                if (EPI == EPI_DACT) {
                    v *= act_derivative(g.act, sv[e]);
                    dot_s = fmaf(v, sv[e], dot_s);
                }
                c[e] = v;
            }
            if (live) {
                if (vec) {
                    if (n < g.N) {
                        *reinterpret_cast<float4 *>(g.out + at) = make_float4(c[0], c[1], c[2], c[3]);
                        if (EPI == EPI_ACT && g.out2)
                            *reinterpret_cast<float4 *>(g.out2 + at) =
                                make_float4(pre[0], pre[1], pre[2], pre[3]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (n + e < g.N) {
                            g.out[at + e] = c[e];
                            if (EPI == EPI_ACT && g.out2) g.out2[at + e] = pre[e];
                        }
                }
            }
            if (EPI == EPI_DACT) {
                dot_h = group_sum<16>(dot_h);
                dot_s = group_sum<16>(dot_s);
                if (live && (lane & 15) == 0) {
                    g.pdot_main[(size_t)p * g.pdot_ld + pslot] = dot_h;
                    g.pdot_act[(size_t)p * g.pdot_ld + pslot] = dot_s;
                }
            }
        }
    }
}

// Work distribution.  A full tile costs ~200 us whatever it holds, and `slots` workgroups
// run at a time (3 per CU), so a launch of T tiles takes ceil(T / slots) rounds.  Measured
// with per-workgroup timestamps (tools/clock_gemm.py, P = 16384 rows ragged over 4 buckets,
// K = N = 1024, 2.37 GHz in-kernel): 768 tiles start at t = 0 and end at 158-237 us (67 us
// per tile and CU = 83 % MFMA-busy); the other 272 tiles then ran one per CU for another
// 150-190 us, 363-381 us end to end against 253 us of work.  The last, partial round is
// therefore cut into half-height tiles (64 x 128): twice as many workgroups (544, ~2 per
// CU, 78 us each), 300-313 us end to end.  The split is decided on the device -- bucket
// sizes never visit the host -- and workgroup ids are dispatched in order, so ids
// [0, main) take the full rounds, ids [main, main + 2 R) the halves of the R remaining
// tiles, and the rest of the (worst-case sized) grid exits at once.
template <bool BN_LAYOUT, int EPI, bool EXT, bool KTAIL>
__global__ __launch_bounds__(GG_THREADS, GG_WAVES_M == 2 ? 2 : 1) void grouped_gemm_kernel(GroupedArgs g) {
    // As | Bs; the epilogue reuses the whole buffer as four per-wave C staging areas
    // A | B images (two or, with an activation epilogue, three parts each)
    constexpr int IMG_FLOATS = GgStages<EPI>::value * ((EPI == EPI_ACT) ? 3 : 2) * (GG_AIMG + GG_BIMG) / 4;
    constexpr int EPI_FLOATS = (GG_THREADS / 64) * 32 * (64 + 4);       // per-wave C staging
    __shared__ __attribute__((aligned(16))) float smem[IMG_FLOATS > EPI_FLOATS ? IMG_FLOATS : EPI_FLOATS];

    const int n_col_tiles = (g.N + GG_BN - 1) / GG_BN;
    int row_tiles = 0;
    for (int i = 0; i < g.G; i++)
        row_tiles += (g.offsets[i + 1] - g.offsets[i] + GG_BM - 1) / GG_BM;
    const int total = row_tiles * n_col_tiles;
    const int main_tiles = (total / g.slots) * g.slots;
    const int rest = total - main_tiles;
    int id = blockIdx.x;
    int half = -1, logical;
    // XCD-aware order inside each part: workgroups are dealt round-robin over the 8 XCDs
    // (each with its own 4 MiB L2); xcd_remap gives every XCD a contiguous run of logical
    // tiles, enumerated column-tile fastest, so the column tiles of one row tile (same
    // 512 KiB A panel) run back to back on one L2.
    if (id < main_tiles) {
        logical = (int)xcd_remap((unsigned)id, (unsigned)main_tiles);
    } else {
        id -= main_tiles;
        if (id >= 2 * rest) return;
        const int h = (int)xcd_remap((unsigned)id, (unsigned)(2 * rest));
        logical = main_tiles + (h >> 1);
        half = h & 1;
    }
    const int col_tile = logical % n_col_tiles;
    int bucket = -1, row_lo = 0, row_hi = 0;
    {
        int tile = logical / n_col_tiles;
        for (int i = 0; i < g.G; i++) {
            const int lo = g.offsets[i], hi = g.offsets[i + 1];
            const int tiles = (hi - lo + GG_BM - 1) / GG_BM;
            if (tile < tiles) {
                bucket = i;
                row_lo = lo + tile * GG_BM;
                row_hi = min(hi, row_lo + GG_BM);
                break;
            }
            tile -= tiles;
        }
    }
    if (bucket < 0) return;  // uniform for the workgroup
#ifdef GG_STAMP
    // diagnostic build only (tools/clock_gemm.py): per-workgroup start / end on the 100 MHz
    // wall clock and the shader clock, to a buffer nothing else reads
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (half < 0) {
        gemm_tile<GG_BM, BN_LAYOUT, EPI, EXT, KTAIL>(g, smem, bucket, row_lo, row_hi, col_tile);
    } else {
        row_lo += (GG_BM / 2) * half;
        row_hi = min(row_hi, row_lo + GG_BM / 2);
        if (row_lo < row_hi)    // (a ragged tile may have no rows in its second half)
            gemm_tile<GG_BM / 2, BN_LAYOUT, EPI, EXT, KTAIL>(g, smem, bucket, row_lo, row_hi,
                                                             col_tile);
    }
#ifdef GG_STAMP
    if (EPI == EPI_PLAIN && g.pdot_main && threadIdx.x == 0) {
        unsigned long long *st = reinterpret_cast<unsigned long long *>(g.pdot_main);
        st[12 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime() - st_t0;
        st[12 * blockIdx.x + 1] = st_r0;
        st[12 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
        st[12 * blockIdx.x + 3] = (unsigned long long)(half + 1);
    }
#endif
}

// y[t, :] = bias + sum_{j < k} rows[pos[t * k + j], :]  -- the un-bucketing of the routed
// FFN (reference: `y[mask] += ...` per block, lora_ffn.py:107-111): a gather in a fixed
// order instead of a scatter-add, so the result is deterministic.
__global__ __launch_bounds__(256) void rows_combine_kernel(
    const float *__restrict__ rows, const int32_t *__restrict__ pos,
    const float *__restrict__ bias, float *__restrict__ out, int n_tokens, int k, int d4) {
    const int t = blockIdx.x;
    for (int c = threadIdx.x; c < d4; c += 256) {
        float4 acc = bias ? reinterpret_cast<const float4 *>(bias)[c]
                          : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < k; j++) {
            const int p = pos[(size_t)t * k + j];
            const float4 v = reinterpret_cast<const float4 *>(rows + (size_t)p * d4 * 4)[c];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4 *>(out + (size_t)t * d4 * 4)[c] = acc;
    }
}

}  // namespace spt

using namespace spt;

// Workgroups per round: GG_SLOTS_PER_CU per CU (three fit: 166 VGPRs, 40 KiB LDS; two measured faster).
static int resident_slots() {
    static int slots = 0;
    if (slots == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return -1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return -1;
        slots = GG_SLOTS_PER_CU * cus;
    }
    return slots;
}

static int launch_grouped(GroupedArgs g, int epilogue, void *stream) {
    if (!g.a || !g.w || !g.offsets || !g.out) return SPT_EINVAL;
    if (g.P <= 0 || g.K <= 0 || g.N <= 0 || g.G <= 0 || g.lda < g.K) return SPT_EINVAL;
    if (g.K % 4 != 0 || g.lda % 4 != 0) return SPT_ESHAPE;       // float4 rows of A
    if (g.ldk != 1 && g.ldn != 1) return SPT_EUNSUP;
    if (g.ldk == 1 && (g.ldn % 4 != 0 || g.gstride % 4 != 0)) return SPT_ESHAPE;
    if (g.ldk != 1 && (g.ldk % 4 != 0 || g.N % 4 != 0 || g.gstride % 4 != 0)) return SPT_ESHAPE;
    const bool ext = g.a2 != nullptr;
    if (ext) {
        if (!g.b2 || g.R <= 0 || g.R > GG_BK) return SPT_EINVAL;
        if (g.R % 4 != 0 || g.lda2 % 4 != 0 || g.lda2 < g.R || g.b2_ldn % 4 != 0 ||
            g.b2_gstride % 4 != 0)
            return SPT_ESHAPE;
    }
    if (epilogue < EPI_PLAIN || epilogue > EPI_DACT) return SPT_EINVAL;
    if (epilogue != EPI_PLAIN && (g.act < ACT_RELU || g.act > ACT_SILU)) return SPT_EUNSUP;
    const unsigned row_tiles = (unsigned)((g.P + GG_BM - 1) / GG_BM + g.G);
    const unsigned col_tiles = (unsigned)((g.N + GG_BN - 1) / GG_BN);
    if ((unsigned long long)row_tiles * col_tiles > 0x3FFFFFFFull) return SPT_EUNSUP;
    if (epilogue == EPI_DACT) {
        if ((!g.h_in && !g.s_in) || !g.pdot_main || !g.pdot_act) return SPT_EINVAL;
        if (!g.s_in && g.act != ACT_RELU) return SPT_EINVAL;   // only ReLU is a function of h
        if (g.pdot_ld < (int)(2 * col_tiles)) return SPT_ESHAPE;
    }
    // worst case: every tile in the halved last round
    dim3 grid(2 * row_tiles * col_tiles);
    hipStream_t s = (hipStream_t)stream;
    g.slots = resident_slots();
    if (g.slots <= 0) return SPT_EINVAL;
#define SPT_GG(BN, EPI, EXT)                                                                  \
    do {                                                                                      \
        if (k_tail)                                                                           \
            hipLaunchKernelGGL((grouped_gemm_kernel<BN, EPI, EXT, true>), grid,               \
                               dim3(GG_THREADS), 0, s, g);                                    \
        else                                                                                  \
            hipLaunchKernelGGL((grouped_gemm_kernel<BN, EPI, EXT, false>), grid,              \
                               dim3(GG_THREADS), 0, s, g);                                    \
    } while (0)
#define SPT_GG_EPI(BN, EXT)                                   \
    do {                                                      \
        if (epilogue == EPI_PLAIN) SPT_GG(BN, EPI_PLAIN, EXT); \
        else if (epilogue == EPI_ACT) SPT_GG(BN, EPI_ACT, EXT); \
        else SPT_GG(BN, EPI_DACT, EXT);                        \
    } while (0)
    const bool k_tail = (g.K % GG_BK) != 0;
    if (g.ldk == 1) {
        if (ext) SPT_GG_EPI(false, true); else SPT_GG_EPI(false, false);
    } else {
        if (ext) SPT_GG_EPI(true, true); else SPT_GG_EPI(true, false);
    }
#undef SPT_GG_EPI
#undef SPT_GG
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_grouped_gemm(const float *a, const int32_t *gather, const float *w,
                                const float *bias, const float *rowscale,
                                const int32_t *offsets, float *out, int n_rows, int k, int n,
                                int n_groups, int lda, long long w_group_stride, int w_ldn,
                                int w_ldk, void *stream) {
    GroupedArgs g = {};
    g.a = a; g.gather = gather; g.w = w; g.bias = bias; g.rowscale = rowscale;
    g.offsets = offsets; g.out = out;
    g.P = n_rows; g.K = k; g.N = n; g.G = n_groups; g.lda = lda;
    g.gstride = w_group_stride; g.ldn = w_ldn; g.ldk = w_ldk;
    return launch_grouped(g, EPI_PLAIN, stream);
}

extern "C" int spt_grouped_gemm_pdot_width(int n) { return n > 0 ? 2 * ((n + GG_BN - 1) / GG_BN) : 0; }

extern "C" int spt_grouped_gemm_fused(const SptGroupedGemm *d, void *stream) {
    if (!d) return SPT_EINVAL;
    GroupedArgs g = {};
    g.a = d->a; g.gather = d->gather; g.w = d->w; g.bias = d->bias; g.rowscale = d->rowscale;
    g.offsets = d->offsets; g.out = d->out;
    g.P = d->n_rows; g.K = d->k; g.N = d->n; g.G = d->n_groups; g.lda = d->lda;
    g.gstride = d->w_group_stride; g.ldn = d->w_ldn; g.ldk = d->w_ldk;
    g.a2 = d->a2; g.gather2 = d->gather2; g.b2 = d->b2; g.lda2 = d->lda2; g.R = d->r;
    g.b2_gstride = d->b2_group_stride; g.b2_ldn = d->b2_ldn;
    g.act = d->activation; g.out2 = d->out2; g.h_in = d->h_in; g.s_in = d->s_in;
    g.pdot_main = d->pdot_main; g.pdot_act = d->pdot_act; g.pdot_ld = d->pdot_ld;
    return launch_grouped(g, d->epilogue, stream);
}

extern "C" int spt_rows_combine(const float *rows, const int32_t *pos, const float *bias,
                                float *out, int n_tokens, int k, int d, void *stream) {
    if (!rows || !pos || !out) return SPT_EINVAL;
    if (n_tokens <= 0 || k <= 0 || d <= 0) return SPT_EINVAL;
    if (d % 4 != 0) return SPT_ESHAPE;
    hipLaunchKernelGGL(rows_combine_kernel, dim3((unsigned)n_tokens), dim3(256), 0,
                       (hipStream_t)stream, rows, pos, bias, out, n_tokens, k, d / 4);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
