// grouped_gemm.hip -- token-bucketed grouped GEMM for the routed FFN and every frozen LoRA
// linear: fp32 semantics on the bf16 matrix cores.
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
// Arithmetic: an fp32 operand is split x = hi + lo (hi = RNE bf16 of x, lo = RNE bf16 of
// x - hi) and a product is three v_mfma_f32_32x32x16_bf16 (lo.hi + hi.lo + hi.hi, fp32
// accumulation): <= 2^-16 relative error per product at 1/5 of the time of the fp32 MFMA.
// One place cannot live with 2^-16: the pre-activation of a ReLU that lies within the error of
// zero (its derivative flips: an O(1) error in one token's gradients).  There the epilogue
// recomputes the element in plain fp32 (gg_epilogue: "near the kink").
//
// Two operand paths, one tile shape (128 x 128 per 256-thread workgroup, each wave a 64 x 64
// quadrant = 2 x 2 MFMA tiles, K in steps of 32), one epilogue:
//
//   * IMAGE path (gemm_tile_img): both operands arrive PRE-SPLIT as bf16 images in global
//     memory (spt_split_bf16: [row][k / 32][hi | lo][32], 128 contiguous bytes per row and
//     k-step) and go global -> LDS with global_load_lds_dwordx4, no registers, no VALU, no
//     ds_write in the k-loop; two LDS stages, the next step's 32 KiB in flight while the
//     current one is contracted, one barrier per k-step.  LDS tiles are conflict-free by an
//     XOR swizzle applied on the SOURCE address of each lane (the LDS side of an LDS-DMA is
//     lane-linear) and again on the fragment reads.
//   * REGISTER path (gemm_tile_regs): fp32 operands, split while they are staged
//     (global -> VGPR -> cvt -> ds_write).  Round 1's kernel; kept for callers without images
//     and for K % 32 != 0.
//
// Why the image path: per-phase shader-clock stamps of the register path
// had a k-step at 4390 cycles of which 1355 were the split + 16 ds_write_b64 per lane and
// 768 matrix-pipe time: the VGPR -> LDS store path, not the matrix cores, set the pace.
#include "spt_common.h"
#include <stdlib.h>

namespace spt {

constexpr int GG_THREADS = 256;
constexpr int GG_BM = 128;
constexpr int GG_BN = 128;
constexpr int GG_BK = 32;
// workgroups per CU the round arithmetic of the kernels counts on (both paths: two fit)
constexpr int GG_SLOTS_PER_CU = 2;
// behind the operand tiles either kernel's LDS array carries four rows of 128 floats: for the
// ReLU epilogue |a2 row|^2 and |b2 row|^2 of the K extension's operands (written while they
// are staged) and rowscale * |a row| of the tile's rows, and rowscale of the tile's rows itself
// (1 without one) -- both loaded in the prologue: the scaling after the k-loop and every row of the
// epilogue read them from LDS (as global loads inside the epilogue's row loop each was a
// memory latency in the open, eight per 32 rows)
constexpr int GG_EXTRAS = 4 * 128;
// The ReLU queue is cut into segments with a counter each (16 words apart: a line of their own),
// a workgroup appends to segment blockIdx % GG_FIX_SEGS: ONE returning atomic counter takes
// ~88 increments / us from the whole chip (MI355X_MICROARCH.md, "dequeue"), and a GEMM queues
// ~10,000 elements -- measured 105 us on one counter.
constexpr int GG_FIX_SEGS = 256;

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((ext_vector_type(8))) __bf16 gg_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 gg_bf16x2;
typedef __attribute__((ext_vector_type(2))) float gg_f32x2;
typedef short gg_v4s16 __attribute__((ext_vector_type(4)));

// ---- splits (whole-vector conversions: one v_cvt_pk_bf16_f32 per pair) ----
__device__ __forceinline__ void gg_split2(float a, float b, unsigned &hi, unsigned &lo) {
    const gg_f32x2 x = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, gg_bf16x2));
    const gg_f32x2 hf = {__builtin_bit_cast(float, hi << 16),
                         __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(x - hf, gg_bf16x2));
}
__device__ __forceinline__ uint2 gg_tr_b64(const char *p) {      // ds_read_b64_tr_b16
    const gg_v4s16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) gg_v4s16 *)(p));
    return __builtin_bit_cast(uint2, r);
}
__device__ __forceinline__ f32x16 gg_mma(const uint4 &a, const uint4 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gg_bf16x8, a),
                                                   __builtin_bit_cast(gg_bf16x8, b), c, 0, 0, 0);
}
struct GgFrag { uint4 hi, lo; };
__device__ __forceinline__ f32x16 gg_mma3(const GgFrag &a, const GgFrag &b, f32x16 c) {
    c = gg_mma(a.lo, b.hi, c);      // small terms first
    c = gg_mma(a.hi, b.lo, c);
    return gg_mma(a.hi, b.hi, c);
}
enum { EPI_PLAIN = 0, EPI_ACT = 1, EPI_DACT = 2 };
enum { ACT_RELU = 0, ACT_GELU = 1, ACT_SILU = 2 };

struct GroupedArgs {
    const float *a;         // [*, K] row-major, leading dimension lda (register path)
    const int32_t *gather;  // [P] or null
    const float *w;
    const float *bias;      // [G, N] or null
    const float *rowscale;  // [P] or null
    const int32_t *offsets; // [G + 1]
    float *out;             // [P, N] row-major
    int P, K, N, G;
    int lda;
    // K in segments (register path): A(row, kk) = a[(kk / a_seg_k) * a_seg_stride + row * lda + kk % a_seg_k]
    // -- the k-loop walks several matrices [*, a_seg_k] that lie a_seg_stride floats apart (the sum
    // of the q / k / v products of a backward as ONE contraction); 0: one matrix [*, K]
    int a_seg_k;
    long long a_seg_stride;
    long long ldo;          // row stride of out / out2 / h_in / s_in (>= N; launch_grouped: 0 -> N)
    int accumulate;         // EPI_PLAIN: out += result (the sum of several products in one buffer)
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
    long long b2_seg;       // != 0: column j of b2 lies in segment j / 16, b2_seg floats apart (register path)
    // epilogue
    int act;
    float *out2;            // EPI_ACT: pre-activation (null: not kept, e.g. ReLU)
    const float *h_in;      // EPI_DACT: activated values [P, N] (used when s_in is null: ReLU)
    const float *s_in;      // EPI_DACT: pre-activation values [P, N] or null
    float *pdot_main;       // EPI_DACT: [P, pdot_ld]: sum_n v[p, n] * h[p, n], v = the value before act'
    float *pdot_act;        // EPI_DACT: [P, pdot_ld]: sum_n out[p, n] * s[p, n] per half tile
    int pdot_ld;
    // EPI_ACT with ReLU: row norms of a [rows of a] and of the weight [G, N] (upper bounds do),
    // from which the epilogue tells which pre-activations lie within the split's error of zero
    const float *a_norm;
    const float *w_norm;
    unsigned *fix_count;    // ... and where their (row, column) pairs are queued for
    int2 *fix_list;         // relu_fix_kernel (null / full: recomputed in the epilogue itself):
    int fix_cap;            // GG_FIX_SEGS segments of fix_cap entries, one counter each
    int slots;              // workgroups resident at a time (CUs x occupancy)
    // image path: pre-split operands (spt_split_bf16), byte strides of one image row
    const char *a_img;
    const char *w_img;
    long long a_rowb, w_rowb;
    long long w_grow;       // image rows / 128-byte blocks from W_g(0, 0) to W_{g+1}(0, 0)
    int w_gblk;
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

// v[p, n] as a plain fp32 computation by one wave: rowscale (sum_k a[src, k] W_g(n, k) + bias)
// + sum_j a2[src2, j] b2[n, j], every lane 4 k's per pass, butterfly sum.  All lanes return it.
// BATCHED: the k-contiguous weights' four passes of loads in flight together (relu_fix_kernel; the
// epilogue's own rare fallback keeps the compact loop: inlined there the batched form cost the
// activation GEMM registers)
template <bool BATCHED = false>
__device__ __forceinline__ float gg_exact_preact(const GroupedArgs &g, int bucket, int p, int n,
                                                 int lane) {
    const long long src = g.gather ? g.gather[p] : p;
    const float *ar = g.a + src * g.lda;
    const float *wr = g.w + (size_t)bucket * g.gstride + (size_t)n * g.ldn;
    float part = 0.0f;
    if (BATCHED && g.ldk == 1) {
        // four passes' loads in flight together (clamped addresses, zeroed past K): as one load pair
        // per pass an element cost the wave four dependent round trips -- relu_fix_kernel took 22 us
        // for ~40 elements per segment.  Same order of additions as the plain loop below.
        for (int k0 = 4 * lane; k0 < g.K; k0 += 1024) {
            float4 x[4], y[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int k = min(k0 + 256 * i, g.K - 4);
                x[i] = *reinterpret_cast<const float4 *>(ar + k);
                y[i] = *reinterpret_cast<const float4 *>(wr + k);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (k0 + 256 * i < g.K) {
                    part = fmaf(x[i].x, y[i].x, part); part = fmaf(x[i].y, y[i].y, part);
                    part = fmaf(x[i].z, y[i].z, part); part = fmaf(x[i].w, y[i].w, part);
                }
            }
        }
    } else {
        for (int k = 4 * lane; k < g.K; k += 256) {
            const float4 x = *reinterpret_cast<const float4 *>(ar + k);
            float4 y;
            if (g.ldk == 1) {
                y = *reinterpret_cast<const float4 *>(wr + k);
            } else {            // n-contiguous weights: a strided column
                y.x = wr[(size_t)k * g.ldk]; y.y = wr[(size_t)(k + 1) * g.ldk];
                y.z = wr[(size_t)(k + 2) * g.ldk]; y.w = wr[(size_t)(k + 3) * g.ldk];
            }
            part = fmaf(x.x, y.x, part); part = fmaf(x.y, y.y, part);
            part = fmaf(x.z, y.z, part); part = fmaf(x.w, y.w, part);
        }
    }
    float side = 0.0f;
    if (g.a2 && 4 * lane < g.R) {
        const long long src2 = g.gather2 ? g.gather2[p] : p;
        const float4 x = *reinterpret_cast<const float4 *>(g.a2 + src2 * g.lda2 + 4 * lane);
        const float4 y = *reinterpret_cast<const float4 *>(
            g.b2 + (size_t)bucket * g.b2_gstride + (size_t)n * g.b2_ldn + 4 * lane);
        side = fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, x.w * y.w)));
    }
    part = group_sum<64>(part);
    side = group_sum<64>(side);
    const float bb = g.bias ? g.bias[(size_t)bucket * g.N + n] : 0.0f;
    const float rr = g.rowscale ? g.rowscale[p] : 1.0f;
    return fmaf(rr, part + bb, side);
}

// acc = rowscale * acc, in the MFMA C layout (row = 32 i + (r & 3) + 8 (r >> 2) + 4 (l >> 5)).
// Runs between the k-loop and the K extension: nothing is ever divided by rowscale (a router
// coefficient may be 0: 2 sigmoid(logit) underflows below logit -104).
template <int NI>
__device__ __forceinline__ void gg_scale_rows(const GroupedArgs &g, f32x16 (&acc)[NI][2],
                                              const float *extras, int wm) {
    if (!g.rowscale) return;
    const int fh = (threadIdx.x & 63) >> 5;
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float rs = extras[384 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh];
            acc[i][0][r] *= rs;
            acc[i][1][r] *= rs;
        }
}

// Epilogue of a (32 NI x 2) x 128 tile.  MFMA C layout: acc[i][j][r] =
// C[32 i + (r & 3) + 8 (r >> 2) + 4 (l >> 5)][32 j + (l & 31)] of the wave's (32 NI) x 64
// part.  Each wave transposes it through LDS, 32 rows at a time, into a row layout -- 16
// lanes x float4 = one 64-column row segment -- so that every global access (the stores, the
// h / s tiles of EPI_DACT, the bias) is 16 bytes per lane and 256 contiguous bytes per row,
// and a row dot is a 16-lane DPP reduction.  The caller has synchronised the workgroup: the
// operand tiles in `smem` are dead.
template <int NI, int EPI>
__device__ __forceinline__ void gg_epilogue(const GroupedArgs &g, float *smem,
                                            f32x16 (&acc)[NI][2], int bucket, int row_lo,
                                            int row_hi, int col_tile, int wm, int wn,
                                            const float *extras, bool has_ext) {
    constexpr int CS_ROW = 64 + 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *cs = smem + wave * (32 * CS_ROW);
    const bool relu_exact = EPI == EPI_ACT && g.act == ACT_RELU && g.a_norm != nullptr;
    const int ccol = lane & 31, chalf = lane >> 5;
    const int rrow = lane >> 4, rcol = 4 * (lane & 15);
    const int pslot = 2 * col_tile + (wave & 1);   // this wave's half tile of columns
    const float *sh = (EPI == EPI_DACT) ? (g.s_in ? g.s_in : g.h_in) : nullptr;
    const int n = col_tile * GG_BN + wn + rcol;
    const bool vec = (g.N & 3) == 0;
    const bool vec_row = (g.ldo & 3) == 0;        // 16-byte aligned pieces of the N-wide rows
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
    // ReLU epilogue: this lane's four columns' share of the error bound (see below)
    float wn4[4] = {0.f, 0.f, 0.f, 0.f}, rn4[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_ACT && relu_exact) {
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (n + e < g.N) {
                wn4[e] = g.w_norm[(size_t)bucket * g.N + n + e];
                rn4[e] = has_ext ? extras[128 + wn + rcol + e] : 0.0f;
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
        // (not unrolled: the body is long -- with the ReLU recomputation inlined eight times the
        // activation epilogues ran 30-45 us slower at the FFN shape; and requesting the saved
        // activations of all eight row groups of EPI_DACT ahead of an unrolled loop was measured
        // 15 us SLOWER than loading them inside it, 175 against 160 us)
        // EPI_DACT: the saved activations of row group t + 1 are requested while group t is worked on
        // (rows past the bucket end: clamped, never used)
        auto load_saved = [&](int t) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (EPI == EPI_DACT) {
                const int p = min(row_lo + wm + 32 * i + rrow + 4 * t, row_hi - 1);
                const size_t at = (size_t)p * g.ldo + n;
                if (vec_row && n + 3 < g.N) {
                    v = *reinterpret_cast<const float4 *>(sh + at);
                } else {
                    v.x = n + 0 < g.N ? sh[at + 0] : 0.f; v.y = n + 1 < g.N ? sh[at + 1] : 0.f;
                    v.z = n + 2 < g.N ? sh[at + 2] : 0.f; v.w = n + 3 < g.N ? sh[at + 3] : 0.f;
                }
            }
            return v;
        };
        float4 sv_next = load_saved(0);
#pragma unroll 1
        for (int t = 0; t < 8; t++) {
            const int row = rrow + 4 * t;
            const int p = row_lo + wm + 32 * i + row;
            const bool live = p < row_hi;
            const float4 c4 = *reinterpret_cast<const float4 *>(&cs[row * CS_ROW + rcol]);
            float c[4] = {c4.x, c4.y, c4.z, c4.w};
            const float b[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
            const float rs = extras[384 + wm + 32 * i + row];
            const size_t at = (size_t)p * g.ldo + n;
            const float sv[4] = {sv_next.x, sv_next.y, sv_next.z, sv_next.w};
            if (EPI == EPI_DACT && t < 7) sv_next = load_saved(t + 1);
            float dot_h = 0.0f, dot_s = 0.0f;
            float pre[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float v = fmaf(rs, b[e], c[e]);         // the accumulators are scaled already
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
            if (EPI == EPI_ACT && relu_exact) {
                // ---- near the kink: |v| within the error bound of the split products ----
                // per product <= 2^-16 |a| |b| (hi.hi + hi.lo + lo.hi of RNE bf16 parts), so
                // |error of v| <= 2^-16 (rs |a_p| |w_n| + |a2_p| |b2_n|) by Cauchy-Schwarz; a
                // pre-activation inside 1.5 x that bound is recomputed as a plain fp32 dot product
                // by a whole wave (6e-4 of the elements at K = 1024: ~10 per tile)
                unsigned cand = 0;
                if (live) {
                    const float an = extras[256 + wm + 32 * i + row];      // rowscale |a row|
                    const float un2 = has_ext ? extras[wm + 32 * i + row] : 0.0f;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float bound = 2.2888e-5f * (an * wn4[e] + sqrtf(un2 * rn4[e]));
                        cand |= (n + e < g.N && fabsf(pre[e]) <= bound ? 1u : 0u) << e;
                    }
                }
                // queue them for relu_fix_kernel (one wave per element, all in parallel) ...
                if (cand && g.fix_list) {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (cand & (1u << e)) {
                            const unsigned seg = blockIdx.x % GG_FIX_SEGS;
                            const unsigned slot = atomicAdd(g.fix_count + 16 * seg, 1u);
                            if (slot < (unsigned)g.fix_cap) {
                                g.fix_list[(size_t)seg * g.fix_cap + slot] =
                                    make_int2(p, (n + e) | (bucket << 24));
                                cand &= ~(1u << e);
                            }
                        }
                }
                // ... or, without a queue (or with a full one), recompute here, wave by wave
                unsigned long long todo = __ballot(cand != 0);
                while (todo) {
                    const int owner = __ffsll(todo) - 1;
                    const unsigned m = __shfl(cand, owner, 64);
                    const int e = __ffs(m) - 1;
                    const float exact = gg_exact_preact(g, bucket, __shfl(p, owner, 64),
                                                        __shfl(n, owner, 64) + e, lane);
                    if (lane == owner) {
#pragma unroll
                        for (int ee = 0; ee < 4; ee++)
                            if (ee == e) {
                                pre[ee] = exact;
                                c[ee] = act_forward(g.act, exact);
                            }
                        cand &= ~(1u << e);
                    }
                    todo = __ballot(cand != 0);
                }
            }
            if (EPI == EPI_PLAIN && g.accumulate && live) {
                if (vec_row && n + 3 < g.N) {
                    const float4 o4 = *reinterpret_cast<const float4 *>(g.out + at);
                    c[0] += o4.x; c[1] += o4.y; c[2] += o4.z; c[3] += o4.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (n + e < g.N) c[e] += g.out[at + e];
                }
            }
            if (live) {
                if (vec_row && n + 3 < g.N) {
                    {
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

// =============================================================================== image path
// Global image of a [rows, cols] fp32 matrix (spt_split_bf16): row-major, every 32 columns
// one 128-byte block [hi: 32 bf16 | lo: 32 bf16].  One k-step of a k-contiguous operand tile
// is then ONE full 128-byte line per row, and a k-row of an n-contiguous weight tile
// (128 n) is 512 contiguous bytes.
//
// LDS tiles (16 KiB each, two stages of A | B = 64 KiB per workgroup):
//   KC  [rows][128 B]  k-contiguous operands (A; B of "BT" weights; the K extension).
//       The 16-byte chunk c = 4 part + (k / 8) of row r sits at chunk c ^ ((r >> 1) & 7):
//       a ds_read_b128 lane group (16 rows, one chunk each) then covers all 64 banks.
//   NC  [32 k][512 B]  n-contiguous weights ("BN"), read with ds_read_b64_tr_b16.
//       The 64-byte segment s = 2 (n / 32) + part of k-row k sits at segment s ^ (k & 3):
//       the four k-rows of a transposing read fall on four different 64-byte bank groups.
// An LDS-DMA writes base + 16 lane, so the swizzle is applied to the SOURCE: the lane that
// fills physical chunk pc of row r fetches logical chunk pc ^ swz(r) of that row.
constexpr int GI_TILE = 16384;                  // bytes of one operand tile
constexpr int GI_STAGE = 2 * GI_TILE;           // A | B
constexpr int GI_LDS_FLOATS = 2 * GI_STAGE / 4; // two stages

__device__ __forceinline__ void gg_glds16(const char *src, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)src,
        (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// Per-lane state of the image path's k-loop.
template <int NI, int NA>
struct GiLane {
    const char *a_src[NA];      // LDS-DMA sources of this lane, k-step 0
    const char *b_src[4];
    long long w_rowb;
    int wave, wm, wn, frow, fh, lane;
};

// fragments: lane l holds A[row l & 31][k = 16 q2 + 8 (l >> 5) + 0..7] and
// B[k = same][col l & 31], 16 bytes per part
__device__ __forceinline__ uint4 gi_frag_kc(const char *tile, int row, int q2, int part, int fh) {
    const int c = 4 * part + 2 * q2 + fh;
    return *reinterpret_cast<const uint4 *>(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
}
__device__ __forceinline__ uint4 gi_frag_nc(const char *tile, int ncol0, int q2, int part, int fh,
                                            int lane) {
    // lane 4 q + p of a 16-lane group supplies k-row q, columns 4 p .. 4 p + 3 of the
    // group's 16 n; lane i receives column i of the four k-rows: two blocks = 8 k's
    const int gl = lane & 15, q = gl >> 2, p = gl & 3, nhalf = (lane >> 4) & 1;
    const int seg = 2 * (ncol0 >> 5) + part;
    const char *ptr = tile + (16 * q2 + 8 * fh + q) * 512 + ((seg ^ q) << 6) + nhalf * 32 + p * 8;
    const uint2 lo = gg_tr_b64(ptr), hi = gg_tr_b64(ptr + 4 * 512);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
// The same fragment of an fp32 tile [row][32 k floats] (the A32 form: the activation goes
// global -> LDS by LDS-DMA as it is, 128 bytes per row and k-step like an image row, and is split
// HERE, on its way into the matrix cores): the lane's 8 floats are chunks 4 q2 + 2 fh and + 1 of
// its row, same swizzle, two conflict-free ds_read_b128; 24 VALU per fragment (96 per k-step and
// wave at NI = 2, under 24 MFMAs).  What it buys: no pre-split image of the activation -- no
// split pass, no second copy of the activation in memory, and every GEMM (the backward's dX
// products too, whose operand nobody else reads) takes the LDS-DMA k-loop.
__device__ __forceinline__ GgFrag gi_frag_a32(const char *tile, int row, int q2, int fh) {
    const int c0 = 4 * q2 + 2 * fh, sw = (row >> 1) & 7;
    const float4 x = *reinterpret_cast<const float4 *>(tile + row * 128 + ((c0 ^ sw) << 4));
    const float4 y = *reinterpret_cast<const float4 *>(tile + row * 128 + (((c0 | 1) ^ sw) << 4));
    GgFrag f;
    gg_split2(x.x, x.y, f.hi.x, f.lo.x);
    gg_split2(x.z, x.w, f.hi.y, f.lo.y);
    gg_split2(y.x, y.y, f.hi.z, f.lo.z);
    gg_split2(y.z, y.w, f.hi.w, f.lo.w);
    return f;
}
template <int NI>
struct GiFrags {
    GgFrag a[NI], b[2];
};
template <int NI, int NA, bool KC_B, bool A32 = false>
__device__ __forceinline__ void gi_read(const GiLane<NI, NA> &c, const char *As, const char *Bs,
                                        int q2, GiFrags<NI> &f) {
#pragma unroll
    for (int i = 0; i < NI; i++) {
        if constexpr (A32) {
            f.a[i] = gi_frag_a32(As, c.wm + 32 * i + c.frow, q2, c.fh);
        } else {
            f.a[i].hi = gi_frag_kc(As, c.wm + 32 * i + c.frow, q2, 0, c.fh);
            f.a[i].lo = gi_frag_kc(As, c.wm + 32 * i + c.frow, q2, 1, c.fh);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        if constexpr (KC_B) {
            f.b[j].hi = gi_frag_kc(Bs, c.wn + 32 * j + c.frow, q2, 0, c.fh);
            f.b[j].lo = gi_frag_kc(Bs, c.wn + 32 * j + c.frow, q2, 1, c.fh);
        } else {
            f.b[j].hi = gi_frag_nc(Bs, c.wn + 32 * j, q2, 0, c.fh, c.lane);
            f.b[j].lo = gi_frag_nc(Bs, c.wn + 32 * j, q2, 1, c.fh, c.lane);
        }
    }
}
template <int NI>
__device__ __forceinline__ void gi_mma(const GiFrags<NI> &f, f32x16 (&acc)[NI][2]) {
#pragma unroll
    for (int i = 0; i < NI; i++) {
        acc[i][0] = gg_mma3(f.a[i], f.b[0], acc[i][0]);
        acc[i][1] = gg_mma3(f.a[i], f.b[1], acc[i][1]);
    }
}
template <int NI, int NA, bool KC_B>
__device__ __forceinline__ void gi_contract(const GiLane<NI, NA> &c, const char *As, const char *Bs,
                                            int q2, f32x16 (&acc)[NI][2]) {
    GiFrags<NI> f;
    gi_read<NI, NA, KC_B>(c, As, Bs, q2, f);
    gi_mma<NI>(f, acc);
}
// k-step k0 .. k0 + 31 into the LDS stage at `dma` (instruction t = wave + 4 j: 1 KiB each)
template <int NI, int NA, bool BN_LAYOUT>
__device__ __forceinline__ void gi_stage(const GiLane<NI, NA> &c, char *dma, int k0) {
    const size_t ka = (size_t)(k0 >> 5) * 128;
    // (ablation builds, tools/variant.sh: wrong results, for timing -- which operand's stream
    // the k-loop waits for)
#ifdef GG_ABL_NO_A
    if (k0 < 64)
#endif
#pragma unroll
    for (int j = 0; j < NA; j++) gg_glds16(c.a_src[j] + ka, dma + (c.wave + 4 * j) * 1024);
#ifdef GG_ABL_NO_B
    if (k0 < 64)
#endif
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if constexpr (!BN_LAYOUT)
            gg_glds16(c.b_src[j] + ka, dma + GI_TILE + (c.wave + 4 * j) * 1024);
        else
            gg_glds16(c.b_src[j] + (size_t)k0 * c.w_rowb, dma + GI_TILE + (c.wave + 4 * j) * 1024);
    }
}
// One k-step: issue the LDS-DMAs of the NEXT step into `dma`, contract the stage at `cur`.
// The two pointers are __restrict__ on purpose: inlined, they give the DMA stores and the
// ds_reads alias scopes, and only with those does hipcc (ROCm 7.2) leave out the
// `s_waitcnt vmcnt(0)` it otherwise puts in front of the first ds_read behind an LDS-DMA in
// flight -- which would wait for the next tile before the current one is contracted.
template <int NI, int NA, bool BN_LAYOUT, bool A32>
__device__ __forceinline__ void gi_step(const GiLane<NI, NA> &c, char *__restrict__ dma,
                                        const char *__restrict__ cur, bool prefetch, int k_next,
                                        f32x16 (&acc)[NI][2]
#ifdef GG_STAMP_DMA
                                        // (diagnostic build, tools/micro/gemm_stamps.py: cycles of a step
                                        // spent issuing its LDS-DMAs / reading fragments / in its MFMAs)
                                        , unsigned long long (&st)[4]
#endif
                                        ) {
    // The order is pinned: the fragments of the step's second half are requested BEFORE the
    // MFMAs of the first, so that their LDS latency passes behind 12 MFMAs.  The wait for the
    // first half stands in front of those reads -- behind them hipcc makes it lgkmcnt(0) (no
    // counted wait is available to it while an LDS-DMA is pending), a wait for the reads just
    // issued.  Left alone, hipcc also sinks each group of reads to its first use and hoists the
    // barrier that follows (with its vmcnt(0), the wait for the NEXT tile) above half the MFMAs.
    GiFrags<NI> f0, f1;
#ifdef GG_STAMP_DMA
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    if (prefetch) gi_stage<NI, NA, BN_LAYOUT>(c, dma, k_next);
#ifdef GG_STAMP_DMA
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
    gi_read<NI, NA, !BN_LAYOUT, A32>(c, cur, cur + GI_TILE, 0, f0);
    __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0) only
    __builtin_amdgcn_sched_barrier(0);
    gi_read<NI, NA, !BN_LAYOUT, A32>(c, cur, cur + GI_TILE, 1, f1);
    __builtin_amdgcn_sched_barrier(0);
#ifdef GG_STAMP_DMA
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
    gi_mma<NI>(f0, acc);
    __builtin_amdgcn_sched_barrier(0);
    gi_mma<NI>(f1, acc);
    __builtin_amdgcn_sched_barrier(0);
#ifdef GG_STAMP_DMA
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    st[0] += t1 - t0; st[1] += t2 - t1; st[2] += t3 - t2;
#endif
}

template <int BM, bool BN_LAYOUT, int EPI, bool EXT, bool A32>
__device__ __forceinline__ void gemm_tile_img(const GroupedArgs &g, float *smem, int bucket,
                                              int row_lo, int row_hi, int col_tile) {
    constexpr int NI = BM / 64;                 // 32-row sub-blocks per wave
    constexpr int NA = BM / 32;                 // LDS-DMA instructions per wave: A tile
#ifdef GG_STAMP_DMA
    const unsigned long long t_enter = __builtin_amdgcn_s_memtime();
#endif
    char *const lds0 = reinterpret_cast<char *>(smem);
    const int n0 = col_tile * GG_BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    GiLane<NI, NA> c;
    c.lane = lane;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.wm = (c.wave >> 1) * (BM / 2);            // the wave's origin inside the tile
    c.wn = (c.wave & 1) * 64;
    c.frow = lane & 31;
    c.fh = lane >> 5;
    c.w_rowb = g.w_rowb;
    const int wave = c.wave, wm = c.wm, wn = c.wn;

    float *const extras = smem + GI_LDS_FLOATS;
    if (tid < BM) {
        const int p = min(row_lo + tid, row_hi - 1);
        const float rs = g.rowscale ? g.rowscale[p] : 1.0f;
        extras[384 + tid] = rs;
        if (EPI == EPI_ACT && g.act == ACT_RELU && g.a_norm)
            extras[256 + tid] = g.a_norm[g.gather ? g.gather[p] : p] * rs;
    }
    // ---- LDS-DMA sources.  KC tile: instruction t = wave + 4 j covers rows 8 t .. 8 t + 7,
    // lane -> row 8 t + (lane >> 3), physical chunk lane & 7.  Rows past the bucket end and
    // columns past N are computed but never stored: clamped, not predicated. ----
#pragma unroll
    for (int j = 0; j < NA; j++) {
        const int r = 8 * (wave + 4 * j) + (lane >> 3);
        const int p = min(row_lo + r, row_hi - 1);
        const long long src = g.gather ? g.gather[p] : p;
        // (A32: the fp32 rows themselves, a_rowb = 4 lda; 32 k = 128 bytes per k-step either way)
        c.a_src[j] = (A32 ? reinterpret_cast<const char *>(g.a) : g.a_img) + src * g.a_rowb +
                     (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    const char *wg = g.w_img + (long long)bucket * g.w_grow * g.w_rowb + (size_t)bucket * g.w_gblk * 128;
    if constexpr (!BN_LAYOUT) {
        // W_g(n, k), k contiguous: image row = n, block = k / 32
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = 8 * (wave + 4 * j) + (lane >> 3);
            const long long n = min(n0 + r, g.N - 1);
            c.b_src[j] = wg + n * g.w_rowb + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
        }
    } else {
        // W_g(n, k), n contiguous: image row = k, block = n / 32.  NC tile: instruction t
        // covers k-rows 2 t, 2 t + 1; lane -> k-row 2 t + (lane >> 5), physical chunk lane & 31
        const int nblocks = (g.N + 31) >> 5;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int kr = 2 * (wave + 4 * j) + (lane >> 5);
            const int pchunk = lane & 31;
            const int seg = (pchunk >> 2) ^ (kr & 3);               // logical segment
            const int nb = min((n0 >> 5) + (seg >> 1), nblocks - 1);
            c.b_src[j] = wg + (long long)kr * g.w_rowb + (size_t)nb * 128 + (seg & 1) * 64 +
                         ((pchunk & 3) << 4);
        }
    }

    f32x16 acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // ---- k-loop: the LDS-DMAs of step t + 1 are in flight while step t is contracted; the
    // barrier at the end of a step (with the vmcnt(0) / lgkmcnt(0) hipcc puts in front of it)
    // both publishes stage (t + 1) & 1 and retires the reads of stage t & 1 ----
    gi_stage<NI, NA, BN_LAYOUT>(c, lds0, 0);
    __syncthreads();
#ifdef GG_STAMP_DMA
    unsigned long long st[4] = {0, 0, 0, 0};
    const unsigned long long loop0 = __builtin_amdgcn_s_memtime();
#define GG_ST , st
#define GG_BAR(x)                                                       \
    do {                                                                \
        const unsigned long long b0 = __builtin_amdgcn_s_memtime();     \
        __syncthreads();                                                \
        st[3] += __builtin_amdgcn_s_memtime() - b0;                     \
    } while (0)
#else
#define GG_ST
#define GG_BAR(x) __syncthreads()
#endif
    for (int k0 = 0; k0 < g.K; k0 += 2 * GG_BK) {
        gi_step<NI, NA, BN_LAYOUT, A32>(c, lds0 + GI_STAGE, lds0, k0 + GG_BK < g.K, k0 + GG_BK, acc GG_ST);
        GG_BAR(0);
        if (k0 + GG_BK < g.K) {
            gi_step<NI, NA, BN_LAYOUT, A32>(c, lds0, lds0 + GI_STAGE, k0 + 2 * GG_BK < g.K,
                                            k0 + 2 * GG_BK, acc GG_ST);
            GG_BAR(0);
        }
    }
#ifdef GG_STAMP_DMA
    const unsigned long long loop1 = __builtin_amdgcn_s_memtime();
#endif
#undef GG_ST
#undef GG_BAR

    gg_scale_rows<NI>(g, acc, extras, wm);

    // ---- K extension: fp32 a2 [*, R] and b2 [n][R] split while staged into two KC tiles
    // (R <= 32: one k-step, zero beyond R) ----
    if (EXT) {
        char *As = lds0, *Bs = lds0 + GI_TILE;
        const int kq = tid & 7, k = 4 * kq;                 // 8 lanes per row: 4 k's each
        auto put4 = [&](char *tile, int r, const float4 &v) {
            unsigned h0, l0, h1, l1;
            gg_split2(v.x, v.y, h0, l0);
            gg_split2(v.z, v.w, h1, l1);
            const int sw = (r >> 1) & 7;
            char *row = tile + r * 128 + (kq & 1) * 8;
            *reinterpret_cast<uint2 *>(row + (((kq >> 1) ^ sw) << 4)) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(row + (((4 + (kq >> 1)) ^ sw) << 4)) = make_uint2(l0, l1);
        };
#pragma unroll
        for (int u = 0; u < BM / 32; u++) {
            const int r = (tid >> 3) + 32 * u;
            const int p = row_lo + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < row_hi && k < g.R) {
                const long long src = g.gather2 ? g.gather2[p] : p;
                v = *reinterpret_cast<const float4 *>(g.a2 + src * g.lda2 + k);
            }
            put4(As, r, v);
            const float ss = group_sum<8>(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
            if (kq == 0) extras[r] = ss;                    // |a2 row|^2, for the epilogue
        }
#pragma unroll
        for (int u = 0; u < GG_BN / 32; u++) {
            const int r = (tid >> 3) + 32 * u;
            const int n = n0 + r;
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < g.N && k < g.R)
                b = *reinterpret_cast<const float4 *>(g.b2 + (size_t)bucket * g.b2_gstride +
                                                      (size_t)n * g.b2_ldn + k);
            put4(Bs, r, b);
            const float ss = group_sum<8>(b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w);
            if (kq == 0) extras[128 + r] = ss;              // |b2 row|^2
        }
        __syncthreads();
        gi_contract<NI, NA, true>(c, As, Bs, 0, acc);
        if (g.R > 16) gi_contract<NI, NA, true>(c, As, Bs, 1, acc);
        __syncthreads();
    }
#ifdef GG_ABL_NO_EPI
    if (acc[0][0][0] == 123.456f) g.out[0] = acc[0][1][3] + acc[NI - 1][0][5];
    return;
#endif
#ifdef GG_STAMP_DMA
    const unsigned long long epi0 = __builtin_amdgcn_s_memtime();
#endif
    gg_epilogue<NI, EPI>(g, smem, acc, bucket, row_lo, row_hi, col_tile, wm, wn, extras, EXT);
#ifdef GG_STAMP_DMA
    if (lane == 0) {
        // eight counters per wave behind the launch's output (the caller allocates the room):
        // the loop's four phases; loop; entry -> loop; loop -> epilogue; epilogue
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(g.out + (size_t)g.P * g.ldo) +
                                  ((size_t)blockIdx.x * 4 + wave) * 8;
        dst[0] = st[0]; dst[1] = st[1]; dst[2] = st[2]; dst[3] = st[3];
        dst[4] = loop1 - loop0; dst[5] = loop0 - t_enter; dst[6] = epi0 - loop1;
        dst[7] = __builtin_amdgcn_s_memtime() - epi0;
    }
#endif
}

struct GgWork {
    int bucket, row_lo, row_hi, col_tile, half;
};

// (Round 4 also built a persistent form of the kernel above whose finished tile waited in registers
// and was stored, a dword per lane straight from the accumulator layout, in the shadow of the next
// tile's MFMAs: 483 against 428 us at 4096 tiles, 56.2 against 55.0 ms in the step -- four times the
// store instructions of the LDS-transposed epilogue cost more than the overlap returns.  Not kept.)

// ======================================================= image path, loader / consumer ring (round 4)
// What a k-step of the kernels above costs a wave (a -DGG_STAMP_DMA build, tools/micro/gemm_stamps.py,
// 4096 tiles of K = N = 1024): 2,354 cycles = 930 issuing its eight LDS-DMAs (a global_load_lds
// costs its issuer ~120 cycles beside MFMAs and ds_reads; a wave that does nothing else issues one
// in 25-60: MI355X_MICROARCH.md, "LDS-DMA piece issue cost", "ldsdma-fill") + 250 reading fragments
// + 752 in its 24 MFMAs + 250 at the barrier: the waves that own the matrix pipe spend more time
// feeding LDS than multiplying.  Here the two jobs are different WAVES.  One workgroup of eight waves
// per CU, resident for the whole launch:
//   waves 4-7, the loaders: walk the workgroup's k-steps (across tile boundaries), fill a ring of
//     three 48 KiB stages (A 128 x 32 k | B 256 x 32 k, the layouts of the kernels above) by LDS-DMA,
//     twelve instructions per wave and step, and publish a stage -- FULL[slot] += 1 per wave, an
//     LDS word -- behind a counted vmcnt that leaves the next step's DMAs in flight; a slot is
//     refilled when FREE[slot] says the four consumers have its fragments in registers.  The LoRA K
//     extension is one more step of the tile whose stage the loaders fill with ds_writes (fp32
//     a2 / b2 split on the way), so the consumers see nothing special.
//   waves 0-3, the consumers: a 128 x 256 tile, each wave 64 x 128 (2 x 4 MFMA tiles, 128
//     accumulator registers, 48 MFMAs per k-step): half a step's fragments are read while the other
//     half's 24 MFMAs run; the next stage's FULL word is requested before an MFMA group and looked
//     at behind it.  Arithmetic intensity against L2 is 1.33 x the 128 x 128 tile's (48 KiB per
//     2 x 2 x 768 MFMA-cycles), the LDS reads per MFMA 0.5 instead of 0.67.
// Consumer w and loader w + 4 share a SIMD (a workgroup's waves go round the SIMDs), so the matrix
// pipe and the vector-memory issue port of a SIMD belong to different waves.
// Plain epilogue (bias, rowscale, K extension), stores straight from the accumulator layout by
// buffer stores.  The activation epilogues keep the kernels above.
constexpr int GR_THREADS = 512;
constexpr int GR_BM = 128, GR_BN = 256;
constexpr int GR_A_TILE = GR_BM * 128, GR_B_TILE = GR_BN * 128, GR_STAGE = GR_A_TILE + GR_B_TILE;
constexpr int GR_SLOTS = 3;
constexpr int GR_FLAGS = GR_SLOTS * GR_STAGE;       // FULL[s] at +4 s, FREE[s] at +32 + 4 s
constexpr int GR_RS = GR_FLAGS + 64;                // 4 consumer waves x 64 row scales
constexpr int GR_LDS = GR_RS + 4 * 64 * 4;          // 148,544 B
constexpr int gr_waitcnt(int vm, int lgkm) {        // s_waitcnt immediate (expcnt: no wait)
    return (vm & 0xF) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 0xF) << 8);
}
__device__ __forceinline__ unsigned gr_lds_addr(const void *p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}
// Flag words: plain LDS operations in asm, so that hipcc neither counts them nor drains an LDS-DMA
// in flight in front of them (cdna_hip_programming.md 5.7)
__device__ __forceinline__ unsigned gr_flag_read(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void gr_flag_add(unsigned addr) {
    const unsigned one = 1u;
    asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(one) : "memory");
}
// wave-uniform wait for flag >= need (counters only grow); a wait that lasts seconds means a broken
// protocol: trap (a launch failure the host sees) rather than hang the GPU
__device__ __forceinline__ void gr_wait(unsigned addr, unsigned need) {
    for (unsigned spin = 0;; spin++) {
        const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)gr_flag_read(addr));
        if ((int)(v - need) >= 0) return;
        __builtin_amdgcn_s_sleep(2);
        if (spin > (1u << 25)) __builtin_trap();
    }
}
// the tile list of a launch, item `id` (every wave of the workgroup walks the same list)
struct GrPlan { int n_col_tiles, total; };
struct GrTile { int bucket, row_lo, row_hi, n0; };
__device__ __forceinline__ GrPlan gr_plan(const GroupedArgs &g) {
    GrPlan pl;
    pl.n_col_tiles = (g.N + GR_BN - 1) / GR_BN;
    int row_tiles = 0;
    for (int i = 0; i < g.G; i++) row_tiles += (g.offsets[i + 1] - g.offsets[i] + GR_BM - 1) / GR_BM;
    pl.total = row_tiles * pl.n_col_tiles;
    return pl;
}
__device__ __forceinline__ GrTile gr_tile(const GroupedArgs &g, const GrPlan &pl, int id) {
    // (gridDim.x is a multiple of 8: id % 8 is the XCD of the workgroup that walks it)
    const int logical = (int)xcd_remap((unsigned)id, (unsigned)pl.total);
    GrTile t;
    t.bucket = -1;
    t.row_lo = t.row_hi = 0;
    t.n0 = (logical % pl.n_col_tiles) * GR_BN;
    int tile = logical / pl.n_col_tiles;
    for (int i = 0; i < g.G; i++) {
        const int lo = g.offsets[i], hi = g.offsets[i + 1];
        const int tiles = (hi - lo + GR_BM - 1) / GR_BM;
        if (tile < tiles) {
            t.bucket = i;
            t.row_lo = lo + tile * GR_BM;
            t.row_hi = min(hi, t.row_lo + GR_BM);
            break;
        }
        tile -= tiles;
    }
    return t;
}

// (diagnostic build -DGR_STAMP, tools/micro/gemm_stamps.py: where a loader / a consumer wave's cycles
// go; the sums are written behind the launch's output)
#ifdef GR_STAMP
#define GR_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#define GR_ACC(slot, a, b) st[slot] += (b) - (a)
#else
#define GR_T(x)
#define GR_ACC(slot, a, b)
#endif
template <bool BN_LAYOUT, bool EXT, bool A32>
__device__ __forceinline__ void gr_loader(const GroupedArgs &g, char *lds, int lw, int lane) {
#ifdef GR_STAMP
    unsigned long long st[4] = {0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    const GrPlan pl = gr_plan(g);
    const unsigned flags = gr_lds_addr(lds + GR_FLAGS);
    const int nk = g.K / GG_BK;
    unsigned t = 0;                                     // the workgroup's k-steps so far
    for (int id = blockIdx.x; id < pl.total; id += gridDim.x) {
        const GrTile tl = gr_tile(g, pl, id);
        if (tl.bucket < 0) continue;
        // ---- LDS-DMA sources of this wave: instruction lw + 4 j of a tile's 16 (A) / 32 (B) ----
        const char *a_src[4], *b_src[8];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = 8 * (lw + 4 * j) + (lane >> 3);
            const int p = min(tl.row_lo + r, tl.row_hi - 1);
            const long long src = g.gather ? g.gather[p] : p;
            a_src[j] = (A32 ? reinterpret_cast<const char *>(g.a) : g.a_img) + src * g.a_rowb +
                       (((lane & 7) ^ ((r >> 1) & 7)) << 4);
        }
        const char *wg = g.w_img + (long long)tl.bucket * g.w_grow * g.w_rowb + (size_t)tl.bucket * g.w_gblk * 128;
        if constexpr (!BN_LAYOUT) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = 8 * (lw + 4 * j) + (lane >> 3);
                const long long n = min(tl.n0 + r, g.N - 1);
                b_src[j] = wg + n * g.w_rowb + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
            }
        } else {
            // NC tile [32 k][1024 B]: instruction t = one k-row; lane = physical 16-byte chunk
            const int nblocks = (g.N + 31) >> 5;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int kr = lw + 4 * j;
                const int seg = (lane >> 2) ^ (kr & 3);
                const int nb = min((tl.n0 >> 5) + (seg >> 1), nblocks - 1);
                b_src[j] = wg + (long long)kr * g.w_rowb + (size_t)nb * 128 + (seg & 1) * 64 + ((lane & 3) << 4);
            }
        }
        // Through REGISTERS, not by LDS-DMA: a -DGR_STAMP build of the first form of this loader
        // (twelve global_load_lds per wave and step) spent 1,700 cycles per step issuing them --
        // 140 each even in a wave that does nothing else: the CU's global -> LDS path moves about
        // 30 bytes per clock, and the kernels above sit on the same ceiling (64 KiB per 2,350
        // cycles).  global_load_dwordx4 (64 B / clk / CU) + ds_write_b128 (79 B / clk / CU) is twice
        // that, and a loader wave has the registers: two steps' worth, 96, so that step t + 2 is
        // requested as soon as step t has left for LDS.  Same LDS image (the swizzle sits in the
        // source address), same flags; hipcc counts these loads itself.
        uint4 ra0[4], rb0[8], ra1[4], rb1[8];
#define GR_FETCH(KS, VA, VB)                                                                        \
    do {                                                                                            \
        const size_t ka_ = (size_t)(KS) * 128, kb_ = BN_LAYOUT ? (size_t)(KS) * GG_BK * g.w_rowb : ka_; \
        _Pragma("unroll") for (int j = 0; j < 4; j++) VA[j] = *reinterpret_cast<const uint4 *>(a_src[j] + ka_); \
        _Pragma("unroll") for (int j = 0; j < 8; j++) VB[j] = *reinterpret_cast<const uint4 *>(b_src[j] + kb_); \
    } while (0)
#define GR_STEP(KS, VA, VB)                                                                         \
    do {                                                                                            \
        const int slot = (int)(t % GR_SLOTS);                                                       \
        const unsigned use = t / GR_SLOTS;                                                          \
        GR_T(l0);                                                                                   \
        if (use > 0) gr_wait(flags + 32 + 4 * slot, 4 * use);                                       \
        GR_T(l1);                                                                                   \
        char *dst = lds + slot * GR_STAGE + 16 * lane;                                              \
        _Pragma("unroll") for (int j = 0; j < 4; j++)                                               \
            *reinterpret_cast<uint4 *>(dst + (lw + 4 * j) * 1024) =                                 \
                make_uint4(VA[j].x, VA[j].y, VA[j].z, VA[j].w); /* (whole-element copies keep the array in scratch) */ \
        _Pragma("unroll") for (int j = 0; j < 8; j++)                                               \
            *reinterpret_cast<uint4 *>(dst + GR_A_TILE + (lw + 4 * j) * 1024) =                     \
                make_uint4(VB[j].x, VB[j].y, VB[j].z, VB[j].w);                                     \
        __builtin_amdgcn_s_waitcnt(gr_waitcnt(63, 0)); /* the ds_writes have landed */              \
        if (lane == 0) gr_flag_add(flags + 4 * slot);                                               \
        GR_T(l2);                                                                                   \
        if ((KS) + 2 < nk) GR_FETCH((KS) + 2, VA, VB);                                              \
        GR_T(l3);                                                                                   \
        GR_ACC(0, l0, l1); GR_ACC(1, l1, l2); GR_ACC(2, l2, l3);                                    \
        t++;                                                                                        \
    } while (0)
        GR_FETCH(0, ra0, rb0);
        if (nk > 1) GR_FETCH(1, ra1, rb1);
        for (int ks = 0; ks < nk; ks += 2) {
            GR_STEP(ks, ra0, rb0);
            if (ks + 1 < nk) GR_STEP(ks + 1, ra1, rb1);
        }
#undef GR_STEP
#undef GR_FETCH
        if (EXT) {
            // the K extension's step: fp32 a2 [*, R] / b2 [n][R] split into two KC tiles (zero past R)
            const int slot = (int)(t % GR_SLOTS);
            const unsigned use = t / GR_SLOTS;
            if (use > 0) gr_wait(flags + 32 + 4 * slot, 4 * use);
            char *As = lds + slot * GR_STAGE, *Bs = As + GR_A_TILE;
            const int ltid = 64 * lw + lane, kq = ltid & 7, k = 4 * kq;
            auto put4 = [&](char *tile, int r, const float4 &v) {
                unsigned h0, l0, h1, l1;
                gg_split2(v.x, v.y, h0, l0);
                gg_split2(v.z, v.w, h1, l1);
                const int sw = (r >> 1) & 7;
                char *row = tile + r * 128 + (kq & 1) * 8;
                *reinterpret_cast<uint2 *>(row + (((kq >> 1) ^ sw) << 4)) = make_uint2(h0, h1);
                *reinterpret_cast<uint2 *>(row + (((4 + (kq >> 1)) ^ sw) << 4)) = make_uint2(l0, l1);
            };
#pragma unroll
            for (int u = 0; u < GR_BM / 32; u++) {
                const int r = (ltid >> 3) + 32 * u;
                const int p = tl.row_lo + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p < tl.row_hi && k < g.R) {
                    const long long src = g.gather2 ? g.gather2[p] : p;
                    v = *reinterpret_cast<const float4 *>(g.a2 + src * g.lda2 + k);
                }
                put4(As, r, v);
            }
#pragma unroll
            for (int u = 0; u < GR_BN / 32; u++) {
                const int r = (ltid >> 3) + 32 * u;
                const int n = tl.n0 + r;
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < g.N && k < g.R)
                    b = *reinterpret_cast<const float4 *>(g.b2 + (size_t)tl.bucket * g.b2_gstride +
                                                          (size_t)n * g.b2_ldn + k);
                put4(Bs, r, b);
            }
            __builtin_amdgcn_s_waitcnt(gr_waitcnt(63, 0));          // the ds_writes have landed
            if (lane == 0) gr_flag_add(flags + 4 * slot);
            t++;
        }
    }
#ifdef GR_STAMP
    if (lane == 0) {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(g.out + (size_t)g.P * g.ldo) +
                                  ((size_t)blockIdx.x * 8 + 4 + lw) * 6;
        dst[0] = st[0]; dst[1] = st[1]; dst[2] = st[2]; dst[3] = st[3];
        dst[4] = __builtin_amdgcn_s_memtime() - t_begin; dst[5] = t;
    }
#endif
}

template <int NJ>
struct GrFrags { GgFrag a[2], b[NJ]; };
// B fragment of an NC tile [32 k][1024 B] (256 n): gi_frag_nc with the longer k-row
__device__ __forceinline__ uint4 gr_frag_nc(const char *tile, int ncol0, int q2, int part, int fh, int lane) {
    const int gl = lane & 15, q = gl >> 2, p = gl & 3, nhalf = (lane >> 4) & 1;
    const int seg = 2 * (ncol0 >> 5) + part;
    const char *ptr = tile + (16 * q2 + 8 * fh + q) * 1024 + ((seg ^ q) << 6) + nhalf * 32 + p * 8;
    const uint2 lo = gg_tr_b64(ptr), hi = gg_tr_b64(ptr + 4 * 1024);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
// half h (16 k) of a stage: the consumer's 2 A and 4 B fragments
template <bool KC_B, bool A_F32>
__device__ __forceinline__ void gr_read(GrFrags<4> &f, const char *As, const char *Bs, int h, int wm, int wn,
                                        int frow, int fh, int lane) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
        if constexpr (A_F32) {
            f.a[i] = gi_frag_a32(As, wm + 32 * i + frow, h, fh);
        } else {
            f.a[i].hi = gi_frag_kc(As, wm + 32 * i + frow, h, 0, fh);
            f.a[i].lo = gi_frag_kc(As, wm + 32 * i + frow, h, 1, fh);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if constexpr (KC_B) {
            f.b[j].hi = gi_frag_kc(Bs, wn + 32 * j + frow, h, 0, fh);
            f.b[j].lo = gi_frag_kc(Bs, wn + 32 * j + frow, h, 1, fh);
        } else {
            f.b[j].hi = gr_frag_nc(Bs, wn + 32 * j, h, 0, fh, lane);
            f.b[j].lo = gr_frag_nc(Bs, wn + 32 * j, h, 1, fh, lane);
        }
    }
}
__device__ __forceinline__ void gr_mma(const GrFrags<4> &f, f32x16 (&acc)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = gg_mma3(f.a[i], f.b[j], acc[i][j]);
}

template <bool BN_LAYOUT, bool EXT, bool A32>
__device__ __forceinline__ void gr_consumer(const GroupedArgs &g, char *lds, int cw, int lane) {
#ifdef GR_STAMP
    unsigned long long st[4] = {0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    const GrPlan pl = gr_plan(g);
    const unsigned flags = gr_lds_addr(lds + GR_FLAGS);
    const int nk = g.K / GG_BK;
    const int wm = (cw >> 1) * 64, wn = (cw & 1) * 128;
    const int frow = lane & 31, fh = lane >> 5;
    float *rsbuf = reinterpret_cast<float *>(lds + GR_RS) + 64 * cw;
    unsigned t = 0;
    for (int id = blockIdx.x; id < pl.total; id += gridDim.x) {
        const GrTile tl = gr_tile(g, pl, id);
        if (tl.bucket < 0) continue;
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
        // this wave's 64 row scales, through its own LDS words (read back per accumulator row)
        if (g.rowscale) {
            rsbuf[lane] = g.rowscale[min(tl.row_lo + wm + lane, tl.row_hi - 1)];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        GrFrags<4> f0, f1;
        GR_T(c0);
        gr_wait(flags + 4 * (t % GR_SLOTS), 4 * (t / GR_SLOTS + 1));
        GR_T(c1);
        GR_ACC(0, c0, c1);
        {
            const char *As = lds + (t % GR_SLOTS) * GR_STAGE;
            gr_read<!BN_LAYOUT, A32>(f0, As, As + GR_A_TILE, 0, wm, wn, frow, fh, lane);
        }
        for (int ks = 0; ks < nk; ks++, t++) {
            const int slot = (int)(t % GR_SLOTS);
            const char *As = lds + slot * GR_STAGE;
            const bool more = ks + 1 < nk;
            const unsigned nslot = (t + 1) % GR_SLOTS, nneed = 4 * ((t + 1) / GR_SLOTS + 1);
            // the next stage's FULL word: asked for now, looked at behind the MFMAs
            unsigned fv = 0;
            if (more) asm volatile("ds_read_b32 %0, %1" : "=v"(fv) : "v"(flags + 4 * nslot) : "memory");
            gr_read<!BN_LAYOUT, A32>(f1, As, As + GR_A_TILE, 1, wm, wn, frow, fh, lane);
            __builtin_amdgcn_sched_barrier(0);
            gr_mma(f0, acc);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fv)::"memory");   // f1 (and fv) have landed
            if (lane == 0) gr_flag_add(flags + 32 + 4 * slot);           // the stage may be refilled
            if (more) {
                GR_T(c2);
                if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)fv) - nneed) < 0)
                    gr_wait(flags + 4 * nslot, nneed);
                GR_T(c3);
                GR_ACC(1, c2, c3);
                const char *An = lds + nslot * GR_STAGE;
                gr_read<!BN_LAYOUT, A32>(f0, An, An + GR_A_TILE, 0, wm, wn, frow, fh, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            gr_mma(f1, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        // rowscale applies to the main product alone: before the K extension
        if (g.rowscale) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float rs = rsbuf[32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh];
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j][r] *= rs;
                }
        }
        if (EXT) {
            const int slot = (int)(t % GR_SLOTS);
            gr_wait(flags + 4 * slot, 4 * (t / GR_SLOTS + 1));
            const char *As = lds + slot * GR_STAGE;
            gr_read<true, false>(f0, As, As + GR_A_TILE, 0, wm, wn, frow, fh, lane);
            if (g.R > 16) gr_read<true, false>(f1, As, As + GR_A_TILE, 1, wm, wn, frow, fh, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) gr_flag_add(flags + 32 + 4 * slot);
            gr_mma(f0, acc);
            if (g.R > 16) gr_mma(f1, acc);
            t++;
        }
        if (g.bias) {
            const float *bp = g.bias + (size_t)tl.bucket * g.N + tl.n0 + wn + frow;
            float b4[4];
#pragma unroll
            for (int j = 0; j < 4; j++) b4[j] = tl.n0 + wn + 32 * j + frow < g.N ? bp[32 * j] : 0.0f;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float rs = g.rowscale ? rsbuf[32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh] : 1.0f;
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j][r] = fmaf(rs, b4[j], acc[i][j][r]);
                }
        }
        // ---- stores, from the accumulator layout: a dword per lane, two 128-byte row segments per
        // instruction; rows past the bucket end fall outside the descriptor, columns past N get an
        // offset beyond it (no branch round a store) ----
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            g.out + (long long)tl.row_lo * g.ldo, 0, (int)((long long)(tl.row_hi - tl.row_lo) * g.ldo * 4),
            0x00020000);
        const unsigned ldo4 = (unsigned)(g.ldo * 4);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int col = tl.n0 + wn + 32 * j + frow;
            const unsigned voff = col < g.N ? (unsigned)(wm + 4 * fh) * ldo4 + 4u * (unsigned)col : 0x80000000u;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    // (through a float: __builtin_bit_cast of an ext_vector ELEMENT took element 0 for
                    // every r -- hipcc 7.2 -- and every row of a tile got its first row's values.  The
                    // row goes into the VECTOR offset: the range check does not see the scalar one)
                    const float v = acc[i][j][r];
                    __builtin_amdgcn_raw_buffer_store_b32(
                        __builtin_bit_cast(unsigned, v), rsrc,
                        voff + (unsigned)(32 * i + (r & 3) + 8 * (r >> 2)) * ldo4, 0, 0);
                }
        }
    }
#ifdef GR_STAMP
    if (lane == 0) {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(g.out + (size_t)g.P * g.ldo) +
                                  ((size_t)blockIdx.x * 8 + cw) * 6;
        dst[0] = st[0]; dst[1] = st[1]; dst[2] = st[2]; dst[3] = st[3];
        dst[4] = __builtin_amdgcn_s_memtime() - t_begin; dst[5] = t;
    }
#endif
}

template <bool BN_LAYOUT, bool EXT, bool A32>
__global__ __launch_bounds__(GR_THREADS, 1) void grouped_gemm_ring_kernel(GroupedArgs g) {
    extern __shared__ __attribute__((aligned(1024))) char gr_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 16) reinterpret_cast<unsigned *>(gr_smem + GR_FLAGS)[threadIdx.x] = 0u;
    __syncthreads();
    if (wave < 4)
        gr_consumer<BN_LAYOUT, EXT, A32>(g, gr_smem, wave, lane);
    else
        gr_loader<BN_LAYOUT, EXT, A32>(g, gr_smem, wave - 4, lane);
}

// ============================================================================ register path
// fp32 operands, split while they are staged.  LDS images of a tile: bf16, one image per part
// of the split (hi, lo and -- GEMMs in front of a ReLU -- mid).  k-contiguous tiles: rows of
// 32 k (64 B + 16 pad): a fragment (8 k of one row) is one conflict-free ds_read_b128 per
// part.  n-contiguous weight tiles stay [k][n] (rows of 128 n, 256 B + 64 pad) and are read
// with the transposing ds_read_b64_tr_b16.
constexpr int GG_KQ = GG_BK / 4;                 // float4 per tile row
constexpr int GG_RPP = GG_THREADS / GG_KQ;       // tile rows staged per pass
constexpr int GG_NU = GG_BN / GG_RPP;            // float4 of B per thread per k-step
constexpr int GG_BNK = GG_THREADS / 32;          // k rows of an n-contiguous B tile staged per pass
constexpr int GG_ROWB = GG_BK * 2 + 16;          // bytes per row of a k-contiguous image: 80
constexpr int GG_BNROWB = GG_BN * 2 + 64;        // bytes per k-row of an n-contiguous image: 320
constexpr int GG_AIMG = GG_BM * GG_ROWB;         // one part of the A tile: 10240 B
constexpr int GG_BIMG = GG_BN * GG_ROWB;         // one part of a B tile (either orientation)
static_assert(GG_BK * GG_BNROWB == GG_BIMG, "both B orientations fit the same slot");
constexpr int GG_REGS_LDS_FLOATS = 2 * (GG_AIMG + GG_BIMG) / 4;   // A | B, hi | lo each

// One output tile of BM x 128: BM = 128 (each wave a 64 x 64 quadrant) or BM = 64 (each
// wave 32 x 64), same B tile, same LDS image, same epilogue.
template <int BM, bool BN_LAYOUT, int EPI, bool EXT, bool KTAIL>
__device__ __forceinline__ void gemm_tile_regs(const GroupedArgs &g, float *smem, int bucket,
                                               int row_lo, int row_hi, int col_tile) {
    constexpr int NI = BM / 64;                  // 32-row sub-blocks per wave
    constexpr int NUA = BM / GG_RPP;             // float4 of A per thread per k-step
    constexpr int NPART = 2;                     // hi, lo
    char *const lds0 = reinterpret_cast<char *>(smem);
    char *As = lds0;                                      // [NPART][GG_AIMG]: hi | lo
    char *Bs = As + NPART * GG_AIMG;                      // [NPART][GG_BIMG]
    const int n0 = col_tile * GG_BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = (wave >> 1) * (BM / 2);       // the wave's origin inside the tile
    const int wn = (wave & 1) * 64;
    const float *wg = g.w + (size_t)bucket * g.gstride;

    float *const extras = smem + GG_REGS_LDS_FLOATS;
    if (tid < BM) {
        const int p = min(row_lo + tid, row_hi - 1);
        const float rs = g.rowscale ? g.rowscale[p] : 1.0f;
        extras[384 + tid] = rs;
        if (EPI == EPI_ACT && g.act == ACT_RELU && g.a_norm)
            extras[256 + tid] = g.a_norm[g.gather ? g.gather[p] : p] * rs;
    }
    // ---- staging assignment: tile = rows x GG_KQ float4 along k ----
    // (8 lanes per row; the two rows of a 16-lane ds_write_b64 group are 4 apart: 320 bytes = 64
    // mod 128, so their 64-byte pieces fall on disjoint banks -- adjacent rows, 80 bytes apart,
    // share four: SQ_LDS_BANK_CONFLICT was a third of the LDS-array cycles)
    static_assert(GG_KQ == 8, "row interleave below assumes 8 lanes per tile row");
    const int s_slot = (tid >> 3) & 7;
    const int s_row = (tid >> 6) * 8 + (((s_slot & 1) << 2) | (s_slot >> 1)), s_kq = tid & 7;
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
    // and the K extension), else [k][n], read with the transposing ds_read_b64_tr_b16 ----
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
    // ---- staging: four fp32 values -> the 8-byte pieces of both parts of an image ----
    auto put4 = [&](char *img, int part_stride, int off, const float4 &v) {
        unsigned h0, l0, h1, l1;
        gg_split2(v.x, v.y, h0, l0);
        gg_split2(v.z, v.w, h1, l1);
        *reinterpret_cast<uint2 *>(img + off) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(img + part_stride + off) = make_uint2(l0, l1);
    };
    auto contract = [&](int kmax, bool bt_image) {        // k = 0 .. kmax of the staged tiles
        for (int q2 = 0; q2 < (kmax + 15) / 16; q2++) mfma_group16(q2, bt_image);
    };

    // ---- software pipeline: the global loads of k-step t+1 are in flight while the MFMAs
    // of step t run; registers -> LDS happens at the top of the next step ----
    float4 av[NUA], bv[GG_NU];
    // segments of K (a_seg_k > 0; the calls below visit k0 = 0, 32, 64 .. in order): the offset of
    // the current segment's matrix, less the k's that lie before it
    long long a_adj = 0;
    int seg_left = g.a_seg_k > 0 ? g.a_seg_k : 0x7FFFFFFF;
    // KTAIL == false (K % GG_BK == 0): no predicate anywhere in the loads.  The predicated
    // form compiles into branches around the loads, 8 per k-step.
    auto load_tile = [&](int k0) {
        const int k = k0 + 4 * s_kq;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < NUA; u++) {
            if constexpr (KTAIL)
                av[u] = k < g.K ? *reinterpret_cast<const float4 *>(a_src[u] + a_adj + k) : zero;
            else
                av[u] = *reinterpret_cast<const float4 *>(a_src[u] + a_adj + k);
        }
        seg_left -= GG_BK;
        if (seg_left <= 0) { a_adj += g.a_seg_stride - g.a_seg_k; seg_left = g.a_seg_k; }
        if constexpr (!BN_LAYOUT) {
#pragma unroll
            for (int u = 0; u < GG_NU; u++) {
                if constexpr (KTAIL)
                    bv[u] = k < g.K ? *reinterpret_cast<const float4 *>(b_src[u] + k) : zero;
                else
                    bv[u] = *reinterpret_cast<const float4 *>(b_src[u] + k);
            }
        } else {
            // W_g(n, k), n contiguous: thread -> k row (tid >> 5) + 8 u, n-quad tid & 31
#pragma unroll
            for (int u = 0; u < GG_NU; u++) {
                const int kk = k0 + (tid >> 5) + GG_BNK * u;
                if constexpr (KTAIL)
                    bv[u] = kk < g.K ? *reinterpret_cast<const float4 *>(bn_src + (size_t)kk * g.ldk)
                                     : zero;
                else
                    bv[u] = *reinterpret_cast<const float4 *>(bn_src + (size_t)kk * g.ldk);
            }
        }
    };
    // registers -> LDS images (component-wise: a struct copy of av[u] keeps the whole array in
    // scratch memory)
    auto stage_store = [&]() {
#pragma unroll
        for (int u = 0; u < NUA; u++)
            put4(As, GG_AIMG, (s_row + GG_RPP * u) * GG_ROWB + 8 * s_kq,
                 make_float4(av[u].x, av[u].y, av[u].z, av[u].w));
        if (!BN_LAYOUT) {
#pragma unroll
            for (int u = 0; u < GG_NU; u++)
                put4(Bs, GG_BIMG, (s_row + GG_RPP * u) * GG_ROWB + 8 * s_kq,
                     make_float4(bv[u].x, bv[u].y, bv[u].z, bv[u].w));
        } else {
            // n-contiguous weights keep their orientation in LDS: Bs[k][n] (transposing them
            // into the [n][k] image needs 4-byte writes 4 rows apart: 16-way bank conflicts)
#pragma unroll
            for (int u = 0; u < GG_NU; u++)
                put4(Bs, GG_BIMG, ((tid >> 5) + GG_BNK * u) * GG_BNROWB + 8 * (tid & 31),
                     make_float4(bv[u].x, bv[u].y, bv[u].z, bv[u].w));
        }
    };
    if (g.K > 0) load_tile(0);
    for (int k0 = 0; k0 < g.K; k0 += GG_BK) {
        __syncthreads();  // previous tile fully consumed
        stage_store();
        __syncthreads();
        if (k0 + GG_BK < g.K) load_tile(k0 + GG_BK);     // the registers are free again
#pragma unroll
        for (int q2 = 0; q2 < GG_BK / 16; q2++) mfma_group16(q2, !BN_LAYOUT);
    }

    gg_scale_rows<NI>(g, acc, extras, wm);

    // ---- K extension on top: acc += A2 . B2_g^T, 32 columns of the two operands per k-step (zero
    // beyond R; R > 32 -- several rank-16 side products side by side -- only without the ReLU
    // epilogue, whose `extras` hold the norms of ONE step's rows) ----
    if (EXT) {
        for (int r0 = 0; r0 < g.R; r0 += GG_BK) {
            __syncthreads();              // the last k-step's tiles are consumed
            const int k = r0 + 4 * s_kq;
#pragma unroll
            for (int u = 0; u < NUA; u++) {
                const int r = s_row + GG_RPP * u;
                const int p = row_lo + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p < row_hi && k < g.R) {
                    const int src = g.gather2 ? g.gather2[p] : p;
                    v = *reinterpret_cast<const float4 *>(g.a2 + (size_t)src * g.lda2 + k);
                }
                put4(As, GG_AIMG, r * GG_ROWB + 8 * s_kq, v);
                const float ss = group_sum<8>(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
                if (s_kq == 0) extras[r] = ss;              // |a2 row|^2, for the epilogue
            }
#pragma unroll
            for (int u = 0; u < GG_NU; u++) {
                const int r = s_row + GG_RPP * u;
                const int n = n0 + r;
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < g.N && k < g.R)
                    b = *reinterpret_cast<const float4 *>(
                        g.b2 + (size_t)bucket * g.b2_gstride + (size_t)n * g.b2_ldn +
                        (g.b2_seg ? (size_t)(k >> 4) * g.b2_seg + (k & 15) : (size_t)k));
                put4(Bs, GG_BIMG, r * GG_ROWB + 8 * s_kq, b);
                const float ss = group_sum<8>(b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w);
                if (s_kq == 0) extras[128 + r] = ss;        // |b2 row|^2
            }
            __syncthreads();
            contract(min(GG_BK, g.R - r0), true);
        }
    }
    __syncthreads();   // all waves are done with the operand tiles
    gg_epilogue<NI, EPI>(g, smem, acc, bucket, row_lo, row_hi, col_tile, wm, wn, extras, EXT);
}

// Work distribution.  `slots` workgroups run at a time (two per CU), so a launch of T tiles
// takes ceil(T / slots) rounds.  Measured on the register path with per-workgroup timestamps
// (P = 16384 rows ragged over 4 buckets, K = N = 1024): the tiles of the last, partial round
// ran one per CU for as long as a full round -- it is therefore cut into half-height tiles
// (64 x 128): twice as many workgroups.  The split is decided on the device -- bucket sizes
// never visit the host -- and workgroup ids are dispatched in order, so ids [0, main) take
// the full rounds, ids [main, main + 2 R) the halves of the R remaining tiles, and the rest of
// the (worst-case sized) grid exits at once.
__device__ __forceinline__ GgWork gg_find_work(const GroupedArgs &g) {
    GgWork w;
    w.bucket = -1;
    w.row_lo = w.row_hi = w.col_tile = 0;
    w.half = -1;
    const int n_col_tiles = (g.N + GG_BN - 1) / GG_BN;
    int row_tiles = 0;
    for (int i = 0; i < g.G; i++)
        row_tiles += (g.offsets[i + 1] - g.offsets[i] + GG_BM - 1) / GG_BM;
    const int total = row_tiles * n_col_tiles;
    const int main_tiles = (total / g.slots) * g.slots;
    const int rest = total - main_tiles;
    int id = blockIdx.x;
    int logical;
    // XCD-aware order inside each part: workgroups are dealt round-robin over the 8 XCDs
    // (each with its own 4 MiB L2); xcd_remap gives every XCD a contiguous run of logical
    // tiles, enumerated column-tile fastest, so the column tiles of one row tile (same A
    // panel) run back to back on one L2.
    if (id < main_tiles) {
        logical = (int)xcd_remap((unsigned)id, (unsigned)main_tiles);
    } else {
        id -= main_tiles;
        if (id >= 2 * rest) return w;
        const int h = (int)xcd_remap((unsigned)id, (unsigned)(2 * rest));
        logical = main_tiles + (h >> 1);
        w.half = h & 1;
    }
    w.col_tile = logical % n_col_tiles;
    int tile = logical / n_col_tiles;
    for (int i = 0; i < g.G; i++) {
        const int lo = g.offsets[i], hi = g.offsets[i + 1];
        const int tiles = (hi - lo + GG_BM - 1) / GG_BM;
        if (tile < tiles) {
            w.bucket = i;
            w.row_lo = lo + tile * GG_BM;
            w.row_hi = min(hi, w.row_lo + GG_BM);
            break;
        }
        tile -= tiles;
    }
    if (w.bucket >= 0 && w.half >= 0) {
        w.row_lo += (GG_BM / 2) * w.half;
        w.row_hi = min(w.row_hi, w.row_lo + GG_BM / 2);
        if (w.row_lo >= w.row_hi) w.bucket = -1;   // a ragged tile may have no second half
    }
    return w;
}

template <bool BN_LAYOUT, int EPI, bool EXT, bool KTAIL>
__global__ __launch_bounds__(GG_THREADS, 2) void grouped_gemm_kernel(GroupedArgs g) {
    // A | B images (hi | lo each), then the ReLU epilogue's extras; the epilogue reuses the
    // images as four per-wave C staging areas
    static_assert(GG_REGS_LDS_FLOATS >= (GG_THREADS / 64) * 32 * (64 + 4), "");
    __shared__ __attribute__((aligned(16))) float smem[GG_REGS_LDS_FLOATS + GG_EXTRAS];
    const GgWork w = gg_find_work(g);
    if (w.bucket < 0) return;  // uniform for the workgroup
    if (w.half < 0)
        gemm_tile_regs<GG_BM, BN_LAYOUT, EPI, EXT, KTAIL>(g, smem, w.bucket, w.row_lo, w.row_hi, w.col_tile);
    else
        gemm_tile_regs<GG_BM / 2, BN_LAYOUT, EPI, EXT, KTAIL>(g, smem, w.bucket, w.row_lo, w.row_hi,
                                                             w.col_tile);
}

template <bool BN_LAYOUT, int EPI, bool EXT, bool A32>
__global__ __launch_bounds__(GG_THREADS, 2) void grouped_gemm_img_kernel(GroupedArgs g) {
    // ALL of the kernel's LDS in one array (a second __shared__ object beside an LDS-DMA
    // target can make hipcc drain vmcnt before every ds_read: cdna guide, section 5)
    __shared__ __attribute__((aligned(1024))) float smem[GI_LDS_FLOATS + GG_EXTRAS];
    const GgWork w = gg_find_work(g);
    if (w.bucket < 0) return;
    if (w.half < 0)
        gemm_tile_img<GG_BM, BN_LAYOUT, EPI, EXT, A32>(g, smem, w.bucket, w.row_lo, w.row_hi, w.col_tile);
    else
        gemm_tile_img<GG_BM / 2, BN_LAYOUT, EPI, EXT, A32>(g, smem, w.bucket, w.row_lo, w.row_hi,
                                                           w.col_tile);
}

// fp32 [rows, cols] (leading dimension ld) -> the bf16 image [row][cols / 32][hi | lo][32]:
// a thread converts 8 consecutive columns (32 bytes in, 16 + 16 bytes out); columns past
// `cols` inside the last block are zero.
__global__ __launch_bounds__(256) void split_bf16_kernel(const float *__restrict__ src,
                                                         char *__restrict__ img, long long rows,
                                                         int cols, long long ld, int blocks) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long chunk_row = idx >> 2;               // (row, block)
    const int c = (int)(idx & 3);
    const long long row = chunk_row / blocks;
    const int kb = (int)(chunk_row - row * blocks);
    if (row >= rows) return;
    const int col = kb * 32 + c * 8;
    const float *p = src + row * ld + col;
    float x[8];
    if (col + 8 <= cols) {
        const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = col + e < cols ? p[e] : 0.0f;
    }
    uint4 hi, lo;
    gg_split2(x[0], x[1], hi.x, lo.x);
    gg_split2(x[2], x[3], hi.y, lo.y);
    gg_split2(x[4], x[5], hi.z, lo.z);
    gg_split2(x[6], x[7], hi.w, lo.w);
    char *dst = img + (chunk_row * 128) + c * 16;
    *reinterpret_cast<uint4 *>(dst) = hi;
    *reinterpret_cast<uint4 *>(dst + 64) = lo;
}

// The queue of the GEMM in front of a ReLU: one wave per queued (row, column), recomputed in
// fp32 and written over what the epilogue stored.  (In the epilogue itself each costs the wave
// that found it a serial ~2 us of dependent loads: 80 us of a 120 us GEMM at the FFN shape.)
__global__ __launch_bounds__(256) void relu_fix_kernel(GroupedArgs g) {
    // grid = 4 GG_FIX_SEGS workgroups: 16 waves per segment
    const int lane = threadIdx.x & 63;
    const unsigned seg = blockIdx.x % GG_FIX_SEGS;
    const unsigned wave = (blockIdx.x / GG_FIX_SEGS) * 4 + (threadIdx.x >> 6);
    const unsigned count = min(g.fix_count[16 * seg], (unsigned)g.fix_cap);
    for (unsigned i = wave; i < count; i += 16) {
        const int2 q = g.fix_list[(size_t)seg * g.fix_cap + i];
        const int p = q.x, n = q.y & 0xFFFFFF, bucket = (unsigned)q.y >> 24;
        // an entry is an ADDRESS in the making: one that is not a cell of this launch's output (a
        // stale counter or list, as the out-of-order memset node of round 3 produced) is skipped,
        // not dereferenced
        if (p < 0 || p >= g.P || n >= g.N || bucket >= g.G) continue;
        const float exact = gg_exact_preact<true>(g, bucket, p, n, lane);
        if (lane == 0) {
            g.out[(size_t)p * g.ldo + n] = act_forward(g.act, exact);
            if (g.out2) g.out2[(size_t)p * g.ldo + n] = exact;
        }
    }
}

// y[t, :] = bias + sum_{j < k} rows[pos[t * k + j], :]  -- the un-bucketing of the routed
// FFN (reference: `y[mask] += ...` per block, lora_ffn.py:107-111): a gather in a fixed
// order instead of a scatter-add, so the result is deterministic.
// side != null: out[t, :] += sum_{j < ns} side[t, j] * side_w[j, :] on top (a per-token rank-ns
// product: the router's share of the routed FFN's input gradient, d logit . W_router, which would
// otherwise be a library GEMM with a 4-wide contraction and an addition pass)
__global__ __launch_bounds__(256) void rows_combine_kernel(
    const float *__restrict__ rows, const int32_t *__restrict__ pos,
    const float *__restrict__ bias, float *__restrict__ out, int n_tokens, int k, int d4,
    const float *__restrict__ side, const float *__restrict__ side_w, int ns) {
    const int t = blockIdx.x;
    for (int c = threadIdx.x; c < d4; c += 256) {
        float4 acc = bias ? reinterpret_cast<const float4 *>(bias)[c]
                          : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < k; j++) {
            const int p = pos[(size_t)t * k + j];
            const float4 v = reinterpret_cast<const float4 *>(rows + (size_t)p * d4 * 4)[c];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (side) {
            for (int j = 0; j < ns; j++) {
                const float s = side[(size_t)t * ns + j];
                const float4 w = reinterpret_cast<const float4 *>(side_w + (size_t)j * d4 * 4)[c];
                acc.x = fmaf(s, w.x, acc.x); acc.y = fmaf(s, w.y, acc.y);
                acc.z = fmaf(s, w.z, acc.z); acc.w = fmaf(s, w.w, acc.w);
            }
        }
        reinterpret_cast<float4 *>(out + (size_t)t * d4 * 4)[c] = acc;
    }
}

}  // namespace spt

using namespace spt;

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

// The image path takes a GEMM when both images are given, K is a whole number of k-steps and
// the group offsets of the weight fall on image rows / 128-byte blocks.
// -> 0: register path; 1: both operands as images; 2 ("A32"): the weight as an image, the
// activation as its fp32 rows (16-byte aligned, lda % 4 == 0: one LDS-DMA lane moves 16 bytes)
static int image_path(GroupedArgs &g, int epilogue) {
    if (!g.w_img || (!g.a_img && !g.a)) return 0;
    if (g.K % GG_BK != 0 || g.a_seg_k > 0 || (g.a2 && (g.R > GG_BK || g.b2_seg != 0))) return 0;
    const long long row_len = g.ldk == 1 ? g.ldn : g.ldk;       // elements of one weight row
    if (row_len <= 0 || g.gstride % 32 != 0 || row_len % 32 != 0) return 0;
    g.w_grow = g.gstride / row_len;
    g.w_gblk = (int)((g.gstride % row_len) / 32);
    g.w_rowb = (row_len / 32) * 128;
    if (g.a_img) {
        g.a_rowb = (long long)(g.K / 32) * 128;
        return 1;
    }
    if ((reinterpret_cast<uintptr_t>(g.a) & 15) != 0 || g.lda % 4 != 0) return 0;
    g.a_rowb = (long long)g.lda * 4;
    return 2;
}

static int launch_grouped(GroupedArgs g, int epilogue, void *stream) {
    if ((!g.a && !g.a_img) || (!g.w && !g.w_img) || !g.offsets || !g.out) return SPT_EINVAL;
    if (g.P <= 0 || g.K <= 0 || g.N <= 0 || g.G <= 0) return SPT_EINVAL;
    if (g.a_seg_k < 0 || g.lda < (g.a_seg_k > 0 ? g.a_seg_k : g.K)) return SPT_EINVAL;
    if (g.a_seg_k > 0) {
        if (g.a_seg_k % GG_BK != 0 || g.K % g.a_seg_k != 0 || g.a_seg_stride % 4 != 0) return SPT_ESHAPE;
        if (epilogue != EPI_PLAIN || !g.a) return SPT_EUNSUP;
    }
    if (g.ldo == 0) g.ldo = g.N;
    if (g.ldo < g.N) return SPT_EINVAL;
    if (g.accumulate && epilogue != EPI_PLAIN) return SPT_EUNSUP;
    if (g.K % 4 != 0 || g.lda % 4 != 0) return SPT_ESHAPE;       // float4 rows of A
    if (g.ldk != 1 && g.ldn != 1) return SPT_EUNSUP;
    if (g.ldk == 1 && (g.ldn % 4 != 0 || g.gstride % 4 != 0)) return SPT_ESHAPE;
    if (g.ldk != 1 && (g.ldk % 4 != 0 || g.N % 4 != 0 || g.gstride % 4 != 0)) return SPT_ESHAPE;
    const bool ext = g.a2 != nullptr;
    if (ext) {
        if (!g.b2 || g.R <= 0 || g.R > 2 * GG_BK) return SPT_EINVAL;
        if (g.R > GG_BK && epilogue != EPI_PLAIN) return SPT_EUNSUP;
        if (g.b2_seg != 0 && (g.b2_seg % 4 != 0 || g.R % 16 != 0 || epilogue != EPI_PLAIN)) return SPT_ESHAPE;
        if (g.R % 4 != 0 || g.lda2 % 4 != 0 || g.lda2 < g.R || g.b2_ldn % 4 != 0 ||
            g.b2_gstride % 4 != 0)
            return SPT_ESHAPE;
    }
    if (epilogue < EPI_PLAIN || epilogue > EPI_DACT) return SPT_EINVAL;
    if (epilogue != EPI_PLAIN && (g.act < ACT_RELU || g.act > ACT_SILU)) return SPT_EUNSUP;
    const unsigned row_tiles = (unsigned)((g.P + GG_BM - 1) / GG_BM + g.G);
    const unsigned col_tiles = (unsigned)((g.N + GG_BN - 1) / GG_BN);
    if ((unsigned long long)row_tiles * col_tiles > 0x3FFFFFFFull) return SPT_EUNSUP;
    if (epilogue == EPI_ACT && g.act == ACT_RELU) {
        // the near-the-kink recomputation reads the fp32 operands
        if (!g.a_norm || !g.w_norm || !g.a || !g.w) return SPT_EINVAL;
    }
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
    const int img = image_path(g, epilogue);
    if (!img && (!g.a || !g.w)) return SPT_EUNSUP;              // images only, but not usable
    // the loader / consumer ring: plain epilogue, image path, nothing to add into; one workgroup per CU
    // OPT-IN (SPT_GEMM_RING=1).  Alone it is 3-5 % faster than the kernels above at every size
    // (tools/micro/gemm_abl.py: 411 against 424 us at 4096 tiles); inside the configs[2] step it is
    // 4.8 ms SLOWER (58.3 against 53.5 ms, GEMM 30.8 against 27.7): 128 x 256 tiles on 256 resident
    // workgroups quantise badly at the step's shapes (FFN down: 520 tiles = 3 rounds for 2.03, q / k / v
    // 768 = 3 for 3, o 256 = 1), and its stamps show why the lead was small to begin with -- with the
    // loads in dedicated waves a k-step is still 2,900 cycles for 1,536 of MFMAs, because the loader's
    // twelve vector loads take 1,370 cycles to issue (the CU's L2 -> CU path runs at ~34 B / clk on
    // these 8-rows-per-instruction accesses, LDS-DMA or not) and its 48 KiB of ds_write_b128 another
    // 730 on the LDS port the consumers read through.
    // (read at every call: tests switch it inside one process)
    const char *ring_env = getenv("SPT_GEMM_RING");
    const bool ring_on = ring_env && ring_env[0] == '1';
    if (img && epilogue == EPI_PLAIN && !g.accumulate && ring_on && (g.slots % 16) == 0 &&
        (long long)GR_BM * g.ldo * 4 < 0x7fffffffLL && (!ext || g.R <= GG_BK)) {
        const dim3 rgrid(g.slots / GG_SLOTS_PER_CU);
#define SPT_GR(BN, EXT)                                                                              \
    do {                                                                                             \
        if (img == 1) {                                                                              \
            SPT_HIP_TRY(hipFuncSetAttribute((const void *)grouped_gemm_ring_kernel<BN, EXT, false>,  \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, GR_LDS));    \
            hipLaunchKernelGGL((grouped_gemm_ring_kernel<BN, EXT, false>), rgrid, dim3(GR_THREADS), GR_LDS, s, g); \
        } else {                                                                                     \
            SPT_HIP_TRY(hipFuncSetAttribute((const void *)grouped_gemm_ring_kernel<BN, EXT, true>,   \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, GR_LDS));    \
            hipLaunchKernelGGL((grouped_gemm_ring_kernel<BN, EXT, true>), rgrid, dim3(GR_THREADS), GR_LDS, s, g);  \
        }                                                                                            \
    } while (0)
        if (g.ldk == 1) {
            if (ext) SPT_GR(false, true); else SPT_GR(false, false);
        } else {
            if (ext) SPT_GR(true, true); else SPT_GR(true, false);
        }
#undef SPT_GR
        SPT_LAUNCH_CHECK();
        return SPT_OK;
    }
#define SPT_GG(BN, EPI, EXT)                                                                  \
    do {                                                                                      \
        if (img == 1)                                                                         \
            hipLaunchKernelGGL((grouped_gemm_img_kernel<BN, EPI, EXT, false>), grid,          \
                               dim3(GG_THREADS), 0, s, g);                                    \
        else if (img == 2)                                                                    \
            hipLaunchKernelGGL((grouped_gemm_img_kernel<BN, EPI, EXT, true>), grid,           \
                               dim3(GG_THREADS), 0, s, g);                                    \
        else if (k_tail)                                                                      \
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
    const bool relu_queue = epilogue == EPI_ACT && g.act == ACT_RELU && g.fix_list && g.G < 256 &&
                            g.N < (1 << 24);
    if (relu_queue) {
        SPT_ZERO_WORDS(g.fix_count, GG_FIX_SEGS * 16, s);
        SPT_LAUNCH_CHECK();
    } else
        g.fix_list = nullptr;
    if (g.ldk == 1) {
        if (ext) SPT_GG_EPI(false, true); else SPT_GG_EPI(false, false);
    } else {
        if (ext) SPT_GG_EPI(true, true); else SPT_GG_EPI(true, false);
    }
#undef SPT_GG_EPI
#undef SPT_GG
    SPT_LAUNCH_CHECK();
    if (relu_queue) {
        hipLaunchKernelGGL(relu_fix_kernel, dim3(4 * GG_FIX_SEGS), dim3(256), 0, s, g);
        SPT_LAUNCH_CHECK();
    }
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
    g.a_img = reinterpret_cast<const char *>(d->a_image);
    g.w_img = reinterpret_cast<const char *>(d->w_image);
    g.a_norm = d->a_norm; g.w_norm = d->w_norm;
    g.ldo = d->ldo;
    g.accumulate = d->accumulate;
    g.a_seg_k = d->a_seg_k; g.a_seg_stride = d->a_seg_stride;
    g.b2_seg = d->b2_seg_stride;
    const long long header = GG_FIX_SEGS * 64;
    if (d->relu_queue && d->relu_queue_bytes >= header + GG_FIX_SEGS * 8) {
        g.fix_count = reinterpret_cast<unsigned *>(d->relu_queue);
        g.fix_list = reinterpret_cast<int2 *>(reinterpret_cast<char *>(d->relu_queue) + header);
        const long long cap = (d->relu_queue_bytes - header) / 8 / GG_FIX_SEGS;
        g.fix_cap = (int)(cap > 0x7FFFFF ? 0x7FFFFF : cap);
    }
    return launch_grouped(g, d->epilogue, stream);
}

extern "C" int spt_grouped_gemm_image_path(const SptGroupedGemm *d) {
    if (!d) return 0;
    GroupedArgs g = {};
    g.K = d->k; g.gstride = d->w_group_stride; g.ldn = d->w_ldn; g.ldk = d->w_ldk;
    g.act = d->activation;
    g.a = d->a; g.lda = d->lda;
    g.a2 = d->a2; g.R = d->r; g.a_seg_k = d->a_seg_k; g.b2_seg = d->b2_seg_stride;
    g.a_img = reinterpret_cast<const char *>(d->a_image);
    g.w_img = reinterpret_cast<const char *>(d->w_image);
    return image_path(g, d->epilogue);
}

extern "C" size_t spt_split_bf16_bytes(long long rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (size_t)rows * (size_t)((cols + 31) / 32) * 128;
}

extern "C" int spt_split_bf16(const float *src, void *image, long long rows, int cols,
                              long long ld, void *stream) {
    if (!src || !image) return SPT_EINVAL;
    if (rows <= 0 || cols <= 0 || ld < cols) return SPT_EINVAL;
    if (ld % 4 != 0 || (reinterpret_cast<uintptr_t>(src) & 15) != 0 ||
        (reinterpret_cast<uintptr_t>(image) & 15) != 0)
        return SPT_ESHAPE;
    const int blocks = (cols + 31) / 32;
    const long long threads = rows * blocks * 4;
    if ((threads + 255) / 256 > 0x7FFFFFFFll) return SPT_EUNSUP;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, src, reinterpret_cast<char *>(image), rows, cols, ld,
                       blocks);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_rows_combine(const float *rows, const int32_t *pos, const float *bias,
                                float *out, int n_tokens, int k, int d, void *stream) {
    if (!rows || !pos || !out) return SPT_EINVAL;
    if (n_tokens <= 0 || k <= 0 || d <= 0) return SPT_EINVAL;
    if (d % 4 != 0) return SPT_ESHAPE;
    hipLaunchKernelGGL(rows_combine_kernel, dim3((unsigned)n_tokens), dim3(256), 0,
                       (hipStream_t)stream, rows, pos, bias, out, n_tokens, k, d / 4, nullptr, nullptr, 0);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_rows_combine_side(const float *rows, const int32_t *pos, const float *bias,
                                     const float *side, const float *side_w, int n_side, float *out,
                                     int n_tokens, int k, int d, void *stream) {
    if (!rows || !pos || !out || !side || !side_w) return SPT_EINVAL;
    if (n_tokens <= 0 || k <= 0 || d <= 0 || n_side <= 0 || n_side > 64) return SPT_EINVAL;
    if (d % 4 != 0 || (reinterpret_cast<uintptr_t>(side_w) & 15) != 0) return SPT_ESHAPE;
    hipLaunchKernelGGL(rows_combine_kernel, dim3((unsigned)n_tokens), dim3(256), 0,
                       (hipStream_t)stream, rows, pos, bias, out, n_tokens, k, d / 4, side, side_w, n_side);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
