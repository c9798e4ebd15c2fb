// sddmm_tile.hip -- batched CSR-sampled Q.K^T on the matrix cores: the dense-tile form of
// sddmm.hip for patterns as dense as lookup's (Z / S = 1/8).
//
// Replaces extension/sddmm.cpp:27-69 of the reference (cusparseSDDMM) for the same contract as
// sddmm.hip: out[b, p] = clamp(scale * dot(Q[b, row(p)], K[b, indices[b, p]])).
//
// Why a second form.  The gather form moves one 256-byte K row through LDS per CSR entry: 2.1 GB
// per launch at the benchmark shape for 134 MB of HBM traffic, LDS-bandwidth bound at 0.33 of the
// HBM roofline.  At 1/8 density it is cheaper to compute ALL scores of a row stripe
// with v_mfma_f32_32x32x16_bf16 (fp32 operands split x = hi + lo, three MFMAs per product:
// <= 2^-16 relative error per product, as everywhere in this library) and to pick the CSR's
// entries out of the finished stripe.
//
// Mapping: K-STATIONARY.  A workgroup of 8 waves owns 512 keys of one batch slice (all of them at
// S <= 512); wave w keeps keys 64 w .. 64 w + 63 as split A-operand fragments in REGISTERS for
// the whole launch (64 VGPRs): K is read once, 128 KiB per slice, and never touches LDS.  The
// rows stream past in stripes of 32:
//   * the stripe's Q rows become a bf16 image in LDS (8 KiB, one float4 per thread, double-buffered);
//   * every wave multiplies its two key tiles with it (24 MFMAs) and parks the 32 x 64 scores in
//     the stripe buffer [32 rows][512 keys] (fp32, 64.5 KiB, double-buffered, ds_write_b128);
//   * wave w walks rows 4 w .. 4 w + 3 of the stripe: 64 entries per instruction -- a coalesced
//     load of the column ids (requested three stripes ahead), one ds_read_b32 per entry from the
//     stripe buffer, scale / clamp, a coalesced store.
// LDS traffic per entry: 4 bytes instead of 256.  One barrier per stripe: iteration s forms the
// scores of stripe s + 1 and picks stripe s (the stripe buffers alternate; a buffer is rewritten
// only behind the barrier that every wave reaches after it has finished reading it), and the two
// waves of a SIMD take the two halves in opposite order.
// 512 < S <= 1024: two workgroups per slice, each picks the entries whose column falls in its keys.
//
// Measured (configs[2] attention shape, 256 slices x 512 x 64, 64 entries per row; rocprofv3 and
// tools/micro/time_sddmm.py): 36 us = 0.46 of the HBM roofline on the operator's 134 MB, against
// 51-56 us (0.30-0.33) for the gather form; 19 against 31 us at S = 256, 108 against 190 at
// S = 1024 (128 entries per row).  Where the 36 us go (s_memtime marks of one workgroup,
// -DST_STAMP): 13 k of 69 k cycles before the first stripe (the K fragments: 33.5 MB requested by all
// CUs at once, HBM-bound), then 3.2-3.4 k per stripe for 1.5 k cycles of MFMA per SIMD -- the rest
// is the wave's own chain of LDS round trips, address arithmetic (~170 vector instructions per
// stripe) and the 13-cycle ds_write_b128 of the stripe buffer (64 KiB per stripe and CU), which two
// waves per SIMD do not hide.  Tried on the way and not kept: 256-key workgroups, two to a CU (51 us:
// every entry's 256-byte result line is then written by two workgroups, 4 bytes at a time); four
// extra picker waves doing all global accesses while eight matrix waves only read Q, multiply and
// park (39-41 us: the pickers' ~330 vector instructions per stripe became the critical path); skipping
// the key tiles right of a stripe's largest column id (causal patterns: half of the MFMAs; the
// extra control flow cost more waits than the MFMAs saved: 39 us).
#include "spt_common.h"
#include <stdlib.h>

namespace spt {

typedef __attribute__((ext_vector_type(8))) __bf16 st_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 st_bf16x2;
typedef __attribute__((ext_vector_type(2))) float st_f32x2;
typedef __attribute__((ext_vector_type(16))) float st_f32x16;

constexpr int ST_WAVES = 8;
constexpr int ST_THREADS = 64 * ST_WAVES;
constexpr int ST_E = 64;                            // d_head
constexpr int ST_KS = ST_E / 16;                    // k-steps of the contraction
constexpr int ST_ROWS = 32;                         // rows per stripe = one MFMA tile
constexpr int ST_WKEYS = 64;                        // keys per wave: two MFMA tiles
constexpr int ST_KEYS = ST_WAVES * ST_WKEYS;        // keys per workgroup
constexpr int ST_DLD = ST_KEYS + 4;                 // floats per row of the parked stripe: the 64 lanes'
                                                    // 16-byte writes (row = lane % 32) spread evenly over the banks
constexpr int ST_DBUF = ST_ROWS * ST_DLD;           // floats of the stripe buffer
constexpr int ST_QLD = ST_E * 2 + 16;               // bytes per row of the Q image: E bf16 + 16 (conflict-free b128 reads)
constexpr int ST_QPART = ST_ROWS * ST_QLD;          // one part (hi or lo)
constexpr int ST_QIMG = 2 * ST_QPART;
constexpr int ST_RPW = ST_ROWS / ST_WAVES;          // rows of a stripe a wave picks entries for
constexpr int ST_LDS = 2 * ST_DBUF * 4 + 2 * ST_QIMG;        // 150,528 B: one workgroup per CU

struct StSplit { unsigned hi, lo; };
// two floats -> packed bf16 pairs (first value in the low half): hi = RNE, lo = RNE(x - hi)
__device__ __forceinline__ StSplit st_split2(float a, float b) {
    const st_f32x2 x = {a, b};
    StSplit s;
    s.hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, st_bf16x2));
    const st_f32x2 hf = {__builtin_bit_cast(float, s.hi << 16), __builtin_bit_cast(float, s.hi & 0xffff0000u)};
    s.lo = __builtin_bit_cast(unsigned, __builtin_convertvector(x - hf, st_bf16x2));
    return s;
}
struct StFrag { uint4 hi, lo; };
__device__ __forceinline__ StFrag st_split8(const float4 &a, const float4 &b) {
    const StSplit s0 = st_split2(a.x, a.y), s1 = st_split2(a.z, a.w), s2 = st_split2(b.x, b.y),
                  s3 = st_split2(b.z, b.w);
    StFrag f;
    f.hi = make_uint4(s0.hi, s1.hi, s2.hi, s3.hi);
    f.lo = make_uint4(s0.lo, s1.lo, s2.lo, s3.lo);
    return f;
}
__device__ __forceinline__ st_f32x16 st_mma(const uint4 &a, const uint4 &b, st_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(st_bf16x8, a),
                                                   __builtin_bit_cast(st_bf16x8, b), c, 0, 0, 0);
}
// (a.hi + a.lo)(b.hi + b.lo) without the lo * lo term, small terms first
__device__ __forceinline__ st_f32x16 st_mm(const StFrag &a, const StFrag &b, st_f32x16 c) {
    c = st_mma(a.lo, b.hi, c);
    c = st_mma(a.hi, b.lo, c);
    return st_mma(a.hi, b.hi, c);
}

struct StPicks { int start[ST_RPW], end[ST_RPW], col[ST_RPW]; };
#ifdef ST_STAMP
#define ST_MARK() do { if (stamps && lane == 0 && n_marks < 128) stamps[wave * 128 + n_marks] = (unsigned)__builtin_amdgcn_s_memtime(); n_marks++; } while (0)
#else
#define ST_MARK() do {} while (0)
#endif

// LONG = rows of more than 64 entries exist (the kernel looks at `indptr` and branches once).
// Every global access of the row loop is UNCONDITIONAL in the instantiation without them -- loads of
// clamped addresses, column ids and results through buffer instructions whose out-of-range lanes
// the hardware drops -- because a load or store under a branch (even the `s_cbranch_execz` around a
// predicated store) makes hipcc wait with vmcnt(0) at the next use of ANY load: the Q rows and
// column ids requested for later stripes, and every store, drained in every iteration.
template <bool LONG>
__device__ __forceinline__ void sddmm_tile_body(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ query, const float *__restrict__ key, float *__restrict__ out, int S,
    int nnz, int key_blocks, float scale, float clampv, int q_heads, int k_heads, char *smem) {
    float *dbuf = reinterpret_cast<float *>(smem);              // [2][ST_ROWS][ST_DLD]
    char *qimg = smem + 2 * ST_DBUF * 4;                        // [2][hi | lo][ST_ROWS][ST_QLD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c32 = lane & 31, h = lane >> 5;
    // consecutive logical ids (the key blocks of one slice, then the next slice) share an XCD
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / key_blocks, key0 = (bid - b * key_blocks) * ST_KEYS;
    const DenseView qv = dense_view(b, S, ST_E, q_heads), kv = dense_view(b, S, ST_E, k_heads);
    const float *q_b = query + qv.base, *k_b = key + kv.base;
    const int n_stripes = (S + ST_ROWS - 1) / ST_ROWS;
    // this slice's column ids and results, nnz * 4 bytes each
    const __amdgpu_buffer_rsrc_t idx_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int32_t *>(indices + (size_t)b * nnz), 0, nnz * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(out + (size_t)b * nnz, 0, nnz * 4, 0x00020000);
#ifdef ST_STAMP
    // diagnostic build: workgroup 0 stores s_memtime marks where its first 1024 results belong, and
    // no results (tools/micro/time_sddmm.py, ST_STAMPS=1)
    unsigned *stamps = bid == 0 ? reinterpret_cast<unsigned *>(out) : nullptr;
    if (bid == 0) out_rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000);
    if (stamps && lane == 0) stamps[wave * 128] = (unsigned)__builtin_amdgcn_s_memtime();
    int n_marks = 1;
#endif

    // thread t stages elements 4 (t % 16) .. + 4 of row t / 16 of a stripe's Q image
    const int qrow = tid >> 4, qe4 = (tid & 15) * 4;
    auto load_q = [&](int s) {
        s = min(s, n_stripes - 1);
        return *reinterpret_cast<const float4 *>(q_b + (size_t)min(s * ST_ROWS + qrow, S - 1) * qv.ld + qe4);
    };
    auto store_q = [&](int s, const float4 &x) {
        char *img = qimg + (s & 1) * ST_QIMG + qrow * ST_QLD + qe4 * 2;
        const StSplit a = st_split2(x.x, x.y), c = st_split2(x.z, x.w);
        *reinterpret_cast<uint2 *>(img) = make_uint2(a.hi, c.hi);
        *reinterpret_cast<uint2 *>(img + ST_QPART) = make_uint2(a.lo, c.lo);
    };
    // The entries this wave picks in stripe s: rows 4 w .. 4 w + 3.  Their five row pointers are ONE
    // vector load (lane l: indptr[row + min(l, 4)]), read back with v_readlane -- as scalar loads
    // they were four dependent round trips per stripe.
    auto load_ptrs = [&](int s) {
        s = min(s, n_stripes - 1);
        return indptr[min(s * ST_ROWS + wave * ST_RPW + min(lane, ST_RPW), S)];     // (row >= S: empty)
    };
    auto request = [&](int ptrs) {
        StPicks p;
#pragma unroll
        for (int rr = 0; rr < ST_RPW; rr++) {
            p.start[rr] = __builtin_amdgcn_readlane(ptrs, rr);
            p.end[rr] = __builtin_amdgcn_readlane(ptrs, rr + 1);
            const int e = p.start[rr] + lane;
            p.col[rr] = (int)__builtin_amdgcn_raw_buffer_load_b32(idx_rsrc, e < p.end[rr] ? e * 4 : -1, 0, 0);
        }
        return p;
    };

    // ---- prologue: everything that does not depend on another load is requested at once ----
    // this wave's keys: A-operand fragments, lane (j, h) = key j, elements 16 ks + 8 h .. + 8
    const int wkey0 = key0 + wave * ST_WKEYS;
    const bool wave_live = wkey0 < S;                           // (wave-uniform)
    float4 kraw[2][ST_KS][2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const float *kp = k_b + (size_t)min(wkey0 + 32 * t + c32, S - 1) * kv.ld + 8 * h;
#pragma unroll
        for (int ks = 0; ks < ST_KS; ks++) {
            kraw[t][ks][0] = *reinterpret_cast<const float4 *>(kp + 16 * ks);
            kraw[t][ks][1] = *reinterpret_cast<const float4 *>(kp + 16 * ks + 4);
        }
    }
    const int p0 = load_ptrs(0), p1 = load_ptrs(1), p2 = load_ptrs(2);
    int ptr_a = load_ptrs(3), ptr_b;                            // (row pointers of the next request)
    const float4 q0 = load_q(0), q1 = load_q(1);
    float4 q_a = load_q(2), q_b2;                               // (q_a: the stripe the coming iteration stores)
    StPicks pk_a = request(p0), pk_b = request(p1), pk_c = request(p2), pk_d;
    StFrag kf[2][ST_KS];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int ks = 0; ks < ST_KS; ks++) kf[t][ks] = st_split8(kraw[t][ks][0], kraw[t][ks][1]);
    store_q(0, q0);
    store_q(1, q1);
    __syncthreads();

    // Scores of stripe s -> stripe buffer s % 2, in two halves so that the wave's own LDS traffic
    // runs under its MFMAs: `fetch` requests the eight Q fragments (a wave that picks first does
    // that before picking), `scores` contracts the first key tile, starts the second, stores the
    // first while the second's MFMAs run, then stores the second.
    StFrag qf[ST_KS];
    auto fetch = [&](int s) {
#ifndef ST_ABL_NO_MMA
        const char *qh = qimg + (s & 1) * ST_QIMG + c32 * ST_QLD + 16 * h;
#pragma unroll
        for (int ks = 0; ks < ST_KS; ks++) {
            qf[ks].hi = *reinterpret_cast<const uint4 *>(qh + 32 * ks);
            qf[ks].lo = *reinterpret_cast<const uint4 *>(qh + ST_QPART + 32 * ks);
        }
#endif
    };
    auto scores = [&](int s) {
#ifndef ST_ABL_NO_MMA
        // lane (i, h) holds row i, keys 8 g + 4 h + u of its tile in register 4 g + u
        float *drow = dbuf + (s & 1) * ST_DBUF + c32 * ST_DLD + wave * ST_WKEYS + 4 * h;
        auto park = [&](const st_f32x16 &d, int t) {
#ifdef ST_ABL_NO_PARK
            asm volatile("" ::"v"(d));
            return;
#endif
#pragma unroll
            for (int g = 0; g < 4; g++)
                *reinterpret_cast<float4 *>(drow + 32 * t + 8 * g) =
                    make_float4(d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]);
        };
        st_f32x16 d0, d1;
#pragma unroll
        for (int r = 0; r < 16; r++) { d0[r] = 0.f; d1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < ST_KS; ks++) d0 = st_mm(kf[0][ks], qf[ks], d0);
        d1 = st_mm(kf[1][0], qf[0], d1);
        __builtin_amdgcn_sched_barrier(0);
        park(d0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 1; ks < ST_KS; ks++) d1 = st_mm(kf[1][ks], qf[ks], d1);
        park(d1, 1);
#endif
    };
    // the entries of stripe s out of stripe buffer s % 2
    auto pick = [&](int s, const StPicks &pk) {
#ifndef ST_ABL_NO_PICK
        const float *dcur = dbuf + (s & 1) * ST_DBUF + wave * ST_RPW * ST_DLD;
        auto value = [&](int rr, int col) {
            float v = dcur[rr * ST_DLD + (col - key0)] * scale;
            if (clampv > 0.0f) v = fminf(fmaxf(v, -clampv), clampv);
            return v;
        };
        auto mine = [&](int col) { return (unsigned)(col - key0) < (unsigned)ST_KEYS && col < S; };
        // the first 64 entries of the four rows: four LDS reads in flight, then four stores
        float v[ST_RPW];
        int off[ST_RPW];
#pragma unroll
        for (int rr = 0; rr < ST_RPW; rr++) {
            const int e = pk.start[rr] + lane;
            const bool take = e < pk.end[rr] && mine(pk.col[rr]);
            off[rr] = take ? e * 4 : -1;
            v[rr] = value(rr, take ? pk.col[rr] : key0);
        }
#pragma unroll
        for (int rr = 0; rr < ST_RPW; rr++)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[rr]), out_rsrc, off[rr], 0, 0);
        if (LONG) {
#pragma unroll
            for (int rr = 0; rr < ST_RPW; rr++) {
                for (int e0 = pk.start[rr] + SPT_WAVE; e0 < pk.end[rr]; e0 += SPT_WAVE) {
                    const int e = e0 + lane;
                    const int col = (int)__builtin_amdgcn_raw_buffer_load_b32(idx_rsrc, e < pk.end[rr] ? e * 4 : -1, 0, 0);
                    const bool take = e < pk.end[rr] && mine(col);
                    const float val = value(rr, take ? col : key0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), out_rsrc,
                                                          take ? e * 4 : -1, 0, 0);
                }
            }
        }
#endif
    };
    if (wave_live) {
        fetch(0);
        scores(0);
    }
    __syncthreads();
    ST_MARK();
    // Software pipeline, one barrier per iteration: iteration s forms the scores of stripe s + 1 and
    // picks the entries of stripe s.  The two waves of a SIMD (w and w + 4) take the two halves in
    // OPPOSITE order, so that one's MFMAs run under the other's loads, LDS reads and stores.
    // Global requests run ahead of their use: column ids three stripes (requested in iteration s
    // for stripe s + 3), Q rows loaded in iteration s for stripe s + 3 and written to the image at
    // the end of iteration s + 1, row pointers for stripe s + 4.  Unrolled four times by hand: the
    // register sets rotate by NAME -- rotated by copies, every copy was a wait for its load.
    const bool scores_first = wave < ST_WAVES / 2;
    auto iteration = [&](int s, const StPicks &cur, StPicks &fresh, const float4 &q_store, float4 &q_load,
                         const int ptr_cur, int &ptr_new) {
        ST_MARK();
        q_load = load_q(s + 3);
        ptr_new = load_ptrs(s + 4);
        fresh = request(ptr_cur);
        const bool more = s + 1 < n_stripes && wave_live;
        if (more) fetch(s + 1);
        if (scores_first) {
            if (more) scores(s + 1);
            ST_MARK();
            pick(s, cur);
        } else {
            pick(s, cur);
            ST_MARK();
            if (more) scores(s + 1);
        }
        ST_MARK();
        if (s + 2 < n_stripes) store_q(s + 2, q_store);
        ST_MARK();
        __syncthreads();
    };
    for (int s = 0; s < n_stripes; s += 4) {
        // (the set of stripe s, the set the request for stripe s + 3 goes into)
        iteration(s, pk_a, pk_d, q_a, q_b2, ptr_a, ptr_b);
        if (s + 1 >= n_stripes) break;
        iteration(s + 1, pk_b, pk_a, q_b2, q_a, ptr_b, ptr_a);
        if (s + 2 >= n_stripes) break;
        iteration(s + 2, pk_c, pk_b, q_a, q_b2, ptr_a, ptr_b);
        if (s + 3 >= n_stripes) break;
        iteration(s + 3, pk_d, pk_c, q_b2, q_a, ptr_b, ptr_a);
    }
}

// (one workgroup per CU = two waves per SIMD: up to 256 registers each)
__global__ __launch_bounds__(ST_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void sddmm_tile_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ query, const float *__restrict__ key, float *__restrict__ out, int S,
    int nnz, int key_blocks, float scale, float clampv, int q_heads, int k_heads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the longest row (indptr is shared by all slices)
    int longest = 0;
    for (int r = threadIdx.x; r < S; r += ST_THREADS) longest = max(longest, indptr[r + 1] - indptr[r]);
    longest = __syncthreads_or(longest > SPT_WAVE);
    if (!longest)
        sddmm_tile_body<false>(indptr, indices, query, key, out, S, nnz, key_blocks, scale, clampv, q_heads,
                               k_heads, smem);
    else
        sddmm_tile_body<true>(indptr, indices, query, key, out, S, nnz, key_blocks, scale, clampv, q_heads,
                              k_heads, smem);
}

// the shapes of this form; everything else is for the gather form (sddmm.hip)
bool sddmm_tile_takes(int B, int S, int E, int nnz) {
    // (S > 1024: every one of a slice's key blocks walks all of its column ids, and rows of S / 8
    // entries take the long-row loop -- 480 us against the gather form's 433 at S = 2048, Z = 256)
    if (E != ST_E || S < 2 * ST_WKEYS || S > 2 * ST_KEYS || B <= 0 || nnz <= 0) return false;
    // all S x S scores of a slice are computed: worth it from a density of 1 / 16 (lookup: 1 / 8)
    if ((long long)nnz * 16 < (long long)S * S) return false;
    const long long nblk = (long long)B * ((S + ST_KEYS - 1) / ST_KEYS);
    // (a workgroup needs ~30 us whatever the grid: below ~5/8 of the CUs the gather form's finer
    // split wins -- 64 workgroups: 29 us against 20.6)
    if (nblk < 160 || nblk > 0x7FFFFFFFll) return false;
    return !getenv("SPT_SDDMM_GATHER");                         // (A/B switch: tools/README.md)
}

// -> SPT_OK, or SPT_EUNSUP when the shape is one for the gather form
int sddmm_tile_launch(const int32_t *indptr, const int32_t *indices, const float *query,
                      const float *key, float *out, int B, int S, int E, int nnz, float scale,
                      float clampv, int q_heads, int k_heads, hipStream_t s) {
    if (!sddmm_tile_takes(B, S, E, nnz)) return SPT_EUNSUP;
    const int key_blocks = (S + ST_KEYS - 1) / ST_KEYS;
    const long long nblk = (long long)B * key_blocks;
    SPT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&sddmm_tile_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS));
    hipLaunchKernelGGL(sddmm_tile_kernel, dim3((unsigned)nblk), dim3(ST_THREADS), ST_LDS, s, indptr,
                       indices, query, key, out, S, nnz, key_blocks, scale, clampv, q_heads, k_heads);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

}  // namespace spt
