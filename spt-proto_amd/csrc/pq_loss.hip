// pq_loss.hip -- the PQ codebook training loss and its gradient as two kernels, gfx950.
//
// Reference: PQBase.forward(mode='train'), naive_gpt/layers/basic/quantizer.py:80-111,
// which the recipe runs for q and k of every layer on every step
// (script/4-sparse-tuning-0.py:71-78).  For each D-dim sub-vector z of subspace m, with
// W = weight[m] ([C, D]):
//
//   d_c  = sum_i |z_i - W_ci|                        (cdist, extension/cdist.cu:47-51)
//   best = argmin_c d_c  (strict '<', ascending c)   (cdist.cu:52-54)
//   zq   = W[best]                                   (gather, quantizer.py:86-90)
//   soft = softmax_c(-log max(d_c, 1e-5))            (quantizer.py:98-101)
//   zw   = sum_c soft_c W_c                          (quantizer.py:102)
//   loss = mean((zw - zq)^2) + mean((z - zq)^2)      (quantizer.py:105-110)
//
// The reference materialises z three times ([NQ,M,D] -> [M,NQ,D] copies), the
// [M,NQ,C] distance tensor four times (distance, clamp, log, softmax) and launches ~25
// kernels forward + backward; at BERT-large / S=512 / batch 16 that is 3.1 ms per layer
// per step, 3x everything else in the attention.  Here the forward reads z once and
// writes one float per block; the backward reads z once, recomputes the 16 distances of
// each sub-vector in registers and writes grad_z plus one [M,C,D] slab per block.
// HBM-bound: 4 B/element forward, 8 B/element backward.
//
// Layout: z is any contiguous tensor whose last dimension is M*D (the [N,S,H,E] head
// layout as well as [B,S,E]): sub-vector j starts at element j*D and belongs to subspace
// j % M.  The loss is a mean over all sub-vectors, so their order does not matter.
//
// Mapping: one lane per sub-vector, grid-stride with a stride that is a multiple of M so
// a lane's subspace never changes and its slice of grad_weight ([C][D] = 128 floats for
// D = 8) lives in registers for the whole kernel.  The codebook sits in LDS with a
// padded subspace stride (the 8 lanes of a token read the same codeword of 8 different
// subspaces: unpadded that is an 8-way bank conflict, see pq_encode_heads_kernel).
//
// Numerics: softmax(-log d) is evaluated as (dmin/d_c) / sum_c (dmin/d_c), which is what
// exp(-log d_c - max) is up to rounding; sums run in a different order than torch's.
// Parity is therefore "fp32 within 1e-3 rel" (tests/test_gpu_pq_loss.py), not bit-exact.
#include "spt_common.h"
#include <stdlib.h>

namespace spt {

constexpr int PQL_THREADS = 256;
constexpr int PQL_C = 16;

template <int D>
struct PQSub {
    float z[D];
    float d[PQL_C];     // raw L1 distances
    float soft[PQL_C];  // softmax(-log clamp(d))
    float zw[D];
    float zq[D];
    int best;
};

template <int D>
__device__ __forceinline__ void pq_sub_forward(const float *__restrict__ zp,
                                               const float *__restrict__ tm, PQSub<D> &s) {
#pragma unroll
    for (int i = 0; i < D / 4; i++) {
        const float4 v = reinterpret_cast<const float4 *>(zp)[i];
        s.z[4 * i + 0] = v.x; s.z[4 * i + 1] = v.y; s.z[4 * i + 2] = v.z; s.z[4 * i + 3] = v.w;
    }
    int best_i = 0;
    float best_d = 1e13f;
#pragma unroll
    for (int c = 0; c < PQL_C; c++) {
        const float4 *tp = reinterpret_cast<const float4 *>(tm + c * D);
        float r = 0.0f;
#pragma unroll
        for (int i = 0; i < D / 4; i++) {
            const float4 tv = tp[i];
            r += fabsf(s.z[4 * i + 0] - tv.x);
            r += fabsf(s.z[4 * i + 1] - tv.y);
            r += fabsf(s.z[4 * i + 2] - tv.z);
            r += fabsf(s.z[4 * i + 3] - tv.w);
        }
        s.d[c] = r;
        const bool cond = r < best_d;
        best_i = cond ? c : best_i;
        best_d = cond ? r : best_d;
    }
    s.best = best_i;
    const float dmin = fmaxf(best_d, 1e-5f);
    float wsum = 0.0f;
#pragma unroll
    for (int c = 0; c < PQL_C; c++) {
        s.soft[c] = dmin * __builtin_amdgcn_rcpf(fmaxf(s.d[c], 1e-5f));   // 1 ulp
        wsum += s.soft[c];
    }
    const float inv = __builtin_amdgcn_rcpf(wsum);
#pragma unroll
    for (int i = 0; i < D; i++) s.zw[i] = 0.0f;
#pragma unroll
    for (int c = 0; c < PQL_C; c++) {
        s.soft[c] *= inv;
        const float4 *tp = reinterpret_cast<const float4 *>(tm + c * D);
#pragma unroll
        for (int i = 0; i < D / 4; i++) {
            const float4 tv = tp[i];
            s.zw[4 * i + 0] = fmaf(s.soft[c], tv.x, s.zw[4 * i + 0]);
            s.zw[4 * i + 1] = fmaf(s.soft[c], tv.y, s.zw[4 * i + 1]);
            s.zw[4 * i + 2] = fmaf(s.soft[c], tv.z, s.zw[4 * i + 2]);
            s.zw[4 * i + 3] = fmaf(s.soft[c], tv.w, s.zw[4 * i + 3]);
        }
    }
    const float4 *bp = reinterpret_cast<const float4 *>(tm + best_i * D);
#pragma unroll
    for (int i = 0; i < D / 4; i++) {
        const float4 tv = bp[i];
        s.zq[4 * i + 0] = tv.x; s.zq[4 * i + 1] = tv.y; s.zq[4 * i + 2] = tv.z; s.zq[4 * i + 3] = tv.w;
    }
}

__device__ __forceinline__ void pq_stage_table(const float *__restrict__ table, float *tab,
                                               int M, int CD, int tstride) {
    for (int i = threadIdx.x; i < M * CD; i += PQL_THREADS) {
        const int mm = i / CD;
        tab[mm * tstride + (i - mm * CD)] = table[i];
    }
    __syncthreads();
}

// partial[block] = sum over the block's sub-vectors of |zw - zq|^2 + |z - zq|^2
template <int D>
__global__ __launch_bounds__(PQL_THREADS) void pq_loss_forward_kernel(
    const float *__restrict__ z, const float *__restrict__ table, float *__restrict__ partial,
    int total, int M, int32_t *__restrict__ codes, int S, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tab = reinterpret_cast<float *>(smem);
    constexpr int CD = PQL_C * D;
    const int tstride = CD + 4;
    pq_stage_table(table, tab, M, CD, tstride);
    __shared__ float wsum[PQL_THREADS / 64];

    const int first = blockIdx.x * PQL_THREADS + threadIdx.x;
    const float *tm = tab + (first % M) * tstride;
    float acc = 0.0f;
    for (int j = first; j < total; j += gridDim.x * PQL_THREADS) {
        PQSub<D> s;
        pq_sub_forward<D>(z + (size_t)j * D, tm, s);
        if (codes) {
            // the argmin IS the PQ code (same distances, same scan as pq_encode_heads_kernel):
            // z [N, S, H, M D] -> codes [N H, S, M], the input of lookup
            const int vec = j / M, m = j - vec * M;
            const int ns = vec / H, h = vec - ns * H;
            const int n = ns / S, sq = ns - n * S;
            codes[((size_t)(n * H + h) * S + sq) * M + m] = s.best;
        }
#pragma unroll
        for (int i = 0; i < D; i++) {
            const float e1 = s.zw[i] - s.zq[i];
            const float e2 = s.z[i] - s.zq[i];
            acc = fmaf(e1, e1, acc);
            acc = fmaf(e2, e2, acc);
        }
    }
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < PQL_THREADS / 64; w++) t += wsum[w];
        partial[blockIdx.x] = t;
    }
}

__global__ void pq_loss_finish_kernel(const float *__restrict__ partial, float *__restrict__ loss,
                                      int nblk, float inv_count) {
    // one wave, fixed order: deterministic
    float acc = 0.0f;
    for (int b = threadIdx.x; b < nblk; b += 64) acc += partial[b];
    acc = group_sum<64>(acc);
    if (threadIdx.x == 0) loss[0] = acc * inv_count;
}

// grad_z and one [M][C][D] slab of grad_weight per block.
//
// NL lanes per sub-vector (a DPP quad holds one or two sub-vectors): lane `part` owns the
// codewords {CL * part .. CL * part + CL - 1}, CL = 16 / NL, i.e. CL * D accumulators of
// grad_weight instead of 16 * D (which, with the loop temporaries, does not fit a 256-VGPR
// budget: 340 registers + scratch, one wave per SIMD, measured 153 us at the BASELINE size).
// The lanes of a sub-vector exchange through DPP quad permutations: the argmin, the softmax
// denominator, the partial zw and <soft, gs>, and the pieces of grad_z.
//   NL = 2 (rounds 1-2): 64 accumulators, ~190 VGPRs, two waves per SIMD: 65 us -- a quarter of
//   the VALU rate its ~1,250 instructions per sub-vector need: one wave's dependent chains and LDS
//   latencies with one other wave to cover them.
//   NL = 4 (round 3): 32 accumulators, four waves per SIMD; the per-lane duplicates (z, gzw, the
//   quad sums) cost ~15 % more instructions.
__device__ __forceinline__ float quad_x1(float v) { return dpp_mov<0xB1>(v); }      // lane ^ 1
__device__ __forceinline__ float quad_x2(float v) { return dpp_mov<0x4E>(v); }      // lane ^ 2
__device__ __forceinline__ int quad_x1(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
}
__device__ __forceinline__ int quad_x2(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
}
// floats from one lane's codewords to the next lane's in the backward kernel's LDS table
__host__ __device__ constexpr int pq_part_stride(int D, int NL) {
    return (PQL_C / NL) * D + ((NL == 2 && D == 8) ? 32 : 0);
}
typedef float pq_v2f __attribute__((ext_vector_type(2)));
// +1 or -1 with the sign bit of e (one v_bfi_b32)
__device__ __forceinline__ float pq_copysign1(float e) {
    return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, e) & 0x80000000u) | 0x3f800000u);
}
template <int NL>
__device__ __forceinline__ float sub_sum(float v) {          // sum over the NL lanes of a sub-vector
    v += quad_x1(v);
    if (NL == 4) v += quad_x2(v);
    return v;
}

// WREG: the lane's CL codewords live in REGISTERS for the whole kernel (its subspace never changes):
// with them in LDS every sub-vector cost each lane 4 passes of ds_read_b128 over its codewords --
// 66 reads per lane and sub-vector, two-way bank-conflicted (the two halves of a subspace lie 256
// bytes apart): ~28 us of LDS time per launch at the bench shape, serialised with the arithmetic
// by the two-waves-per-SIMD occupancy.  Only the best codeword (a data-dependent row) is still
// read from LDS.
template <int D, int NL, bool WREG>
__global__ __launch_bounds__(PQL_THREADS, NL == 4 ? 3 : 2) void pq_loss_backward_kernel(
    const float *__restrict__ z, const float *__restrict__ table,
    const float *__restrict__ grad_loss, float *__restrict__ grad_z,
    float *__restrict__ partial, int total, int M, float inv_count, int accumulate) {
    constexpr int CL = PQL_C / NL;                  // codewords per lane
    constexpr int DL = D / NL;                      // elements of grad_z a lane stores
    static_assert(NL == 2 || NL == 4, "two or four lanes per sub-vector");
    static_assert(DL == 1 || DL == 2 || DL == 4, "grad_z pieces of 4, 8 or 16 bytes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tab = reinterpret_cast<float *>(smem);
    constexpr int CD = PQL_C * D;
    // the table in LDS: subspace m at m * tstride, the CL codewords of lane `part` at part * PSTRIDE.
    // The 16 lanes of one ds_read_b128 group are 16 / NL sub-vectors of consecutive subspaces; with
    // the parts 256 bytes apart (D = 8, two lanes) both lanes of a sub-vector read the same banks:
    // every codeword read two-way conflicted, 66 reads per lane and sub-vector.  96 floats apart
    // (32 banks) and subspaces 4 banks apart, the 16 addresses of a group fall on 16 different
    // four-bank windows.
    constexpr int PSTRIDE = pq_part_stride(D, NL);
    constexpr int tstride = NL * PSTRIDE + 4;
    for (int i = threadIdx.x; i < M * CD; i += PQL_THREADS) {
        const int mm = i / CD, e = i - mm * CD;
        const int c = e / D;
        tab[mm * tstride + (c / CL) * PSTRIDE + (c % CL) * D + (e - c * D)] = table[i];
    }
    __syncthreads();
    float *red = tab + M * tstride;  // [waves][M][CD]

    const int gtid = blockIdx.x * PQL_THREADS + threadIdx.x;
    const int part = gtid & (NL - 1);
    const int first = gtid / NL;
    const int m = first % M;
    const float *tsub = tab + m * tstride;          // the whole subspace (for zq)
    const float *tm = tsub + part * PSTRIDE;        // this lane's codewords
    const float g = grad_loss[0] * inv_count;

    pq_v2f gt[CL][D / 2];
#pragma unroll
    for (int c = 0; c < CL; c++)
#pragma unroll
        for (int p = 0; p < D / 2; p++) gt[c][p] = pq_v2f{0.0f, 0.0f};

    float4 wr[WREG ? CL : 1][D / 4];
    if constexpr (WREG) {
#pragma unroll
        for (int c = 0; c < CL; c++)
#pragma unroll
            for (int i = 0; i < D / 4; i++) wr[c][i] = reinterpret_cast<const float4 *>(tm + c * D)[i];
    }
    // codeword c of this lane, float4 i: from the registers or from LDS
    auto cw = [&](int c, int i) -> float4 {
        if constexpr (WREG) return wr[c][i];
        else return reinterpret_cast<const float4 *>(tm + c * D)[i];
    };
    const int stride = gridDim.x * (PQL_THREADS / NL);
    float4 znext[D / 4];
    if (first < total) {
#pragma unroll
        for (int i = 0; i < D / 4; i++)
            znext[i] = reinterpret_cast<const float4 *>(z + (size_t)first * D)[i];
    }
#ifdef PQ_ABL_ONE_ITERATION
    total = min(total, stride);
#endif
    for (int j = first; j < total; j += stride) {
        // element pairs (2 p, 2 p + 1) as one 64-bit operand: the products and sums below are
        // v_pk_fma_f32 / v_pk_add_f32, two lanes of arithmetic per issued instruction
        pq_v2f zp[D / 2];
#pragma unroll
        for (int i = 0; i < D / 4; i++) {
            zp[2 * i + 0] = pq_v2f{znext[i].x, znext[i].y};
            zp[2 * i + 1] = pq_v2f{znext[i].z, znext[i].w};
        }
        if (j + stride < total) {   // next sub-vector: in flight during this one's arithmetic
#pragma unroll
            for (int i = 0; i < D / 4; i++)
                znext[i] = reinterpret_cast<const float4 *>(z + (size_t)(j + stride) * D)[i];
        }
        if constexpr (!WREG) asm volatile("" ::: "memory");   // keep the codebook in LDS, not in registers
        // codeword c of this lane as D / 2 pairs
        auto cwp = [&](int c, int p) -> pq_v2f {
            const float4 tv = cw(c, p >> 1);
            return (p & 1) ? pq_v2f{tv.z, tv.w} : pq_v2f{tv.x, tv.y};
        };
        // ---- forward, recomputed: distances, argmin, soft assignment, zw, zq ----------
        float d[CL], soft[CL];
        int best_i = 0;
        float best_d = 1e13f;
#pragma unroll
        for (int c = 0; c < CL; c++) {
            float r = 0.0f;                 // (the forward kernel's order of additions: same argmin)
#pragma unroll
            for (int p = 0; p < D / 2; p++) {
                const pq_v2f e = zp[p] - cwp(c, p);
                r += fabsf(e.x);
                r += fabsf(e.y);
            }
            d[c] = r;
            const bool cond = r < best_d;
            best_i = cond ? (part * CL + c) : best_i;
            best_d = cond ? r : best_d;
        }
        {   // smaller distance, then smaller index: the sequential strict-'<' scan's pick
            auto merge = [&](float od, int oi) {
                const bool take = (od < best_d) || (od == best_d && oi < best_i);
                best_d = take ? od : best_d;
                best_i = take ? oi : best_i;
            };
            merge(quad_x1(best_d), quad_x1(best_i));
            if (NL == 4) merge(quad_x2(best_d), quad_x2(best_i));
        }
        // 1 / max(d, 1e-5) by v_rcp_f32 (1 ulp): used for the soft assignment and again
        // for d(-log d)/dd; sixteen IEEE divisions were a third of the kernel
        const float dmin = fmaxf(best_d, 1e-5f);
        float rd[CL];
        float wsum = 0.0f;
#pragma unroll
        for (int c = 0; c < CL; c++) {
            rd[c] = __builtin_amdgcn_rcpf(fmaxf(d[c], 1e-5f));
            soft[c] = dmin * rd[c];
            wsum += soft[c];
        }
        wsum = sub_sum<NL>(wsum);
        const float inv = __builtin_amdgcn_rcpf(wsum);
        pq_v2f zw[D / 2];
#pragma unroll
        for (int p = 0; p < D / 2; p++) zw[p] = pq_v2f{0.0f, 0.0f};
        if constexpr (!WREG) asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < CL; c++) {
            soft[c] *= inv;
            const pq_v2f sc = pq_v2f{soft[c], soft[c]};
#pragma unroll
            for (int p = 0; p < D / 2; p++) zw[p] = __builtin_elementwise_fma(sc, cwp(c, p), zw[p]);
        }
        pq_v2f gzw[D / 2], hard[D / 2], gz[D / 2];
        {
            const float4 *bp = reinterpret_cast<const float4 *>(tsub + (best_i / CL) * PSTRIDE + (best_i % CL) * D);
            const float g2 = 2.0f * g;
#pragma unroll
            for (int p = 0; p < D / 2; p++) {
                const float4 tv = bp[p >> 1];
                const pq_v2f zq = (p & 1) ? pq_v2f{tv.z, tv.w} : pq_v2f{tv.x, tv.y};
                const pq_v2f zws = pq_v2f{sub_sum<NL>(zw[p].x), sub_sum<NL>(zw[p].y)};
                gzw[p] = g2 * (zws - zq);
                const pq_v2f e2 = g2 * (zp[p] - zq);         // d/dz of the second term
                hard[p] = -gzw[p] - e2;                      // d/dzq of both terms -> W[best]
                gz[p] = part == 0 ? e2 : pq_v2f{0.0f, 0.0f}; // counted once per sub-vector
            }
        }
        // ---- softmax backward: ga_c = soft_c (gs_c - <soft, gs>), gs_c = <gzw, W_c> ----
        if constexpr (!WREG) asm volatile("" ::: "memory");
        float gs[CL];
        float dot = 0.0f;
#pragma unroll
        for (int c = 0; c < CL; c++) {
            pq_v2f r2 = gzw[0] * cwp(c, 0);
#pragma unroll
            for (int p = 1; p < D / 2; p++) r2 = __builtin_elementwise_fma(gzw[p], cwp(c, p), r2);
            gs[c] = r2.x + r2.y;
            dot = fmaf(soft[c], gs[c], dot);
        }
        dot = sub_sum<NL>(dot);
        if constexpr (!WREG) asm volatile("" ::: "memory");
#ifndef PQ_ABL_NO_PASS4
#pragma unroll
        for (int c = 0; c < CL; c++) {
            // a = -log(max(d, 1e-5)): da/dd = -1/d where d >= 1e-5 (torch.clamp passes the
            // gradient at the boundary), 0 below
            const float ga = soft[c] * (gs[c] - dot);
            const float gd = (d[c] >= 1e-5f) ? -ga * rd[c] : 0.0f;
            const pq_v2f gd2 = pq_v2f{gd, gd}, ngd2 = pq_v2f{-gd, -gd};
            const pq_v2f sc = pq_v2f{soft[c], soft[c]};
            // (one fma per element with a 0 / 1 factor instead of a select and an add)
            const float is_best = (part * CL + c == best_i) ? 1.0f : 0.0f;
            const pq_v2f ib = pq_v2f{is_best, is_best};
#pragma unroll
            for (int p = 0; p < D / 2; p++) {
                // cdist backward, extension/cdist.cu:113-119,167-174: sg = (z - w > 0) ? gd : -gd.
                // With e = w - z and s = copysign(1, e) (one v_bfi_b32; w == z gives e = +0: s = +1,
                // the "not greater" side) sg = -s gd exactly, so both accumulations are ONE fma each:
                //   gz += sg = fma(s, -gd, gz),   gt -= sg = fma(s, gd, gt)
                const pq_v2f e = cwp(c, p) - zp[p];
                const pq_v2f sgn = pq_v2f{pq_copysign1(e.x), pq_copysign1(e.y)};
                gz[p] = __builtin_elementwise_fma(sgn, ngd2, gz[p]);
                pq_v2f t = __builtin_elementwise_fma(sc, gzw[p], gt[c][p]);
                t = __builtin_elementwise_fma(ib, hard[p], t);
                gt[c][p] = __builtin_elementwise_fma(sgn, gd2, t);
            }
        }
#else
        gz[0].x += dot + gs[0];
#endif
        // grad_z: lane `part` of the sub-vector stores elements DL part .. DL part + DL - 1
        float gzs[D];
#pragma unroll
        for (int p = 0; p < D / 2; p++) {
            gzs[2 * p] = sub_sum<NL>(gz[p].x);
            gzs[2 * p + 1] = sub_sum<NL>(gz[p].y);
        }
        float *gp = grad_z + (size_t)j * D + part * DL;
        float o[DL];
#pragma unroll
        for (int e = 0; e < DL; e++) {
            // (a select chain over the lane's part: no dynamically indexed register array)
            float v = gzs[e];
#pragma unroll
            for (int q = 1; q < NL; q++) v = part == q ? gzs[q * DL + e] : v;
            o[e] = v;
        }
        // accumulate: grad_z already holds the gradient that reached z over another path (the
        // attention's grad_q / grad_k): added here instead of by an elementwise pass of its own
        if constexpr (DL == 4) {
            float4 v = make_float4(o[0], o[1], o[2], o[3]);
            if (accumulate) {
                const float4 b = *reinterpret_cast<const float4 *>(gp);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            *reinterpret_cast<float4 *>(gp) = v;
        } else if constexpr (DL == 2) {
            float2 v = make_float2(o[0], o[1]);
            if (accumulate) {
                const float2 b = *reinterpret_cast<const float2 *>(gp);
                v.x += b.x; v.y += b.y;
            }
            *reinterpret_cast<float2 *>(gp) = v;
        } else {
            *gp = accumulate ? o[0] + *gp : o[0];
        }
    }

    // lanes l, l + NL M, l + 2 NL M .. of a wave hold the same (subspace, part): butterfly over
    // those, then the waves of the block through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int step = NL * M; step < 64; step <<= 1) {
#pragma unroll
        for (int c = 0; c < CL; c++)
#pragma unroll
            for (int p = 0; p < D / 2; p++) {
                gt[c][p].x += __shfl_xor(gt[c][p].x, step, 64);
                gt[c][p].y += __shfl_xor(gt[c][p].y, step, 64);
            }
    }
    if (lane < NL * M) {
        float *dst = red + ((size_t)wave * M + m) * CD + part * CL * D;
#pragma unroll
        for (int c = 0; c < CL; c++)
#pragma unroll
            for (int p = 0; p < D / 2; p++) {
                dst[c * D + 2 * p] = gt[c][p].x;
                dst[c * D + 2 * p + 1] = gt[c][p].y;
            }
    }
    __syncthreads();
    float *out = partial + (size_t)blockIdx.x * M * CD;
    for (int e = threadIdx.x; e < M * CD; e += PQL_THREADS) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < PQL_THREADS / 64; w++) t += red[(size_t)w * M * CD + e];
        out[e] = t;
    }
}

// grad_table[e] = sum_b partial[b][e]: 64 elements x 16 slab groups per block, fixed order
__global__ __launch_bounds__(1024) void pq_loss_table_reduce_kernel(
    const float *__restrict__ partial, float *__restrict__ grad_table, int nblk, int elems) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    float acc = 0.0f;
    if (e < elems) {
        // four loads in flight per thread (one dependent chain of 32 loads was the kernel's time)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int b = grp;
        for (; b + 48 < nblk; b += 64) {
            a0 += partial[(size_t)b * elems + e];
            a1 += partial[(size_t)(b + 16) * elems + e];
            a2 += partial[(size_t)(b + 32) * elems + e];
            a3 += partial[(size_t)(b + 48) * elems + e];
        }
        for (; b < nblk; b += 16) a0 += partial[(size_t)b * elems + e];
        acc = (a0 + a1) + (a2 + a3);
    }
    red[grp][lane] = acc;
    __syncthreads();
    if (grp == 0 && e < elems) {
        float t = 0.0f;
#pragma unroll
        for (int g = 0; g < 16; g++) t += red[g][lane];
        grad_table[e] = t;
    }
}

static bool pq_loss_shape_ok(long long n_vectors, int M, int C, int D) {
    if (C != PQL_C || (D != 4 && D != 8)) return false;
    if (M <= 0 || M > 32 || (M & (M - 1)) != 0) return false;
    return n_vectors * M < 0x7FFFFFFFLL - 4 * 1024 * 1024;
}

static int pq_loss_blocks(long long total, int cap = 512) {
    long long nblk = (total + PQL_THREADS - 1) / PQL_THREADS;
    if (nblk > cap) nblk = cap;  // `cap / 256` blocks per CU; the grid stride cap * 256 / NL is a multiple of M
    return (int)nblk;
}
// lanes per sub-vector of the backward (pq_loss_backward_kernel): 2; SPT_PQ_LANES=4 runs the
// four-lane form with the codewords in registers (round 3: measured 80-87 us against 74 -- the
// kernel is bound by the instructions it issues, not by occupancy or LDS, and four lanes issue
// 15 % more); the workspace is sized for either
static int pq_backward_lanes() {
    static int lanes = 0;
    if (lanes == 0) {
        const char *e = getenv("SPT_PQ_LANES");
        lanes = (e && e[0] == '4') ? 4 : 2;
    }
    return lanes;
}
static int pq_backward_blocks(long long total) {
    const int nl = pq_backward_lanes();
    return pq_loss_blocks((long long)nl * total, nl == 4 ? 768 : 512);
}

}  // namespace spt

using namespace spt;

extern "C" int64_t spt_pq_loss_workspace_bytes(int64_t n_vectors, int n_subspaces,
                                               int n_codewords, int d_code) {
    if (n_vectors <= 0 || !pq_loss_shape_ok(n_vectors, n_subspaces, n_codewords, d_code)) return 0;
    const int nblk = pq_loss_blocks(4 * n_vectors * n_subspaces, 1024);   // backward: upper bound
    return (int64_t)nblk * n_subspaces * n_codewords * d_code * (int64_t)sizeof(float);
}

// `parts` tensors of n_vectors rows each, back to back in z (q and k of one attention): one pass
// over all of them; every part's loss is a mean over ITS sub-vectors and `loss` their sum
static int pq_loss_forward_any(const float *z, const float *table, float *loss, void *workspace,
                               int64_t n_vectors, int parts, int n_subspaces, int n_codewords, int d_code,
                               int32_t *codes, int seq_length, int n_heads, void *stream) {
    if (!z || !table || !loss || !workspace) return SPT_EINVAL;
    if (n_vectors <= 0 || parts <= 0 || parts > 4 || n_subspaces <= 0 || n_codewords <= 0 || d_code <= 0)
        return SPT_EINVAL;
    if (!pq_loss_shape_ok(n_vectors * parts, n_subspaces, n_codewords, d_code)) return SPT_EUNSUP;
    if (codes && (seq_length <= 0 || n_heads <= 0 || n_vectors % ((int64_t)seq_length * n_heads) != 0))
        return SPT_ESHAPE;
    const int total = (int)(n_vectors * parts * n_subspaces);
    const int nblk = pq_loss_blocks(total);
    const size_t lds = (size_t)n_subspaces * (n_codewords * d_code + 4) * sizeof(float);
    const float inv_count = 1.0f / ((float)(n_vectors * n_subspaces) * (float)d_code);
    float *partial = reinterpret_cast<float *>(workspace);
    hipStream_t s = (hipStream_t)stream;
    if (d_code == 4)
        hipLaunchKernelGGL((pq_loss_forward_kernel<4>), dim3(nblk), dim3(PQL_THREADS), lds, s, z,
                           table, partial, total, n_subspaces, codes, seq_length, n_heads);
    else
        hipLaunchKernelGGL((pq_loss_forward_kernel<8>), dim3(nblk), dim3(PQL_THREADS), lds, s, z,
                           table, partial, total, n_subspaces, codes, seq_length, n_heads);
    SPT_LAUNCH_CHECK();
    hipLaunchKernelGGL(pq_loss_finish_kernel, dim3(1), dim3(64), 0, s, partial, loss, nblk,
                       inv_count);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_pq_loss_forward(const float *z, const float *table, float *loss,
                                   void *workspace, int64_t n_vectors, int n_subspaces,
                                   int n_codewords, int d_code, void *stream) {
    return pq_loss_forward_any(z, table, loss, workspace, n_vectors, 1, n_subspaces, n_codewords,
                               d_code, nullptr, 0, 0, stream);
}

extern "C" int spt_pq_loss_forward_codes(const float *z, const float *table, float *loss,
                                         void *workspace, int32_t *codes, int batch,
                                         int seq_length, int n_heads, int n_subspaces,
                                         int n_codewords, int d_code, void *stream) {
    if (!codes || batch <= 0 || seq_length <= 0 || n_heads <= 0) return SPT_EINVAL;
    return pq_loss_forward_any(z, table, loss, workspace, (int64_t)batch * seq_length * n_heads, 1,
                               n_subspaces, n_codewords, d_code, codes, seq_length, n_heads, stream);
}

extern "C" int spt_pq_loss_forward_codes_parts(const float *z, const float *table, float *loss,
                                               void *workspace, int32_t *codes, int parts, int batch,
                                               int seq_length, int n_heads, int n_subspaces,
                                               int n_codewords, int d_code, void *stream) {
    if (!codes || batch <= 0 || seq_length <= 0 || n_heads <= 0) return SPT_EINVAL;
    return pq_loss_forward_any(z, table, loss, workspace, (int64_t)batch * seq_length * n_heads, parts,
                               n_subspaces, n_codewords, d_code, codes, seq_length, n_heads, stream);
}

static int pq_loss_backward_any(const float *z, const float *table, const float *grad_loss,
                                float *grad_z, float *grad_table, void *workspace,
                                int64_t n_vectors, int parts, int n_subspaces, int n_codewords,
                                int d_code, int accumulate, void *stream);

extern "C" int spt_pq_loss_backward(const float *z, const float *table, const float *grad_loss,
                                    float *grad_z, float *grad_table, void *workspace,
                                    int64_t n_vectors, int n_subspaces, int n_codewords,
                                    int d_code, int accumulate, void *stream) {
    return pq_loss_backward_any(z, table, grad_loss, grad_z, grad_table, workspace, n_vectors, 1,
                                n_subspaces, n_codewords, d_code, accumulate, stream);
}

extern "C" int spt_pq_loss_backward_parts(const float *z, const float *table, const float *grad_loss,
                                          float *grad_z, float *grad_table, void *workspace,
                                          int64_t n_vectors, int parts, int n_subspaces,
                                          int n_codewords, int d_code, int accumulate, void *stream) {
    return pq_loss_backward_any(z, table, grad_loss, grad_z, grad_table, workspace, n_vectors, parts,
                                n_subspaces, n_codewords, d_code, accumulate, stream);
}

static int pq_loss_backward_any(const float *z, const float *table, const float *grad_loss,
                                float *grad_z, float *grad_table, void *workspace,
                                int64_t n_vectors, int parts, int n_subspaces, int n_codewords,
                                int d_code, int accumulate, void *stream) {
    if (!z || !table || !grad_loss || !grad_z || !grad_table || !workspace) return SPT_EINVAL;
    if (n_vectors <= 0 || parts <= 0 || parts > 4 || n_subspaces <= 0 || n_codewords <= 0 || d_code <= 0)
        return SPT_EINVAL;
    if (!pq_loss_shape_ok(n_vectors * parts, n_subspaces, n_codewords, d_code)) return SPT_EUNSUP;
    const int total = (int)(n_vectors * parts * n_subspaces);
    const int nblk = pq_backward_blocks(total);
    const bool four = pq_backward_lanes() == 4;
    const int CD = n_codewords * d_code;
    const size_t lds = ((size_t)n_subspaces * ((four ? 4 : 2) * pq_part_stride(d_code, four ? 4 : 2) + 4) +
                        (size_t)(PQL_THREADS / 64) * n_subspaces * CD) * sizeof(float);
    if (lds > 64 * 1024) return SPT_EUNSUP;
    const float inv_count = 1.0f / ((float)(n_vectors * n_subspaces) * (float)d_code);   // (a mean per part)
    float *partial = reinterpret_cast<float *>(workspace);
    hipStream_t s = (hipStream_t)stream;
    if (d_code == 4)
        if (four)
            hipLaunchKernelGGL((pq_loss_backward_kernel<4, 4, true>), dim3(nblk), dim3(PQL_THREADS), lds, s, z,
                               table, grad_loss, grad_z, partial, total, n_subspaces, inv_count, accumulate);
        else
            hipLaunchKernelGGL((pq_loss_backward_kernel<4, 2, false>), dim3(nblk), dim3(PQL_THREADS), lds, s, z,
                               table, grad_loss, grad_z, partial, total, n_subspaces, inv_count, accumulate);
    else if (four)
        hipLaunchKernelGGL((pq_loss_backward_kernel<8, 4, true>), dim3(nblk), dim3(PQL_THREADS), lds, s, z,
                           table, grad_loss, grad_z, partial, total, n_subspaces, inv_count, accumulate);
    else
        hipLaunchKernelGGL((pq_loss_backward_kernel<8, 2, false>), dim3(nblk), dim3(PQL_THREADS), lds, s, z,
                           table, grad_loss, grad_z, partial, total, n_subspaces, inv_count, accumulate);
    SPT_LAUNCH_CHECK();
    const int elems = n_subspaces * CD;
    hipLaunchKernelGGL(pq_loss_table_reduce_kernel, dim3((elems + 63) / 64), dim3(1024), 0, s,
                       partial, grad_table, nblk, elems);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
