// fused_attention.hip -- the forward of the sparse attention core as ONE launch (gfx950).
//
// Reference: SparseVanillaAttentionV2._get_attn / _apply_attn
// (naive_gpt/layers/sparse/attention.py:106-142): sddmm -> scale, clamp -> softmax -> spmm,
// four launches and three [B, S*Z] round trips through HBM (the authors'
// own abandoned attempt is legacy/sparse_mha.cu).  Here one workgroup owns a (sample, head)
// slice for the whole chain:
//
//   phase A   K slice -> LDS; per CSR row: 64 dot products (gather4.h quad mapping), scale,
//             clamp, masked exp, row sum over the 16 lanes that hold the row, normalise --
//             all in registers; the clamped scores and the probabilities are each written
//             once (the backward needs both: softmax VJP and the clamp mask)
//   phase B   V slice -> the same LDS tile; y rows = sum_p P[row, p] V[col(p)] with the
//             probabilities read back from L2 (this workgroup wrote them microseconds ago)
//   layout    y is written either as [B, S, E] or -- y_transposed -- as [B, E, S], the
//             memory layout the reference's `y.transpose(1, 2).contiguous()` produces
//             (DESIGN.md "Reference quirks" 4), through an LDS staging buffer of 64 rows so
//             that the stores stay 256-byte contiguous.
//
// Shapes: E == 64, uniform rows of Z = nnz / S <= 64 entries with Z % 4 == 0 (what lookup
// produces: row r owns entries [r Z, (r+1) Z)), S * E * 4 <= 128 KiB.  Everything else goes
// through the separate operators.  Numerics are those of the separate kernels: the same
// fmaf order in the dot products, expf, 1e-9 denominator clamp (softmax.cu:30), the same
// accumulation order in the product; only the association of the launches changed.
#include "gather4.h"

namespace spt {

// 512 threads: two waves per SIMD with a 256-VGPR budget.  At 1024 threads (128 VGPRs) the
// backward kernel spilled 150-180 bytes per lane inside its row loop: 163 us against 123 us;
// the forward does not care (124 vs 120 us).
#ifndef FA_THREADS_VALUE
#define FA_THREADS_VALUE 512
#endif
constexpr int FA_THREADS = FA_THREADS_VALUE;
constexpr int FA_OROWS = 64;             // rows per transposed write-out phase
constexpr int FA_OLD = 64 + 4;           // padded row of the staging buffer (16-byte aligned)

// One chunk of FA_OROWS = 64 rows of a [S][64] slice = 1024 float4: FA_NPT per thread.
constexpr int FA_NPT = 1024 / FA_THREADS;
constexpr int FA_NW = FA_THREADS / SPT_WAVE;      // waves per workgroup
constexpr int FA_GPS = 16 / FA_NW;                // 4-row groups per wave per 64-row sweep
struct Chunk { float4 v[FA_NPT]; };
struct ChunkLoader {
    const float *src;     // slice base
    int ld, S, row, c4;   // this thread's first row inside a chunk and its float4 column
    __device__ __forceinline__ ChunkLoader(const float *base, int ld_, int S_, int tid)
        : src(base), ld(ld_), S(S_), row(tid >> 4), c4(tid & 15) {}
    __device__ __forceinline__ Chunk load(int chunk) const {
        Chunk c;
#pragma unroll
        for (int u = 0; u < FA_NPT; u++) {
            const int r = chunk * FA_OROWS + row + (FA_THREADS / 16) * u;
            c.v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < S) c.v[u] = *reinterpret_cast<const float4 *>(src + (size_t)r * ld + 4 * c4);
        }
        return c;
    }
    __device__ __forceinline__ void store(float *tile, int chunk, const Chunk &c) const {
#pragma unroll
        for (int u = 0; u < FA_NPT; u++) {
            const int r = chunk * FA_OROWS + row + (FA_THREADS / 16) * u;
            if (r < S) *reinterpret_cast<float4 *>(tile + (size_t)r * 64 + 4 * c4) =
                make_float4(c.v[u].x, c.v[u].y, c.v[u].z, c.v[u].w);
        }
    }
};

// out rows = sum_p val[row, p] * X[col(p)] for uniform rows, X staged into `tile` (whole, or
// CAUSAL: 64 rows at a time two chunks ahead of the rows that use them).  Output rows go to
// out_b + row * ld_out, or (YT) to out_b[e * S + row] through the 64-row LDS transposer obuf.
// All FA_THREADS threads of the workgroup must call it; on entry nobody may still read `tile`.
template <bool YT, bool CAUSAL>
__device__ __forceinline__ void product_phase(float *tile, float *obuf,
                                              const float *__restrict__ x_base, int ld_x,
                                              const int32_t *__restrict__ idx_b,
                                              const float *__restrict__ val_b,
                                              float *__restrict__ out_b, int ld_out, int S, int Z) {
    constexpr int LPE = 4, E = 64, R = 4;
    constexpr int NW = FA_NW;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nphases = (S + FA_OROWS - 1) / FA_OROWS;
    const ChunkLoader xload(x_base, ld_x, S, tid);
    Chunk pre = {};
    if (CAUSAL) {
        xload.store(tile, 0, xload.load(0));
        pre = xload.load(1);
    } else {
        stage_rows(tile, x_base, ld_x, S, E, tid, FA_THREADS);
    }
    __syncthreads();
    const Lane4<LPE> L;
    const int ngroups = (S + R - 1) / R;
    const int eoff = 16 * L.t + 4 * L.qs;
    const bool have = eoff < Z;
    struct PSeg { int4 ix; float4 pv; };
    auto load_seg4 = [&](int g) {
        PSeg sg;
        sg.ix = make_int4(0, 0, 0, 0);
        sg.pv = make_float4(0.f, 0.f, 0.f, 0.f);
        const int row = g * R + L.j;
        if (g < ngroups && row < S && have) {
            const size_t at = (size_t)row * Z + eoff;
            sg.ix = *reinterpret_cast<const int4 *>(idx_b + at);
            sg.pv = *reinterpret_cast<const float4 *>(val_b + at);
        }
        return sg;
    };
    // software pipeline: the entries of the next sweep's rows are requested before this
    // sweep's arithmetic, across the hand-over barriers
    // a sweep = 64 rows = 16 groups: FA_GPS groups per wave, then the chunk hand-over
    auto group_of = [&](int it) { return (it / FA_GPS) * 16 + (it % FA_GPS) * NW + wave; };
    PSeg nxt = load_seg4(group_of(0));
    for (int it = 0; it < nphases * FA_GPS; it++) {
        const int p = it / FA_GPS;
        const bool sweep_end = (it % FA_GPS) == FA_GPS - 1;
        const int r0 = p * FA_OROWS;
        const int g = group_of(it);
        const int row = g * R + L.j;
        const PSeg cur = nxt;
        nxt = load_seg4(group_of(it + 1));
        if (g < ngroups) {
            const int idx[4] = {cur.ix.x, cur.ix.y, cur.ix.z, cur.ix.w};
            const float val[4] = {cur.pv.x, cur.pv.y, cur.pv.z, cur.pv.w};
            float4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const int col = (s >> 2) == 0   ? quad_bcast_i<0>(idx[s & 3])
                                : (s >> 2) == 1 ? quad_bcast_i<1>(idx[s & 3])
                                : (s >> 2) == 2 ? quad_bcast_i<2>(idx[s & 3])
                                                : quad_bcast_i<3>(idx[s & 3]);
                const float pw = (s >> 2) == 0   ? quad_bcast_f<0>(val[s & 3])
                                 : (s >> 2) == 1 ? quad_bcast_f<1>(val[s & 3])
                                 : (s >> 2) == 2 ? quad_bcast_f<2>(val[s & 3])
                                                 : quad_bcast_f<3>(val[s & 3]);
                const float *vrow = tile + (size_t)col * E;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float4 x4 = *reinterpret_cast<const float4 *>(vrow + L.choff[i]);
                    acc[i].x = fmaf(pw, x4.x, acc[i].x);
                    acc[i].y = fmaf(pw, x4.y, acc[i].y);
                    acc[i].z = fmaf(pw, x4.z, acc[i].z);
                    acc[i].w = fmaf(pw, x4.w, acc[i].w);
                }
            }
            // sum the four entry groups of each row (lane offsets 16 and 32)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                acc[i].x += lane_xor_bperm<16>(acc[i].x);
                acc[i].y += lane_xor_bperm<16>(acc[i].y);
                acc[i].z += lane_xor_bperm<16>(acc[i].z);
                acc[i].w += lane_xor_bperm<16>(acc[i].w);
                acc[i].x += lane_xor_bperm<32>(acc[i].x);
                acc[i].y += lane_xor_bperm<32>(acc[i].y);
                acc[i].z += lane_xor_bperm<32>(acc[i].z);
                acc[i].w += lane_xor_bperm<32>(acc[i].w);
            }
            if (L.t == 0 && row < S) {
                float *dst = YT ? obuf + (size_t)(row - r0) * FA_OLD : out_b + (size_t)row * ld_out;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    *reinterpret_cast<float4 *>(dst + L.choff[i]) =
                        make_float4(acc[i].x, acc[i].y, acc[i].z, acc[i].w);
            }
        }
        if (!sweep_end) continue;
        if (CAUSAL && p + 1 < nphases) {
            xload.store(tile, p + 1, pre);
            pre = xload.load(p + 2);
        }
        if (YT || (CAUSAL && p + 1 < nphases)) __syncthreads();
        if (YT) {
            // transposed output, FA_OROWS rows at a time: obuf[s - r0][e] -> out_b[e * S + s];
            // 16 lanes take 16 consecutive s of one e (64 contiguous bytes per store,
            // 2-way LDS bank conflicts at a row stride of 68 words)
            const int nrows = min(S, r0 + FA_OROWS) - r0;
            for (int i = tid; i < E * 16; i += FA_THREADS) {
                const int e = i >> 4, sl = i & 15;
#pragma unroll
                for (int u = 0; u < FA_OROWS / 16; u++) {
                    const int sr = sl + 16 * u;
                    if (sr < nrows) out_b[(size_t)e * S + r0 + sr] = obuf[sr * FA_OLD + e];
                }
            }
            __syncthreads();
        }
    }
}

// CAUSAL: every column id is <= its row (what lookup guarantees), so row r only needs rows
// <= r of the K / V slice.  The slice is then staged 64 rows at a time, two chunks ahead of
// the rows being processed: the loads of chunk p+2 are in flight while the rows of chunk
// p+1 are computed, and only the first 16 KiB of each 128 KiB slice is waited for (whole-
// tile staging is 2 x 15 us of a 120 us kernel during which no wave computes).
// With CAUSAL == false the slices are staged whole and `scores` of entries with col > row
// hold the true dot products; with CAUSAL they are unspecified (their probability is 0
// either way).
template <bool YT, bool CAUSAL>
__global__ __launch_bounds__(FA_THREADS) void sparse_attention_forward_kernel(
    const int32_t *__restrict__ indices, const float *__restrict__ q,
    const float *__restrict__ k, const float *__restrict__ v, float *__restrict__ scores,
    float *__restrict__ attn, float *__restrict__ y, int S, int Z, float scale, float clampv,
    int heads) {
    constexpr int LPE = 4, E = 64, R = 4;
    constexpr int NW = FA_NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tile = reinterpret_cast<float *>(smem);                 // [S][E]
    int *lptr = reinterpret_cast<int *>(tile + (size_t)S * E);     // [S + 1] uniform indptr
    float *obuf = reinterpret_cast<float *>(lptr + ((S + 1 + 3) & ~3));   // [FA_OROWS][FA_OLD]
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int nnz = S * Z;
    const DenseView dv = dense_view(b, S, E, heads);
    const int32_t *idx_b = indices + (size_t)b * nnz;
    float *sc_b = scores + (size_t)b * nnz;
    float *at_b = attn + (size_t)b * nnz;
    const int nphases = (S + FA_OROWS - 1) / FA_OROWS;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const ChunkLoader kload(k + dv.base, dv.ld, S, tid);
    Chunk pre = {};
    if (CAUSAL) {
        kload.store(tile, 0, kload.load(0));
        pre = kload.load(1);
    } else {
        stage_rows(tile, k + dv.base, dv.ld, S, E, tid, FA_THREADS);
    }
    for (int i = tid; i <= S; i += FA_THREADS) lptr[i] = i * Z;
    __syncthreads();

    // ---- phase A: scores + softmax, R = 4 rows per wave at a time ----
    {
        const Lane4<LPE> L;
        const int ngroups = (S + R - 1) / R;
        const int eoff = 16 * L.t + 4 * L.qs;          // this lane's 4 entries inside a row
        const bool have = eoff < Z;                    // Z % 4 == 0: all four or none
        const float *q_b = q + dv.base;

        auto load_idx = [&](int g) {
            const int row = g * R + L.j;
            int4 ix = make_int4(0, 0, 0, 0);
            if (g < ngroups && row < S && have)
                ix = *reinterpret_cast<const int4 *>(idx_b + (size_t)row * Z + eoff);
            return ix;
        };
        struct QRow { float4 c[4]; };
        auto load_q = [&](int g) {
            QRow qr;
            const int row = g * R + L.j;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                qr.c[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g < ngroups && row < S)
                    qr.c[i] = *reinterpret_cast<const float4 *>(q_b + (size_t)row * dv.ld + L.choff[i]);
            }
            return qr;
        };
        // a sweep = 64 rows = one chunk = 16 groups, FA_GPS per wave: sweep p needs chunks <= p
        // (every wave runs all sweeps: the chunk hand-over below is a workgroup barrier)
        auto group_of = [&](int it) { return (it / FA_GPS) * 16 + (it % FA_GPS) * NW + wave; };
        int4 ixn = load_idx(group_of(0));
        QRow qn = load_q(group_of(0));
        for (int it = 0; it < nphases * FA_GPS; it++) {
            const int p = it / FA_GPS;
            const bool sweep_end = (it % FA_GPS) == FA_GPS - 1;
            const int g = group_of(it);
            const int row = g * R + L.j;
            const int4 ix4 = ixn;
            const QRow qc = qn;
            ixn = load_idx(group_of(it + 1));
            qn = load_q(group_of(it + 1));
            if (g < ngroups) {
            const int idx[4] = {ix4.x, ix4.y, ix4.z, ix4.w};
            float res[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const int col = (s >> 2) == 0   ? quad_bcast_i<0>(idx[s & 3])
                                : (s >> 2) == 1 ? quad_bcast_i<1>(idx[s & 3])
                                : (s >> 2) == 2 ? quad_bcast_i<2>(idx[s & 3])
                                                : quad_bcast_i<3>(idx[s & 3]);
                const float *krow = tile + (size_t)col * E;
                float part = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float4 k4 = *reinterpret_cast<const float4 *>(krow + L.choff[i]);
                    part = fmaf(qc.c[i].x, k4.x, part);
                    part = fmaf(qc.c[i].y, k4.y, part);
                    part = fmaf(qc.c[i].z, k4.z, part);
                    part = fmaf(qc.c[i].w, k4.w, part);
                }
                const float tot = group_sum<LPE>(part);
                res[s & 3] = (L.qs == (s >> 2)) ? tot : res[s & 3];
            }
            // scale, clamp (attention.py:125-127), masked exp and the row sum
            // (extension/softmax.cu:19-31: entries with col > row do not take part)
            float o[4], ex[4];
            float part = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                float val = res[kk] * scale;
                if (clampv > 0.0f) val = fminf(fmaxf(val, -clampv), clampv);
                o[kk] = val;
                const bool keep = have && idx[kk] <= row;
                ex[kk] = keep ? expf(val) : 0.0f;
                part += ex[kk];
            }
            part = group_sum<4>(part);                 // the quad
            part += lane_xor_bperm<16>(part);          // the four entry groups of the row
            part += lane_xor_bperm<32>(part);
            const float inv = 1.0f / fmaxf(1e-9f, part);
            if (have && row < S) {
                const size_t at = (size_t)row * Z + eoff;
                *reinterpret_cast<float4 *>(sc_b + at) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4 *>(at_b + at) =
                    make_float4(inv * ex[0], inv * ex[1], inv * ex[2], inv * ex[3]);
            }
            }   // g < ngroups
            if (CAUSAL && sweep_end && p + 1 < nphases) {
                kload.store(tile, p + 1, pre);         // chunk p+1: needed by the next sweep
                pre = kload.load(p + 2);               // in flight during the next sweep
                __syncthreads();
            }
        }
    }

    // ---- phase B: y = P V ----
    __syncthreads();      // every wave is done with the K tile; the P rows are visible
    product_phase<YT, CAUSAL>(tile, obuf, v + dv.base, dv.ld, idx_b, at_b,
                              y + (size_t)b * S * E, E, S, Z);
}

// The row-wise half of the backward as one launch (the column-wise half -- grad_k, grad_v --
// are the two transposed products):
//   dP     = sddmm(dY, V)                                   kernels/spmm.py:25-40 (grad_values)
//   dS     = softmax VJP (1e-9 clamp on sum y dy, extension/softmax.cu:49-81), then the clamp
//            mask and the scale of attention.py:125-127 -> gradient wrt the RAW scores
//   grad_q = spmm(dS, K)                                    kernels/sddmm.py:25-41
// Same structure as the forward: phase A on the V slice with everything between the dot
// products and dS in registers, phase B (product_phase) on the K slice.  GT: dY arrives in the
// reference's output layout [B, E, S] (what autograd hands back for `y.view(...)` of the
// transposed forward output); each 64-row sweep first transposes its [64 e][64 s] chunk through
// LDS, and the rows are also written out as [B, S, E] for the grad_v product, which replaces
// the separate transpose-copy launch.
template <bool GT, bool CAUSAL>
__global__ __launch_bounds__(FA_THREADS) void sparse_attention_backward_rows_kernel(
    const int32_t *__restrict__ indices, const float *__restrict__ gy,
    const float *__restrict__ v, const float *__restrict__ k,
    const float *__restrict__ scores, const float *__restrict__ attn,
    float *__restrict__ grad_raw, float *__restrict__ grad_q, float *__restrict__ gy_rows, int S,
    int Z, float scale, float clampv, int heads) {
    constexpr int LPE = 4, E = 64, R = 4;
    constexpr int NW = FA_NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tile = reinterpret_cast<float *>(smem);                 // [S][E]
    float *obuf = tile + (size_t)S * E;                            // [E][FA_OLD]: dY^T chunk
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int nnz = S * Z;
    const DenseView dv = dense_view(b, S, E, heads);
    const int32_t *idx_b = indices + (size_t)b * nnz;
    const float *sc_b = scores + (size_t)b * nnz;
    const float *at_b = attn + (size_t)b * nnz;
    float *gr_b = grad_raw + (size_t)b * nnz;
    const float *gy_b = gy + (size_t)b * S * E;
    const int nphases = (S + FA_OROWS - 1) / FA_OROWS;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const ChunkLoader vload(v + dv.base, dv.ld, S, tid);
    Chunk pre = {};
    if (CAUSAL) {
        vload.store(tile, 0, vload.load(0));
        pre = vload.load(1);
    } else {
        stage_rows(tile, v + dv.base, dv.ld, S, E, tid, FA_THREADS);
    }
    // (the first barrier of the sweep loop publishes the tile)

    {
        const Lane4<LPE> L;
        const int ngroups = (S + R - 1) / R;
        const int eoff = 16 * L.t + 4 * L.qs;
        const bool have = eoff < Z;
        struct Ent { int4 ix; float4 p, c; };
        auto load_ent = [&](int g) {
            Ent en;
            en.ix = make_int4(0, 0, 0, 0);
            en.p = make_float4(0.f, 0.f, 0.f, 0.f);
            en.c = make_float4(0.f, 0.f, 0.f, 0.f);
            const int row = g * R + L.j;
            if (g < ngroups && row < S && have) {
                const size_t at = (size_t)row * Z + eoff;
                en.ix = *reinterpret_cast<const int4 *>(idx_b + at);
                en.p = *reinterpret_cast<const float4 *>(at_b + at);
                en.c = *reinterpret_cast<const float4 *>(sc_b + at);
            }
            return en;
        };
        struct QRow { float4 c[4]; };
        auto load_q_rows = [&](int g) {          // !GT: dY rows straight from [B, S, E]
            QRow qr;
            const int row = g * R + L.j;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                qr.c[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!GT && g < ngroups && row < S)
                    qr.c[i] = *reinterpret_cast<const float4 *>(gy_b + (size_t)row * E + L.choff[i]);
            }
            return qr;
        };
        // GT: this thread's float4s of the next [64 e][64 s] chunk of dY^T
        const int te = tid >> 4, ts4 = 4 * (tid & 15);
        auto load_gt = [&](int p) {
            Chunk c = {};
            const int s0 = p * FA_OROWS + ts4;
            if (GT && p < nphases) {
#pragma unroll
                for (int u = 0; u < FA_NPT; u++) {
                    const float *src = gy_b + (size_t)(te + (FA_THREADS / 16) * u) * S + s0;
                    if (s0 + 4 <= S && (S & 3) == 0) {
                        c.v[u] = *reinterpret_cast<const float4 *>(src);
                    } else {
                        c.v[u].x = s0 + 0 < S ? src[0] : 0.f; c.v[u].y = s0 + 1 < S ? src[1] : 0.f;
                        c.v[u].z = s0 + 2 < S ? src[2] : 0.f; c.v[u].w = s0 + 3 < S ? src[3] : 0.f;
                    }
                }
            }
            return c;
        };
        auto group_of = [&](int it) { return (it / FA_GPS) * 16 + (it % FA_GPS) * NW + wave; };
        Ent entn = load_ent(group_of(0));
        QRow qn = load_q_rows(group_of(0));
        Chunk gtn = load_gt(0);
        for (int it = 0; it < nphases * FA_GPS; it++) {
            const int p = it / FA_GPS;
            const bool sweep_start = (it % FA_GPS) == 0, sweep_end = (it % FA_GPS) == FA_GPS - 1;
            const int r0 = p * FA_OROWS;
            const int g = group_of(it);
            const int row = g * R + L.j;
            if (sweep_start) {
                if (GT) {
#pragma unroll
                    for (int u = 0; u < FA_NPT; u++)
                        *reinterpret_cast<float4 *>(obuf + (te + (FA_THREADS / 16) * u) * FA_OLD + ts4) =
                            make_float4(gtn.v[u].x, gtn.v[u].y, gtn.v[u].z, gtn.v[u].w);
                    gtn = load_gt(p + 1);
                }
                __syncthreads();  // dY^T chunk (and, first sweep, the V tile) is in LDS
            }
            const Ent ent = entn;
            QRow qc = qn;
            entn = load_ent(group_of(it + 1));
            qn = load_q_rows(group_of(it + 1));
            if (g < ngroups) {
                if (GT) {
                    const int rl = min(row, S - 1) - r0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float *col = obuf + (size_t)L.choff[i] * FA_OLD + rl;
                        qc.c[i] = make_float4(col[0], col[FA_OLD], col[2 * FA_OLD], col[3 * FA_OLD]);
                    }
                    if (gy_rows && L.t == 0 && row < S) {
                        float *dst = gy_rows + ((size_t)b * S + row) * E;
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            *reinterpret_cast<float4 *>(dst + L.choff[i]) =
                                make_float4(qc.c[i].x, qc.c[i].y, qc.c[i].z, qc.c[i].w);
                    }
                }
                const int idx[4] = {ent.ix.x, ent.ix.y, ent.ix.z, ent.ix.w};
                float res[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 16; s++) {
                    const int col = (s >> 2) == 0   ? quad_bcast_i<0>(idx[s & 3])
                                    : (s >> 2) == 1 ? quad_bcast_i<1>(idx[s & 3])
                                    : (s >> 2) == 2 ? quad_bcast_i<2>(idx[s & 3])
                                                    : quad_bcast_i<3>(idx[s & 3]);
                    const float *vrow = tile + (size_t)col * E;
                    float part = 0.0f;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float4 k4 = *reinterpret_cast<const float4 *>(vrow + L.choff[i]);
                        part = fmaf(qc.c[i].x, k4.x, part);
                        part = fmaf(qc.c[i].y, k4.y, part);
                        part = fmaf(qc.c[i].z, k4.z, part);
                        part = fmaf(qc.c[i].w, k4.w, part);
                    }
                    const float tot = group_sum<LPE>(part);
                    res[s & 3] = (L.qs == (s >> 2)) ? tot : res[s & 3];
                }
                const float pv[4] = {ent.p.x, ent.p.y, ent.p.z, ent.p.w};
                const float cv[4] = {ent.c.x, ent.c.y, ent.c.z, ent.c.w};
                float part = 0.0f;
                bool keep[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    keep[kk] = have && idx[kk] <= row;
                    part += keep[kk] ? pv[kk] * res[kk] : 0.0f;
                }
                part = group_sum<4>(part);
                part += lane_xor_bperm<16>(part);
                part += lane_xor_bperm<32>(part);
                const float sum = fmaxf(1e-9f, part);          // softmax.cu:69
                float o[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const float yv = keep[kk] ? pv[kk] * (res[kk] - sum) : 0.0f;
                    o[kk] = (fabsf(cv[kk]) < clampv) ? yv * scale : 0.0f;
                }
                if (have && row < S)
                    *reinterpret_cast<float4 *>(gr_b + (size_t)row * Z + eoff) =
                        make_float4(o[0], o[1], o[2], o[3]);
            }
            if (!sweep_end) continue;
            if (CAUSAL && p + 1 < nphases) {
                vload.store(tile, p + 1, pre);
                pre = vload.load(p + 2);
            }
            if (GT || CAUSAL) __syncthreads();   // obuf may be overwritten, chunk p+1 is staged
        }
    }

    // ---- phase B: grad_q = dS K ----
    __syncthreads();      // every wave is done with the V tile; the dS rows are visible
    product_phase<false, CAUSAL>(tile, obuf, k + dv.base, dv.ld, idx_b, gr_b, grad_q + dv.base,
                                 dv.ld, S, Z);
}

}  // namespace spt

using namespace spt;

extern "C" int spt_sparse_attention_forward(const int32_t *indices, const float *q,
                                            const float *k, const float *v, float *scores,
                                            float *attn, float *y, int batch_size,
                                            int seq_length, int d_head, int nnz, float scale,
                                            float clampv, int heads, int y_transposed,
                                            int causal, void *stream) {
    if (!indices || !q || !k || !v || !scores || !attn || !y) return SPT_EINVAL;
    if (batch_size <= 0 || seq_length <= 0 || d_head <= 0 || nnz <= 0 || heads < 0)
        return SPT_EINVAL;
    if (d_head != 64) return SPT_EUNSUP;
    if (nnz % seq_length != 0) return SPT_ESHAPE;
    const int Z = nnz / seq_length;
    if (Z > 64 || Z % 4 != 0) return SPT_EUNSUP;
    if (heads > 0 && batch_size % heads != 0) return SPT_ESHAPE;
    const size_t tile = (size_t)seq_length * d_head * sizeof(float);
    if (tile > 128 * 1024) return SPT_EUNSUP;
    const size_t lds = tile + (size_t)((seq_length + 1 + 3) & ~3) * sizeof(int) +
                       (y_transposed ? (size_t)FA_OROWS * FA_OLD * sizeof(float) : 0);
    if (lds > 160 * 1024) return SPT_EUNSUP;
    hipStream_t s = (hipStream_t)stream;
#define SPT_FA(YT, CA)                                                                        \
    do {                                                                                      \
        SPT_HIP_TRY(hipFuncSetAttribute(                                                      \
            reinterpret_cast<const void *>(&sparse_attention_forward_kernel<YT, CA>),         \
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
        hipLaunchKernelGGL((sparse_attention_forward_kernel<YT, CA>),                         \
                           dim3((unsigned)batch_size), dim3(FA_THREADS), lds, s, indices, q,  \
                           k, v, scores, attn, y, seq_length, Z, scale, clampv, heads);       \
    } while (0)
    if (y_transposed) {
        if (causal) SPT_FA(true, true); else SPT_FA(true, false);
    } else {
        if (causal) SPT_FA(false, true); else SPT_FA(false, false);
    }
#undef SPT_FA
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_sparse_attention_backward_rows(const int32_t *indices, const float *grad_y,
                                                  const float *v, const float *k,
                                                  const float *scores, const float *attn,
                                                  float *grad_raw, float *grad_q,
                                                  float *grad_y_rows, int batch_size,
                                                  int seq_length, int d_head, int nnz,
                                                  float scale, float clampv, int heads,
                                                  int grad_y_transposed, int causal,
                                                  void *stream) {
    if (!indices || !grad_y || !v || !k || !scores || !attn || !grad_raw || !grad_q)
        return SPT_EINVAL;
    if (batch_size <= 0 || seq_length <= 0 || d_head <= 0 || nnz <= 0 || heads < 0)
        return SPT_EINVAL;
    if (d_head != 64) return SPT_EUNSUP;
    if (nnz % seq_length != 0) return SPT_ESHAPE;
    const int Z = nnz / seq_length;
    if (Z > 64 || Z % 4 != 0) return SPT_EUNSUP;
    if (heads > 0 && batch_size % heads != 0) return SPT_ESHAPE;
    const size_t tile = (size_t)seq_length * d_head * sizeof(float);
    if (tile > 128 * 1024) return SPT_EUNSUP;
    const size_t lds = tile + (size_t)64 * FA_OLD * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define SPT_FB(GT, CA)                                                                        \
    do {                                                                                      \
        SPT_HIP_TRY(hipFuncSetAttribute(                                                      \
            reinterpret_cast<const void *>(&sparse_attention_backward_rows_kernel<GT, CA>),   \
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
        hipLaunchKernelGGL((sparse_attention_backward_rows_kernel<GT, CA>),                   \
                           dim3((unsigned)batch_size), dim3(FA_THREADS), lds, s, indices,     \
                           grad_y, v, k, scores, attn, grad_raw, grad_q, grad_y_rows,         \
                           seq_length, Z, scale, clampv, heads);                              \
    } while (0)
    if (grad_y_transposed) {
        if (causal) SPT_FB(true, true); else SPT_FB(true, false);
    } else {
        if (causal) SPT_FB(false, true); else SPT_FB(false, false);
    }
#undef SPT_FB
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
