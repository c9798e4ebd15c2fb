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

#ifndef FA_THREADS_VALUE
#define FA_THREADS_VALUE 1024
#endif
constexpr int FA_THREADS = FA_THREADS_VALUE;
constexpr int FA_OROWS = 64;             // rows per transposed write-out phase
constexpr int FA_OLD = 64 + 4;           // padded row of the staging buffer (16-byte aligned)

// One chunk of FA_OROWS = 64 rows of a [S][64] slice: one float4 per thread.
struct ChunkLoader {
    const float *src;     // slice base
    int ld, S, row, c4;   // this thread's row inside a chunk and float4 column
    __device__ __forceinline__ ChunkLoader(const float *base, int ld_, int S_, int tid)
        : src(base), ld(ld_), S(S_), row(tid >> 4), c4(tid & 15) {}
    __device__ __forceinline__ float4 load(int chunk) const {
        const int r = chunk * FA_OROWS + row;
        if (r < S) return *reinterpret_cast<const float4 *>(src + (size_t)r * ld + 4 * c4);
        return make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ void store(float *tile, int chunk, const float4 &v) const {
        const int r = chunk * FA_OROWS + row;
        if (r < S) *reinterpret_cast<float4 *>(tile + (size_t)r * 64 + 4 * c4) =
            make_float4(v.x, v.y, v.z, v.w);
    }
};

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
    static_assert(FA_THREADS == 16 * FA_OROWS, "one float4 per thread per 64-row chunk");
    constexpr int LPE = 4, E = 64, R = 4;
    constexpr int NW = FA_THREADS / SPT_WAVE;
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
    float4 pre = make_float4(0.f, 0.f, 0.f, 0.f);
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
        int4 ixn = load_idx(wave);
        QRow qn = load_q(wave);
        // NW waves x R rows = 64 rows per sweep = one chunk: sweep p needs chunks <= p
        // (every wave runs all nphases sweeps: the chunk hand-over below is a workgroup barrier)
        for (int p = 0; p < nphases; p++) {
            const int g = p * NW + wave;
            const int row = g * R + L.j;
            const int4 ix4 = ixn;
            const QRow qc = qn;
            ixn = load_idx(g + NW);
            qn = load_q(g + NW);
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
            if (CAUSAL && p + 1 < nphases) {
                kload.store(tile, p + 1, pre);         // chunk p+1: needed by the next sweep
                pre = kload.load(p + 2);               // in flight during the next sweep
                __syncthreads();
            }
        }
    }

    // ---- phase B: y = P V ----
    __syncthreads();      // every wave is done with the K tile; the P rows are visible
    const ChunkLoader vload(v + dv.base, dv.ld, S, tid);
    if (CAUSAL) {
        vload.store(tile, 0, vload.load(0));
        pre = vload.load(1);
    } else {
        stage_rows(tile, v + dv.base, dv.ld, S, E, tid, FA_THREADS);
    }
    __syncthreads();
    float *y_b = y + (size_t)b * S * E;
    {
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
                sg.pv = *reinterpret_cast<const float4 *>(at_b + at);
            }
            return sg;
        };
        // the same software pipeline as phase A: the entries of the next sweep's rows are
        // requested before this sweep's arithmetic, across the hand-over barriers
        PSeg nxt = load_seg4(wave);
        for (int p = 0; p < nphases; p++) {
            const int r0 = p * FA_OROWS;
            const int g = p * NW + wave;
            const int row = g * R + L.j;
            const PSeg cur = nxt;
            nxt = load_seg4(g + NW);
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
                    float *dst = YT ? obuf + (size_t)(row - r0) * FA_OLD : y_b + (size_t)row * E;
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        *reinterpret_cast<float4 *>(dst + L.choff[i]) =
                            make_float4(acc[i].x, acc[i].y, acc[i].z, acc[i].w);
                }
            }
            if (CAUSAL && p + 1 < nphases) {
                vload.store(tile, p + 1, pre);
                pre = vload.load(p + 2);
            }
            if (YT || (CAUSAL && p + 1 < nphases)) __syncthreads();
            if (YT) {
                // transposed output, FA_OROWS rows at a time: obuf[s - r0][e] -> y_b[e * S + s];
                // 16 lanes take 16 consecutive s of one e (64 contiguous bytes per store,
                // 2-way LDS bank conflicts at a row stride of 68 words)
                const int nrows = min(S, r0 + FA_OROWS) - r0;
                for (int i = tid; i < E * 16; i += FA_THREADS) {
                    const int e = i >> 4, sl = i & 15;
#pragma unroll
                    for (int u = 0; u < FA_OROWS / 16; u++) {
                        const int sr = sl + 16 * u;
                        if (sr < nrows) y_b[(size_t)e * S + r0 + sr] = obuf[sr * FA_OLD + e];
                    }
                }
                __syncthreads();
            }
        }
    }
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
