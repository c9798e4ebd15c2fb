// gather4.h -- the row-gather engine behind the fast sddmm / spmm kernels (E = 64, 128).
//
// One CSR entry = one dense row of E floats gathered from an [S, E] tile (K for sddmm,
// X for spmm; LDS-resident when it fits).  Mapping of a wave, E = 16 * LPE:
//
//   * LPE lanes (4 at E=64, 8 at E=128) co-operate on one entry; each lane reads four
//     16-byte chunks of the row, so an entry costs ONE cross-lane broadcast of its
//     column id (DPP quad_perm, pure VALU) instead of the 16 ds_bpermute of a
//     16-lanes-per-entry mapping, and the dot product needs log2(LPE) DPP adds.
//   * the 64 / LPE lane groups work on R = 16 / LPE DIFFERENT CSR rows at once
//     (row slot j = group % R), four groups per row, each owning 16 consecutive
//     entries of the row's current 64-entry chunk: a lane loads 4 consecutive column
//     ids / values with one 16-byte load and stores 4 consecutive sddmm results with
//     one 16-byte store.
//   * bank conflicts: a ds_read_b128 is served in groups of 16 lanes = 16 / LPE
//     entries of DIFFERENT row slots.  Row slot j reads quarter (i + j) % 4 of its row
//     at read i, so the entries of one hardware group always cover disjoint banks,
//     whatever their columns are.  No padding, no swizzle of the tile.
//
// Algorithmic traffic is unchanged; what this mapping buys is VALU and
// LDS-instruction count: ~25 VALU + 4 ds_read_b128 per 16 gathered rows per lane.
#ifndef SPT_GATHER4_H
#define SPT_GATHER4_H

#include "spt_common.h"

namespace spt {

// broadcast the value of lane (quad base + SRC) to the four lanes of each quad
template <int SRC>
__device__ __forceinline__ int quad_bcast_i(int v) {
    constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
    return __builtin_amdgcn_update_dpp(0, v, ctrl, 0xF, 0xF, false);
}
template <int SRC>
__device__ __forceinline__ float quad_bcast_f(float v) {
    return __builtin_bit_cast(float, quad_bcast_i<SRC>(__builtin_bit_cast(int, v)));
}

// value of lane (lane ^ MASK), through the LDS crossbar (no LDS memory is touched)
template <int MASK>
__device__ __forceinline__ float lane_xor_bperm(float v) {
    const int src = (lane_id() ^ MASK) << 2;
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, v)));
}

enum GatherMode { G_SDDMM = 0, G_SPMM = 1, G_SPMM_PERM = 2 };

template <int LPE>
struct Lane4 {
    static constexpr int R = 16 / LPE;   // CSR rows in flight per wave
    int sub, qs, j, t;
    int choff[4];                        // float offset of this lane's i-th chunk in a row
    __device__ __forceinline__ Lane4() {
        const int lane = lane_id();
        const int grp = lane / LPE;
        sub = lane % LPE;
        qs = sub & 3;
        j = grp % R;
        t = grp / R;
#pragma unroll
        for (int i = 0; i < 4; i++) choff[i] = 4 * (((i + j) & 3) * LPE + sub);
    }
};

struct Seg4 {          // this lane's 4 consecutive entries of the current chunk
    int idx[4];
    float val[4];
};

// load column ids (and values) of entries e0 .. e0+3, zero beyond `end`
template <int MODE>
__device__ __forceinline__ Seg4 load_seg(const int32_t *__restrict__ idx_b,
                                         const int32_t *__restrict__ perm_b,
                                         const float *__restrict__ val_b, int e0, int end) {
    Seg4 s;
    const bool full = (e0 + 4 <= end) && ((e0 & 3) == 0);
    if (full) {
        const int4 v = *reinterpret_cast<const int4 *>(idx_b + e0);
        s.idx[0] = v.x; s.idx[1] = v.y; s.idx[2] = v.z; s.idx[3] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) s.idx[k] = (e0 + k < end) ? idx_b[e0 + k] : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) s.val[k] = 0.0f;
    if (MODE == G_SPMM) {
        if (full) {
            const float4 v = *reinterpret_cast<const float4 *>(val_b + e0);
            s.val[0] = v.x; s.val[1] = v.y; s.val[2] = v.z; s.val[3] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) s.val[k] = (e0 + k < end) ? val_b[e0 + k] : 0.0f;
        }
    } else if (MODE == G_SPMM_PERM) {
        // transposed structure: the value of entry e lives at values[perm[e]]
        int p[4];
        if (full) {
            const int4 v = *reinterpret_cast<const int4 *>(perm_b + e0);
            p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) p[k] = (e0 + k < end) ? perm_b[e0 + k] : -1;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            s.val[k] = (e0 + k < end && p[k] >= 0) ? val_b[p[k]] : 0.0f;
    }
    return s;
}

// Process the CSR rows  rows_first + g * R + j  (g = g_first, g_first + g_stride, ...)
// of one batch.  `tile` is the [S, E] gather source (LDS or global), `dense_b` the
// batch's per-row dense operand: Q rows for sddmm (read), Y rows for spmm (written).
template <int LPE, int MODE>
__device__ __forceinline__ void gather_rows(const int32_t *__restrict__ ptr,     // [nrows+1]
                                            const int32_t *__restrict__ idx_b,   // [nnz]
                                            const int32_t *__restrict__ perm_b,  // [nnz] | null
                                            const float *__restrict__ val_b,     // [nnz] | null
                                            const float *__restrict__ tile,      // [S, E]
                                            const float *__restrict__ q_b,       // sddmm: [S, E]
                                            float *__restrict__ out_b,  // sddmm: [nnz]; spmm: [S,E]
                                            int g_first, int g_stride, int nrows, float scale,
                                            float clampv, int ld_q = 16 * LPE,
                                            int ld_out = 16 * LPE) {
    constexpr int E = 16 * LPE;
    constexpr int R = 16 / LPE;
    const Lane4<LPE> L;
    const int ngroups = (nrows + R - 1) / R;

    auto row_bounds = [&](int g, int &start, int &end) {
        const int row = g * R + L.j;
        start = 0;
        end = 0;
        if (g < ngroups && row < nrows) {
            start = ptr[row];
            end = ptr[row + 1];
        }
    };

    int start, end, nstart, nend;
    row_bounds(g_first, start, end);
    row_bounds(g_first + g_stride, nstart, nend);
    Seg4 seg = load_seg<MODE>(idx_b, perm_b, val_b, start + 16 * L.t + 4 * L.qs, end);

    // this lane's four chunks of a dense query row (sddmm only)
    struct QRow { float4 c[4]; };
    auto load_q = [&](int g) {
        QRow qr;
        const int row = g * R + L.j;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            qr.c[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == G_SDDMM && g < ngroups && row < nrows)
                qr.c[i] = *reinterpret_cast<const float4 *>(q_b + (size_t)row * ld_q + L.choff[i]);
        }
        return qr;
    };
    QRow qcur = load_q(g_first);

    for (int g = g_first; g < ngroups; g += g_stride) {
        const int row = g * R + L.j;
        // ---- prefetch: first chunk + query row of the next group, bounds of the one after
        const Seg4 nseg = load_seg<MODE>(idx_b, perm_b, val_b, nstart + 16 * L.t + 4 * L.qs, nend);
        const QRow qnext = load_q(g + g_stride);
        int nnstart, nnend;
        row_bounds(g + 2 * g_stride, nnstart, nnend);

        float4 q[4];
#pragma unroll
        for (int i = 0; i < 4; i++) q[i] = qcur.c[i];
        float4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);

        // number of 64-entry chunks = max over the R rows in flight (wave-uniform)
        int len = end - start;
        int maxlen = 0;
#pragma unroll
        for (int jj = 0; jj < R; jj++)
            maxlen = max(maxlen, __builtin_amdgcn_readlane(len, jj * LPE));
        const int nchunks = (maxlen + 63) >> 6;

        for (int c = 0; c < nchunks; c++) {
            const int e0 = start + 64 * c + 16 * L.t + 4 * L.qs;
            if (c > 0) seg = load_seg<MODE>(idx_b, perm_b, val_b, e0, end);
            float res[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const int col = (s >> 2) == 0   ? quad_bcast_i<0>(seg.idx[s & 3])
                                : (s >> 2) == 1 ? quad_bcast_i<1>(seg.idx[s & 3])
                                : (s >> 2) == 2 ? quad_bcast_i<2>(seg.idx[s & 3])
                                                : quad_bcast_i<3>(seg.idx[s & 3]);
                const float *krow = tile + (size_t)col * E;
                if (MODE == G_SDDMM) {
                    float part = 0.0f;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float4 k4 = *reinterpret_cast<const float4 *>(krow + L.choff[i]);
                        part = fmaf(q[i].x, k4.x, part);
                        part = fmaf(q[i].y, k4.y, part);
                        part = fmaf(q[i].z, k4.z, part);
                        part = fmaf(q[i].w, k4.w, part);
                    }
                    const float tot = group_sum<LPE>(part);
                    res[s & 3] = (L.qs == (s >> 2)) ? tot : res[s & 3];
                } else {
                    const float v = (s >> 2) == 0   ? quad_bcast_f<0>(seg.val[s & 3])
                                    : (s >> 2) == 1 ? quad_bcast_f<1>(seg.val[s & 3])
                                    : (s >> 2) == 2 ? quad_bcast_f<2>(seg.val[s & 3])
                                                    : quad_bcast_f<3>(seg.val[s & 3]);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float4 x4 = *reinterpret_cast<const float4 *>(krow + L.choff[i]);
                        acc[i].x = fmaf(v, x4.x, acc[i].x);
                        acc[i].y = fmaf(v, x4.y, acc[i].y);
                        acc[i].z = fmaf(v, x4.z, acc[i].z);
                        acc[i].w = fmaf(v, x4.w, acc[i].w);
                    }
                }
            }
            if (MODE == G_SDDMM && L.sub < 4) {
                // this lane holds entries e0 .. e0+3
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float v = res[k] * scale;
                    if (clampv > 0.0f) v = fminf(fmaxf(v, -clampv), clampv);
                    o[k] = v;
                }
                if ((e0 + 4 <= end) && ((e0 & 3) == 0)) {
                    *reinterpret_cast<float4 *>(out_b + e0) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (e0 + k < end) out_b[e0 + k] = o[k];
                }
            }
        }
        if (MODE != G_SDDMM) {
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
            if (L.t == 0 && row < nrows) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    *reinterpret_cast<float4 *>(out_b + (size_t)row * ld_out + L.choff[i]) = acc[i];
            }
        }
        seg = nseg;
        qcur = qnext;
        start = nstart; end = nend;
        nstart = nnstart; nend = nnend;
    }
}

// ---- transposed (PERM) products: skewed row lengths ------------------------------------
//
// Rows of a transposed attention pattern are its key columns: their lengths are far from
// uniform (column 0 of a causal top-Z pattern carries ~2400 of the 32768 entries at S=512,
// the next ones ~400, the median 30).  Two measures keep the waves of a workgroup level:
//   * row groups are handed out dynamically (an LDS ticket counter), heavy groups first;
//   * a group whose longest row would cost more 64-entry rounds than handling its rows
//     one by one at 256 entries per round runs in WIDE mode: all 64 / LPE lane groups
//     work on ONE row; row slot j takes entries [64j, 64j+64) of each 256-entry round.
//     Slot j holds quarter (i + j) % 4 of the output row in acc[i], so the four slots are
//     summed with three DPP row rotations and a static register shift:
//         tot[i] = acc[i] + ror4(acc[i+1]) + ror8(acc[i+2]) + ror12(acc[i+3]).
template <int LPE>
__device__ __forceinline__ void accumulate16(const Seg4 &seg, const float *__restrict__ tile,
                                             const int (&choff)[4], float4 (&acc)[4]) {
    constexpr int E = 16 * LPE;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const int col = (s >> 2) == 0   ? quad_bcast_i<0>(seg.idx[s & 3])
                        : (s >> 2) == 1 ? quad_bcast_i<1>(seg.idx[s & 3])
                        : (s >> 2) == 2 ? quad_bcast_i<2>(seg.idx[s & 3])
                                        : quad_bcast_i<3>(seg.idx[s & 3]);
        const float v = (s >> 2) == 0   ? quad_bcast_f<0>(seg.val[s & 3])
                        : (s >> 2) == 1 ? quad_bcast_f<1>(seg.val[s & 3])
                        : (s >> 2) == 2 ? quad_bcast_f<2>(seg.val[s & 3])
                                        : quad_bcast_f<3>(seg.val[s & 3]);
        const float *krow = tile + (size_t)col * E;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float4 x4 = *reinterpret_cast<const float4 *>(krow + choff[i]);
            acc[i].x = fmaf(v, x4.x, acc[i].x);
            acc[i].y = fmaf(v, x4.y, acc[i].y);
            acc[i].z = fmaf(v, x4.z, acc[i].z);
            acc[i].w = fmaf(v, x4.w, acc[i].w);
        }
    }
}

template <int ROR>
__device__ __forceinline__ float4 ror_f4(const float4 &v) {
    constexpr int ctrl = 0x120 + ROR;  // DPP row_ror:ROR
    return make_float4(dpp_mov<ctrl>(v.x), dpp_mov<ctrl>(v.y), dpp_mov<ctrl>(v.z),
                       dpp_mov<ctrl>(v.w));
}

__device__ __forceinline__ void add_f4(float4 &a, const float4 &b) {
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
}

__device__ __forceinline__ void reduce_t(float4 (&acc)[4]) {
    // sum over the four entry groups of a row slot (lane offsets 16 and 32)
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
}

// LPE == 4 only (E = 64): R = 4 row slots.  `ticket` is an LDS word zeroed by the caller.
template <int MODE>
__device__ __forceinline__ void gather_rows_dynamic(
    const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx_b,
    const int32_t *__restrict__ perm_b, const float *__restrict__ val_b,
    const float *__restrict__ tile, float *__restrict__ out_b, int *ticket, int nrows,
    int ld_out = 64) {
    constexpr int LPE = 4, R = 4;  // E = 64
    const Lane4<LPE> L;
    const int ngroups = (nrows + R - 1) / R;
    const int lane = lane_id();

    // Software pipeline over tickets: the bounds of the group after next and the first
    // segment of the next group are requested before the current group is computed (a
    // freshly drawn group otherwise costs two dependent HBM latencies before its first
    // gather).  A wave therefore holds up to three tickets; the tail is at most two
    // groups per wave.
    auto draw = [&]() {
        int g = 0;
        if (lane == 0) g = atomicAdd(ticket, 1);
        return __builtin_amdgcn_readfirstlane(g);
    };
    auto bounds = [&](int g, int &st, int &en) {
        const int row = g * R + L.j;
        st = 0;
        en = 0;
        if (g < ngroups && row < nrows) {
            st = ptr[row];
            en = ptr[row + 1];
        }
    };
    int g = draw(), g1 = draw(), g2;
    int start, end, start1, end1, start2, end2;
    bounds(g, start, end);
    bounds(g1, start1, end1);
    Seg4 seg0 = load_seg<MODE>(idx_b, perm_b, val_b, start + 16 * L.t + 4 * L.qs, end);

    for (; g < ngroups; g = g1, g1 = g2, start = start1, end = end1, start1 = start2,
                        end1 = end2) {
        g2 = draw();
        bounds(g2, start2, end2);
        // first (normal-mode) segment of the next group: in flight during this group
        const Seg4 seg_next =
            load_seg<MODE>(idx_b, perm_b, val_b, start1 + 16 * L.t + 4 * L.qs, end1);
        const int row = g * R + L.j;
        const int len = end - start;
        int lens[R], starts[R];
        int maxlen = 0, wide_rounds = 0;
#pragma unroll
        for (int jj = 0; jj < R; jj++) {
            lens[jj] = __builtin_amdgcn_readlane(len, jj * LPE);
            starts[jj] = __builtin_amdgcn_readlane(start, jj * LPE);
            maxlen = max(maxlen, lens[jj]);
            wide_rounds += (lens[jj] + 255) >> 8;
        }
        const int normal_rounds = (maxlen + 63) >> 6;

        if (normal_rounds <= wide_rounds + 1) {
            // ---- normal: four rows side by side, 64 entries of each per round ----
            float4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            Seg4 seg = seg0;
            for (int c = 0; c < normal_rounds; c++) {
                const Seg4 nseg = load_seg<MODE>(idx_b, perm_b, val_b,
                                                 start + 64 * (c + 1) + 16 * L.t + 4 * L.qs, end);
                accumulate16<LPE>(seg, tile, L.choff, acc);
                seg = nseg;
            }
            reduce_t(acc);
            if (L.t == 0 && row < nrows) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    *reinterpret_cast<float4 *>(out_b + (size_t)row * ld_out + L.choff[i]) = acc[i];
            }
        } else {
            // ---- wide: one row at a time, 256 entries per round -------------------
#pragma unroll
            for (int jj = 0; jj < R; jj++) {
                const int wrow = g * R + jj;
                const int wstart = starts[jj], wend = wstart + lens[jj];
                float4 acc[4];
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                const int rounds = (lens[jj] + 255) >> 8;
                const int lane_off = 64 * L.j + 16 * L.t + 4 * L.qs;
                Seg4 seg = load_seg<MODE>(idx_b, perm_b, val_b, wstart + lane_off, wend);
                for (int c = 0; c < rounds; c++) {
                    const Seg4 nseg = load_seg<MODE>(idx_b, perm_b, val_b,
                                                     wstart + 256 * (c + 1) + lane_off, wend);
                    accumulate16<LPE>(seg, tile, L.choff, acc);
                    seg = nseg;
                }
                reduce_t(acc);
                // slot j holds quarter (i + j) % 4 in acc[i]: rotate-and-shift sum
                float4 tot[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    tot[i] = acc[i];
                    add_f4(tot[i], ror_f4<4>(acc[(i + 1) & 3]));
                    add_f4(tot[i], ror_f4<8>(acc[(i + 2) & 3]));
                    add_f4(tot[i], ror_f4<12>(acc[(i + 3) & 3]));
                }
                if (L.t == 0 && L.j == 0 && wrow < nrows) {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        *reinterpret_cast<float4 *>(out_b + (size_t)wrow * ld_out + L.choff[i]) = tot[i];
                }
            }
        }
        seg0 = seg_next;
    }
}

}  // namespace spt

#endif  // SPT_GATHER4_H
