// sparse_rows.h -- how one wave walks its share of CSR rows (shared by sddmm / spmm).
//
// A wave owns rows  first, first + stride, first + 2*stride, ...  (< S) of one batch.
// Row bounds are fetched 64 rows at a time (lane j keeps indptr of the j-th row) and
// broadcast with v_readlane, so the inner loop has no dependent indptr load.  The
// first 64-entry chunk of the NEXT row (column ids, values) is requested before the
// current row is computed: HBM latency hides under the current row's LDS gathers.
#ifndef SPT_SPARSE_ROWS_H
#define SPT_SPARSE_ROWS_H

#include "spt_common.h"

namespace spt {

struct RowChunk {
    int start;      // first CSR entry of the row
    int end;        // one past the last
    int idx;        // lane's column id of entry start + lane (0 when past the end)
    float val;      // lane's value of that entry (0 when past the end; unused by sddmm)
};

template <bool WITH_VALUES>
__device__ __forceinline__ RowChunk fetch_row(const int32_t *__restrict__ idx_b,
                                              const float *__restrict__ val_b, int start,
                                              int end) {
    RowChunk c;
    c.start = start;
    c.end = end;
    const int lane = lane_id();
    const bool in = lane < end - start;
    c.idx = in ? idx_b[start + lane] : 0;
    c.val = 0.0f;
    if (WITH_VALUES) c.val = in ? val_b[start + lane] : 0.0f;
    return c;
}

// Calls body(row, chunk) for every row of this wave, with the next row's first chunk
// already in flight.  `row_aux(row)` lets the caller prefetch per-row operands too.
template <bool WITH_VALUES, typename Prefetch, typename Body>
__device__ __forceinline__ void for_each_row(const int32_t *__restrict__ indptr,
                                             const int32_t *__restrict__ idx_b,
                                             const float *__restrict__ val_b, int first,
                                             int stride, int S, Prefetch prefetch, Body body) {
    const int lane = lane_id();
    for (int batch0 = first; batch0 < S; batch0 += SPT_WAVE * stride) {
        // bounds of up to 64 rows: lane j <-> row batch0 + j*stride
        const int my_row = batch0 + lane * stride;
        int my_start = 0, my_end = 0;
        if (my_row < S) {
            my_start = indptr[my_row];
            my_end = indptr[my_row + 1];
        }
        const int nrows = min(SPT_WAVE, (S - batch0 + stride - 1) / stride);
        RowChunk next = fetch_row<WITH_VALUES>(idx_b, val_b,
                                               __builtin_amdgcn_readlane(my_start, 0),
                                               __builtin_amdgcn_readlane(my_end, 0));
        auto next_aux = prefetch(batch0);
        for (int j = 0; j < nrows; j++) {
            const RowChunk cur = next;
            const auto cur_aux = next_aux;
            const int row = batch0 + j * stride;
            if (j + 1 < nrows) {
                const int ns = __builtin_amdgcn_readlane(my_start, j + 1);
                const int ne = __builtin_amdgcn_readlane(my_end, j + 1);
                next = fetch_row<WITH_VALUES>(idx_b, val_b, ns, ne);
                next_aux = prefetch(row + stride);
            }
            body(row, cur, cur_aux);
        }
    }
}

}  // namespace spt

#endif  // SPT_SPARSE_ROWS_H
