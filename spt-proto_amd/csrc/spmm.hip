// spmm.hip -- batched CSR x dense and CSR^T x dense for gfx950 (no hipSPARSE).
//
// Replaces extension/spmm.cpp:27-69 of the reference (cusparseSpMM, op = N or T).
//   N: y[b,r,:]             = sum_{p in row r} values[b,p] * x[b, indices[b,p], :]
//   T: y[b,indices[b,p],:] += values[b,p] * x[b, row(p), :]
//
// N uses the same group-per-entry gather as sddmm.hip: LPE = E/4 lanes read one whole
// x row per entry (float4 each), 64/LPE entries per wave-instruction, per-lane float4
// accumulators, one cross-group butterfly per output row.  With X LDS-resident
// (S*E*4 <= 128 KiB) HBM traffic is the algorithmic 2*S*E*4 + 2*nnz*4 bytes per batch.
//
// T is a scatter.  cuSPARSE does it with global atomics; global float atomics on
// MI355X run at ~1.3 TB/s of added bytes (MI355X_MICROARCH "Global float atomics"),
// 6x under the HBM stream this op should be bound by.  Here the batch's whole output
// [S,E] is an LDS accumulator: entries are added with ds_add_f32 (LDS atomics, no
// HBM traffic) and the tile is written once, coalesced, zeros included.  Lanes are
// rotated over the four 16-word quarters of a row so that the two entry groups of a
// half-wave always hit disjoint LDS banks.
#include "spt_common.h"

namespace spt {

constexpr int SP_THREADS = 256;
constexpr int SP_THREADS_LDS = 1024;

// ------------------------------------------------------------------ N (gather)

template <int LPE, bool XLDS>
__device__ __forceinline__ void spmm_row(const int32_t *__restrict__ idx_b,
                                         const float *__restrict__ val_b,
                                         const float *__restrict__ xbase,  // global or LDS
                                         float *__restrict__ yrow, int start, int end, int E) {
    constexpr int EPS = SPT_WAVE / LPE;
    const int lane = lane_id();
    const int sub = lane & (LPE - 1);
    const int grp = lane / LPE;
    const bool sub_live = (4 * sub) < E;

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p0 = start; p0 < end; p0 += SPT_WAVE) {
        const int remaining = end - p0;
        const bool in = lane < remaining;
        const int my_idx = in ? idx_b[p0 + lane] : 0;
        const float my_val = in ? val_b[p0 + lane] : 0.0f;  // 0 weight: padding adds nothing
        const int nsteps = min(LPE, (remaining + EPS - 1) / EPS);
        auto step = [&](int s) {
            const int e = s * EPS + grp;
            const int col = __shfl(my_idx, e, SPT_WAVE);
            const float v = __shfl(my_val, e, SPT_WAVE);
            if (sub_live) {
                const float4 x4 =
                    *reinterpret_cast<const float4 *>(xbase + (size_t)col * E + 4 * sub);
                acc.x = fmaf(v, x4.x, acc.x);
                acc.y = fmaf(v, x4.y, acc.y);
                acc.z = fmaf(v, x4.z, acc.z);
                acc.w = fmaf(v, x4.w, acc.w);
            }
        };
        if (remaining >= SPT_WAVE) {
#pragma unroll
            for (int s = 0; s < LPE; s++) step(s);
        } else {
            for (int s = 0; s < nsteps; s++) step(s);
        }
    }
    // sum the EPS groups (lanes with equal `sub`)
    if constexpr (LPE <= 1) { acc.x += butterfly_partner<1>(acc.x); acc.y += butterfly_partner<1>(acc.y); acc.z += butterfly_partner<1>(acc.z); acc.w += butterfly_partner<1>(acc.w); }
    if constexpr (LPE <= 2) { acc.x += butterfly_partner<2>(acc.x); acc.y += butterfly_partner<2>(acc.y); acc.z += butterfly_partner<2>(acc.z); acc.w += butterfly_partner<2>(acc.w); }
    if constexpr (LPE <= 4) { acc.x += __shfl_xor(acc.x, 4, SPT_WAVE); acc.y += __shfl_xor(acc.y, 4, SPT_WAVE); acc.z += __shfl_xor(acc.z, 4, SPT_WAVE); acc.w += __shfl_xor(acc.w, 4, SPT_WAVE); }
    if constexpr (LPE <= 8) { acc.x += __shfl_xor(acc.x, 8, SPT_WAVE); acc.y += __shfl_xor(acc.y, 8, SPT_WAVE); acc.z += __shfl_xor(acc.z, 8, SPT_WAVE); acc.w += __shfl_xor(acc.w, 8, SPT_WAVE); }
    if constexpr (LPE <= 16) { acc.x += butterfly_partner<16>(acc.x); acc.y += butterfly_partner<16>(acc.y); acc.z += butterfly_partner<16>(acc.z); acc.w += butterfly_partner<16>(acc.w); }
    if constexpr (LPE <= 32) { acc.x += butterfly_partner<32>(acc.x); acc.y += butterfly_partner<32>(acc.y); acc.z += butterfly_partner<32>(acc.z); acc.w += butterfly_partner<32>(acc.w); }
    if (grp == 0 && sub_live) *reinterpret_cast<float4 *>(yrow + 4 * sub) = acc;
}

template <int LPE>
__global__ __launch_bounds__(SP_THREADS) void spmm_n_global_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const float *__restrict__ x, float *__restrict__ y,
    int B, int S, int E, int nnz, int tiles_per_batch, int rows_per_block) {
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / tiles_per_batch;
    const int tile = lid - b * tiles_per_batch;
    if (b >= B) return;
    const int wave = threadIdx.x >> 6;
    const int row0 = tile * rows_per_block;
    const int row1 = min(S, row0 + rows_per_block);
    const float *xb = x + (size_t)b * S * E;
    for (int r = row0 + wave; r < row1; r += SP_THREADS / SPT_WAVE) {
        spmm_row<LPE, false>(indices + (size_t)b * nnz, values + (size_t)b * nnz, xb,
                             y + ((size_t)b * S + r) * E, indptr[r], indptr[r + 1], E);
    }
}

template <int LPE>
__global__ __launch_bounds__(SP_THREADS_LDS) void spmm_n_lds_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const float *__restrict__ x, float *__restrict__ y,
    int B, int S, int E, int nnz, int splits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xtile = reinterpret_cast<float *>(smem);  // [S][E]
    const int b = blockIdx.x / splits;
    const int part = blockIdx.x - b * splits;
    const int tid = threadIdx.x;
    {
        const float4 *src = reinterpret_cast<const float4 *>(x + (size_t)b * S * E);
        float4 *dst = reinterpret_cast<float4 *>(xtile);
        const int n4 = (S * E) >> 2;
        for (int i = tid; i < n4; i += SP_THREADS_LDS) dst[i] = src[i];
    }
    __syncthreads();
    const int wave = tid >> 6;
    constexpr int NW = SP_THREADS_LDS / SPT_WAVE;
    for (int r = part * NW + wave; r < S; r += splits * NW) {
        spmm_row<LPE, true>(indices + (size_t)b * nnz, values + (size_t)b * nnz, xtile,
                            y + ((size_t)b * S + r) * E, indptr[r], indptr[r + 1], E);
    }
}

// ------------------------------------------------------------------ T (scatter)

// LDS accumulator layout: word(col, e) = col * E + e (no padding: a row of E = 64
// words spans both 32-bank halves exactly twice; conflicts are avoided by the lane
// rotation below, not by padding).
//
// One group of 16 lanes per entry (E <= 64: four 16-word quarters per row, lane `sub`
// owns words sub, sub+16, sub+32, sub+48).  At add-instruction j group g touches
// quarter (j + g) & 3, so the two groups of each half-wave are always on different
// 16-bank halves: no bank conflict for any column pair.
template <int QUARTERS>  // ceil(E / 16), 1..4
__global__ __launch_bounds__(SP_THREADS_LDS) void spmm_t_lds_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const float *__restrict__ x, float *__restrict__ y,
    int B, int S, int E, int nnz, int splits, int atomic_flush) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *acc = reinterpret_cast<float *>(smem);  // [S][E]
    const int b = blockIdx.x / splits;
    const int part = blockIdx.x - b * splits;
    const int tid = threadIdx.x;
    const int n = S * E;
    for (int i = tid; i < (n >> 2); i += SP_THREADS_LDS)
        reinterpret_cast<float4 *>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int sub = lane & 15;
    const int grp = lane >> 4;
    constexpr int NW = SP_THREADS_LDS / SPT_WAVE;
    const int32_t *idx_b = indices + (size_t)b * nnz;
    const float *val_b = values + (size_t)b * nnz;

    for (int r = part * NW + wave; r < S; r += splits * NW) {
        const int start = indptr[r], end = indptr[r + 1];
        // this lane's four words of x[b, r, :], rotated by the group id
        const float *xrow = x + ((size_t)b * S + r) * E;
        float xq[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int e = 16 * k + sub;
            xq[k] = (k < QUARTERS && e < E) ? xrow[e] : 0.0f;
        }
        float xr[4];  // xr[j] = xq[(j + grp) & 3]
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = (j + grp) & 3;
            xr[j] = (k & 2) ? ((k & 1) ? xq[3] : xq[2]) : ((k & 1) ? xq[1] : xq[0]);
        }
        for (int p0 = start; p0 < end; p0 += SPT_WAVE) {
            const int remaining = end - p0;
            const bool in = lane < remaining;
            const int my_idx = in ? idx_b[p0 + lane] : 0;
            const float my_val = in ? val_b[p0 + lane] : 0.0f;
            const int nsteps = min(16, (remaining + 3) >> 2);
            for (int s = 0; s < nsteps; s++) {
                const int e = s * 4 + grp;
                const int col = __shfl(my_idx, e, SPT_WAVE);
                const float v = __shfl(my_val, e, SPT_WAVE);
                if (e < remaining) {
                    float *arow = acc + (size_t)col * E + sub;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int k = (j + grp) & 3;
                        if (k < QUARTERS && (16 * k + sub) < E) atomicAdd(arow + 16 * k, v * xr[j]);
                    }
                }
            }
        }
    }
    __syncthreads();
    float *yb = y + (size_t)b * n;
    if (atomic_flush) {
        for (int i = tid; i < n; i += SP_THREADS_LDS) {
            const float v = acc[i];
            if (v != 0.0f) atomicAdd(yb + i, v);
        }
    } else {
        for (int i = tid; i < (n >> 2); i += SP_THREADS_LDS)
            reinterpret_cast<float4 *>(yb)[i] = reinterpret_cast<const float4 *>(acc)[i];
    }
}

// fallback for shapes whose [S,E] tile does not fit LDS: global float atomics into a
// zeroed y.  One group of LPE lanes per entry, float4 per lane.
template <int LPE>
__global__ __launch_bounds__(SP_THREADS) void spmm_t_global_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const float *__restrict__ x, float *__restrict__ y,
    int B, int S, int E, int nnz, int tiles_per_batch, int rows_per_block) {
    constexpr int EPS = SPT_WAVE / LPE;
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / tiles_per_batch;
    const int tile = lid - b * tiles_per_batch;
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane & (LPE - 1);
    const int grp = lane / LPE;
    const bool sub_live = (4 * sub) < E;
    const int row0 = tile * rows_per_block;
    const int row1 = min(S, row0 + rows_per_block);
    const int32_t *idx_b = indices + (size_t)b * nnz;
    const float *val_b = values + (size_t)b * nnz;
    float *yb = y + (size_t)b * S * E;
    for (int r = row0 + wave; r < row1; r += SP_THREADS / SPT_WAVE) {
        const int start = indptr[r], end = indptr[r + 1];
        float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sub_live)
            x4 = *reinterpret_cast<const float4 *>(x + ((size_t)b * S + r) * E + 4 * sub);
        for (int p0 = start; p0 < end; p0 += SPT_WAVE) {
            const int remaining = end - p0;
            const bool in = lane < remaining;
            const int my_idx = in ? idx_b[p0 + lane] : 0;
            const float my_val = in ? val_b[p0 + lane] : 0.0f;
            const int nsteps = min(LPE, (remaining + EPS - 1) / EPS);
            for (int s = 0; s < nsteps; s++) {
                const int e = s * EPS + grp;
                const int col = __shfl(my_idx, e, SPT_WAVE);
                const float v = __shfl(my_val, e, SPT_WAVE);
                if (e < remaining && sub_live) {
                    float *dst = yb + (size_t)col * E + 4 * sub;
                    atomicAdd(dst + 0, v * x4.x);
                    atomicAdd(dst + 1, v * x4.y);
                    atomicAdd(dst + 2, v * x4.z);
                    atomicAdd(dst + 3, v * x4.w);
                }
            }
        }
    }
}

}  // namespace spt

using namespace spt;

extern "C" int spt_spmm_forward(int trans_lhs, const int32_t *indptr, const int32_t *indices,
                                const float *values, const float *x, float *y,
                                int batch_size, int seq_length, int d_head, int nnz,
                                void *stream) {
    if (!indptr || !indices || !values || !x || !y) return SPT_EINVAL;
    if (batch_size <= 0 || seq_length <= 0 || d_head <= 0 || nnz < 0) return SPT_EINVAL;
    if (d_head % 4 != 0) return SPT_ESHAPE;
    if (d_head > 256) return SPT_EUNSUP;
    const int B = batch_size, S = seq_length, E = d_head;
    hipStream_t s = (hipStream_t)stream;
    const size_t ybytes = (size_t)B * S * E * sizeof(float);
    if (nnz == 0) {
        SPT_HIP_TRY(hipMemsetAsync(y, 0, ybytes, s));
        return SPT_OK;
    }
    const int LPE = pow2_ceil(E / 4);
    const size_t tile_bytes = (size_t)S * E * sizeof(float);
    const bool fits = tile_bytes <= 128 * 1024 && S >= 64;
    int splits = 1;
    while ((long long)B * splits < 256 && splits < 8) splits <<= 1;

    if (!trans_lhs) {
        if (fits && B >= 32) {
            dim3 grid((unsigned)(B * splits));
#define SPT_SPN_LDS(L)                                                                        \
    do {                                                                                      \
        SPT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&spmm_n_lds_kernel<L>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize,           \
                                        (int)tile_bytes));                                    \
        hipLaunchKernelGGL((spmm_n_lds_kernel<L>), grid, dim3(SP_THREADS_LDS), tile_bytes, s, \
                           indptr, indices, values, x, y, B, S, E, nnz, splits);              \
    } while (0)
            switch (LPE) {
                case 1: SPT_SPN_LDS(1); break;
                case 2: SPT_SPN_LDS(2); break;
                case 4: SPT_SPN_LDS(4); break;
                case 8: SPT_SPN_LDS(8); break;
                case 16: SPT_SPN_LDS(16); break;
                case 32: SPT_SPN_LDS(32); break;
                default: SPT_SPN_LDS(64); break;
            }
#undef SPT_SPN_LDS
        } else {
            const int rows_per_block = 16;
            const int tiles = (S + rows_per_block - 1) / rows_per_block;
            const long long nblk = (long long)B * tiles;
            if (nblk > 0x7FFFFFFFLL) return SPT_EUNSUP;
            dim3 grid((unsigned)nblk);
#define SPT_SPN_G(L)                                                                       \
    hipLaunchKernelGGL((spmm_n_global_kernel<L>), grid, dim3(SP_THREADS), 0, s, indptr,    \
                       indices, values, x, y, B, S, E, nnz, tiles, rows_per_block)
            switch (LPE) {
                case 1: SPT_SPN_G(1); break;
                case 2: SPT_SPN_G(2); break;
                case 4: SPT_SPN_G(4); break;
                case 8: SPT_SPN_G(8); break;
                case 16: SPT_SPN_G(16); break;
                case 32: SPT_SPN_G(32); break;
                default: SPT_SPN_G(64); break;
            }
#undef SPT_SPN_G
        }
    } else {
        if (fits && E <= 64) {
            const int atomic_flush = splits > 1;
            if (atomic_flush) SPT_HIP_TRY(hipMemsetAsync(y, 0, ybytes, s));
            dim3 grid((unsigned)(B * splits));
            const int quarters = (E + 15) / 16;
#define SPT_SPT_LDS(QQ)                                                                        \
    do {                                                                                       \
        SPT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&spmm_t_lds_kernel<QQ>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize,            \
                                        (int)tile_bytes));                                     \
        hipLaunchKernelGGL((spmm_t_lds_kernel<QQ>), grid, dim3(SP_THREADS_LDS), tile_bytes, s, \
                           indptr, indices, values, x, y, B, S, E, nnz, splits, atomic_flush); \
    } while (0)
            switch (quarters) {
                case 1: SPT_SPT_LDS(1); break;
                case 2: SPT_SPT_LDS(2); break;
                case 3: SPT_SPT_LDS(3); break;
                default: SPT_SPT_LDS(4); break;
            }
#undef SPT_SPT_LDS
        } else {
            SPT_HIP_TRY(hipMemsetAsync(y, 0, ybytes, s));
            const int rows_per_block = 16;
            const int tiles = (S + rows_per_block - 1) / rows_per_block;
            const long long nblk = (long long)B * tiles;
            if (nblk > 0x7FFFFFFFLL) return SPT_EUNSUP;
            dim3 grid((unsigned)nblk);
#define SPT_SPT_G(L)                                                                       \
    hipLaunchKernelGGL((spmm_t_global_kernel<L>), grid, dim3(SP_THREADS), 0, s, indptr,    \
                       indices, values, x, y, B, S, E, nnz, tiles, rows_per_block)
            switch (LPE) {
                case 1: SPT_SPT_G(1); break;
                case 2: SPT_SPT_G(2); break;
                case 4: SPT_SPT_G(4); break;
                case 8: SPT_SPT_G(8); break;
                case 16: SPT_SPT_G(16); break;
                case 32: SPT_SPT_G(32); break;
                default: SPT_SPT_G(64); break;
            }
#undef SPT_SPT_G
        }
    }
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
