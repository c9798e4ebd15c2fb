// sddmm.hip -- batched CSR-sampled Q.K^T for gfx950 (no hipSPARSE).
//
// Replaces extension/sddmm.cpp:27-69 of the reference (cusparseSDDMM with per-call
// descriptor + workspace allocation).  out[b,p] = dot(Q[b,row(p)], K[b,indices[b,p]]).
//
// Mapping.  A group of LPE = E/4 consecutive lanes owns one CSR entry: lane `sub`
// holds float4 #sub of the query row (registers, fixed for the whole row) and reads
// float4 #sub of the gathered key row, so one wave-instruction gathers 64/LPE whole
// key rows of E*4 contiguous bytes (E = 64: four 256-B rows).  The E-long dot is a
// DPP butterfly inside the group.  Results of LPE consecutive steps are kept by the
// lane whose `sub` equals the step, which turns the output of a 64-entry chunk into
// ONE coalesced 256-B store; the chunk's 64 column ids are ONE coalesced load,
// redistributed by ds_bpermute.
//
// Two variants:
//   KLDS = true : the batch's whole K [S,E] (<= 128 KiB of the CU's 160 KiB LDS) is
//                 staged once per workgroup; gathers are conflict-free ds_read_b128
//                 (a 256-B row covers all 64 banks).  HBM traffic = the algorithmic
//                 2*S*E*4 + 2*nnz*4 bytes per batch.
//   KLDS = false: any shape; key rows are gathered from global memory (L2 hits when
//                 the batch's blocks share an XCD, see xcd_remap).
#include "gather4.h"
#include "sparse_rows.h"

namespace spt {

constexpr int SD_THREADS = 256;        // gather-from-global variant
constexpr int SD_THREADS_LDS = 512;    // LDS-resident variant: one block per CU, 2 waves per SIMD (256 VGPRs each: no spills)

__device__ __forceinline__ float epilogue(float v, float scale, float clampv) {
    v *= scale;
    if (clampv > 0.0f) v = fminf(fmaxf(v, -clampv), clampv);
    return v;
}

// One CSR row: dot of the (register-resident) query row with every gathered key row.
// `chunk` holds the row's first 64 column ids (already loaded); longer rows load on.
template <int LPE>
__device__ __forceinline__ void sddmm_row(const int32_t *__restrict__ idx_b,
                                          const float4 q4,
                                          const float *__restrict__ kbase,  // global or LDS
                                          float *__restrict__ out_b, const RowChunk &chunk,
                                          int E, float scale, float clampv) {
    constexpr int EPS = SPT_WAVE / LPE;  // entries per step
    const int lane = lane_id();
    const int sub = lane & (LPE - 1);
    const int grp = lane / LPE;
    const bool sub_live = (4 * sub) < E;
    const float *krow0 = kbase + 4 * sub;

    int my_idx = chunk.idx;
    for (int p0 = chunk.start; p0 < chunk.end; p0 += SPT_WAVE) {
        const int remaining = chunk.end - p0;
        if (p0 != chunk.start) my_idx = (lane < remaining) ? idx_b[p0 + lane] : 0;
        float res = 0.0f;
        auto step = [&](int s) {
            const int e = s * EPS + grp;
            const int col = __shfl(my_idx, e, SPT_WAVE);
            float part = 0.0f;
            if (sub_live) {
                const float4 k4 = *reinterpret_cast<const float4 *>(krow0 + (size_t)col * E);
                part = q4.x * k4.x;
                part = fmaf(q4.y, k4.y, part);
                part = fmaf(q4.z, k4.z, part);
                part = fmaf(q4.w, k4.w, part);
            }
            const float tot = group_sum<LPE>(part);
            res = (sub == s) ? tot : res;
        };
        if (remaining >= SPT_WAVE) {
            // full chunk: fixed trip count, fully unrolled so the LPE gathers overlap
#pragma unroll
            for (int s = 0; s < LPE; s++) step(s);
        } else {
            const int nsteps = (remaining + EPS - 1) / EPS;
            for (int s = 0; s < nsteps; s++) step(s);
        }
        // lane (grp, sub) now holds entry e = sub * EPS + grp of this chunk
        const int e = sub * EPS + grp;
        if (e < remaining) out_b[p0 + e] = epilogue(res, scale, clampv);
    }
}

template <int LPE>
__device__ __forceinline__ void sddmm_rows(const int32_t *__restrict__ indptr,
                                           const int32_t *__restrict__ idx_b,
                                           const float *__restrict__ q_b,
                                           const float *__restrict__ kbase,
                                           float *__restrict__ out_b, int first, int stride,
                                           int S, int E, float scale, float clampv) {
    const int sub = lane_id() & (LPE - 1);
    const bool sub_live = (4 * sub) < E;
    auto load_q = [&](int row) {
        float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sub_live && row < S)
            q4 = *reinterpret_cast<const float4 *>(q_b + (size_t)row * E + 4 * sub);
        return q4;
    };
    for_each_row<false>(indptr, idx_b, nullptr, first, stride, S, load_q,
                        [&](int row, const RowChunk &chunk, const float4 &q4) {
                            sddmm_row<LPE>(idx_b, q4, kbase, out_b, chunk, E, scale, clampv);
                        });
}

template <int LPE>
__global__ __launch_bounds__(SD_THREADS) void sddmm_global_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ query, const float *__restrict__ key, float *__restrict__ out,
    int B, int S, int E, int nnz, int tiles_per_batch, int rows_per_block, float scale,
    float clampv) {
    // consecutive logical ids (= same batch) share an XCD and therefore an L2
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / tiles_per_batch;
    const int tile = lid - b * tiles_per_batch;
    if (b >= B) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row0 = tile * rows_per_block;
    const int row1 = min(S, row0 + rows_per_block);
    sddmm_rows<LPE>(indptr, indices + (size_t)b * nnz, query + (size_t)b * S * E,
                    key + (size_t)b * S * E, out + (size_t)b * nnz, row0 + wave,
                    SD_THREADS / SPT_WAVE, row1, E, scale, clampv);
}

template <int LPE>
__global__ __launch_bounds__(SD_THREADS_LDS) void sddmm_lds_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ query, const float *__restrict__ key, float *__restrict__ out,
    int B, int S, int E, int nnz, int splits, float scale, float clampv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *ktile = reinterpret_cast<float *>(smem);  // [S][E]
    const int b = blockIdx.x / splits;
    const int part = blockIdx.x - b * splits;
    const int tid = threadIdx.x;

    stage_tile(ktile, key + (size_t)b * S * E, (S * E) >> 2, tid, SD_THREADS_LDS);
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = SD_THREADS_LDS / SPT_WAVE;
    // interleave rows over (split, wave) so that causal patterns stay balanced
    sddmm_rows<LPE>(indptr, indices + (size_t)b * nnz, query + (size_t)b * S * E, ktile,
                    out + (size_t)b * nnz, part * NW + wave, splits * NW, S, E, scale, clampv);
}

// ---- fast path, E = 16 * LPE in {64, 128}: see gather4.h ----------------------------

template <int LPE>
__global__ __launch_bounds__(SD_THREADS_LDS) void sddmm_g4_lds_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ query, const float *__restrict__ key, float *__restrict__ out,
    int S, int nnz, int splits, float scale, float clampv, int q_heads, int k_heads) {
    constexpr int E = 16 * LPE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *ktile = reinterpret_cast<float *>(smem);  // [S][E]
    const int b = blockIdx.x / splits;
    const int part = blockIdx.x - b * splits;
    const int tid = threadIdx.x;
    const DenseView kv = dense_view(b, S, E, k_heads), qv = dense_view(b, S, E, q_heads);
    stage_rows(ktile, key + kv.base, kv.ld, S, E, tid, SD_THREADS_LDS);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = SD_THREADS_LDS / SPT_WAVE;
    gather_rows<LPE, G_SDDMM>(indptr, indices + (size_t)b * nnz, nullptr, nullptr, ktile,
                              query + qv.base, out + (size_t)b * nnz, part * NW + wave,
                              splits * NW, S, scale, clampv, qv.ld);
}

template <int LPE>
__global__ __launch_bounds__(SD_THREADS) void sddmm_g4_global_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ query, const float *__restrict__ key, float *__restrict__ out,
    int B, int S, int nnz, int blocks_per_batch, float scale, float clampv) {
    constexpr int E = 16 * LPE;
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / blocks_per_batch;
    const int part = lid - b * blocks_per_batch;
    if (b >= B) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NW = SD_THREADS / SPT_WAVE;
    gather_rows<LPE, G_SDDMM>(indptr, indices + (size_t)b * nnz, nullptr, nullptr,
                              key + (size_t)b * S * E, query + (size_t)b * S * E,
                              out + (size_t)b * nnz, part * NW + wave, blocks_per_batch * NW, S,
                              scale, clampv);
}

template <int LPE>
static int sddmm_g4_launch(const int32_t *indptr, const int32_t *indices, const float *query,
                           const float *key, float *out, int B, int S, int nnz, float scale,
                           float clampv, int q_heads, int k_heads, hipStream_t s) {
    constexpr int E = 16 * LPE;
    const size_t kbytes = (size_t)S * E * sizeof(float);
    if (kbytes <= 128 * 1024 && (long long)B * 8 >= 256) {
        int splits = 1;
        while ((long long)B * splits < 256 && splits < 8) splits <<= 1;
        SPT_HIP_TRY(hipFuncSetAttribute(
            reinterpret_cast<const void *>(&sddmm_g4_lds_kernel<LPE>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kbytes));
        hipLaunchKernelGGL((sddmm_g4_lds_kernel<LPE>), dim3((unsigned)(B * splits)),
                           dim3(SD_THREADS_LDS), kbytes, s, indptr, indices, query, key, out, S,
                           nnz, splits, scale, clampv, q_heads, k_heads);
    } else {
        if (q_heads > 0 || k_heads > 0) return SPT_EUNSUP;  // head layout: LDS path only
        // one block per 64 CSR rows of a batch
        int bpb = (S + 63) / 64;
        if (bpb < 1) bpb = 1;
        const long long nblk = (long long)B * bpb;
        if (nblk > 0x7FFFFFFFLL) return SPT_EUNSUP;
        hipLaunchKernelGGL((sddmm_g4_global_kernel<LPE>), dim3((unsigned)nblk), dim3(SD_THREADS),
                           0, s, indptr, indices, query, key, out, B, S, nnz, bpb, scale, clampv);
    }
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

// the dense-tile form (sddmm_tile.hip): SPT_EUNSUP for the shapes it leaves to the kernels above
int sddmm_tile_launch(const int32_t *indptr, const int32_t *indices, const float *query,
                      const float *key, float *out, int B, int S, int E, int nnz, float scale,
                      float clampv, int q_heads, int k_heads, hipStream_t s);
bool sddmm_tile_takes(int B, int S, int E, int nnz);

}  // namespace spt

using namespace spt;

extern "C" int spt_sddmm_form(int batch_size, int seq_length, int d_head, int nnz) {
    return sddmm_tile_takes(batch_size, seq_length, d_head, nnz) ? 1 : 0;
}

extern "C" int spt_sddmm_forward(const int32_t *indptr, const int32_t *indices,
                                 const float *query, const float *key, float *out,
                                 int batch_size, int seq_length, int d_head, int nnz,
                                 float scale, float clampv, int q_heads, int k_heads,
                                 void *stream) {
    if (!indptr || !indices || !query || !key || !out) return SPT_EINVAL;
    if (batch_size <= 0 || seq_length <= 0 || d_head <= 0 || nnz < 0) return SPT_EINVAL;
    if (d_head % 4 != 0) return SPT_ESHAPE;
    if (d_head > 256) return SPT_EUNSUP;
    if (nnz == 0) return SPT_OK;
    const int B = batch_size, S = seq_length, E = d_head;
    const int LPE = pow2_ceil(E / 4);
    hipStream_t s = (hipStream_t)stream;
    if (q_heads < 0 || k_heads < 0) return SPT_EINVAL;
    if ((q_heads > 0 && B % q_heads != 0) || (k_heads > 0 && B % k_heads != 0)) return SPT_ESHAPE;
    {
        const int rc = sddmm_tile_launch(indptr, indices, query, key, out, B, S, E, nnz, scale, clampv,
                                         q_heads, k_heads, s);
        if (rc != SPT_EUNSUP) return rc;
    }
    if (E == 64)
        return sddmm_g4_launch<4>(indptr, indices, query, key, out, B, S, nnz, scale, clampv,
                                  q_heads, k_heads, s);
    if (E == 128)
        return sddmm_g4_launch<8>(indptr, indices, query, key, out, B, S, nnz, scale, clampv,
                                  q_heads, k_heads, s);
    if (q_heads > 0 || k_heads > 0) return SPT_EUNSUP;

    const size_t kbytes = (size_t)S * E * sizeof(float);
    const bool use_lds = kbytes <= 128 * 1024 && (long long)B * 8 >= 256 && S >= 64;
    if (use_lds) {
        // one 1024-thread block per CU; split a batch's rows over several blocks only
        // when there are fewer batches than CUs
        int splits = 1;
        while ((long long)B * splits < 256 && splits < 8) splits <<= 1;
        dim3 grid((unsigned)(B * splits));
#define SPT_SD_LDS(L)                                                                        \
    do {                                                                                     \
        SPT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&sddmm_lds_kernel<L>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize,          \
                                        (int)kbytes));                                       \
        hipLaunchKernelGGL((sddmm_lds_kernel<L>), grid, dim3(SD_THREADS_LDS), kbytes, s,     \
                           indptr, indices, query, key, out, B, S, E, nnz, splits, scale,    \
                           clampv);                                                          \
    } while (0)
        switch (LPE) {
            case 1: SPT_SD_LDS(1); break;
            case 2: SPT_SD_LDS(2); break;
            case 4: SPT_SD_LDS(4); break;
            case 8: SPT_SD_LDS(8); break;
            case 16: SPT_SD_LDS(16); break;
            case 32: SPT_SD_LDS(32); break;
            default: SPT_SD_LDS(64); break;
        }
#undef SPT_SD_LDS
    } else {
        const int rows_per_block = 16;
        const int tiles = (S + rows_per_block - 1) / rows_per_block;
        const long long nblk = (long long)B * tiles;
        if (nblk > 0x7FFFFFFFLL) return SPT_EUNSUP;
        dim3 grid((unsigned)nblk);
#define SPT_SD_G(L)                                                                         \
    hipLaunchKernelGGL((sddmm_global_kernel<L>), grid, dim3(SD_THREADS), 0, s, indptr,      \
                       indices, query, key, out, B, S, E, nnz, tiles, rows_per_block, scale, \
                       clampv)
        switch (LPE) {
            case 1: SPT_SD_G(1); break;
            case 2: SPT_SD_G(2); break;
            case 4: SPT_SD_G(4); break;
            case 8: SPT_SD_G(8); break;
            case 16: SPT_SD_G(16); break;
            case 32: SPT_SD_G(32); break;
            default: SPT_SD_G(64); break;
        }
#undef SPT_SD_G
    }
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
