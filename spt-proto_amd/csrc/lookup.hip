// lookup.hip -- causal, bucketed approximate top-Z key selection from PQ codes.
//
// Replaces extension/lookup.cu:10-84 of the reference (64-thread blocks, four
// "worker" threads per query row walking the key columns serially and appending to
// per-slot shared-memory lists).  The reference selector is order dependent and
// has two saturation quirks; parity demands the identical output, so this kernel
// computes the reference's result in closed form (verified against
// the literal emulation oracle/spt_oracle.c):
//
//   column c <= row belongs to worker tx = c % 4 and slot s = min(3, matches / (M/4));
//   list L[s][tx] = those columns in ascending order, n = |L|;  Q = Z / 4;
//   worker 0,1 keep min(n, Q) entries, worker 2,3 keep min(n, Q-1);
//   entry Q-1 of worker 0 (1) is replaced by the LAST column of L[s][3] (L[s][2])
//   when that list has >= Q entries and its last column is larger;
//   worker tx emits slot 3, 2, 1, 0 kept entries at output positions tx, tx+4, ...
//   while position < min(row+1, Z); everything else stays 0.
//
// MI355X mapping: one wave per query row, lane l owns columns 64w + l (so a lane's
// worker id l % 4 never changes), membership of the 16 lists is four wave ballots
// per 64-column window, ranks inside a list are mbcnt prefix popcounts.  Key codes of
// the batch are packed to uint16 pairs in LDS once per block (the reference compares
// uint16 truncations, lookup.cu:22,43).  The output row is assembled in LDS and
// written with one coalesced store, zeros included, so no memset pass is needed.
#include "spt_common.h"

namespace spt {

constexpr int LK_THREADS = 256;
constexpr int LK_WAVES = LK_THREADS / SPT_WAVE;
constexpr int LK_ROWS = 16;  // query rows per block

template <int M2>
struct Codes {
    uint32_t w[M2];
};

// number of equal uint16 halves between a (per-lane key) and b (wave-uniform query)
template <int M2>
__device__ __forceinline__ int match_count(const Codes<M2> &a, const Codes<M2> &b) {
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < M2; i++) {
        const uint32_t x = a.w[i] ^ b.w[i];
        cnt += ((x & 0xFFFFu) == 0u) + ((x >> 16) == 0u);
    }
    return cnt;
}

__device__ __forceinline__ int sel4(int v0, int v1, int v2, int v3, int s) {
    const int lo = (s & 1) ? v1 : v0;
    const int hi = (s & 1) ? v3 : v2;
    return (s & 2) ? hi : lo;
}

template <int M2>
__global__ __launch_bounds__(LK_THREADS) void lookup_forward_kernel(
    const int32_t *__restrict__ query, const int32_t *__restrict__ key,
    int32_t *__restrict__ out, int B, int S, int M, int Z, int tiles_per_batch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *kcodes = reinterpret_cast<uint32_t *>(smem);                 // [cols][M2]
    int32_t *rowbuf = reinterpret_cast<int32_t *>(smem) + (size_t)S * M2;  // [LK_WAVES][Z]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // heavy (late) row tiles first: they take longest, so they should start earliest
    const int b = blockIdx.x % B;
    const int tile = tiles_per_batch - 1 - (blockIdx.x / B);
    const int row0 = tile * LK_ROWS;
    const int ncols = row0 + LK_ROWS;  // columns any row of this block may look at

    // ---- pack this batch's key codes (columns < ncols) into uint16 pairs ----
    {
        const int32_t *ksrc = key + (size_t)b * S * M;
        for (int i = tid; i < ncols * M2; i += LK_THREADS) {
            const int col = i / M2, d = i - col * M2;
            const int k0 = 2 * d, k1 = 2 * d + 1;
            const uint32_t lo = (uint32_t)ksrc[(size_t)col * M + k0] & 0xFFFFu;
            // an odd M pads the last half so that it can never match (query pad = 0)
            const uint32_t hi = (k1 < M) ? ((uint32_t)ksrc[(size_t)col * M + k1] & 0xFFFFu) : 0xFFFFu;
            kcodes[i] = lo | (hi << 16);
        }
    }
    __syncthreads();

    const int tx = lane & 3;
    const unsigned long long wm_own = 0x1111111111111111ull << tx;   // lanes of my worker
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int Q = Z >> 2;
    const int div = M >> 2;  // matches per slot, lookup.cu:62
    const int cap = (tx < 2) ? Q : Q - 1;
    int32_t *myrow = rowbuf + wave * Z;

    for (int r = wave; r < LK_ROWS; r += LK_WAVES) {
        const int gy = row0 + r;
        const int limit = min(gy + 1, Z);
        const int nwin = (gy >> 6) + 1;

        // query codes of this row: wave-uniform
        Codes<M2> qc;
        {
            const int32_t *qsrc = query + ((size_t)b * S + gy) * M;
#pragma unroll
            for (int d = 0; d < M2; d++) {
                const uint32_t lo = (uint32_t)qsrc[2 * d] & 0xFFFFu;
                const uint32_t hi = (2 * d + 1 < M) ? ((uint32_t)qsrc[2 * d + 1] & 0xFFFFu) : 0u;
                qc.w[d] = lo | (hi << 16);
            }
        }

        // zero the staging row
        for (int i = lane; i < Z; i += SPT_WAVE) myrow[i] = 0;

        // ---------------- pass 1: sizes of the 16 lists, last column of tx=2,3 ----
        // n_own[s]  : |L[s][my tx]|          (per lane, identical for equal tx)
        // n_par[s]  : |L[s][3 - my tx]|      (partner worker: 0<->3, 1<->2)
        // last_par[s]: last column of the partner list
        int n_own1 = 0, n_own2 = 0, n_own3 = 0;
        int n_par0 = 0, n_par1 = 0, n_par2 = 0, n_par3 = 0;
        int lp0 = 0, lp1 = 0, lp2 = 0, lp3 = 0;
        const unsigned long long wm_par = 0x1111111111111111ull << (3 - tx);
        for (int w = 0; w < nwin; w++) {
            const int col = (w << 6) + lane;
            int slot = 4;  // invalid
            if (col <= gy) {
                Codes<M2> kc;
#pragma unroll
                for (int d = 0; d < M2; d++) kc.w[d] = kcodes[col * M2 + d];
                const int cnt = match_count<M2>(kc, qc);
                slot = (cnt >= div) + (cnt >= 2 * div) + (cnt >= 3 * div);
            }
            const unsigned long long m0 = __ballot(slot == 0);
            const unsigned long long m1 = __ballot(slot == 1);
            const unsigned long long m2 = __ballot(slot == 2);
            const unsigned long long m3 = __ballot(slot == 3);
            n_own1 += __popcll(m1 & wm_own);
            n_own2 += __popcll(m2 & wm_own); n_own3 += __popcll(m3 & wm_own);
            const unsigned long long p0 = m0 & wm_par, p1 = m1 & wm_par;
            const unsigned long long p2 = m2 & wm_par, p3 = m3 & wm_par;
            n_par0 += __popcll(p0); n_par1 += __popcll(p1);
            n_par2 += __popcll(p2); n_par3 += __popcll(p3);
            if (p0) lp0 = (w << 6) + 63 - __clzll(p0);
            if (p1) lp1 = (w << 6) + 63 - __clzll(p1);
            if (p2) lp2 = (w << 6) + 63 - __clzll(p2);
            if (p3) lp3 = (w << 6) + 63 - __clzll(p3);
        }
        // kept entries per slot for my worker and the output offset of each slot
        const int k3 = min(n_own3, cap), k2 = min(n_own2, cap), k1 = min(n_own1, cap);
        const int off3 = 0, off2 = k3, off1 = k3 + k2, off0 = k3 + k2 + k1;

        // ---------------- pass 2: ranks and placement -------------------------------
        int b0 = 0, b1 = 0, b2 = 0, b3 = 0;  // entries of L[s][my tx] before this window
        for (int w = 0; w < nwin; w++) {
            const int col = (w << 6) + lane;
            int slot = 4;
            if (col <= gy) {
                Codes<M2> kc;
#pragma unroll
                for (int d = 0; d < M2; d++) kc.w[d] = kcodes[col * M2 + d];
                const int cnt = match_count<M2>(kc, qc);
                slot = (cnt >= div) + (cnt >= 2 * div) + (cnt >= 3 * div);
            }
            const unsigned long long m0 = __ballot(slot == 0) & wm_own;
            const unsigned long long m1 = __ballot(slot == 1) & wm_own;
            const unsigned long long m2 = __ballot(slot == 2) & wm_own;
            const unsigned long long m3 = __ballot(slot == 3) & wm_own;
            if (slot < 4) {
                const unsigned long long mine =
                    (slot & 2) ? ((slot & 1) ? m3 : m2) : ((slot & 1) ? m1 : m0);
                const int rank = sel4(b0, b1, b2, b3, slot) + __popcll(mine & lt_mask);
                if (rank < cap) {
                    int val = col;
                    if (tx < 2 && rank == Q - 1) {
                        // reference quirk: the partner worker's cursor saturates on
                        // this word; the later (= larger) column survives
                        const int npar = sel4(n_par0, n_par1, n_par2, n_par3, slot);
                        const int lpar = sel4(lp0, lp1, lp2, lp3, slot);
                        if (npar >= Q) val = max(val, lpar);
                    }
                    const int pos = tx + 4 * (sel4(off0, off1, off2, off3, slot) + rank);
                    if (pos < limit) myrow[pos] = val;
                }
            }
            b0 += __popcll(m0); b1 += __popcll(m1); b2 += __popcll(m2); b3 += __popcll(m3);
        }

        // ---------------- coalesced store of the row (zeros included) ---------------
        __builtin_amdgcn_wave_barrier();
        int32_t *dst = out + ((size_t)b * S + gy) * Z;
        for (int i = lane; i < Z; i += SPT_WAVE) dst[i] = myrow[i];
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace spt

using namespace spt;

extern "C" int spt_lookup_forward(const int32_t *query, const int32_t *key, int32_t *out,
                                  int batch_size, int seq_length, int n_subspaces,
                                  int sparsity, void *stream) {
    if (!query || !key || !out) return SPT_EINVAL;
    if (batch_size <= 0 || seq_length <= 0 || n_subspaces <= 0 || sparsity <= 0)
        return SPT_EINVAL;
    const int S = seq_length, M = n_subspaces;
    if (S % 16 != 0 || S % sparsity != 0) return SPT_ESHAPE;  // lookup.cu:103-104
    const int Z = S / sparsity;
    if (Z % 16 != 0) return SPT_ESHAPE;                        // lookup.cu:106
    if (M < 4 || M > 16) return SPT_EUNSUP;                    // lookup.cu:167-169
    if (S > 65536) return SPT_EUNSUP;                          // uint16 columns, lookup.cu:32
    const int M2 = (M + 1) / 2;
    const size_t lds = (size_t)S * M2 * 4 + (size_t)LK_WAVES * Z * 4;
    if (lds > 160 * 1024) return SPT_EUNSUP;
    const int tiles = S / LK_ROWS;
    const long long nblk = (long long)batch_size * tiles;
    if (nblk > 0x7FFFFFFFLL) return SPT_EUNSUP;
    dim3 grid((unsigned)nblk);
    hipStream_t s = (hipStream_t)stream;
#define SPT_LK(MM2)                                                                          \
    do {                                                                                     \
        if (lds > 64 * 1024)                                                                 \
            SPT_HIP_TRY(hipFuncSetAttribute(                                                 \
                reinterpret_cast<const void *>(&lookup_forward_kernel<MM2>),                 \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
        hipLaunchKernelGGL((lookup_forward_kernel<MM2>), grid, dim3(LK_THREADS), lds, s,     \
                           query, key, out, batch_size, S, M, Z, tiles);                                 \
    } while (0)
    switch (M2) {
        case 2: SPT_LK(2); break;
        case 3: SPT_LK(3); break;
        case 4: SPT_LK(4); break;
        case 5: SPT_LK(5); break;
        case 6: SPT_LK(6); break;
        case 7: SPT_LK(7); break;
        case 8: SPT_LK(8); break;
        default: return SPT_EUNSUP;
    }
#undef SPT_LK
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
