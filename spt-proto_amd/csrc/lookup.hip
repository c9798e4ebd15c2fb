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
// MI355X mapping: one wave per query row.  The row's candidate columns are cut into
// 64 contiguous segments of whole 4-column groups, one per lane, so a lane meets its
// columns in ascending order and the rank of a column inside its (slot, worker) list
// is: (entries of that list in lower lanes) + (a running count in the lane).  The
// first term is ONE wave-wide exclusive scan of the 16 per-lane list sizes (packed
// two per register, DPP row_shr / row_bcast, no LDS); no per-column cross-lane work.
// PQ codes are nibble-packed (one 32-bit word per token at M <= 8, C <= 16: a match
// count is xor + 3 bit-ops + popcount); any code outside [0,16) switches the block
// to the exact uint16 comparison of the reference (lookup.cu:22,43).  The output
// row is assembled in LDS with ds_max (the two writers of a saturated word resolve
// to the larger column, as in the reference) and stored coalesced, zeros included.
#include <type_traits>

#include "spt_common.h"

namespace spt {

constexpr int LK_THREADS = 256;
constexpr int LK_WAVES = LK_THREADS / SPT_WAVE;
constexpr int LK_ROWS = 16;  // query rows per block

// slot = min(3, matches / (M / 4)) (lookup.cu:61-63) as a multiply-shift with magic =
// ceil(32 / (M / 4)): exact for matches <= 16 and M / 4 in 1..4.  (Written out: min() of an
// unsigned and __umul24's int resolves to the double overload -- three fp64 instructions.)
__device__ __forceinline__ unsigned slot_of(unsigned matches, unsigned magic) {
    const unsigned t = (unsigned)__umul24(matches, magic) >> 5;
    return t < 3u ? t : 3u;
}

__device__ __forceinline__ int sel4(int v0, int v1, int v2, int v3, int s) {
    const int lo = (s & 1) ? v1 : v0;
    const int hi = (s & 1) ? v3 : v2;
    return (s & 2) ? hi : lo;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}

// wave-wide inclusive prefix sum (GCN sequence: 4 shifts inside each row of 16 lanes,
// then lane 15 -> next row, lane 31 -> upper half).  Fields packed in v must not carry.
// LANES = 64: over the wave; LANES = 32: independently over each half of the wave.
template <int LANES>
__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v) {
    v += dpp_u32<0x111, 0xF>(v);  // row_shr:1
    v += dpp_u32<0x112, 0xF>(v);  // row_shr:2
    v += dpp_u32<0x114, 0xF>(v);  // row_shr:4
    v += dpp_u32<0x118, 0xF>(v);  // row_shr:8
    v += dpp_u32<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
    if (LANES == 64) v += dpp_u32<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
    return v;
}

// value of the last lane of this lane's segment (LANES = 64: wave-uniform)
template <int LANES>
__device__ __forceinline__ unsigned segment_last(unsigned v) {
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
    if (LANES == 64) return hi;
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)v, 31);
    return (lane_id() & 32) ? hi : lo;
}

// ---- code representations -------------------------------------------------------
// NIB: W words per token, 4 bits per code; pad nibbles are 0xF in keys and 0 in
// queries so that they never match.  U16: W = ceil(M/2) words, 16 bits per code.
template <int W>
struct Code {
    uint32_t w[W];
};

template <int W, bool NIB>
__device__ __forceinline__ int match_count(const Code<W> &k, const Code<W> &q) {
    int cnt = 0;
    if (NIB) {
        int diff = 0;
#pragma unroll
        for (int i = 0; i < W; i++) {
            uint32_t x = k.w[i] ^ q.w[i];
            x |= x >> 1;
            x |= x >> 2;
            diff += __popc(x & 0x11111111u);
        }
        cnt = 8 * W - diff;
    } else {
#pragma unroll
        for (int i = 0; i < W; i++) {
            const uint32_t x = k.w[i] ^ q.w[i];
            cnt += ((x & 0xFFFFu) == 0u) + ((x >> 16) == 0u);
        }
    }
    return cnt;
}

template <int W, bool NIB>
__device__ __forceinline__ Code<W> pack_codes(const int32_t *__restrict__ src, int M,
                                              bool is_key) {
    Code<W> c;
    if (NIB) {
#pragma unroll
        for (int i = 0; i < W; i++) {
            uint32_t word = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = 8 * i + j;
                const uint32_t nib = (k < M) ? ((uint32_t)src[k] & 0xFu) : (is_key ? 0xFu : 0u);
                word |= nib << (4 * j);
            }
            c.w[i] = word;
        }
    } else {
#pragma unroll
        for (int i = 0; i < W; i++) {
            const int k0 = 2 * i, k1 = 2 * i + 1;
            const uint32_t lo = (k0 < M) ? ((uint32_t)src[k0] & 0xFFFFu) : (is_key ? 0xFFFFu : 0u);
            const uint32_t hi = (k1 < M) ? ((uint32_t)src[k1] & 0xFFFFu) : (is_key ? 0xFFFFu : 0u);
            c.w[i] = lo | (hi << 16);
        }
    }
    return c;
}

// One query row, one wave.  kcodes: LDS, Code<W> per column.
//
// Register budget of the inner loops (per 4-column group): pass 1 counts into one
// register per worker (four 8-bit fields, one per slot) and remembers the 3-bit slot of
// every column in a 64-bit word, so pass 2 neither re-reads LDS nor re-compares codes
// (rows with more than 20 columns per lane recompute instead).  Ranks live in 16-bit
// fields, two per register, exactly as the wave scan leaves them.  Output words are
// written with plain LDS stores: the only word with two writers is the one behind the
// reference's cursor-saturation quirk, fixed up after the loop by one lane.
//
// LANES lanes per query row: 64 (one row per wave) or 32 (two rows per wave, rows gy and
// gy + 1 in the lower / upper half).  The work per row is mostly fixed cost (scans, list
// offsets, fix-up, row store: ~600 instructions whatever the row length), so at S <= 1024,
// where 32 lanes still hold a row in <= 5 groups each, two rows per wave nearly halve it.
// `gy`, `qsrc`, `myrow`, `dst` are per-lane values (uniform inside a segment).
template <int W, bool NIB, int LANES>
__device__ __forceinline__ void lookup_row(const uint32_t *__restrict__ kcodes,
                                           const int32_t *__restrict__ qsrc,
                                           int32_t *__restrict__ myrow,
                                           int32_t *__restrict__ dst, int gy, int gy_max, int M,
                                           int Z) {
    const int lane = lane_id() & (LANES - 1);       // lane inside the row's segment
    const int Q = Z >> 2;
    const int limit = min(gy + 1, Z);
    const int ngroups = (gy + 4) >> 2;              // 4-column groups holding a candidate
    // groups per lane: from the longest row of the wave, so that the loops stay uniform
    const int gpl = (((gy_max + 4) >> 2) + LANES - 1) / LANES;
    const int g0 = lane * gpl;
    const int g1 = min(ngroups, g0 + gpl);
    const bool keep_slots = gpl <= 5;               // 20 columns x 3 bits
    // slot = min(3, matches / (M / 4)) (lookup.cu:61-63) as a multiply-shift: exact for
    // matches <= 16 and M / 4 in 1..4
    const int div = M >> 2;
    const unsigned magic = (32u + div - 1) / div;

    const Code<W> qc = pack_codes<W, NIB>(qsrc, M, false);  // uniform inside the segment

    for (int i = lane; i < Z; i += LANES) myrow[i] = 0;

    // slots of the 4 columns of group g (4 = not a candidate).  The group's codes are
    // 4*W consecutive LDS words, 16-byte aligned: W ds_read_b128.
    auto group_slots = [&](int g, int (&slot)[4]) {
        uint32_t raw[4 * W];
        const uint4 *src = reinterpret_cast<const uint4 *>(kcodes + (size_t)g * 4 * W);
#pragma unroll
        for (int i = 0; i < W; i++) {
            const uint4 t = src[i];
            raw[4 * i + 0] = t.x; raw[4 * i + 1] = t.y; raw[4 * i + 2] = t.z; raw[4 * i + 3] = t.w;
        }
#pragma unroll
        for (int tx = 0; tx < 4; tx++) {
            Code<W> kc;
#pragma unroll
            for (int d = 0; d < W; d++) kc.w[d] = raw[tx * W + d];
            const unsigned cnt = (unsigned)match_count<W, NIB>(kc, qc);
            const int sl = (int)min(3u, (cnt * magic) >> 5);
            slot[tx] = (4 * g + tx <= gy) ? sl : 4;
        }
    };

    // ---- pass 1: per-lane sizes of the 16 (worker, slot) lists -------------------------
    unsigned cnt8[4] = {0u, 0u, 0u, 0u};      // worker tx: four 8-bit counters (slot 0..3)
    int last2[4] = {-1, -1, -1, -1};          // worker 2: last column seen per slot
    int last3[4] = {-1, -1, -1, -1};          // worker 3
    unsigned long long saved = 0ull;
    for (int g = g0; g < g1; g++) {
        int slot[4];
        group_slots(g, slot);
#pragma unroll
        for (int tx = 0; tx < 4; tx++) {
            // slot 4 (not a candidate) shifts the 1 out of the register
            cnt8[tx] += (unsigned)((1ull << (8 * slot[tx])) & 0xFFFFFFFFull);
        }
#pragma unroll
        for (int sl = 0; sl < 4; sl++) {
            last2[sl] = (slot[2] == sl) ? 4 * g + 2 : last2[sl];
            last3[sl] = (slot[3] == sl) ? 4 * g + 3 : last3[sl];
        }
        if (keep_slots) {
            const unsigned four = (unsigned)slot[0] | ((unsigned)slot[1] << 3) |
                                  ((unsigned)slot[2] << 6) | ((unsigned)slot[3] << 9);
            saved |= (unsigned long long)four << (12 * (g - g0));
        }
    }

    // ---- exclusive scan over the lanes of the row; totals are uniform per row -----------
    unsigned posA[4], posB[4];   // worker tx: ranks of slots (0,1) and (2,3), 16 bits each
    int n[4][4];                 // list sizes
#pragma unroll
    for (int tx = 0; tx < 4; tx++) {
        const unsigned ownA = (cnt8[tx] & 0xFFu) | ((cnt8[tx] & 0xFF00u) << 8);
        const unsigned ownB = ((cnt8[tx] >> 16) & 0xFFu) | ((cnt8[tx] >> 24) << 16);
        const unsigned incA = wave_inclusive_scan<LANES>(ownA);
        const unsigned incB = wave_inclusive_scan<LANES>(ownB);
        posA[tx] = incA - ownA;
        posB[tx] = incB - ownB;
        const unsigned totA = segment_last<LANES>(incA);
        const unsigned totB = segment_last<LANES>(incB);
        n[tx][0] = totA & 0xFFFF;
        n[tx][1] = totA >> 16;
        n[tx][2] = totB & 0xFFFF;
        n[tx][3] = totB >> 16;
    }
    // kept entries per list and output offset of each slot (slot 3 first)
    int off[4][4];
#pragma unroll
    for (int tx = 0; tx < 4; tx++) {
        const int cap = (tx < 2) ? Q : Q - 1;
        const int k3 = min(n[tx][3], cap), k2 = min(n[tx][2], cap), k1 = min(n[tx][1], cap);
        off[tx][3] = 0;
        off[tx][2] = k3;
        off[tx][1] = k3 + k2;
        off[tx][0] = k3 + k2 + k1;
    }

    // ---- pass 2: placement -------------------------------------------------------------------
    for (int g = g0; g < g1; g++) {
        int slot[4];
        if (keep_slots) {
            const unsigned four = (unsigned)(saved >> (12 * (g - g0))) & 0xFFFu;
            slot[0] = four & 7; slot[1] = (four >> 3) & 7; slot[2] = (four >> 6) & 7;
            slot[3] = (four >> 9) & 7;
        } else {
            group_slots(g, slot);
        }
#pragma unroll
        for (int tx = 0; tx < 4; tx++) {
            const int sl = slot[tx];
            const unsigned both = (sl & 2) ? posB[tx] : posA[tx];
            const int rank = (int)((sl & 1) ? (both >> 16) : (both & 0xFFFFu));
            const unsigned inc = (sl & 1) ? 0x10000u : 1u;
            posA[tx] += (sl < 2) ? inc : 0u;
            posB[tx] += (sl == 2 || sl == 3) ? inc : 0u;
            const int cap = (tx < 2) ? Q : Q - 1;
            if (sl < 4 && rank < cap) {
                const int p =
                    tx + 4 * (sel4(off[tx][0], off[tx][1], off[tx][2], off[tx][3], sl) + rank);
                if (p < limit) myrow[p] = 4 * g + tx;
            }
        }
    }

    // ---- reference quirk: the cursor of worker 2 (3) saturates on the word that holds
    // entry Q-1 of worker 1 (0); its LAST candidate of the slot survives there if larger.
    // Rare (both lists need >= Q entries): the reduction runs only when some row of the
    // wave needs it.
#pragma unroll
    for (int sl = 0; sl < 4; sl++) {
#pragma unroll
        for (int tx = 2; tx < 4; tx++) {
            const int ptx = 3 - tx;
            const bool need = n[tx][sl] >= Q && n[ptx][sl] >= Q;     // uniform per row
            if (__builtin_amdgcn_ballot_w64(need) != 0ull) {          // uniform per wave
                int lastcol = (tx == 2) ? last2[sl] : last3[sl];
#pragma unroll
                for (int d = 1; d < LANES; d <<= 1)
                    lastcol = max(lastcol, __shfl_xor(lastcol, d, SPT_WAVE));
                const int p = ptx + 4 * (off[ptx][sl] + Q - 1);
                __builtin_amdgcn_wave_barrier();
                if (need && lane == 0 && p < limit) myrow[p] = max(myrow[p], lastcol);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }

    // ---- coalesced store of the row (zeros included) -----------------------------------
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < Z; i += LANES) dst[i] = myrow[i];
    __builtin_amdgcn_wave_barrier();
}

template <int WN, int WU, int LANES>  // words per token (nibble, uint16 form); lanes per row
__global__ __launch_bounds__(LK_THREADS) void lookup_forward_kernel(
    const int32_t *__restrict__ query, const int32_t *__restrict__ key,
    int32_t *__restrict__ out, int B, int S, int M, int Z, int tiles_per_batch) {
    constexpr int RPW = SPT_WAVE / LANES;   // rows per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *kcodes = reinterpret_cast<uint32_t *>(smem);                 // [cols][W]
    int32_t *rowbuf = reinterpret_cast<int32_t *>(smem) + (size_t)S * WU;  // [LK_WAVES * RPW][Z]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seg = (tid & 63) / LANES;     // which of the wave's rows this lane works on
    // heavy (late) row tiles first: they take longest, so they should start earliest
    const int b = blockIdx.x % B;
    const int tile = tiles_per_batch - 1 - (blockIdx.x / B);
    const int row0 = tile * LK_ROWS;
    const int ncols = row0 + LK_ROWS;  // columns any row of this block may look at
    const int32_t *ksrc = key + (size_t)b * S * M;
    const int32_t *qsrc = query + ((size_t)b * S + row0) * M;

    // ---- can this block use the 4-bit form?  (all codes it touches in [0, 16)) --------
    int wide = 0;
    for (int i = tid; i < ncols * M; i += LK_THREADS) wide |= ksrc[i];
    for (int i = tid; i < LK_ROWS * M; i += LK_THREADS) wide |= qsrc[i];
    const bool use_u16 = __syncthreads_or((wide & ~0xF) != 0);

    int32_t *myrow = rowbuf + (wave * RPW + seg) * Z;
    if (!use_u16) {
        for (int col = tid; col < ncols; col += LK_THREADS) {
            const Code<WN> c = pack_codes<WN, true>(ksrc + (size_t)col * M, M, true);
#pragma unroll
            for (int d = 0; d < WN; d++) kcodes[col * WN + d] = c.w[d];
        }
        __syncthreads();
        for (int r = wave * RPW; r < LK_ROWS; r += LK_WAVES * RPW) {
            const int gy = row0 + r + seg;
            lookup_row<WN, true, LANES>(kcodes, qsrc + (size_t)(r + seg) * M, myrow,
                                        out + ((size_t)b * S + gy) * Z, gy, row0 + r + RPW - 1,
                                        M, Z);
        }
    } else {
        for (int col = tid; col < ncols; col += LK_THREADS) {
            const Code<WU> c = pack_codes<WU, false>(ksrc + (size_t)col * M, M, true);
#pragma unroll
            for (int d = 0; d < WU; d++) kcodes[col * WU + d] = c.w[d];
        }
        __syncthreads();
        for (int r = wave * RPW; r < LK_ROWS; r += LK_WAVES * RPW) {
            const int gy = row0 + r + seg;
            lookup_row<WU, false, LANES>(kcodes, qsrc + (size_t)(r + seg) * M, myrow,
                                         out + ((size_t)b * S + gy) * Z, gy, row0 + r + RPW - 1,
                                         M, Z);
        }
    }
}

// ---- rows-on-lanes form (S <= 512) ---------------------------------------------------------
// The kernel above gives a wave one (or two) query rows and pays ~600 instructions of fixed cost
// per row (wave scans for the list ranks, offsets, fix-up, row store): 107 us at the bench shape,
// a quarter of the fine-tune step once attention moved to the matrix cores.  Here a wave takes
// 64 consecutive rows, one per LANE, and walks the key columns in ascending order, so the rank
// of a column inside its (slot, worker) list is simply the lane's running counter: no
// cross-lane work at all, the key codes are wave-uniform (one broadcast LDS read per 4
// columns).  Kept candidates are parked per list as one byte (column / 4: the worker is the
// list's) in a lane-private LDS region; when the row is complete the lane knows its list sizes,
// hence the reference's output positions, and assembles its row in LDS for a coalesced store.
// Same closed form as lookup_row (see the file header), same quirks, bit-identical output.
template <int W, bool NIB>
__device__ __forceinline__ int rows_match(const uint32_t *__restrict__ kc, const Code<W> &qc,
                                          const int32_t *__restrict__ kraw,
                                          const int32_t (&qraw)[16], int M) {
    if (NIB) {
        Code<W> k;
#pragma unroll
        for (int d = 0; d < W; d++) k.w[d] = kc[d];
        return match_count<W, true>(k, qc);
    }
    int cnt = 0;                       // exact uint16 comparison (lookup.cu:22,43)
    for (int m = 0; m < M; m++) cnt += ((kraw[m] ^ qraw[m]) & 0xFFFF) == 0;
    return cnt;
}

#ifndef LR_WAVES_VALUE
#define LR_WAVES_VALUE 4
#endif
constexpr int LR_WAVES = LR_WAVES_VALUE;     // waves per 64-row block: 2 or 4
constexpr int LR_WPW = 4 / LR_WAVES;         // workers per wave
constexpr int LR_THREADS = LR_WAVES * SPT_WAVE;
template <int W>
__global__ __launch_bounds__(LR_THREADS) void lookup_rows_kernel(
    const int32_t *__restrict__ query, const int32_t *__restrict__ key,
    int32_t *__restrict__ out, int B, int S, int M, int Z, int blocks_per_batch, int list_pitch,
    int out_pitch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & (SPT_WAVE - 1);
    // The four workers' lists are independent until the final fix-up, so the block's waves
    // split them: all walk all column groups, wave wv only looks at its workers' columns of
    // each.  A fraction of the walk per wave and more waves per CU for the same LDS: a lone
    // wave per SIMD ran this loop at ~10 cycles per instruction (78 us with one wave per block;
    // pairing heavy and light row blocks in one wave, equal work for every wave: 86 us; two
    // waves per block 61 us).
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x % B;
    const int blk = blocks_per_batch - 1 - (blockIdx.x / B);       // heavy row blocks first
    const int r0 = blk * SPT_WAVE, gy = r0 + lane;
    const int ncols = min(S, r0 + SPT_WAVE);
    const int ngroups = (ncols + 3) >> 2;
    const int Q = Z >> 2;
    uint32_t *kcodes = reinterpret_cast<uint32_t *>(smem);                    // [4 ngroups][W]
    const size_t kbytes = (size_t)(((S + 3) & ~3) + 4) * W * 4;     // + one group: the prefetch
    unsigned char *lists = reinterpret_cast<unsigned char *>(smem) + kbytes +
                           (size_t)lane * list_pitch;    // [16 lists][Q] bytes + 8 last-group bytes
    unsigned short *orow = reinterpret_cast<unsigned short *>(
        smem + kbytes + (size_t)SPT_WAVE * list_pitch);   // [64][out_pitch]
    unsigned *xch = reinterpret_cast<unsigned *>(
        smem + kbytes + (size_t)SPT_WAVE * list_pitch + (size_t)SPT_WAVE * out_pitch * 2);  // [64][2]
    const int32_t *ksrc = key + (size_t)b * S * M;
    const int32_t *qsrc = query + ((size_t)b * S + min(gy, S - 1)) * M;

    // 4-bit form unless some code this block touches is outside [0, 16).  One pass: every lane
    // loads the codes of its (<= 4) key columns with unconditional 16-byte loads (clamped
    // column: a load per code with a wait each, as a first version had, cost more than the
    // walk), checks them and packs them as nibbles; the packed words are only used if the
    // whole block passed the check.
    int wide = 0;
    int32_t qraw[16];
#pragma unroll
    for (int m = 0; m < 16; m++) qraw[m] = 0;
    if ((M & 3) == 0) {
#pragma unroll
        for (int m4 = 0; m4 < 4; m4++) {
            if (4 * m4 < M) {
                const int4 t = reinterpret_cast<const int4 *>(qsrc)[m4];
                qraw[4 * m4] = t.x; qraw[4 * m4 + 1] = t.y; qraw[4 * m4 + 2] = t.z; qraw[4 * m4 + 3] = t.w;
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < 16; m++)
            if (m < M) qraw[m] = qsrc[m];
    }
#pragma unroll
    for (int m = 0; m < 16; m++) wide |= qraw[m];
    Code<W> qc;
#pragma unroll
    for (int d = 0; d < W; d++) {
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) word |= ((uint32_t)qraw[8 * d + j] & 0xFu) << (4 * j);   // pad: 0
        qc.w[d] = word;
    }
    constexpr int KPL = 512 / LR_THREADS;               // key columns per thread at S <= 512
    int32_t kraw[KPL][16];
#pragma unroll
    for (int i = 0; i < KPL; i++) {
        const int col = min((int)threadIdx.x + LR_THREADS * i, ncols - 1);
        const int32_t *src = ksrc + (size_t)col * M;
#pragma unroll
        for (int m = 0; m < 16; m++) kraw[i][m] = 0;
        if ((M & 3) == 0) {
#pragma unroll
            for (int m4 = 0; m4 < 4; m4++) {
                if (4 * m4 < M) {
                    const int4 t = reinterpret_cast<const int4 *>(src)[m4];
                    kraw[i][4 * m4] = t.x; kraw[i][4 * m4 + 1] = t.y;
                    kraw[i][4 * m4 + 2] = t.z; kraw[i][4 * m4 + 3] = t.w;
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 16; m++)
                if (m < M) kraw[i][m] = src[m];
        }
    }
#pragma unroll
    for (int i = 0; i < KPL; i++) {
        const int col = threadIdx.x + LR_THREADS * i;
#pragma unroll
        for (int m = 0; m < 16; m++) wide |= kraw[i][m];
        if (col < 4 * ngroups) {
#pragma unroll
            for (int d = 0; d < W; d++) {
                uint32_t word = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int m = 8 * d + j;
                    // pad nibbles are 0xF in keys (0 in queries): they never match
                    const uint32_t nibv = (m < M && col < ncols) ? ((uint32_t)kraw[i][m] & 0xFu) : 0xFu;
                    word |= nibv << (4 * j);
                }
                kcodes[col * W + d] = word;
            }
        }
    }
    const bool nib = !__syncthreads_or((wide & ~0xF) != 0);        // (also publishes kcodes)
    unsigned char *lastg = lists + 16 * Q;            // [2 workers][4 slots]: last column / 4

    // slot = min(3, matches / (M / 4)) (lookup.cu:61-63) as a multiply-shift, as in lookup_row
    const int div = M >> 2;
    const unsigned magic = (32u + div - 1) / div;
    // list sizes of this wave's two workers: four 8-bit fields (slot 0..3) each; a list holds
    // at most S / 4 <= 128 columns at S <= 512
    unsigned n8[LR_WPW];
#pragma unroll
    for (int t2 = 0; t2 < LR_WPW; t2++) n8[t2] = 0u;

    // the group's four codes are read together, one group ahead: read one by one they sat
    // behind the byte stores (same LDS, no alias information) with a full LDS latency each
    auto group_codes = [&](int g, uint32_t (&raw)[4 * W]) {
        const uint4 *src = reinterpret_cast<const uint4 *>(kcodes + (size_t)g * 4 * W);
#pragma unroll
        for (int i = 0; i < W; i++) {
            const uint4 t = src[i];
            raw[4 * i + 0] = t.x; raw[4 * i + 1] = t.y; raw[4 * i + 2] = t.z; raw[4 * i + 3] = t.w;
        }
    };
    const unsigned Q4 = 4u * Q;
    const unsigned cap = LR_WPW * wv < 2 ? Q : Q - 1;   // workers 0, 1 keep Q, workers 2, 3 Q - 1
    auto walk = [&](auto nib_tag) {
        constexpr bool NIBF = decltype(nib_tag)::value;
        uint32_t nxt[4 * W];
        if (NIBF) group_codes(0, nxt);
        for (int g = 0; g < ngroups; g++) {
            uint32_t cur[4 * W];
#pragma unroll
            for (int i = 0; i < 4 * W; i++) cur[i] = nxt[i];
            if (NIBF) group_codes(g + 1, nxt);
#pragma unroll
            for (int t2 = 0; t2 < LR_WPW; t2++) {
                const int tx = LR_WPW * wv + t2;
                const int c = 4 * g + tx;
                const int cs = min(c, S - 1);
                uint32_t kw[W];
#pragma unroll
                for (int d = 0; d < W; d++) {
                    // (wave-uniform choice of the worker's column among the group's four)
                    const uint32_t a = cur[(t2)*W + d], bq = cur[(LR_WPW + t2) * W + d];
                    if (LR_WPW == 2) {
                        kw[d] = wv ? bq : a;
                    } else {
                        const uint32_t c2 = cur[2 * W + d], c3 = cur[3 * W + d];
                        kw[d] = wv == 0 ? a : (wv == 1 ? bq : (wv == 2 ? c2 : c3));
                    }
                }
                const unsigned cnt = (unsigned)rows_match<W, NIBF>(kw, qc, ksrc + (size_t)cs * M,
                                                                   qraw, M);
                const unsigned sl = slot_of(cnt, magic);
                const bool active = c <= gy;          // (c < ncols <= S follows for real rows)
                const unsigned sh = sl << 3;
                const unsigned rank = (n8[t2] >> sh) & 0xFFu;
                n8[t2] += (active ? 1u : 0u) << sh;
                if (active && rank < cap) lists[__umul24(sl, Q4) + tx * Q + rank] = (unsigned char)g;
                if (tx >= 2 && active) lastg[(tx - 2) * 4 + sl] = (unsigned char)g;
            }
        }
    };
    if (nib) walk(std::true_type{});
    else walk(std::false_type{});

    // ---- the rows are complete: sizes -> kept counts -> output positions of this wave's workers
    const int limit = min(gy + 1, Z);
    unsigned short *mine = orow + (size_t)lane * out_pitch;
    int n[LR_WPW][4], off[LR_WPW][4];
#pragma unroll
    for (int t2 = 0; t2 < LR_WPW; t2++) {
        const int tx = LR_WPW * wv + t2;
        n[t2][0] = n8[t2] & 0xFF; n[t2][1] = (n8[t2] >> 8) & 0xFF;
        n[t2][2] = (n8[t2] >> 16) & 0xFF; n[t2][3] = n8[t2] >> 24;
        const int k3 = min(n[t2][3], (int)cap), k2 = min(n[t2][2], (int)cap),
                  k1 = min(n[t2][1], (int)cap), k0 = min(n[t2][0], (int)cap);
        off[t2][3] = 0;
        off[t2][2] = k3;
        off[t2][1] = k3 + k2;
        off[t2][0] = k3 + k2 + k1;
        const int total = k3 + k2 + k1 + k0;
        // position index i of worker tx: the slot whose kept range holds it (slot 3 first)
        for (int i = 0; i < Q; i++) {
            const int p = tx + 4 * i;
            const int sl = i < off[t2][2] ? 3 : (i < off[t2][1] ? 2 : (i < off[t2][0] ? 1 : 0));
            const int r = i - sel4(off[t2][0], off[t2][1], off[t2][2], off[t2][3], sl);
            unsigned short v = 0;
            if (i < total && p < limit) v = (unsigned short)(4 * lists[(sl * 4 + tx) * Q + r] + tx);
            mine[p] = v;
        }
    }
#pragma unroll
    for (int t2 = 0; t2 < LR_WPW; t2++) {
        const int tx = LR_WPW * wv + t2;
        if (tx >= 2) xch[2 * lane + (tx - 2)] = n8[t2];
    }
    __syncthreads();
    // reference quirk: the cursor of worker 2 (3) saturates on the word that holds entry Q-1
    // of worker 1 (0); its LAST candidate of the slot survives there if larger.  Applied by the
    // wave that owns workers 0 and 1 (the positions), with wave 1's sizes and last columns.
#pragma unroll
    for (int t2 = 0; t2 < LR_WPW; t2++) {
        const int ptx = LR_WPW * wv + t2;                 // a worker this wave owns
        if (ptx < 2) {
            const int tx = 3 - ptx;                       // the worker whose cursor lands on it
#pragma unroll
            for (int sl = 0; sl < 4; sl++) {
                const int ntx = (int)((xch[2 * lane + (tx - 2)] >> (8 * sl)) & 0xFFu);
                if (ntx >= Q && n[t2][sl] >= Q) {
                    const int p = ptx + 4 * (off[t2][sl] + Q - 1);
                    const int lastcol = 4 * lastg[(tx - 2) * 4 + sl] + tx;
                    if (p < limit) mine[p] = (unsigned short)max((int)mine[p], lastcol);
                }
            }
        }
    }
    __syncthreads();
    // ---- coalesced store, zeros included: the waves take rows in turn ----
    const int nrows = min(SPT_WAVE, S - r0);
    for (int r = wv; r < nrows; r += LR_WAVES) {
        int32_t *dst = out + ((size_t)b * S + r0 + r) * Z;
        const unsigned short *src = orow + (size_t)r * out_pitch;
        for (int i = lane; i < Z; i += SPT_WAVE) dst[i] = src[i];
    }
}


// ---- rows-on-lanes form for long rows (S > 512): two walks instead of parked candidates ------
// Above S = 512 the per-lane candidate lists of lookup_rows_kernel no longer fit: a list holds
// up to Q = Z / 4 entries of 9+ bits, 16 lists per row, 64 rows per block = 128 KiB at Z = 256.
// But the lists are only needed because a candidate's output position depends on the FINAL sizes
// of the higher slots.  So: walk once counting (sizes of the four slot lists of the wave's
// worker, 16-bit fields of one 64-bit register; last column per slot for workers 2 and 3), derive
// the slot offsets, walk again with the same running counters and write every kept candidate
// straight to its output position in the LDS row.  Twice the comparisons, no list memory, no
// position loop; the key codes are stored per worker ([worker][group][W]) so that a wave reads
// the codes of its next four columns with one ds_read_b128 (W = 1).  Same closed form, same
// quirks, bit-identical output (tests: test_lookup_bit_exact at S = 1024 / 2048, M = 8 / 10 / 16).
template <int W>
__global__ __launch_bounds__(256) void lookup_rows_long_kernel(
    const int32_t *__restrict__ query, const int32_t *__restrict__ key,
    int32_t *__restrict__ out, int B, int S, int M, int Z, int blocks_per_batch, int GP,
    int out_pitch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & (SPT_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // = the worker (column mod 4)
    const int b = blockIdx.x % B;
    const int blk = blocks_per_batch - 1 - (blockIdx.x / B);           // heavy row blocks first
    const int r0 = blk * SPT_WAVE, gy = r0 + lane;
    const int ncols = min(S, r0 + SPT_WAVE);
    const int ngroups = (ncols + 3) >> 2;
    const int Q = Z >> 2;
    uint32_t *kcodes = reinterpret_cast<uint32_t *>(smem);                        // [4][GP][W]
    const size_t kbytes = (size_t)4 * GP * W * 4;
    unsigned short *orow = reinterpret_cast<unsigned short *>(smem + kbytes);     // [64][out_pitch]
    unsigned short *lastg_all = orow + (size_t)SPT_WAVE * out_pitch;              // [64][2][4]
    unsigned long long *xch = reinterpret_cast<unsigned long long *>(
        smem + kbytes + (size_t)SPT_WAVE * out_pitch * 2 + (size_t)SPT_WAVE * 16);   // [64][2]
    const int32_t *ksrc = key + (size_t)b * S * M;
    const int32_t *qsrc = query + ((size_t)b * S + min(gy, S - 1)) * M;

    // the block's output rows start as zeros (lookup.cu:107-109)
    for (int i = threadIdx.x; i < SPT_WAVE * out_pitch / 2; i += 256)
        reinterpret_cast<uint32_t *>(orow)[i] = 0u;

    // own query row: raw codes (exact form) and nibbles
    int wide = 0;
    int32_t qraw[16];
#pragma unroll
    for (int m = 0; m < 16; m++) qraw[m] = 0;
    if ((M & 3) == 0) {
#pragma unroll
        for (int m4 = 0; m4 < 4; m4++) {
            if (4 * m4 < M) {
                const int4 t = reinterpret_cast<const int4 *>(qsrc)[m4];
                qraw[4 * m4] = t.x; qraw[4 * m4 + 1] = t.y; qraw[4 * m4 + 2] = t.z; qraw[4 * m4 + 3] = t.w;
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < 16; m++)
            if (m < M) qraw[m] = qsrc[m];
    }
#pragma unroll
    for (int m = 0; m < 16; m++) wide |= qraw[m];
    Code<W> qc;
#pragma unroll
    for (int d = 0; d < W; d++) {
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) word |= ((uint32_t)qraw[8 * d + j] & 0xFu) << (4 * j);   // pad: 0
        qc.w[d] = word;
    }
    // key codes as nibbles, per worker; groups past the last real column (the walk's four-group
    // steps and its read-ahead touch them) hold 0xF nibbles, which match nothing
    const int gfill = min(GP, ((ngroups + 3) & ~3) + 4);
    for (int col = threadIdx.x; col < 4 * gfill; col += 256) {
        uint32_t word[W];
#pragma unroll
        for (int d = 0; d < W; d++) word[d] = 0xFFFFFFFFu;
        if (col < ncols) {
            const int32_t *src = ksrc + (size_t)col * M;
            int32_t kraw[16];
#pragma unroll
            for (int m = 0; m < 16; m++) kraw[m] = 15;
            if ((M & 3) == 0) {
#pragma unroll
                for (int m4 = 0; m4 < 4; m4++) {
                    if (4 * m4 < M) {
                        const int4 t = reinterpret_cast<const int4 *>(src)[m4];
                        kraw[4 * m4] = t.x; kraw[4 * m4 + 1] = t.y; kraw[4 * m4 + 2] = t.z; kraw[4 * m4 + 3] = t.w;
                    }
                }
            } else {
#pragma unroll
                for (int m = 0; m < 16; m++)
                    if (m < M) kraw[m] = src[m];
            }
#pragma unroll
            for (int m = 0; m < 16; m++)
                if (m < M) wide |= kraw[m];
#pragma unroll
            for (int d = 0; d < W; d++) {
                uint32_t wd = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int m = 8 * d + j;
                    wd |= ((m < M) ? ((uint32_t)kraw[m] & 0xFu) : 0xFu) << (4 * j);
                }
                word[d] = wd;
            }
        }
#pragma unroll
        for (int d = 0; d < W; d++) kcodes[((size_t)(col & 3) * GP + (col >> 2)) * W + d] = word[d];
    }
    const bool nib = !__syncthreads_or((wide & ~0xF) != 0);        // (also publishes kcodes, orow)

    // slot = min(3, matches / (M / 4)) (lookup.cu:61-63) as a multiply-shift, as in lookup_row
    const int div = M >> 2;
    const unsigned magic = (32u + div - 1) / div;
    const uint32_t *mycodes = kcodes + (size_t)wv * GP * W;
    auto load4 = [&](int g, uint32_t (&raw)[4 * W]) {
        const uint4 *src = reinterpret_cast<const uint4 *>(mycodes + (size_t)g * W);
#pragma unroll
        for (int i = 0; i < W; i++) {
            const uint4 t = src[i];
            raw[4 * i + 0] = t.x; raw[4 * i + 1] = t.y; raw[4 * i + 2] = t.z; raw[4 * i + 3] = t.w;
        }
    };
    // visit(group, column, slot shift (16 * slot), candidate?) for every column of this wave's worker
    auto walk = [&](auto nib_tag, auto &&visit) {
        constexpr bool NIBF = decltype(nib_tag)::value;
        uint32_t nxt[4 * W];
        if (NIBF) load4(0, nxt);
        for (int g = 0; g < ngroups; g += 4) {
            uint32_t cur[4 * W];
#pragma unroll
            for (int i = 0; i < 4 * W; i++) cur[i] = nxt[i];
            if (NIBF) load4(g + 4, nxt);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int c = 4 * (g + j) + wv;
                unsigned cnt = 0;
                if (NIBF) {
                    Code<W> k;
#pragma unroll
                    for (int d = 0; d < W; d++) k.w[d] = cur[j * W + d];
                    cnt = (unsigned)match_count<W, true>(k, qc);
                } else {                    // exact uint16 comparison (lookup.cu:22,43)
                    const int32_t *kraw = ksrc + (size_t)min(c, S - 1) * M;
                    for (int m = 0; m < M; m++) cnt += ((kraw[m] ^ qraw[m]) & 0xFFFF) == 0;
                }
                const unsigned sl = slot_of(cnt, magic);
                visit(g + j, c, sl << 4, c <= gy);   // (c < ncols <= S follows for real rows)
            }
        }
    };

    // ---- walk 1: list sizes (and the last column / 4 per slot of workers 2, 3) ----
    // Branch-free: the sizes are four 16-bit fields of one 64-bit register; "last group + 1" per
    // slot is a packed 16-bit maximum (groups come in ascending order).
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    unsigned long long n64 = 0ull;
    u16x2 last_lo = {0, 0}, last_hi = {0, 0};
    auto count = [&](int g, int c, unsigned sh, bool active) {
        n64 += (unsigned long long)(active ? 1u : 0u) << sh;
    };
    auto count_last = [&](int g, int c, unsigned sh, bool active) {
        n64 += (unsigned long long)(active ? 1u : 0u) << sh;
        const unsigned long long cand = (unsigned long long)(active ? (unsigned)g + 1u : 0u) << sh;
        last_lo = __builtin_elementwise_max(last_lo, __builtin_bit_cast(u16x2, (unsigned)cand));
        last_hi = __builtin_elementwise_max(last_hi, __builtin_bit_cast(u16x2, (unsigned)(cand >> 32)));
    };
    if (wv >= 2) {
        if (nib) walk(std::true_type{}, count_last);
        else walk(std::false_type{}, count_last);
    } else {
        if (nib) walk(std::true_type{}, count);
        else walk(std::false_type{}, count);
    }
    unsigned short *lastg = lastg_all + (size_t)lane * 8;
    if (wv >= 2) {                       // (read only for lists with >= Q entries)
        lastg[(wv - 2) * 4 + 0] = (unsigned short)(last_lo.x - 1);
        lastg[(wv - 2) * 4 + 1] = (unsigned short)(last_lo.y - 1);
        lastg[(wv - 2) * 4 + 2] = (unsigned short)(last_hi.x - 1);
        lastg[(wv - 2) * 4 + 3] = (unsigned short)(last_hi.y - 1);
    }

    // kept entries per list and the output offset of each slot (slot 3 first)
    const int cap = wv < 2 ? Q : Q - 1;                  // workers 0, 1 keep Q, workers 2, 3 Q - 1
    const int limit = min(gy + 1, Z);
    int n[4], off[4];
#pragma unroll
    for (int sl = 0; sl < 4; sl++) n[sl] = (int)((n64 >> (16 * sl)) & 0xFFFFull);
    off[3] = 0;
    off[2] = min(n[3], cap);
    off[1] = off[2] + min(n[2], cap);
    off[0] = off[1] + min(n[1], cap);
    const unsigned long long off64 = (unsigned long long)off[0] | ((unsigned long long)off[1] << 16) |
                                     ((unsigned long long)off[2] << 32);
    // ---- walk 2: placement (only this wave writes positions = wv mod 4 of a row).  A column
    // that is not kept goes to the row's spare entry Z + 1: no branch around the store. ----
    unsigned short *mine = orow + (size_t)lane * out_pitch;
    unsigned long long r64 = 0ull;
    auto place = [&](int g, int c, unsigned sh, bool active) {
        const int rank = (int)((r64 >> sh) & 0xFFFFull);
        r64 += (unsigned long long)(active ? 1u : 0u) << sh;
        const int p = wv + 4 * ((int)((off64 >> sh) & 0xFFFFull) + rank);
        const bool keep = active && rank < cap && p < limit;
        mine[keep ? p : Z + 1] = (unsigned short)c;
    };
    if (nib) walk(std::true_type{}, place);
    else walk(std::false_type{}, place);

    if (wv >= 2) xch[2 * lane + (wv - 2)] = n64;
    __syncthreads();
    // reference quirk: the cursor of worker 2 (3) saturates on the word that holds entry Q-1
    // of worker 1 (0); its LAST candidate of the slot survives there if larger.
    if (wv < 2) {
        const int tx = 3 - wv;                            // the worker whose cursor lands on ours
        const unsigned long long theirs = xch[2 * lane + (tx - 2)];
#pragma unroll
        for (int sl = 0; sl < 4; sl++) {
            const int ntx = (int)((theirs >> (16 * sl)) & 0xFFFFull);
            if (ntx >= Q && n[sl] >= Q) {
                const int p = wv + 4 * (off[sl] + Q - 1);
                const int lastcol = 4 * (int)lastg[(tx - 2) * 4 + sl] + tx;
                if (p < limit) mine[p] = (unsigned short)max((int)mine[p], lastcol);
            }
        }
    }
    __syncthreads();
    // ---- coalesced store, zeros included: the waves take rows in turn ----
    const int nrows = min(SPT_WAVE, S - r0);
    for (int r = wv; r < nrows; r += 4) {
        int32_t *dst = out + ((size_t)b * S + r0 + r) * Z;
        const unsigned short *src = orow + (size_t)r * out_pitch;
        for (int i = lane; i < Z; i += SPT_WAVE) dst[i] = src[i];
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
    if (S > 32768) return SPT_EUNSUP;                          // 8-bit per-lane counters (ref: uint16 columns, lookup.cu:32)
    const int WU = (M + 1) / 2;
    hipStream_t s = (hipStream_t)stream;
    if (S <= 512) {
        // rows-on-lanes form: column / 4 and the list sizes (<= S / 4) fit in bytes
        const int WN = (M + 7) / 8;
        const int Q = Z / 4;
        const int list_pitch = ((16 * Q + 8 + 3) & ~3) | 4;      // odd number of words
        const int out_pitch = Z + 2;                              // uint16: odd number of words
        const size_t lds = (size_t)(((S + 3) & ~3) + 4) * WN * 4 + (size_t)SPT_WAVE * list_pitch +
                           (size_t)SPT_WAVE * out_pitch * 2 + (size_t)SPT_WAVE * 8;
        const int nb = (S + SPT_WAVE - 1) / SPT_WAVE;
        const long long nblk = (long long)batch_size * nb;
        if (lds <= 64 * 1024 && nblk <= 0x7FFFFFFFLL) {
            if (WN == 1)
                hipLaunchKernelGGL(lookup_rows_kernel<1>, dim3((unsigned)nblk), dim3(LR_THREADS), lds,
                                   s, query, key, out, batch_size, S, M, Z, nb, list_pitch,
                                   out_pitch);
            else
                hipLaunchKernelGGL(lookup_rows_kernel<2>, dim3((unsigned)nblk), dim3(LR_THREADS), lds,
                                   s, query, key, out, batch_size, S, M, Z, nb, list_pitch,
                                   out_pitch);
            SPT_LAUNCH_CHECK();
            return SPT_OK;
        }
    }
    if (S <= 65535) {
        // long rows on lanes: two walks (columns fit 16 bits, so do the list sizes)
        const int WN = (M + 7) / 8;
        const int GP = S / 4 + 8;
        const int out_pitch = Z + 2;                              // uint16: odd number of words
        const size_t lds = (size_t)4 * GP * WN * 4 + (size_t)SPT_WAVE * out_pitch * 2 +
                           (size_t)SPT_WAVE * 16 + (size_t)SPT_WAVE * 16;
        const int nb = (S + SPT_WAVE - 1) / SPT_WAVE;
        const long long nblk = (long long)batch_size * nb;
        if (lds <= 64 * 1024 && nblk <= 0x7FFFFFFFLL) {
            if (WN == 1)
                hipLaunchKernelGGL(lookup_rows_long_kernel<1>, dim3((unsigned)nblk), dim3(256), lds, s,
                                   query, key, out, batch_size, S, M, Z, nb, GP, out_pitch);
            else
                hipLaunchKernelGGL(lookup_rows_long_kernel<2>, dim3((unsigned)nblk), dim3(256), lds, s,
                                   query, key, out, batch_size, S, M, Z, nb, GP, out_pitch);
            SPT_LAUNCH_CHECK();
            return SPT_OK;
        }
    }
    // two rows per wave while 32 lanes hold a row in <= 5 groups of 4 columns each
    const int lanes = (S <= 640) ? 32 : 64;
    const size_t lds = (size_t)S * WU * 4 + (size_t)LK_WAVES * (SPT_WAVE / lanes) * Z * 4;
    if (lds > 160 * 1024) return SPT_EUNSUP;
    const int tiles = S / LK_ROWS;
    const long long nblk = (long long)batch_size * tiles;
    if (nblk > 0x7FFFFFFFLL) return SPT_EUNSUP;
    dim3 grid((unsigned)nblk);
#define SPT_LK2(WN_, WU_, LN_)                                                                \
    do {                                                                                      \
        if (lds > 64 * 1024)                                                                  \
            SPT_HIP_TRY(hipFuncSetAttribute(                                                  \
                reinterpret_cast<const void *>(&lookup_forward_kernel<WN_, WU_, LN_>),        \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
        hipLaunchKernelGGL((lookup_forward_kernel<WN_, WU_, LN_>), grid, dim3(LK_THREADS),    \
                           lds, s, query, key, out, batch_size, S, M, Z, tiles);              \
    } while (0)
#define SPT_LK(WN_, WU_)                                    \
    do {                                                    \
        if (lanes == 32) SPT_LK2(WN_, WU_, 32);             \
        else SPT_LK2(WN_, WU_, 64);                         \
    } while (0)
    switch (WU) {
        case 2: SPT_LK(1, 2); break;
        case 3: SPT_LK(1, 3); break;
        case 4: SPT_LK(1, 4); break;
        case 5: SPT_LK(2, 5); break;
        case 6: SPT_LK(2, 6); break;
        case 7: SPT_LK(2, 7); break;
        case 8: SPT_LK(2, 8); break;
        default: return SPT_EUNSUP;
    }
#undef SPT_LK2
#undef SPT_LK
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
