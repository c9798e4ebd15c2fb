// mfma_attention.hip -- the sparse attention core on the matrix cores (gfx950).
//
// Reference: SparseVanillaAttentionV2._get_attn / _apply_attn
// (naive_gpt/layers/sparse/attention.py:106-142) and its backward through kernels/sddmm.py,
// kernels/spmm.py, extension/softmax.cu.  Same mathematics as fused_attention.hip, different
// machine mapping.  At the densities lookup produces (Z / S = 1/8 at S = 512) a gather kernel
// moves a 256-byte K or V row through LDS per CSR entry and is LDS-bandwidth bound at ~1/5 of
// the HBM roofline.  Here the CSR rows only decide WHICH cells of a dense
// 32 x 32 score tile are alive:
//
//   tile      D[j, i] = sum_e K[j, e] Q[i, e]       v_mfma_f32_32x32x16_bf16, fp32 accumulate
//   cells     p[j, i] = m[i, j] * exp(clamp(scale * D[j, i]))   m = multiplicity of column j in
//             CSR row i (0 for most cells; > 1 only for lookup's zero padding), col <= row only
//   product   Y[i, :] += sum_j p[j, i] V[j, :]      the accumulator tile is the next A operand
//   row       y = Y / max(1e-9, sum_j p)            (softmax.cu:30; no max-subtraction, as there)
//
// fp32 operands are split into two bf16 halves x = hi + lo (hi = RNE(x), lo = RNE(x - hi)) and
// every product is three MFMAs (lo*hi + hi*lo + hi*hi): relative error <= 2^-16 per product,
// 50x inside the 1e-3 parity bar, at 1/5 of the fp32-MFMA cost.
//
// A workgroup of 8 waves owns 256 rows of one (sample, head) slice, a wave 32 rows.  K and V
// stream through LDS 32 keys at a time as bf16 images (K row-major for the A operand, V
// transposed for the B operand, double-buffered, one barrier per tile); a wave skips the
// tiles its rows have no entries in, which includes everything right of the diagonal
// (softmax.cu:19-31 masks col > row).  The multiplicities m come from a one-off pass over the
// CSR (spt_attention_mfma_prepare) that stores, per (32-row tile, 32-key tile), the 1 KiB of
// cell counts already permuted into the accumulator layout: a consumer lane loads its 16
// bytes with one global_load_dwordx4 a tile ahead and touches no LDS for them.
#include "spt_common.h"
#include <stdlib.h>

// The file is compiled twice: for d_head 64 (this is the unit that also holds the cell-tile
// pass and the C entry points) and, with -DMA_E_VALUE=128, for d_head 128 into namespace
// spt::e128 (mfma_attention_e128.o), to which the entry points dispatch.
#ifndef MA_E_VALUE
#define MA_E_VALUE 64
#endif
#if MA_E_VALUE == 64
#define MA_NS_OPEN namespace spt {
#define MA_NS_CLOSE }
#else
#define MA_NS_OPEN namespace spt { namespace e128 {
#define MA_NS_CLOSE } }
#endif

MA_NS_OPEN

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifndef MA_THREADS_VALUE
#define MA_THREADS_VALUE 512
#endif
constexpr int MA_THREADS = MA_THREADS_VALUE;       // 512 or 256
constexpr int MA_WAVES = MA_THREADS / SPT_WAVE;    // 8 or 4
constexpr int MA_E = MA_E_VALUE;                   // d_head: 64 or 128
constexpr int MA_EQ = MA_E / 4;                    // float4 per row of a tile
constexpr int MA_RPT = 32 * MA_EQ / MA_THREADS;    // 32 x E tile: float4s per thread to stage
constexpr int MA_RPP = MA_THREADS / MA_EQ;         // tile rows covered by one float4 per thread
constexpr int MA_KS = MA_E / 16;                   // k-steps of a contraction over e
constexpr int MA_ET = MA_E / 32;                   // 32-column tiles of an [*, E] result
constexpr int MA_BH = MA_E / 64;                   // backward launches: 64 gradient columns each
constexpr int MA_WROWS = 32;                       // rows per wave = one MFMA tile
constexpr int MA_ROWS = MA_WAVES * MA_WROWS;       // 256 rows per workgroup
constexpr int MA_KT = 32;                          // keys per tile
[[maybe_unused]] constexpr int MA_MAXZ = 256;      // entries per row (cell counts are bytes)
[[maybe_unused]] constexpr int MA_MAXNT = 64;      // key tiles: S <= 2048

// LDS images of one key tile, bf16: rows padded so that the operand reads are conflict-free
constexpr int MA_KLD = MA_E * 2 + 16;              // bytes per row of a rows image: E bf16 + 16
constexpr int MA_VLD = 72;                         // bytes per e row of V^T: 32 bf16 + 8
// the forward's tile: K rows image (hi, lo) | V rows image (hi, lo)
constexpr int MA_KH = 0, MA_KL = MA_KT * MA_KLD, MA_VH = 2 * MA_KT * MA_KLD,
              MA_VL = 3 * MA_KT * MA_KLD, MA_IMG = 4 * MA_KT * MA_KLD;         // 18432 B
constexpr int MA_CLD = 9;                          // words per row of the multiplicity tile
constexpr int MA_CELLS = MA_WROWS * MA_KT;         // bytes of one stored cell tile
constexpr int MA_TLD = 36;                         // floats per row of the epilogue's [e][row] tile

struct Split { unsigned hi, lo; };
typedef __attribute__((ext_vector_type(2))) float f32x2;
// two floats -> packed bf16 pairs (first value in the low half): hi = RNE, lo = RNE(x - hi).
// Whole-vector conversions: element-wise (__bf16) casts compile to one v_cvt_pk_bf16_f32 per
// ELEMENT plus re-packing (81 conversions per tile instead of 24).
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2 x = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
}
__device__ __forceinline__ Split split2(float a, float b) {
    const f32x2 x = {a, b};
    Split s;
    s.hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    const f32x2 hf = {__builtin_bit_cast(float, s.hi << 16),
                      __builtin_bit_cast(float, s.hi & 0xffff0000u)};
    s.lo = __builtin_bit_cast(unsigned, __builtin_convertvector(x - hf, bf16x2));
    return s;
}
struct Frag { uint4 hi, lo; };      // 8 bf16 each: one MFMA operand fragment, split
__device__ __forceinline__ Frag split8(float a0, float a1, float a2, float a3, float a4,
                                       float a5, float a6, float a7) {
    const Split s0 = split2(a0, a1), s1 = split2(a2, a3), s2 = split2(a4, a5),
                s3 = split2(a6, a7);
    Frag f;
    f.hi = make_uint4(s0.hi, s1.hi, s2.hi, s3.hi);
    f.lo = make_uint4(s0.lo, s1.lo, s2.lo, s3.lo);
    return f;
}
__device__ __forceinline__ f32x16 mma(const uint4 &a, const uint4 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// (a.hi + a.lo)(b.hi + b.lo) without the lo*lo term, small terms first.  PA / PB = parts of the
// operand: 2 = an fp32 value split in two, 1 = a value that IS a bf16 (bf16 storage: no lo part).
template <int PA, int PB>
__device__ __forceinline__ f32x16 mm(const Frag &a, const Frag &b, f32x16 c) {
    if (PA == 2) c = mma(a.lo, b.hi, c);
    if (PB == 2) c = mma(a.hi, b.lo, c);
    return mma(a.hi, b.hi, c);
}
__device__ __forceinline__ void wave_lds_fence() {
    // LDS executes one wave's instructions in order; this only stops the compiler from
    // moving accesses of different lanes' data across the point
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The lane id, from an instruction the compiler will not move.  For the kernels' RARE branches:
// whatever such a branch derives from the ordinary `lane` is loop-invariant, gets hoisted out of
// the tile loop, and then occupies registers (or scratch) for the whole kernel -- 16 hoisted shift
// counts of the saturation fix were spilled in the prologue of the key-owned backward.
__device__ __forceinline__ int rare_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// accumulator register r of lane half h holds tile row  (r & 3) + 8 (r >> 2) + 4 h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- element types of the dense operands (q, k, v, y and their gradients) ----
// float: the reference's fp32 tensors, every value split in two bf16 parts on its way to the
// matrix cores.  bf16_t (raw 16-bit patterns): bf16 STORAGE -- a stored value is its own hi part
// and has no lo part, so operands that go to LDS unchanged (K, V, Q tiles) are copied, not
// converted, and their products cost one MFMA per part of the OTHER operand.  Values computed in
// fp32 on the way (scaled Q, probabilities, dS, weighted dY) are split in two as before, so the
// only rounding the bf16 path adds is that of its stored outputs.
typedef unsigned short bf16_t;
__device__ __forceinline__ float bf16_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int PARTS = 2;
    typedef float4 raw4;                                 // four consecutive elements, as loaded
    static __device__ __forceinline__ float4 f4(const float4 &r) { return r; }
};
template <> struct Elem<bf16_t> {
    static constexpr int PARTS = 1;
    typedef uint2 raw4;
    static __device__ __forceinline__ float4 f4(const uint2 &r) {
        return make_float4(bf16_lo(r.x), bf16_hi(r.x), bf16_lo(r.y), bf16_hi(r.y));
    }
};
template <typename T>
__device__ __forceinline__ typename Elem<T>::raw4 ld_raw4(const T *p) {
    return *reinterpret_cast<const typename Elem<T>::raw4 *>(p);
}
template <typename T>
__device__ __forceinline__ float4 ld4(const T *p) { return Elem<T>::f4(ld_raw4(p)); }
__device__ __forceinline__ void st4(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void st4(bf16_t *p, const float4 &v) {
    *reinterpret_cast<uint2 *>(p) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
}
__device__ __forceinline__ void st1(float *p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t *p, float v) { *p = (bf16_t)(pack_bf16(v, 0.0f) & 0xffffu); }

// ---- 32-row tiles of a dense operand: global -> registers -> bf16 LDS images ----
// "rows" image [r][e] (144-byte rows): A-operand fragments by read_rows;
// "cols" image [e][r] (72-byte rows):  B-operand fragments in accumulator k order by read_cols.
// An image is a hi part followed (fp32 sources) by a lo part of the same shape.
constexpr int MA_RIMG = MA_KT * MA_KLD;            // one rows image part: 4608 B
constexpr int MA_CIMG = MA_E * MA_VLD;             // one cols image part: 4608 B
// four values of row r at e4 .. e4 + 3 into a rows image: fp32 (split, two parts) ...
__device__ __forceinline__ void put_rows4(char *img, int r, int e4, const float4 &x) {
    const Split a = split2(x.x, x.y), b = split2(x.z, x.w);
    *reinterpret_cast<uint2 *>(img + r * MA_KLD + e4 * 2) = make_uint2(a.hi, b.hi);
    *reinterpret_cast<uint2 *>(img + MA_RIMG + r * MA_KLD + e4 * 2) = make_uint2(a.lo, b.lo);
}
// ... or stored bf16 (copied, one part)
__device__ __forceinline__ void put_rows4(char *img, int r, int e4, const uint2 &x) {
    *reinterpret_cast<uint2 *>(img + r * MA_KLD + e4 * 2) = x;
}
// four fp32 values of column e at rows r4 .. r4 + 3 (an operand stored [E][S]) into a cols image
__device__ __forceinline__ void put_cols4(char *img, int e, int r4, const float4 &x) {
    const Split a = split2(x.x, x.y), b = split2(x.z, x.w);
    *reinterpret_cast<uint2 *>(img + e * MA_VLD + r4 * 2) = make_uint2(a.hi, b.hi);
    *reinterpret_cast<uint2 *>(img + MA_CIMG + e * MA_VLD + r4 * 2) = make_uint2(a.lo, b.lo);
}

// Key tiles of the forward and of the row-owned backward: K rows image | V rows image (each
// hi, lo).  Thread t stages the four elements (row t / (E/4) + MA_RPP u, columns 4 (t % (E/4)) ..)
// for u < MA_RPT.
template <typename T>
struct TileRegs { typename Elem<T>::raw4 kf[MA_RPT], vf[MA_RPT]; };
template <typename T>
struct TileStager {
    const T *k_b, *v_b;
    int ld, S, jl, e4;
    __device__ __forceinline__ TileStager(const T *k, const T *v, int ld_, int S_, int tid)
        : k_b(k), v_b(v), ld(ld_), S(S_), jl(tid / MA_EQ), e4((tid % MA_EQ) * 4) {}
    // Unconditional loads with clamped rows: a load under a branch makes the compiler wait
    // with vmcnt(0) for OLDER loads too (it cannot count what the branch issued), which
    // exposed the full memory latency in every iteration.  Keys >= S read row S - 1: finite
    // data whose cells have multiplicity 0.
    __device__ __forceinline__ TileRegs<T> load(int t) const {
        TileRegs<T> r;
#pragma unroll
        for (int u = 0; u < MA_RPT; u++) {
            const int j = min(t * MA_KT + jl + MA_RPP * u, S - 1);
            r.kf[u] = ld_raw4(k_b + (size_t)j * ld + e4);
            r.vf[u] = ld_raw4(v_b + (size_t)j * ld + e4);
        }
        return r;
    }
    __device__ __forceinline__ void store(char *buf, const TileRegs<T> &r) const {
#pragma unroll
        for (int u = 0; u < MA_RPT; u++) {
            put_rows4(buf + MA_KH, jl + MA_RPP * u, e4, r.kf[u]);
            put_rows4(buf + MA_VH, jl + MA_RPP * u, e4, r.vf[u]);
        }
    }
    // max over this thread's K rows of the tile of the sum of squares of its four columns
    __device__ __forceinline__ float k_share(const TileRegs<T> &r) const {
        float best = 0.f;
#pragma unroll
        for (int u = 0; u < MA_RPT; u++) {
            const float4 x = Elem<T>::f4(r.kf[u]);
            best = fmaxf(best, fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, x.w * x.w))));
        }
        return best;
    }
};

// A-operand fragment of k-step ks from a row-major image: lane (r, h) reads row r, elements
// 8h + 16ks .. + 8 (16 bytes).  NP = parts the image holds (the lo read is skipped for 1).
template <int NP>
__device__ __forceinline__ Frag read_rows(const char *hi, const char *lo, int lane, int ks) {
    const int off = (lane & 31) * MA_KLD + 16 * (lane >> 5) + 32 * ks;
    Frag f;
    f.hi = *reinterpret_cast<const uint4 *>(hi + off);
    f.lo = NP == 2 ? *reinterpret_cast<const uint4 *>(lo + off) : make_uint4(0u, 0u, 0u, 0u);
    return f;
}
// B-operand fragment whose k order matches an accumulator tile used as the A operand
// (k-step s, element jj of lane half h <-> tile row 16s + 8(jj>>2) + 4h + (jj&3)):
// lane (c, h) reads row c of the transposed image at 16s + 4h .. +4 and 16s + 8 + 4h .. +4
__device__ __forceinline__ Frag read_cols(const char *hi, const char *lo, int row, int h, int s) {
    const int off = row * MA_VLD + 32 * s + 8 * h;

    const uint2 a = *reinterpret_cast<const uint2 *>(hi + off),
                b = *reinterpret_cast<const uint2 *>(hi + off + 16),
                c = *reinterpret_cast<const uint2 *>(lo + off),
                d = *reinterpret_cast<const uint2 *>(lo + off + 16);
    Frag f;
    f.hi = make_uint4(a.x, a.y, b.x, b.y);
    f.lo = make_uint4(c.x, c.y, d.x, d.y);
    return f;
}

// The same B-operand fragment from a ROWS image through gfx950's transposing LDS read:
// ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a 4-row x 16-column block of
// 16-bit elements, rows in its elements 0..3; lane 4q + p of the group supplies the address of
// row q, columns 4p .. 4p + 3.  Two blocks (tile rows 16s + 4h .. and 16s + 8 + 4h ..) give
// elements 0..3 and 4..7; the group's columns are 16 (lane >> 4 & 1) .. + 15 of the 32-column
// half `col0`.  No transposed copy of the tile and none of its 2-byte scattered stores.
// (All 64 lanes must be active: the gather crosses lanes.)
typedef short v4s16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr_b64(const char *p) {
    const v4s16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) v4s16 *)(p));
    return __builtin_bit_cast(uint2, r);
}
template <int NP>
__device__ __forceinline__ Frag read_cols_tr(const char *hi, const char *lo, int col0, int lane,
                                             int s) {
    const int gl = lane & 15, q = gl >> 2, p = gl & 3, h = lane >> 5;
    const int off = (16 * s + 4 * h + q) * MA_KLD + (col0 + 16 * ((lane >> 4) & 1) + 4 * p) * 2;
    const uint2 a = lds_tr_b64(hi + off), b = lds_tr_b64(hi + off + 8 * MA_KLD);
    Frag f;
    f.hi = make_uint4(a.x, a.y, b.x, b.y);
    f.lo = make_uint4(0u, 0u, 0u, 0u);
    if (NP == 2) {
        const uint2 c = lds_tr_b64(lo + off), d = lds_tr_b64(lo + off + 8 * MA_KLD);
        f.lo = make_uint4(c.x, c.y, d.x, d.y);
    }
    return f;
}

// A-operand fragment of k-step ks from a COLS image [e][row] (an operand that arrives as
// [E][S]): lane (r, h) wants row r, elements 8h + 16ks .. + 8 -- columns of the image, again the
// transposing read: blocks of image rows e0 .. e0+3 and e0+4 .. e0+7, image columns = tile rows
__device__ __forceinline__ Frag read_rows_tr(const char *hi, const char *lo, int lane, int ks) {
    const int gl = lane & 15, q = gl >> 2, p = gl & 3, h = lane >> 5;
    const int off = (8 * h + 16 * ks + q) * MA_VLD + (16 * ((lane >> 4) & 1) + 4 * p) * 2;
    const uint2 a = lds_tr_b64(hi + off), b = lds_tr_b64(hi + off + 4 * MA_VLD),
                c = lds_tr_b64(lo + off), d = lds_tr_b64(lo + off + 4 * MA_VLD);
    Frag f;
    f.hi = make_uint4(a.x, a.y, b.x, b.y);
    f.lo = make_uint4(c.x, c.y, d.x, d.y);
    return f;
}

// B-operand fragments of a wave's own 32 rows of a dense operand x[row * ld + e]: lane (r, h)
// holds row i0 + r, elements 8h + 16ks .. + 8.
template <typename T>
__device__ __forceinline__ void load_own_rows_raw(float (&x)[MA_E / 2], const T *x_b, int ld, int S,
                                                  int i0, int lane) {
    // rows >= S read row S - 1 (unconditional loads; such rows are never stored)
    const int row = min(i0 + (lane & 31), S - 1), h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < MA_KS; ks++) {
        const T *p = x_b + (size_t)row * ld + 8 * h + 16 * ks;
        const float4 a = ld4(p), b = ld4(p + 4);
        x[8 * ks + 0] = a.x; x[8 * ks + 1] = a.y; x[8 * ks + 2] = a.z; x[8 * ks + 3] = a.w;
        x[8 * ks + 4] = b.x; x[8 * ks + 5] = b.y; x[8 * ks + 6] = b.z; x[8 * ks + 7] = b.w;
    }
}
// split into the four k-step fragments, optionally pre-multiplied (the score scale folded
// into the operand: one multiply per element here instead of one per cell per tile)
__device__ __forceinline__ void split_own_rows(Frag (&f)[MA_KS], const float (&x)[MA_E / 2],
                                               float mult = 1.0f) {
#pragma unroll
    for (int ks = 0; ks < MA_KS; ks++)
        f[ks] = split8(x[8 * ks] * mult, x[8 * ks + 1] * mult, x[8 * ks + 2] * mult,
                       x[8 * ks + 3] * mult, x[8 * ks + 4] * mult, x[8 * ks + 5] * mult,
                       x[8 * ks + 6] * mult, x[8 * ks + 7] * mult);
}

// ---- wide global accesses for operands whose register layout is element-wise ----
// A 4-byte load or store per lane moves 256 bytes per wave instruction; the kernels' own-row
// operands and results (8 KiB per wave and tensor) are therefore passed through a wave-private
// LDS tile of 32 x MA_TLD floats, one 32-column half at a time.
//
// One e-half of a wave's 32-row accumulator tile (column = lane & 31, rows in the registers)
// -> rows of `dst` (row stride ld elements), 8 rows x 32 elements per store instruction.
template <typename T>
__device__ __forceinline__ void store_acc_half(const f32x16 &acc, float mult, float *tile,
                                               T *dst, size_t ld, int rows_left, int lane) {
    const int c32 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; r++) tile[acc_row(r, h) * MA_TLD + c32] = acc[r] * mult;
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int row = (lane >> 3) + 8 * k, e4 = (lane & 7) * 4;
        const float4 o = *reinterpret_cast<const float4 *>(tile + row * MA_TLD + e4);
        if (row < rows_left) st4(dst + (size_t)row * ld + e4, o);
    }
    wave_lds_fence();
}
// A wave's own 32 rows of an operand stored [E][S] (row i0 .. of every e) into the fragment
// order of load_own_rows_raw: x[8 ks + j] = src[(8h + 16ks + j) * S + i0 + (lane & 31)].
template <typename T>
__device__ __forceinline__ void load_own_rows_transposed(float (&x)[MA_E / 2], const T *src, int S,
                                                         int i0, float *tile, int lane) {
    const int c32 = lane & 31, h = lane >> 5;
    const int i4 = min(i0 + (lane & 7) * 4, S - 4) - i0;      // (S % 4 == 0; rows >= S unused)
#pragma unroll
    for (int hf = 0; hf < MA_ET; hf++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int el = (lane >> 3) + 8 * k;
            const float4 v4 = ld4(src + (size_t)(32 * hf + el) * S + i0 + i4);
            *reinterpret_cast<float4 *>(tile + el * MA_TLD + (lane & 7) * 4) = v4;
        }
        wave_lds_fence();
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
#pragma unroll
            for (int j = 0; j < 8; j++)
                x[8 * (2 * hf + kk) + j] = tile[(8 * h + 16 * kk + j) * MA_TLD + c32];
        }
        wave_lds_fence();
    }
}

// ---- cell tiles: multiplicity of every (row, key) cell, per 32 x 32 tile ----
//   masks [batch][RT][2] uint64: [0] bit t set <=> key tile t has a live entry in row tile rt;
//     [1] bit t set <=> some cell of that tile has multiplicity >= 2 ("multi" tile)
//   cells [batch][RT (RT + 1) / 2][64 lanes][16] uint8: tile (rt, t <= rt) at rt (rt + 1) / 2 + t;
//     byte 4 g + u of lane (c, h) = multiplicity of key 32 t + 8 g + 4 h + u in row 32 rt + c --
//     accumulator register 4 g + u of that lane when the tile is computed as D[key, row]
//   cells_t, same indexing: byte 4 g + u of lane (c, h) = multiplicity of key 32 t + c in row
//     32 rt + 8 g + 4 h + u -- the tile computed as D[row, key] (the key-owned backward)
//   A tile that is not "multi" (all counts 0 or 1: every tile of a lookup pattern except those
//     holding its zero padding) stores only 128 bytes per orientation instead: word c of the
//     tile's slot = the 32-bit key mask of row c (cells_t: the row mask of key c); a consumer
//     lane reads the 16 bytes holding its word and spreads its 16 bits back into the byte form.
//     The consumers are bound by the bytes they move: this takes 36 MB off each of them.
//   live = col <= row (softmax.cu:19-31 masks the others) and 0 <= col < S.
// A wave owns a row tile and walks its key tiles in chunks of MB_CHUNK: the chunk's counts
// live in LDS as [tile][row][key] bytes (4 keys per word); every entry whose column falls into
// the chunk does one ds_add_u32 (LDS atomics run at about one lane per clock per CU, so they
// are the kernel's cost: the transposed tiles are gathered bytewise from the same counts
// rather than counted a second time).  (A first
// version bucketed the entries by key tile with a counting sort before counting: 33 us at
// the bench shape, latency-bound on its LDS read-modify-write chains and wave scans.)
// A byte saturates at 255: only Z = 256 can reach it, in a row whose 256 entries are ONE column
// (row 0 of a lookup pattern: column 0, 256 times); then the add that would carry into the
// neighbouring key is taken back and the row's bit is set in the row tile's word of `sat`
//   sat [batch][RT] uint32: bit r set <=> row 32 rt + r holds ONE live cell, of multiplicity 256,
//     stored as 255 -- the consumers add the 1 back (round 4: row_sum of such a row is exact).
constexpr int MB_WAVES = 4;
constexpr int MB_CHUNK = 8;                          // key tiles per pass over the entries
constexpr int MB_CNT_WORDS = MB_CHUNK * MA_WROWS * MA_CLD;     // one orientation: 2304 words
__host__ __device__ __forceinline__ size_t tri(size_t n) { return n * (n + 1) / 2; }
// bytes of a tile's slot: 1 KiB in the full layout, 128 in the compact one (see CellTiles)
__device__ __forceinline__ int cell_slot_bytes(const void *pool) { return pool ? 128 : MA_CELLS; }
__host__ __device__ __forceinline__ size_t prepare_lds_per_wave() {
    return (size_t)MB_CNT_WORDS * 4;                 // 9216 B
}
template <bool SATURATE>
__global__ __launch_bounds__(MB_WAVES * SPT_WAVE) void attention_cell_tiles_kernel(
    const int32_t *__restrict__ indices, unsigned long long *__restrict__ masks,
    unsigned char *__restrict__ cells, unsigned char *__restrict__ cells_t,
    unsigned char *__restrict__ pool, unsigned char *__restrict__ pool_t,
    unsigned *__restrict__ sat, int S, int Z, int NT, int RT, int n_tiles_total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * MB_WAVES + wave;
    if (gw >= n_tiles_total) return;                    // no workgroup barrier below
    const int b = gw / RT, rt = gw - b * RT, i0 = rt * MA_WROWS;
    unsigned *cnt = reinterpret_cast<unsigned *>(smem + wave * prepare_lds_per_wave());
    unsigned satrows = 0;                               // (per lane; OR-reduced at the end)

    // the tile's rows are one contiguous run of 32 * Z ints (Z % 4 == 0)
    const int nrows = min(MA_WROWS, S - i0);
    const int n4 = nrows * Z / 4;
    const int4 *src = reinterpret_cast<const int4 *>(indices + ((size_t)b * S + i0) * Z);
    const int slot_bytes = cell_slot_bytes(pool);
    unsigned char *out = cells + ((size_t)b * tri(RT) + tri(rt)) * slot_bytes;
    unsigned char *out_t = cells_t + ((size_t)b * tri(RT) + tri(rt)) * slot_bytes;
    const int c32 = lane & 31, h = lane >> 5;
    const int nbuckets = min(NT, rt + 1);               // live entries have col <= row
    unsigned long long mask = 0, multi = 0;

    auto count = [&](unsigned *word, unsigned sh, int rl) {
        if (!SATURATE) {
            atomicAdd(word, 1u << sh);
        } else {
            const unsigned old = atomicAdd(word, 1u << sh);
            if (((old >> sh) & 0xffu) == 0xffu) {       // the row's 256th copy of this column
                atomicSub(word, 1u << sh);
                satrows |= 1u << rl;
            }
        }
    };
    for (int t0 = 0; t0 < nbuckets; t0 += MB_CHUNK) {
        uint4 *c4 = reinterpret_cast<uint4 *>(cnt);     // 9 x 64 x 16 bytes
#pragma unroll
        for (int x = 0; x < MB_CNT_WORDS / 4 / SPT_WAVE; x++)
            c4[x * SPT_WAVE + lane] = make_uint4(0u, 0u, 0u, 0u);
        wave_lds_fence();
        for (int x = lane; x < n4; x += SPT_WAVE) {
            const int4 e4 = src[x];
            const int rl = 4 * x / Z;                   // an int4 never straddles rows
            const int c[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int tl = (c[q] >> 5) - t0;
                if (c[q] >= 0 && c[q] <= i0 + rl && c[q] < S && tl >= 0 && tl < MB_CHUNK) {
                    const unsigned jl = c[q] & 31;
                    count(&cnt[(tl * MA_WROWS + rl) * MA_CLD + (jl >> 2)], 8 * (jl & 3), rl);
                }
            }
        }
        wave_lds_fence();
        const int tend = min(MB_CHUNK, nbuckets - t0);
        for (int tl = 0; tl < tend; tl++) {
            // word 2 g + h of row c holds keys 8 g + 4 h .. + 3: exactly this lane's bytes 4 g ..
            const unsigned *row = cnt + (tl * MA_WROWS + c32) * MA_CLD + h;
            uint4 mine = make_uint4(row[0], row[2], row[4], row[6]);
            if (__ballot((mine.x | mine.y | mine.z | mine.w) != 0u) == 0ull) continue;
            mask |= 1ull << (t0 + tl);
            // transposed: byte 4 g + u of lane (c, h) = count of key c in row 8 g + 4 h + u
            const unsigned char *bytes = reinterpret_cast<const unsigned char *>(cnt) +
                                         (size_t)tl * MA_WROWS * MA_CLD * 4 + c32;
            unsigned w[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const unsigned char *p0 = bytes + (8 * g + 4 * h) * MA_CLD * 4;
                w[g] = (unsigned)p0[0] | ((unsigned)p0[MA_CLD * 4] << 8) |
                       ((unsigned)p0[2 * MA_CLD * 4] << 16) | ((unsigned)p0[3 * MA_CLD * 4] << 24);
            }
            uint4 *slot = reinterpret_cast<uint4 *>(out + (size_t)(t0 + tl) * slot_bytes);
            uint4 *slot_t = reinterpret_cast<uint4 *>(out_t + (size_t)(t0 + tl) * slot_bytes);
            const bool big = ((mine.x | mine.y | mine.z | mine.w) & 0xFEFEFEFEu) != 0u;
            // (compact layout: byte forms exist for key tile 0 only -- the caller vouches that
            // no other tile repeats a column, as lookup patterns do not; a tile that breaks the
            // promise is stored as if its counts were 0 / 1 and flagged in the header)
            const bool pooled_ok = pool == nullptr || t0 + tl == 0;
            if (__ballot(big) != 0ull && pooled_ok) {
                multi |= 1ull << (t0 + tl);
                if (pool) {
                    slot = reinterpret_cast<uint4 *>(pool + ((size_t)b * RT + rt) * MA_CELLS);
                    slot_t = reinterpret_cast<uint4 *>(pool_t + ((size_t)b * RT + rt) * MA_CELLS);
                }
                slot[lane] = mine;
                slot_t[lane] = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
                if (__ballot(big) != 0ull) {
                    if (lane == 0) atomicOr(reinterpret_cast<unsigned *>(masks) - 64 + 1, 1u);
                    auto clamp1 = [](unsigned x) { return (x | (x >> 1) | (x >> 2) | (x >> 3) | (x >> 4) |
                                                           (x >> 5) | (x >> 6) | (x >> 7)) & 0x01010101u; };
                    mine = make_uint4(clamp1(mine.x), clamp1(mine.y), clamp1(mine.z), clamp1(mine.w));
#pragma unroll
                    for (int g = 0; g < 4; g++) w[g] = clamp1(w[g]);
                }
                // counts are 0 / 1: 16 of the row's (key's) 32 bits sit in this lane, the other
                // 16 in lane ^ 32; bytes -> nibble by one multiply (no carries: each product
                // term lands on its own bit)
                auto bits16 = [&](unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
                    auto nib = [](unsigned x) { return ((x * 0x01020408u) >> 24) & 0xFu; };
                    return (nib(w0) | (nib(w1) << 8) | (nib(w2) << 16) | (nib(w3) << 24)) << (4 * h);
                };
                unsigned rm = bits16(mine.x, mine.y, mine.z, mine.w);
                unsigned cm = bits16(w[0], w[1], w[2], w[3]);
                rm |= (unsigned)__shfl_xor((int)rm, 32, SPT_WAVE);
                cm |= (unsigned)__shfl_xor((int)cm, 32, SPT_WAVE);
                if (h == 0) {
                    reinterpret_cast<unsigned *>(slot)[c32] = rm;
                    reinterpret_cast<unsigned *>(slot_t)[c32] = cm;
                }
            }
        }
        wave_lds_fence();
    }
#pragma unroll
    for (int off = 1; off < SPT_WAVE; off <<= 1) satrows |= (unsigned)__shfl_xor((int)satrows, off, SPT_WAVE);
    if (lane == 0) {
        masks[2 * (size_t)gw] = mask;
        masks[2 * (size_t)gw + 1] = multi;
        sat[gw] = satrows;
    }
}

// The compact layout's builder (round 3).  Its promise -- only columns < 32 repeat inside a row --
// means every key tile but the first holds 0 / 1 cells: ONE pass over the row tile's entries sets
// bits ([key tile][row] words, ds_or) for the columns >= 32 and counts the columns < 32 in 32-bit
// counters (the padding column 0 pre-summed per lane: a row of S = 2048 holds up to 256 of them);
// the byte-count kernel above walks the entries once per eight key tiles with a ds_add per entry
// (and, at Z = 256, a returning add and a conditional subtract): 241 us per launch at S = 2048,
// Z = 256, B = 64, 6 % of an OPT-1.3B block's step.  Same masks, slots and pool as the kernel above
// (tests/test_gpu_mfma_attention.py compares whole workspaces); a repeated column >= 32 shows as
// popcount(row) != the row's count of such entries and sets the header's promise flag.
__host__ __device__ __forceinline__ size_t cell_bits_lds_per_wave(int NT) {
    return ((size_t)MA_WROWS * 32 + (size_t)(NT > 1 ? NT - 1 : 1) * MA_WROWS + MA_WROWS) * 4;
}
__global__ __launch_bounds__(MB_WAVES * SPT_WAVE) void attention_cell_bits_kernel(
    const int32_t *__restrict__ indices, unsigned long long *__restrict__ masks,
    unsigned char *__restrict__ cells, unsigned char *__restrict__ cells_t,
    unsigned char *__restrict__ pool, unsigned char *__restrict__ pool_t,
    unsigned *__restrict__ sat, int S, int Z, int NT, int RT, int n_tiles_total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * MB_WAVES + wave;
    if (gw >= n_tiles_total) return;                    // no workgroup barrier below
    const int b = gw / RT, rt = gw - b * RT, i0 = rt * MA_WROWS;
    unsigned *cnt0 = reinterpret_cast<unsigned *>(smem + wave * cell_bits_lds_per_wave(NT));   // [row][key < 32]
    unsigned *bits = cnt0 + MA_WROWS * 32;              // [key tile - 1][row]
    const int nbuckets = min(NT, rt + 1);               // live entries have col <= row
    unsigned *nval = bits + (size_t)(NT > 1 ? NT - 1 : 1) * MA_WROWS;   // [row]: entries with col >= 32
    const int nrows = min(MA_WROWS, S - i0);
    const int n4 = nrows * Z / 4;
    const int4 *src = reinterpret_cast<const int4 *>(indices + ((size_t)b * S + i0) * Z);
    const int c32 = lane & 31, h = lane >> 5;

    for (int x = lane; x < MA_WROWS * 32 + (nbuckets - 1) * MA_WROWS; x += SPT_WAVE) cnt0[x] = 0u;
    if (lane < MA_WROWS) nval[lane] = 0u;
    wave_lds_fence();
    for (int x = lane; x < n4; x += SPT_WAVE) {
        const int4 e4 = src[x];
        const int rl = 4 * x / Z;                       // an int4 never straddles rows
        const int c[4] = {e4.x, e4.y, e4.z, e4.w};
        unsigned nz = 0, nv = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (c[q] == 0) {
                nz++;
            } else if (c[q] > 0 && c[q] <= i0 + rl && c[q] < S) {
                if (c[q] < 32) {
                    atomicAdd(&cnt0[rl * 32 + c[q]], 1u);
                } else {
                    atomicOr(&bits[((c[q] >> 5) - 1) * MA_WROWS + rl], 1u << (c[q] & 31));
                    nv++;
                }
            }
        }
        if (nz) atomicAdd(&cnt0[rl * 32], nz);
        if (nv) atomicAdd(&nval[rl], nv);
    }
    wave_lds_fence();
    // a repeated column >= 32 breaks the layout's promise (its second entry set no new bit)
    {
        unsigned pc = 0;
        if (lane < MA_WROWS)
            for (int t = 1; t < nbuckets; t++) pc += __popc(bits[(t - 1) * MA_WROWS + lane]);
        const bool broke = lane < MA_WROWS && pc != nval[lane];
        if (__ballot(broke) != 0ull && lane == 0) atomicOr(reinterpret_cast<unsigned *>(masks) - 64 + 1, 1u);
    }
    unsigned long long mask = 0, multi = 0;
    unsigned char *out = cells + ((size_t)b * tri(RT) + tri(rt)) * 128;
    unsigned char *out_t = cells_t + ((size_t)b * tri(RT) + tri(rt)) * 128;
    // ---- key tile 0: counts ----
    {
        // lane (c, h): its 16 counts of row c (keys 8 g + 4 h + u), saturated to a byte
        unsigned mine[4], tr[4];
        bool big = false, full = false;                 // full: row c holds a column 256 times
#pragma unroll
        for (int g = 0; g < 4; g++) {
            unsigned wm = 0, wt = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const unsigned raw = cnt0[c32 * 32 + 8 * g + 4 * h + u];                   // row c, key ..
                const unsigned a = min(raw, 255u);
                const unsigned t2 = min(cnt0[(8 * g + 4 * h + u) * 32 + c32], 255u);        // key c, row ..
                wm |= a << (8 * u);
                wt |= t2 << (8 * u);
                big |= a > 1u;
                full |= raw > 255u;
            }
            mine[g] = wm;
            tr[g] = wt;
        }
        const unsigned long long fb = __ballot(full);   // (the two lane halves hold disjoint keys)
        if (lane == 0) sat[gw] = (unsigned)fb | (unsigned)(fb >> 32);
        const bool any = (mine[0] | mine[1] | mine[2] | mine[3]) != 0u;
        if (__ballot(any) != 0ull) {
            mask |= 1ull;
            if (__ballot(big) != 0ull) {
                multi |= 1ull;
                reinterpret_cast<uint4 *>(pool + ((size_t)b * RT + rt) * MA_CELLS)[lane] =
                    make_uint4(mine[0], mine[1], mine[2], mine[3]);
                reinterpret_cast<uint4 *>(pool_t + ((size_t)b * RT + rt) * MA_CELLS)[lane] =
                    make_uint4(tr[0], tr[1], tr[2], tr[3]);
            } else {
                // counts are 0 / 1: bytes -> bits (the kernel above, `bits16`)
                auto nib = [](unsigned x) { return ((x * 0x01020408u) >> 24) & 0xFu; };
                auto bits16 = [&](const unsigned (&w)[4]) {
                    return (nib(w[0]) | (nib(w[1]) << 8) | (nib(w[2]) << 16) | (nib(w[3]) << 24)) << (4 * h);
                };
                unsigned rm = bits16(mine), cm = bits16(tr);
                rm |= (unsigned)__shfl_xor((int)rm, 32, SPT_WAVE);
                cm |= (unsigned)__shfl_xor((int)cm, 32, SPT_WAVE);
                if (h == 0) {
                    reinterpret_cast<unsigned *>(out)[c32] = rm;
                    reinterpret_cast<unsigned *>(out_t)[c32] = cm;
                }
            }
        }
    }
    // ---- key tiles 1 ..: bits ----
    for (int t = 1; t < nbuckets; t++) {
        const unsigned *tb = bits + (t - 1) * MA_WROWS;
        const unsigned rm = tb[c32];                    // row c: its 32 keys of this tile
        if (__ballot(rm != 0u) == 0ull) continue;
        mask |= 1ull << t;
        // key c: bit i = row i has it; each half of the wave gathers 16 rows (broadcast reads)
        unsigned cm = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) cm |= ((tb[16 * h + i] >> c32) & 1u) << (16 * h + i);
        cm |= (unsigned)__shfl_xor((int)cm, 32, SPT_WAVE);
        if (h == 0) {
            reinterpret_cast<unsigned *>(out + (size_t)t * 128)[c32] = rm;
            reinterpret_cast<unsigned *>(out_t + (size_t)t * 128)[c32] = cm;
        }
    }
    if (lane == 0) {
        masks[2 * (size_t)gw] = mask;
        masks[2 * (size_t)gw + 1] = multi;
    }
}

// a tile's 16 bytes as loaded -> the four count words of this lane (byte form)
__device__ __forceinline__ uint4 cell_words(const uint4 &v, bool multi, int lane) {
    if (multi) return v;
    const int c = lane & 3;
    const unsigned m = (c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w) >> (4 * (lane >> 5));
    auto spread = [](unsigned n) { return ((n & 0xFu) * 0x00204081u) & 0x01010101u; };
    return make_uint4(spread(m), spread(m >> 8), spread(m >> 16), spread(m >> 24));
}
// Two layouts of the stored tiles (chosen by the caller of spt_attention_mfma_prepare):
//   full     every tile owns a 1 KiB slot, which holds its byte form or (first 128 bytes) its
//            mask form;
//   compact  every tile owns a 128-byte slot for its mask form; a byte form lives in a pool of
//            one 1 KiB slot per ROW TILE and is allowed for key tile 0 only -- what a lookup
//            pattern needs (its only repeated column is the padding column 0), at 1/4 of the
//            memory (S = 512: 66 instead of 272 KiB per slice and step).
// `pool == nullptr` selects the full layout.
// a consumer wave's view of its row tile's cell tiles
struct CellTiles {
    unsigned long long mask, multi;
    unsigned satrows;         // rows of this row tile whose one live cell holds 256 (stored 255)
    int last;                 // the row tile's own index = its last stored key tile
    int stride;               // uint4 per slot
    const uint4 *base;        // tile t of this row tile: base[t * stride + ...]
    const uint4 *pooled;      // compact layout: the byte form of key tile 0, else nullptr
    __device__ __forceinline__ CellTiles(const unsigned long long *masks,
                                         const unsigned char *cells, const unsigned char *pool,
                                         const unsigned *sat, int b, int RT, int rt, bool have) {
        mask = have ? masks[2 * ((size_t)b * RT + rt)] : 0ull;
        multi = have ? masks[2 * ((size_t)b * RT + rt) + 1] : 0ull;
        satrows = have ? sat[(size_t)b * RT + rt] : 0u;
        last = have ? rt : 0;
        stride = cell_slot_bytes(pool) / 16;
        base = reinterpret_cast<const uint4 *>(
            cells + ((size_t)b * tri(RT) + tri(have ? rt : 0)) * cell_slot_bytes(pool));
        pooled = pool ? reinterpret_cast<const uint4 *>(
                            pool + ((size_t)b * RT + (have ? rt : 0)) * MA_CELLS)
                      : nullptr;
    }
    __device__ __forceinline__ bool live(int t) const { return (mask >> t) & 1ull; }
    __device__ __forceinline__ bool is_multi(int t) const { return (multi >> min(t, 63)) & 1ull; }
    // always in bounds (t clamped), so the caller can load unconditionally; ONE 16-byte load
    // either way: the lane's own 16 counts, or the four row masks that include its row's
    __device__ __forceinline__ uint4 load(int t, int lane) const {
        const int tc = min(t, last);
        if (is_multi(tc)) return pooled ? pooled[lane] : base[tc * stride + lane];
        return base[tc * stride + ((lane & 31) >> 2)];
    }
    // (the 16 bytes loaded for tile t) -> count words
    __device__ __forceinline__ uint4 words(const uint4 &v, int t, int lane) const {
        return cell_words(v, is_multi(min(t, last)), lane);
    }
};
// byte u of a cell word as float (v_cvt_f32_ubyte<u>)
template <int U>
__device__ __forceinline__ float cell_count(unsigned word) {
    return (float)((word >> (8 * U)) & 0xffu);
}
// A row that is ONE column 256 times (row 0 of a lookup pattern at Z = 256) keeps a byte count of 255
// and its bit in `sat`.  Nothing in the tile loops knows: with m = 255 the forward's y and the
// backward's P = m e / (m e) are what m = 256 gives (the count cancels), and only the row sum the
// forward REPORTS is 255 / 256 of the true one.  So the forward stores rs * 256 / 255 for such rows,
// and the backward kernels take rs * 255 / 256 back for them where they load the row sums -- once per
// row, outside every loop.  (Round 4's first form fixed the count inside the loops, behind a
// wave-uniform branch: 7-13 % on every Z = 256 launch for one cell per slice.)
constexpr float MA_SAT_UP = 256.0f / 255.0f, MA_SAT_DOWN = 255.0f / 256.0f;

// Row tile rt costs rt + 1 key tiles, so contiguous 256-row blocks would give the workgroups
// of a slice 36 : 100 of the work at S = 512 and, two to a CU, idle CUs at the end.  Folded
// instead: workgroup g of a slice takes row tiles 4g .. 4g+3 (waves 0-3) and the mirror
// images RTpad-4g-4 .. RTpad-4g-1 (waves 4-7; one short and one long wave per SIMD).
__device__ __forceinline__ int folded_row_tile(int g, int blocks_per_batch, int wave) {
    constexpr int HALF = MA_WAVES / 2;
    return wave < HALF ? HALF * g + wave
                       : MA_WAVES * blocks_per_batch - HALF * (g + 1) + (wave - HALF);
}

// ===================================== forward =============================================
// grid: batch * ceil(S / 256) workgroups.

// exp(clamp(scale d)) = exp2(med3(d * scale log2e, -+clamp log2e)); without a clamp the
// exponent is bounded at 2^127 so that dead cells (m = 0) stay 0 and not 0 * inf
struct ScoreMap {
    float sl2, bound;
    __device__ __forceinline__ ScoreMap(float scale, float clampv) {
        const float log2e = 1.4426950408889634f;
        sl2 = scale * log2e;
        bound = clampv > 0.0f ? clampv * log2e : 127.0f;
    }
    // d: the tile value with sl2 already folded into one operand (q or k pre-multiplied)
    __device__ __forceinline__ float exp_of(float d) const {
        return __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(d, -bound, bound));
    }
    // the clamp passes gradient strictly inside (-clamp, clamp) (attention.py:125-127 through
    // spt_softmax_backward_clamped: |clamped| < clamp)
    __device__ __forceinline__ bool inside(float d) const { return fabsf(d) < bound; }
    // the largest value `inside` accepts
    __device__ __forceinline__ float bound_in() const {
        return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, bound) - 1u);
    }
};

// ---- the clamp's gradient mask, decided on the EXACT score (round 4) ----
// `v = clamp(scale * sddmm, -10, 10)` (attention.py:125-127) passes gradient only inside the clamp.
// That mask is a discontinuous function of the score, and a score formed from split-bf16 operands
// carries an error of up to
//     (3 * 2^-18 [the lo * lo term and the rounding of the lo parts] + 3 E * 2^-24 [fp32 accumulation
//     of the 3 E products, worst case]) * sum_e |q_e k_e|  <=  MA_SPLIT_ERR * |q| |k|
// so a cell that close to +-clamp can land on the other side of it than the reference's fp32 score
// (measured before this: 7e-3 of grad_q on a layer with 15 % of its scores on the clamp).  The
// backward kernels therefore recompute such cells -- |d| within eps of the bound, eps from the norm of
// the lane's own row and the slice-wide bound on the streamed operand's row norms that the forward
// left in `bounds` -- as the oracle does: an fp64 dot in ascending e, rounded to fp32, times scale in
// fp32, compared with the clamp.  A tile value on the wrong side is then moved onto the bound or just
// inside it, which `exp_of` and `inside` turn into the exact mask (the value changes by less than its
// own error; moving every recomputed cell to the bound changed probabilities by up to eps = 0.5 %).
// Cost on the common path: a scalar test per tile while no score of the slice can reach the clamp
// (Cauchy-Schwarz on the norms), else the smallest distance of the lane's 16 |d| from the bound and
// a ballot per tile; the recomputation
// sits behind a wave-uniform branch (about 1e-4 of the live cells of a layer with scores of
// standard deviation 10, none at all while the scores are O(1)).
constexpr float MA_SPLIT_ERR = (1.0f + 3.0f * MA_E / 256.0f) * 1.52587890625e-5f;
// bounds [batch][MA_BOUND_SLOTS][2] floats, written by the forward's workgroup g of a slice:
// [0] max over its rows of |q_i|^2, [1] an upper bound of max |k_j|^2 over the keys it staged
constexpr int MA_BOUND_SLOTS = 8;                   // workgroups per slice: S <= 2048
struct ClampGuard {
    float thr;                 // bound - eps: a cell with | |d| - bound | <= eps calls for the recomputation
    float bound, eps;
    bool on;                   // false: no score of this wave can reach thr at all (Cauchy-Schwarz on the
                               // norms) -- the usual case, scores O(1): the tile loop then only tests a scalar
    // own2: |own row|^2 of this lane (unscaled operand); which: 1 = the streamed operand is K, 0 = Q
    __device__ __forceinline__ ClampGuard(const float *bounds, int b, int nslots, int which, float own2,
                                          const ScoreMap &sm, float clampv) {
        thr = __builtin_inff();
        bound = sm.bound;
        eps = 0.0f;
        on = false;
        if (bounds != nullptr && clampv > 0.0f) {
            float other2 = 0.0f;
            for (int g = 0; g < nslots; g++)
                other2 = fmaxf(other2, bounds[((size_t)b * MA_BOUND_SLOTS + g) * 2 + which]);
            // the WAVE's largest own row: one threshold per wave, held in a scalar register
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) own2 = fmaxf(own2, __shfl_xor(own2, off, SPT_WAVE));
            const float reach = sqrtf(own2 * other2) * sm.sl2;      // |d| <= |own| |other| scale log2e
            thr = sm.bound - MA_SPLIT_ERR * reach;
            thr = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, thr)));
            eps = bound - thr;                                      // (as clamp_exact forms it)
            on = __builtin_amdgcn_readfirstlane((int)(reach * 1.0001f >= thr)) != 0;
        }
    }
    __device__ __forceinline__ bool wanted(const f32x16 &d) const {
#ifdef MA_NO_CLAMP_GUARD                                // (A/B builds: what the check costs)
        return false;
#endif
        if (!on) return false;
        // TWO-sided: the smallest distance of the lane's 16 |d| from the bound.  (The first form asked
        // for max |d| >= thr, which every tile answers with yes once scores lie BEYOND the clamp -- ten
        // steps into the bench's training run: max |score| 15-23 in the upper layers, and the key-owned
        // kernel went from 89 to 140-165 us there, all of it the second-level test behind the branch.)
        // (a cheaper necessary condition in front -- max |d| >= thr, 8 instructions and a ballot -- was
        // measured in the step: 105.7 / 80.8 us against 102.0 / 78.3 without it: not kept)
        float m = fminf(fabsf(fabsf(d[0]) - bound), fabsf(fabsf(d[1]) - bound));
#pragma unroll
        for (int r = 2; r < 16; r += 2)
            m = fminf(m, fminf(fabsf(fabsf(d[r]) - bound), fabsf(fabsf(d[r + 1]) - bound)));
        return __ballot(m <= eps) != 0ull;
    }
};
// extension/sddmm.cpp:27-69 as the oracle restates it: sum in fp64, e ascending, rounded once
template <typename T>
__device__ __forceinline__ float exact_dot(const T *a, const T *b) {
    double acc = 0.0;
#pragma unroll 1
    for (int e = 0; e < MA_E; e += 4) {
        const float4 x = ld4(a + e), y = ld4(b + e);
        acc = fma((double)x.x, (double)y.x, acc);
        acc = fma((double)x.y, (double)y.y, acc);
        acc = fma((double)x.z, (double)y.z, acc);
        acc = fma((double)x.w, (double)y.w, acc);
    }
    return (float)acc;
}
// d: the lane's 16 cells (own row x tile rows acc_row(r, h)); cw: their multiplicities as the
// four count words (byte form); dp: the tile's second product, which has nothing to do with the
// mask but is live here.  At the call sites the kernels hold 190-250 registers; this loop inlined
// beside them raised the allocation past 256 (spills inside the tile loop), as a non-inlined call
// the allocator kept the K / V fragments in scratch for the whole kernel, and in FRONT of the second
// product (where 16-32 registers fewer are live) the branch serialised the two MFMA chains: 205
// against 193 us for the backward pair.  So d and dp are parked in a wave-private LDS block for the
// duration of the loop -- a hand-placed spill, paid only inside the rare branch.
constexpr int MA_PARK = 32 * SPT_WAVE * 4;          // bytes per wave: d and dp, [register][lane]
// own: the own operand's slice base (rows of `ld` elements), own_row0: the wave's first own row
// (lane & 31 is added here); tile_rows: the streamed operand's tile; park0: the parking blocks'
// base (the wave's block is chosen here).  Everything lane-dependent is derived from rare_lane().
template <typename T>
__device__ __forceinline__ void clamp_exact(f32x16 &d, f32x16 &dp, char *park0, int wave, const uint4 &cw,
                                            float bound, float thr, const T *own, int own_row0, int S,
                                            const T *tile_rows, int ld, float scale, float clampv) {
    const unsigned w[4] = {cw.x, cw.y, cw.z, cw.w};
    const float eps = bound - thr;
    unsigned near = 0, in = 0;
#pragma unroll
    for (int r = 0; r < 16; r++)
        if (((w[r >> 2] >> (8 * (r & 3))) & 0xffu) != 0u && fabsf(fabsf(d[r]) - bound) <= eps)
            near |= 1u << r;
    if (__ballot(near != 0u) == 0ull) return;       // close to the clamp, but no live cell is
    const int lane = rare_lane();
    float *park = reinterpret_cast<float *>(park0 + wave * MA_PARK) + lane;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        park[r * SPT_WAVE] = d[r];
        park[(16 + r) * SPT_WAVE] = dp[r];
    }
    asm volatile("" ::: "memory");                  // (no store-to-load forwarding past the loop)
    const T *own_row = own + (size_t)min(own_row0 + (lane & 31), S - 1) * ld;
    for (unsigned rem = near; rem != 0u; rem &= rem - 1u) {
        const int r = __builtin_ctz(rem);
        const float sc = exact_dot(own_row, tile_rows + (size_t)acc_row(r, lane >> 5) * ld) * scale;
        if (fabsf(sc) < clampv) in |= 1u << r;
    }
    asm volatile("" ::: "memory");
    const float bound_in = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, bound) - 1u);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        // the tile value moves only where it sits on the wrong side, and then by less than its own
        // error (onto the bound, or to the largest value inside it)
        float v = park[r * SPT_WAVE];
        const bool is_in = (in >> r) & 1u;
        if (((near >> r) & 1u) && is_in != (fabsf(v) < bound)) v = copysignf(is_in ? bound_in : bound, v);
        d[r] = v;
        dp[r] = park[(16 + r) * SPT_WAVE];
    }
}

// two workgroups per CU (4 waves per SIMD, 128 VGPRs): the barrier keeps the waves of one
// workgroup in the same phase, a second workgroup fills the other pipes meanwhile
#if MA_E_VALUE == 64
#define MA_WAVES_PER_EU 4
#else
#define MA_WAVES_PER_EU 2       // (64 + 64 registers of Q fragments and accumulators alone)
#endif
// One key tile per iteration in the forward and in the row-owned backward (two per iteration,
// i.e. fewer barriers and two independent chains, measured 58-71 against 48 us in the forward and
// no faster in the backward; sharing a row tile between the two waves of a SIMD -- alternate key
// tiles, partial grad_q summed through LDS -- measured 100 k against 89 k clocks: the two waves
// of a SIMD run the same program between the same barriers, so their MFMA phases coincide and
// their VALU phases coincide).
constexpr int MK_KTILES = MA_ET;    // 32-column tiles of grad_k per launch of the key-owned kernel
constexpr int MR_QTILES = MA_ET;    // 32-column tiles of grad_q per launch of the row-owned kernel
#define MA_ROWS_WAVES_PER_EU 2
#define MA_KEYS_WAVES_PER_EU 2
template <typename T, bool YT>
__global__ __launch_bounds__(MA_THREADS)
__attribute__((amdgpu_waves_per_eu(MA_WAVES_PER_EU, MA_WAVES_PER_EU)))
void attention_mfma_forward_kernel(
    const unsigned long long *__restrict__ masks, const unsigned char *__restrict__ cells,
    const unsigned char *__restrict__ pool, const unsigned *__restrict__ sat,
    const T *__restrict__ q, const T *__restrict__ k, const T *__restrict__ v,
    T *__restrict__ y, float *__restrict__ row_sum, float *__restrict__ bounds, int S, float scale,
    float clampv, int heads, int blocks_per_batch) {
    constexpr int PI = Elem<T>::PARTS;                  // parts of the K / V images
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *img = smem;                                   // [2][MA_IMG]
    float *stat = reinterpret_cast<float *>(smem + 2 * MA_IMG);   // [waves][32] | [waves][2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c32 = lane & 31;
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / blocks_per_batch;
    const int i0 = MA_WROWS * folded_row_tile(bid % blocks_per_batch, blocks_per_batch, wave);
    const DenseView dv = dense_view(b, S, MA_E, heads);
    const int RT = (S + MA_WROWS - 1) / MA_WROWS;
    const T *k_b = k + dv.base, *v_b = v + dv.base;

    const CellTiles ct(masks, cells, pool, sat, b, RT, i0 / MA_WROWS, i0 < S);
    // tiles any row of this workgroup can see: keys <= its last row (that of wave 7)
    const int last_tile = folded_row_tile(bid % blocks_per_batch, blocks_per_batch, MA_WAVES - 1);
    const int T_ = min(RT, last_tile + 1);
    const ScoreMap sm(scale, clampv);
    // every prologue load is issued before the first result is needed
    const TileStager<T> stager(k_b, v_b, dv.ld, S, tid);
    TileRegs<T> nxt = stager.load(0);
    uint4 mcur = ct.load(0, lane), mnxt;
    Frag qf[MA_KS];
    // for the backward's exact clamp mask (ClampGuard): |q_i|^2 of the own row, and the running
    // maximum of THIS thread's share of |k_j|^2 -- the four columns it stages, over all the rows it
    // stages; summed over the column groups at the end that bounds every staged key's |k_j|^2
    // (max_j sum_g <= sum_g max_j) without a cross-lane sum per tile.  About 3x the true maximum on
    // Gaussian rows (1.75x in the norm): the guard switches on that much earlier than it has to.  The
    // exact maximum -- |k_i|^2 of the own rows, one more read of K in the prologue -- was measured:
    // +3 us on every forward (44 -> 48 in the step), more than the guard's per-tile test costs while
    // it is on, so the loose bound stays.
    float q2 = 0.f, k2part = 0.f;
    {
        float xq[MA_E / 2];
        load_own_rows_raw(xq, q + dv.base, dv.ld, S, i0, lane);
        stager.store(img, nxt);
        if (bounds) {
            k2part = stager.k_share(nxt);
#pragma unroll
            for (int x = 0; x < MA_E / 2; x++) q2 = fmaf(xq[x], xq[x], q2);
        }
        split_own_rows(qf, xq, sm.sl2);
    }
    __syncthreads();

    f32x16 yacc[MA_ET];
#pragma unroll
    for (int eh = 0; eh < MA_ET; eh++)
#pragma unroll
        for (int r = 0; r < 16; r++) yacc[eh][r] = 0.f;
    float rs = 0.f;

    for (int t = 0; t < T_; t++) {
        const char *buf = img + (t & 1) * MA_IMG;
        // the next tile (K, V and this wave's cell counts) is in flight meanwhile
        nxt = stager.load(min(t + 1, T_ - 1));
        mnxt = ct.load(t + 1, lane);
        if (ct.live(t)) {
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; r++) d[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < MA_KS; ks++)
                d = mm<PI, 2>(read_rows<PI>(buf + MA_KH, buf + MA_KL, lane, ks), qf[ks], d);
            // cells: this lane holds row i0 + c32, keys 32t + acc_row(r, h); byte 4g + u of
            // the cell word is the multiplicity of register 4g + u
            const uint4 mm4 = ct.words(mcur, t, lane);
            const unsigned mw[4] = {mm4.x, mm4.y, mm4.z, mm4.w};
            float p[16];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                p[4 * g + 0] = cell_count<0>(mw[g]) * sm.exp_of(d[4 * g + 0]);
                p[4 * g + 1] = cell_count<1>(mw[g]) * sm.exp_of(d[4 * g + 1]);
                p[4 * g + 2] = cell_count<2>(mw[g]) * sm.exp_of(d[4 * g + 2]);
                p[4 * g + 3] = cell_count<3>(mw[g]) * sm.exp_of(d[4 * g + 3]);
                rs += (p[4 * g] + p[4 * g + 1]) + (p[4 * g + 2] + p[4 * g + 3]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const Frag pf = split8(p[8 * s2], p[8 * s2 + 1], p[8 * s2 + 2], p[8 * s2 + 3],
                                       p[8 * s2 + 4], p[8 * s2 + 5], p[8 * s2 + 6], p[8 * s2 + 7]);
#pragma unroll
                for (int eh = 0; eh < MA_ET; eh++)
                    yacc[eh] = mm<2, PI>(pf, read_cols_tr<PI>(buf + MA_VH, buf + MA_VL, 32 * eh, lane, s2),
                                         yacc[eh]);
            }
        }
        if (t + 1 < T_) stager.store(img + ((t + 1) & 1) * MA_IMG, nxt);
        if (bounds) k2part = fmaxf(k2part, stager.k_share(nxt));
        mcur = mnxt;
        __syncthreads();
    }

    if (bounds) {                                       // (kernel argument: uniform)
        // columns: lanes with equal tid % MA_EQ; rows: the other lane bits
        float km = k2part;
#pragma unroll
        for (int off = MA_EQ; off < SPT_WAVE; off <<= 1) km = fmaxf(km, __shfl_xor(km, off, SPT_WAVE));
#pragma unroll
        for (int off = 1; off < MA_EQ; off <<= 1) km += __shfl_xor(km, off, SPT_WAVE);
        float qm = q2 + __shfl_xor(q2, 32, SPT_WAVE);   // the two lane halves hold half a row each
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) qm = fmaxf(qm, __shfl_xor(qm, off, SPT_WAVE));
        float *wb = stat + MA_WAVES * MA_WROWS;
        if (lane == 0) {
            wb[2 * wave] = qm;
            wb[2 * wave + 1] = km;
        }
        __syncthreads();
        if (tid < 2) {
            float best = 0.f;
#pragma unroll
            for (int w = 0; w < MA_WAVES; w++) best = fmaxf(best, wb[2 * w + tid]);
            bounds[((size_t)b * MA_BOUND_SLOTS + bid % blocks_per_batch) * 2 + tid] = best;
        }
    }

    // ---- rows: 1 / max(1e-9, sum); the two lane halves hold disjoint keys of the same row ----
    rs += __shfl_xor(rs, 32, SPT_WAVE);
    const float inv = 1.0f / fmaxf(1e-9f, rs);
    float *wstat = stat + wave * MA_WROWS;
    if (h == 0) {
        wstat[c32] = inv;
        if (i0 + c32 < S) row_sum[(size_t)b * S + i0 + c32] = ((ct.satrows >> c32) & 1u) ? rs * MA_SAT_UP : rs;
    }
    wave_lds_fence();
    T *y_b = y + (size_t)b * S * MA_E;
    if (i0 < S) {
        if (!YT) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int il = acc_row(r, h);
                const float sc = wstat[il];
                if (i0 + il < S) {
#pragma unroll
                    for (int eh = 0; eh < MA_ET; eh++)
                        st1(y_b + (size_t)(i0 + il) * MA_E + 32 * eh + c32, yacc[eh][r] * sc);
                }
            }
        } else {
            // [batch, E, S]: registers 4g .. 4g+3 are four consecutive rows of one e, but
            // stored straight from the accumulator layout every lane writes its own 16-byte
            // piece 4 S bytes from its neighbour's (14 us of partial-line writes).  Through a
            // wave-private LDS tile [e][row] instead: 8 lanes then cover 128 contiguous bytes
            // of one e and a store instruction writes 8 such runs.
            float *tile = reinterpret_cast<float *>(img) + wave * (32 * MA_TLD);
#pragma unroll
            for (int eh = 0; eh < MA_ET; eh++) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int il = 8 * g + 4 * h;
                    *reinterpret_cast<float4 *>(tile + c32 * MA_TLD + il) = make_float4(
                        yacc[eh][4 * g] * wstat[il], yacc[eh][4 * g + 1] * wstat[il + 1],
                        yacc[eh][4 * g + 2] * wstat[il + 2], yacc[eh][4 * g + 3] * wstat[il + 3]);
                }
                wave_lds_fence();
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    const int el = (lane >> 3) + 8 * k4, i4 = (lane & 7) * 4;
                    const float4 o = *reinterpret_cast<const float4 *>(tile + el * MA_TLD + i4);
                    T *dst = y_b + (size_t)(el + 32 * eh) * S + i0 + i4;
                    if (i0 + i4 + 3 < S && (S & 3) == 0) {
                        st4(dst, o);
                    } else {
                        const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                        for (int u = 0; u < 4; u++)
                            if (i0 + i4 + u < S) st1(dst + u, ov[u]);
                    }
                }
                wave_lds_fence();
            }
        }
    }
}

// ===================================== backward ============================================
// Nothing of size nnz was saved: both kernels recompute the score tiles from q, k and the
// probabilities from the forward's row sums.  With P = m exp(s) / row_sum and
//   delta_i = max(1e-9, sum_p P_p dP_p) = max(1e-9, dY_i . Y_i)        (softmax.cu:69; P dP
//             summed over a row is dY_i . (P V)_i = dY_i . Y_i)
//   dS      = scale * P * (dP - delta) inside the clamp, 0 outside      (softmax.cu:75-78)
// the row-owned kernel produces grad_q (+ delta for its sibling), the key-owned one grad_k
// and grad_v:
//   grad_q[i] = sum_j dS[i, j] k[j]     grad_k[j] = sum_i dS[i, j] q[i]     (kernels/sddmm.py)
//   grad_v[j] = sum_i P[i, j] dY[i]     dP[i, j]  = dY[i] . v[j]            (kernels/spmm.py)

// ---- row-owned: same skeleton as the forward and the same images, K rows | V rows (the K
// rows image also feeds grad_q = dS^T K through the transposing read).
// Measured by compiling phases out (80 us whole): without the tile arithmetic 54 us (prologue --
// dY, Y, Q: 100 MB -- 23, the loop's loads 23, epilogue 6, image stores 2, barriers 0); WITH the
// arithmetic but without the loop's loads, stores and barriers 75 us.  So the loop's memory
// traffic hides behind the arithmetic, and the arithmetic (46 us for 16 us of MFMA-pipe time) is
// one wave per SIMD running its MFMA and VALU phases one after the other: at 187 VGPRs a CU holds
// one workgroup, and half of its waves (the short row tiles) finish early. ----
// GT: grad_y and y arrive as [batch, E, S] (the transposed forward output and its gradient)
template <typename T, bool GT>
__global__ __launch_bounds__(MA_THREADS)
__attribute__((amdgpu_waves_per_eu(MA_ROWS_WAVES_PER_EU, MA_ROWS_WAVES_PER_EU)))
void attention_mfma_backward_rows_kernel(
    const unsigned long long *__restrict__ masks, const unsigned char *__restrict__ cells,
    const unsigned char *__restrict__ pool, const unsigned *__restrict__ sat,
    const T *__restrict__ q, const T *__restrict__ k, const T *__restrict__ v,
    const T *__restrict__ gy, const T *__restrict__ y,
    const float *__restrict__ row_sum, const float *__restrict__ bounds, T *__restrict__ grad_q,
    float *__restrict__ delta, int S, float scale, float clampv, int heads, int blocks_per_batch,
    int half) {
    constexpr int PI = Elem<T>::PARTS;                  // parts of the K / V images
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *img = smem;                                   // [2][MA_IMG]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c32 = lane & 31;
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / blocks_per_batch;
    const DenseView dv = dense_view(b, S, MA_E, heads);
    const int RT = (S + MA_WROWS - 1) / MA_WROWS;
    const ScoreMap sm(scale, clampv);
    const int i0 = MA_WROWS * folded_row_tile(bid % blocks_per_batch, blocks_per_batch, wave);
    const CellTiles ct(masks, cells, pool, sat, b, RT, i0 / MA_WROWS, i0 < S);
    const int last_tile = folded_row_tile(bid % blocks_per_batch, blocks_per_batch, MA_WAVES - 1);
    const int T_ = min(RT, last_tile + 1);
    const TileStager<T> stager(k + dv.base, v + dv.base, dv.ld, S, tid);
    TileRegs<T> nxt = stager.load(0);
    uint4 mcur = ct.load(0, lane), mnxt;
    // own rows: dY (B operand of dP = V dY^T), delta = max(1e-9, dY . Y), Q (B operand of D)
    Frag gf[MA_KS], qf[MA_KS];
    float delta_i, q2 = 0.f;
    const int row = i0 + c32;
    float rsum = row_sum[(size_t)b * S + min(row, S - 1)];
    if ((ct.satrows >> c32) & 1u) rsum *= MA_SAT_DOWN;          // (the sum the tiles' counts add up to)
    {
        float xr[MA_E / 2], yr[MA_E / 2], xq[MA_E / 2];
        const size_t ob = (size_t)b * S * MA_E;
        load_own_rows_raw(xq, q + dv.base, dv.ld, S, i0, lane);
        if (GT) {
            // [E][S] operands: wide loads through the wave's LDS tile (the image buffers
            // are free until the first key tile is staged)
            float *tile = reinterpret_cast<float *>(img) + wave * (32 * MA_TLD);
            load_own_rows_transposed(xr, gy + ob, S, i0, tile, lane);
            load_own_rows_transposed(yr, y + ob, S, i0, tile, lane);
            __syncthreads();
        } else {
            load_own_rows_raw(xr, gy + ob, MA_E, S, i0, lane);
            load_own_rows_raw(yr, y + ob, MA_E, S, i0, lane);
        }
        stager.store(img, nxt);
        float dl = 0.f;
#pragma unroll
        for (int x = 0; x < MA_E / 2; x++) dl = fmaf(xr[x], yr[x], dl);
        dl += __shfl_xor(dl, 32, SPT_WAVE);
        delta_i = fmaxf(1e-9f, dl);
        // scale / row_sum is a per-row factor of dS: folded into dY (and delta) once here
        // instead of into every cell: dS = m e (dP' - delta') with dP' = pscale dP
        const float pscale = scale / fmaxf(1e-9f, rsum);
        split_own_rows(gf, xr, pscale);
        split_own_rows(qf, xq, sm.sl2);
        // for the key-owned sibling: the row's weight 1 / row_sum (of the sum its tile counts add up
        // to) and the weighted delta -- it stages 32 rows per iteration and divides nothing
        if (h == 0 && row < S && half == 0) {
            const float w = 1.0f / fmaxf(1e-9f, rsum);
            delta[(size_t)b * S + row] = w;
            delta[((size_t)(gridDim.x / blocks_per_batch) + b) * S + row] = w * delta_i;
        }
        delta_i *= pscale;
        if (bounds) {
#pragma unroll
            for (int x = 0; x < MA_E / 2; x++) q2 = fmaf(xq[x], xq[x], q2);
            q2 += __shfl_xor(q2, 32, SPT_WAVE);
        }
    }
    const ClampGuard cg(bounds, b, blocks_per_batch, 1, q2, sm, clampv);
    __syncthreads();

    f32x16 qacc[MR_QTILES];
#pragma unroll
    for (int e = 0; e < MR_QTILES; e++)
#pragma unroll
        for (int r = 0; r < 16; r++) qacc[e][r] = 0.f;

    for (int t = 0; t < T_; t++) {
        const char *buf = img + (t & 1) * MA_IMG;
        nxt = stager.load(min(t + 1, T_ - 1));
        mnxt = ct.load(t + 1, lane);
        if (ct.live(t)) {
            f32x16 d, dp;
#pragma unroll
            for (int r = 0; r < 16; r++) d[r] = dp[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < MA_KS; ks++) {
                d = mm<PI, 2>(read_rows<PI>(buf + MA_KH, buf + MA_KL, lane, ks), qf[ks], d);
                dp = mm<PI, 2>(read_rows<PI>(buf + MA_VH, buf + MA_VL, lane, ks), gf[ks], dp);
            }
            const uint4 cw = ct.words(mcur, t, lane);
            if (cg.wanted(d))                           // (wave-uniform, rare)
                clamp_exact(d, dp, smem + 2 * MA_IMG, wave, cw, sm.bound, cg.thr, q + dv.base, i0, S,
                            k + dv.base + (size_t)t * MA_KT * dv.ld, dv.ld, scale, clampv);
            const unsigned mw[4] = {cw.x, cw.y, cw.z, cw.w};
            float ds[16];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float m4[4] = {cell_count<0>(mw[g]), cell_count<1>(mw[g]),
                                     cell_count<2>(mw[g]), cell_count<3>(mw[g])};
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    const int r = 4 * g + x;
                    const float pw = m4[x] * sm.exp_of(d[r]);
                    ds[r] = sm.inside(d[r]) ? pw * (dp[r] - delta_i) : 0.0f;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const Frag sf = split8(ds[8 * s2], ds[8 * s2 + 1], ds[8 * s2 + 2], ds[8 * s2 + 3],
                                       ds[8 * s2 + 4], ds[8 * s2 + 5], ds[8 * s2 + 6], ds[8 * s2 + 7]);
#pragma unroll
                for (int eh = 0; eh < MR_QTILES; eh++)
                    qacc[eh] = mm<2, PI>(sf, read_cols_tr<PI>(buf + MA_KH, buf + MA_KL,
                                                              32 * MR_QTILES * half + 32 * eh, lane, s2),
                                         qacc[eh]);
            }
        }
        if (t + 1 < T_) stager.store(img + ((t + 1) & 1) * MA_IMG, nxt);
        mcur = mnxt;
        __syncthreads();
    }
    if (i0 < S) {
        float *tile = reinterpret_cast<float *>(img) + wave * (32 * MA_TLD);
        T *gq_b = grad_q + dv.base + (size_t)i0 * dv.ld + 32 * MR_QTILES * half;
#pragma unroll
        for (int eh = 0; eh < MR_QTILES; eh++)
            store_acc_half(qacc[eh], 1.0f, tile, gq_b + 32 * eh, dv.ld, S - i0, lane);
    }
}

// ---- key-owned: a wave owns 32 keys (K, V fragments and the grad_k, grad_v accumulators in
// registers) and walks the row tiles at or below the diagonal; 32 rows of Q and dY stream
// through LDS as images Q rows | dY rows | Q cols | dY cols, plus the rows' 1 / row_sum and
// delta.  Tiles are computed as D[row, key], so the sums over rows take the accumulator tile
// as the A operand.  Folded like the rows: waves 0-3 own key tiles 4g .. 4g+3 (many row
// tiles), waves 4-7 their mirror images (few).
// images: Q rows | dY (rows image, or -- GT, the operand arriving as [E][S] -- a cols image:
// whichever its source layout fills with 8-byte stores; the other orientation of each operand
// comes from the transposing read) | the rows' scaled delta
constexpr int MK_GSLOT = MA_CIMG > MA_RIMG ? MA_CIMG : MA_RIMG;           // either orientation
constexpr int MK_QR = 0, MK_G = 2 * MA_RIMG, MK_ST = 2 * MA_RIMG + 2 * MK_GSLOT,
              MK_IMG = MK_ST + MA_WROWS * 4;                                // 18560 B at E = 64
template <typename T, bool GT>
struct KeysStager {
    const T *q_b, *gy_b;
    const float *w_b, *wd_b;    // the rows' 1 / row_sum and (1 / row_sum) delta, from the row-owned kernel
    int ld, S, tid;
    struct Regs {
        typename Elem<T>::raw4 qf[MA_RPT], gf[MA_RPT];
        float4 rs[MA_RPT];
        float st_wd;
    };
    // unconditional, clamped loads (see TileStager::load); rows >= S are given weight 0 by
    // store(), so whatever finite data the clamped rows hold never counts.  Raw values only:
    // arithmetic on them here would wait for the loads at the top of the iteration.
    __device__ __forceinline__ Regs load(int rt) const {
        Regs r;
        const int i0 = rt * MA_WROWS;
        const int e4 = (tid % MA_EQ) * 4;
#pragma unroll
        for (int u = 0; u < MA_RPT; u++) {
            const int il = tid / MA_EQ + MA_RPP * u;
            const int row = min(i0 + il, S - 1);
            r.qf[u] = ld_raw4(q_b + (size_t)row * ld + e4);
            if (!GT) {
                r.gf[u] = ld_raw4(gy_b + (size_t)row * MA_E + e4);
                r.rs[u] = make_float4(w_b[row], 0.f, 0.f, 0.f);
            } else {        // [E][S]: four consecutive rows of one e (S % 4 == 0)
                const int e = (tid >> 3) + (MA_THREADS / 8) * u;
                const int i4 = min(i0 + (tid & 7) * 4, S - 4);
                r.gf[u] = ld_raw4(gy_b + (size_t)e * S + i4);
                r.rs[u] = *reinterpret_cast<const float4 *>(w_b + i4);
            }
        }
        const int sr = min(i0 + (tid & (MA_WROWS - 1)), S - 1);
        r.st_wd = wd_b[sr];
        return r;
    }
    // 1 / row_sum is a per-row factor of both P (grad_v) and dS (grad_k): dY is staged
    // pre-multiplied by it (an fp32 product: two parts whatever the storage type), rows >= S
    // by 0, and the rows' delta likewise
    __device__ __forceinline__ void store(char *buf, const Regs &r, int rt) const {
        const int i0 = rt * MA_WROWS;
        const int e4 = (tid % MA_EQ) * 4;
        auto weight = [&](float w, int row) { return row < S ? w : 0.0f; };
#pragma unroll
        for (int u = 0; u < MA_RPT; u++) {
            const int il = tid / MA_EQ + MA_RPP * u;
            put_rows4(buf + MK_QR, il, e4, r.qf[u]);
            const float4 g = Elem<T>::f4(r.gf[u]);
            if (!GT) {
                const float w = weight(r.rs[u].x, i0 + il);
                put_rows4(buf + MK_G, il, e4, make_float4(w * g.x, w * g.y, w * g.z, w * g.w));
            } else {
                const int i4 = (tid & 7) * 4;
                put_cols4(buf + MK_G, (tid >> 3) + (MA_THREADS / 8) * u, i4,
                          make_float4(weight(r.rs[u].x, i0 + i4) * g.x,
                                      weight(r.rs[u].y, i0 + i4 + 1) * g.y,
                                      weight(r.rs[u].z, i0 + i4 + 2) * g.z,
                                      weight(r.rs[u].w, i0 + i4 + 3) * g.w));
            }
        }
        if (tid < MA_WROWS)
            reinterpret_cast<float *>(buf + MK_ST)[tid] = weight(r.st_wd, i0 + tid);
    }
};

// MODE 0: grad_k and grad_v together, 64 columns (`half`) per launch -- d_head 64.
// d_head 128 cannot hold K, V fragments (128 registers) and four accumulator tiles beside the
// tile arithmetic (the combined kernel spilled 41-60 registers: 281 us per launch), so there
// MODE 1: grad_v alone, all 128 columns (needs P only: no V fragments, no dP, no delta);
// MODE 2: grad_k alone, 64 columns per launch.
template <typename T, bool GT, int MODE>
__global__ __launch_bounds__(MA_THREADS)
__attribute__((amdgpu_waves_per_eu(MA_KEYS_WAVES_PER_EU, MA_KEYS_WAVES_PER_EU)))
void attention_mfma_backward_keys_kernel(
    const unsigned long long *__restrict__ masks, const unsigned char *__restrict__ cells_t,
    const unsigned char *__restrict__ pool_t, const unsigned *__restrict__ sat,
    const T *__restrict__ q, const T *__restrict__ k, const T *__restrict__ v,
    const T *__restrict__ gy, const float *__restrict__ row_sum, const float *__restrict__ bounds,
    const float *__restrict__ delta, T *__restrict__ grad_k, T *__restrict__ grad_v,
    int S, float scale, float clampv, int heads, int blocks_per_batch, int half) {
    constexpr int PI = Elem<T>::PARTS;                  // parts of the Q image and of the V fragments
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *img = smem;                                   // [2][MK_IMG]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c32 = lane & 31;
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / blocks_per_batch, g = bid % blocks_per_batch;
    const int kt = folded_row_tile(g, blocks_per_batch, wave);
    const int j0 = kt * MA_KT;
    const DenseView dv = dense_view(b, S, MA_E, heads);
    const int RT = (S + MA_WROWS - 1) / MA_WROWS;
    const bool have = j0 < S;

    constexpr bool WANT_K = MODE != 1, WANT_V = MODE != 2;
    constexpr int NVT = MODE == 1 ? MA_ET : 2;           // 32-column tiles of grad_v produced
    const ScoreMap sm(scale, clampv);
    Frag kf[MA_KS], vf[WANT_K ? MA_KS : 1];
    float k2 = 0.f;
    {
        float xk[MA_E / 2];
        load_own_rows_raw(xk, k + dv.base, dv.ld, S, j0, lane);
        if (WANT_K && bounds) {
#pragma unroll
            for (int x = 0; x < MA_E / 2; x++) k2 = fmaf(xk[x], xk[x], k2);
            k2 += __shfl_xor(k2, 32, SPT_WAVE);
        }
        split_own_rows(kf, xk, sm.sl2);     // the score scale (log2 domain) folded into K
        if constexpr (WANT_K) {             // V only feeds dP, which only dS needs
            float xv[MA_E / 2];
            load_own_rows_raw(xv, v + dv.base, dv.ld, S, j0, lane);
            split_own_rows(vf, xv);         // (stored bf16: the lo parts are zero and unused)
        }
    }
    // Round 4: the exact clamp mask put this kernel five registers past its 256; hipcc's answer was
    // one 16-byte spill of a K / V fragment, reloaded from scratch in every iteration -- a VECTOR
    // memory load, whose wait also waits for the row tile requested at the top of the iteration
    // (vmcnt counts in order): 117.7 - 132.7 us against 106.9.  The last V fragment (8 registers with
    // both parts) is parked in a wave-private LDS block instead and read back once per iteration, an
    // LDS read among two dozen others, on lgkmcnt.
    constexpr bool STASH_V = WANT_K && MA_E == 64 && sizeof(T) == 4;
    char *const vstash = smem + 2 * MK_IMG + MA_WAVES * MA_PARK + wave * 2048 + 16 * lane;
    if constexpr (STASH_V) {
        *reinterpret_cast<uint4 *>(vstash) = make_uint4(vf[MA_KS - 1].hi.x, vf[MA_KS - 1].hi.y,
                                                        vf[MA_KS - 1].hi.z, vf[MA_KS - 1].hi.w);
        *reinterpret_cast<uint4 *>(vstash + 1024) = make_uint4(vf[MA_KS - 1].lo.x, vf[MA_KS - 1].lo.y,
                                                               vf[MA_KS - 1].lo.z, vf[MA_KS - 1].lo.w);
    }
    // the slice's row-tile masks live in registers (lane rt: mask of row tile rt), so that the
    // loop body has no load whose result it needs at once
    // (grad_v needs P alone, which is continuous in the score: no exact mask for MODE 1)
    const ClampGuard cg(WANT_K ? bounds : nullptr, b, blocks_per_batch, 0, k2, sm, clampv);
    // (round 4: ONE register -- this wave's key tile needs two bits per row tile, lane rt holds bit 0:
    // tile (rt, kt) is live, bit 1: it is stored in byte form; the 64-bit masks were four registers
    // of a kernel that has none to spare)
    unsigned bits_reg = 0u;
    if (lane < RT) {
        const unsigned long long m0 = masks[2 * ((size_t)b * RT + lane)];
        const unsigned long long m1 = masks[2 * ((size_t)b * RT + lane) + 1];
        bits_reg = (unsigned)((m0 >> min(kt, 63)) & 1ull) | ((unsigned)((m1 >> min(kt, lane)) & 1ull) << 1);
    }
    // bits of tile (rt, kt) (0 past the last row tile)
    auto mask_of = [&](int rt) -> unsigned {
        return rt < RT ? (unsigned)__builtin_amdgcn_readlane((int)bits_reg, rt) : 0u;
    };
    // is the tile (row tile rt, this wave's key tile) stored in byte form?
    auto multi_of = [&](int rt) {
        return (bool)(((unsigned)__builtin_amdgcn_readlane((int)bits_reg, min(rt, RT - 1)) >> 1) & 1u);
    };
    const int cstride = cell_slot_bytes(pool_t) / 16;
    const uint4 *cell_b = reinterpret_cast<const uint4 *>(
        cells_t + (size_t)b * tri(RT) * cell_slot_bytes(pool_t));
    const uint4 *pool_b = pool_t ? reinterpret_cast<const uint4 *>(pool_t + (size_t)b * RT * MA_CELLS)
                                 : nullptr;
    auto live = [&](unsigned m, int rt) { return have && rt >= kt && rt < RT && (m & 1u); };
    // always in bounds: the row tile clamped to the last one, the key tile to the diagonal
    auto cell_load = [&](int rt) {
        const int rc = min(rt, RT - 1);
        if (multi_of(rt)) return pool_b ? pool_b[rc * 64 + lane] : cell_b[(tri(rc) + min(kt, rc)) * 64 + lane];
        return cell_b[(tri(rc) + min(kt, rc)) * cstride + (c32 >> 2)];
    };

    const int rt0 = (MA_WAVES / 2) * g;                 // the first row tile any wave needs
    const KeysStager<T, GT> stager{q + dv.base, gy + (size_t)b * S * MA_E, delta + (size_t)b * S,
                                   delta + ((size_t)(gridDim.x / blocks_per_batch) + b) * S, dv.ld, S, tid};
    stager.store(img, stager.load(min(rt0, RT - 1)), rt0);
    unsigned mcur_mask = mask_of(rt0);
    uint4 mcur = cell_load(rt0);
    __syncthreads();

    f32x16 kacc[MK_KTILES], vacc[NVT];
#pragma unroll
    for (int e = 0; e < MK_KTILES; e++)
#pragma unroll
        for (int r = 0; r < 16; r++) kacc[e][r] = 0.f;
#pragma unroll
    for (int e = 0; e < NVT; e++)
#pragma unroll
        for (int r = 0; r < 16; r++) vacc[e][r] = 0.f;

    for (int rt = rt0; rt < RT; rt++) {
        const char *buf = img + ((rt - rt0) & 1) * MK_IMG;
        const typename KeysStager<T, GT>::Regs nxt = stager.load(min(rt + 1, RT - 1));
        const unsigned mnxt_mask = mask_of(rt + 1);
        const uint4 mnxt = cell_load(rt + 1);
        if (live(mcur_mask, rt)) {
            f32x16 d, dp;
#pragma unroll
            for (int r = 0; r < 16; r++) d[r] = dp[r] = 0.f;
            Frag vlast;
            if constexpr (STASH_V) {
                vlast.hi = *reinterpret_cast<const uint4 *>(vstash);
                vlast.lo = *reinterpret_cast<const uint4 *>(vstash + 1024);
            }
#pragma unroll
            for (int ks = 0; ks < MA_KS; ks++) {
                d = mm<PI, 2>(read_rows<PI>(buf + MK_QR, buf + MK_QR + MA_RIMG, lane, ks), kf[ks], d);
                if constexpr (WANT_K)
                    dp = mm<2, PI>(GT ? read_rows_tr(buf + MK_G, buf + MK_G + MA_CIMG, lane, ks)
                                      : read_rows<2>(buf + MK_G, buf + MK_G + MA_RIMG, lane, ks),
                                   (STASH_V && ks == MA_KS - 1) ? vlast : vf[ks], dp);
            }
            // this lane: key j0 + c32, rows 8g + 4h + u of the tile in register 4g + u
            const bool mlt = multi_of(rt);
            const uint4 cw = cell_words(mcur, mlt, lane);
            if constexpr (WANT_K) {
                if (cg.wanted(d))                       // (wave-uniform, rare)
                    clamp_exact(d, dp, smem + 2 * MK_IMG, wave, cw, sm.bound, cg.thr, k + dv.base, j0, S,
                                q + dv.base + (size_t)rt * MA_WROWS * dv.ld, dv.ld, scale, clampv);
            }
            const float *st = reinterpret_cast<const float *>(buf + MK_ST);
            const unsigned mw[4] = {cw.x, cw.y, cw.z, cw.w};
            float p[16], ds[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                float4 del4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (WANT_K) del4 = *reinterpret_cast<const float4 *>(st + 8 * g4 + 4 * h);
                const float del[4] = {del4.x, del4.y, del4.z, del4.w};
                const float m4[4] = {cell_count<0>(mw[g4]), cell_count<1>(mw[g4]),
                                     cell_count<2>(mw[g4]), cell_count<3>(mw[g4])};
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int r = 4 * g4 + u;
                    p[r] = m4[u] * sm.exp_of(d[r]);          // x 1 / row_sum: in the dY images
                    ds[r] = 0.0f;
                    if constexpr (WANT_K)                    // x scale: in the epilogue
                        ds[r] = sm.inside(d[r]) ? p[r] * (dp[r] - del[u]) : 0.0f;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const Frag pf = split8(p[8 * s2], p[8 * s2 + 1], p[8 * s2 + 2], p[8 * s2 + 3],
                                       p[8 * s2 + 4], p[8 * s2 + 5], p[8 * s2 + 6], p[8 * s2 + 7]);
                const Frag sf = split8(ds[8 * s2], ds[8 * s2 + 1], ds[8 * s2 + 2], ds[8 * s2 + 3],
                                       ds[8 * s2 + 4], ds[8 * s2 + 5], ds[8 * s2 + 6], ds[8 * s2 + 7]);
                if constexpr (WANT_V) {
#pragma unroll
                    for (int eh = 0; eh < NVT; eh++) {
                        const int col0 = (MODE == 1 ? 0 : 64 * half) + 32 * eh;
                        vacc[eh] = mm<2, 2>(pf, GT ? read_cols(buf + MK_G, buf + MK_G + MA_CIMG,
                                                               c32 + col0, h, s2)
                                                   : read_cols_tr<2>(buf + MK_G, buf + MK_G + MA_RIMG,
                                                                     col0, lane, s2), vacc[eh]);
                    }
                }
                if constexpr (WANT_K) {
#pragma unroll
                    for (int eh = 0; eh < MK_KTILES; eh++)
                        kacc[eh] = mm<2, PI>(sf, read_cols_tr<PI>(buf + MK_QR, buf + MK_QR + MA_RIMG,
                                                                  32 * MK_KTILES * half + 32 * eh, lane, s2),
                                             kacc[eh]);
                }
            }
        }
        if (rt + 1 < RT) stager.store(img + ((rt + 1 - rt0) & 1) * MK_IMG, nxt, rt + 1);
        mcur = mnxt;
        mcur_mask = mnxt_mask;
        __syncthreads();
    }
    if (have) {
        float *tile = reinterpret_cast<float *>(img) + wave * (32 * MA_TLD);
        if constexpr (WANT_K) {
            T *gk_b = grad_k + dv.base + (size_t)j0 * dv.ld + 32 * MK_KTILES * half;
#pragma unroll
            for (int eh = 0; eh < MK_KTILES; eh++)
                store_acc_half(kacc[eh], scale, tile, gk_b + 32 * eh, dv.ld, S - j0, lane);
        }
        if constexpr (WANT_V) {
            T *gv_b = grad_v + dv.base + (size_t)j0 * dv.ld + (MODE == 1 ? 0 : 64 * half);
#pragma unroll
            for (int eh = 0; eh < NVT; eh++)
                store_acc_half(vacc[eh], 1.0f, tile, gv_b + 32 * eh, dv.ld, S - j0, lane);
        }
    }
}

// (Round 3 also built the backward as ONE launch per slice -- S, P, dP, dS once per tile pair, 60
// instead of 84 MFMA-equivalents, dQ carried through LDS: 164.6 us against the two kernels' 165.0 at
// the configs[1] shape, because neither form is bound by the matrix pipe.  DESIGN.md 5.11 keeps the
// measurements and the ablation; the kernel left the tree in round 4.)

static size_t mfma_forward_lds() {
    return 2 * MA_IMG + (size_t)MA_WAVES * (MA_WROWS + 2) * sizeof(float);
}

// ---- spmm (A . x over the CSR) on the matrix cores: the operator form of the P . V product ----
// Replaces extension/spmm.cpp:27-69 (cusparseSpMM, non-transposed) for patterns as dense as
// lookup's: y[b, i, :] = sum_{p in row i} values[b, p] x[b, indices[b, p], :].
//
// The gather form (spmm.hip) moves one 256-byte row of x through LDS per CSR entry: 2.1 GB per
// launch at the configs[1] shape for 134 MB of HBM traffic, LDS-bandwidth bound (27 us at 100 % of
// the LDS rate, 46.9 measured).  Here a wave owns 32 rows, x streams through LDS 32 keys at a time
// as in the attention forward (bf16 images, transposing reads), and the wave's P tile [32 rows x
// 32 keys] is assembled DENSE in a wave-private 4.5 KiB LDS tile: LDS traffic per entry is 4 bytes
// in and 4 bytes out instead of 256.
//   * The wave keeps its rows' entries in registers: lane l holds entry l of each of the 32 rows,
//     packed as key tile << 16 | byte address in the tile.
//   * Duplicates (the same column twice in a row: lookup pads short causal rows with column 0) are
//     merged ONCE, in the prologue: every lane writes its lane id to owner[col], reads it back, and
//     where another lane won adds its value to the winner's (ds_add_f32 into a 64-float scratch, a
//     rare wave-uniform branch) and retires.  After that a (row, column) occurs once, so ...
//   * ... per key tile each row is two vector instructions and one plain ds_write_b32, no branch:
//     a = packed ^ (tile << 16) is the cell's address if the entry belongs to the tile and >= 65536
//     if not; min(a, trash + 4 lane) sends the others to a per-lane trash word.  (The first form --
//     a predicated ds_add_f32 per row and tile -- took 110 us: 41 of them the LDS atomics, 17 the
//     exec-mask round trips.)
//   * The tile is read back in the accumulator layout (and zeroed behind the read), split in two bf16
//     parts and multiplied with the x image.  Key tiles no row of the wave has an entry in are skipped
//     (a 64-bit mask formed once).
//   * Rows of more than 64 entries take a slow loop (their further entries are re-read from global
//     memory for every tile and added with ds_add_f32): the dispatcher only sends patterns whose MEAN
//     row has at most 64 here, so this is for ragged patterns' few long rows.
// Workgroups of four waves (128 rows), three to a CU; x tiles are requested two iterations ahead.
//
// MEASURED (round 4, rocprofv3, configs[1] shape: 256 slices x 512 rows x 64, 64 entries per row, 134 MB
// algorithmic): 65.7 us = 0.26 of the HBM roofline, against 46.3 us (0.36) for the gather form -- the
// dense-tile form does NOT pay for spmm the way it does for sddmm, and is therefore opt-in
// (SPT_SPMM_MFMA=1).  Where the time goes (ablation builds, tools/variant.sh -DSN_ABL_*):
//   22.4 us  the prologue and epilogue alone (entries in, live mask, first x tile, y out: 100 MB of
//            the 134, latency- and HBM-bound, nothing else running meanwhile)
//   +11.5    the duplicate merge (8 LDS round trips per wave; 2.6 us of it not hidden in the full kernel)
//   +26      the 16 tile iterations without the scatter (P tile read / zero, split, 16 transposing
//            reads, 12 MFMAs, x staging, one barrier: a chain of LDS round trips per wave that three
//            waves per SIMD do not cover; the MFMAs themselves are free: 66.6 us with them against 67.7
//            without in an early build)
//   +15      the scatter (32 x [xor, min, ds_write_b32] per tile and wave)
// and 1,024 workgroups on 768 slots (153 VGPRs: three waves per SIMD) run as 1.33 rounds.  The
// first build (one predicated ds_add_f32 per row and tile, eight-wave workgroups, entries loaded row by
// row) took 110 us: 41 of them the LDS atomics, 35 the serialised loads.  What would be needed to beat
// the gather form: <= 128 VGPRs (the 64 entry registers packed two to a register) for two eight-wave
// workgroups per CU in ONE round, and the next row block's entries in flight during the tile loop --
// both against the register budget; the fused forward (attention_mfma_forward_kernel) does the same
// product in 44 us INCLUDING the scores and the softmax because its P tile never leaves registers.
constexpr int SN_THREADS = 256;
constexpr int SN_WAVES = SN_THREADS / SPT_WAVE;
constexpr int SN_ROWS = SN_WAVES * MA_WROWS;
constexpr int SN_RPT = 32 * MA_EQ / SN_THREADS;         // float4s per thread to stage a 32 x E tile
constexpr int SN_RPP = SN_THREADS / MA_EQ;
constexpr int SN_PLD = 36;                              // floats per row of the dense P tile
constexpr int SN_PT = MA_WROWS * SN_PLD * 4;            // bytes of one wave's P tile
constexpr int SN_TRASH = SN_PT;                         // 64 words behind it: one per lane
constexpr int SN_SUM = SN_TRASH + 256;                  // 64 floats: the duplicates' sums
constexpr int SN_OWNER = SN_SUM + 256;                  // SN_NOWN x S bytes (rounded up to 16): owner[col]
constexpr int SN_IMG = 2 * MA_RIMG;                     // one x tile: rows image, hi | lo
constexpr int SN_NOWN = 4;                              // owner arrays: rows deduplicated per LDS round trip
__host__ __device__ constexpr int sn_wave_bytes(int S) { return SN_OWNER + SN_NOWN * (((S + 15) / 16) * 16); }
struct SnRegs { float4 xf[SN_RPT]; };
#if MA_E_VALUE == 64
#define SN_WAVES_PER_EU 3
#else
#define SN_WAVES_PER_EU 2
#endif
__global__ __launch_bounds__(SN_THREADS)
__attribute__((amdgpu_waves_per_eu(SN_WAVES_PER_EU, SN_WAVES_PER_EU)))
void spmm_mfma_kernel(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                      const float *__restrict__ values, const float *__restrict__ x,
                      float *__restrict__ y, int S, int nnz, int x_heads, int y_heads,
                      int blocks_per_batch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *img = smem;                                   // [2][SN_IMG]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c32 = lane & 31;
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / blocks_per_batch;
    const int i0 = MA_WROWS * ((bid % blocks_per_batch) * SN_WAVES + wave);
    const DenseView xv = dense_view(b, S, MA_E, x_heads), yv = dense_view(b, S, MA_E, y_heads);
    const float *x_b = x + xv.base;
    const int KT = (S + MA_KT - 1) / MA_KT;
    char *ptile = smem + 2 * SN_IMG + wave * sn_wave_bytes(S);
    const int32_t *idx_b = indices + (size_t)b * nnz;
    const float *val_b = values + (size_t)b * nnz;

    // the x tiles: thread t stages row t / (E/4) + SN_RPP u, columns 4 (t % (E/4)) .. (clamped rows:
    // keys >= S are finite data no entry points at)
    const int jl = tid / MA_EQ, e4 = (tid % MA_EQ) * 4;
    auto xload = [&](int t) {
        SnRegs r;
#pragma unroll
        for (int u = 0; u < SN_RPT; u++)
            r.xf[u] = ld_raw4(x_b + (size_t)min(t * MA_KT + jl + SN_RPP * u, S - 1) * xv.ld + e4);
        return r;
    };
    auto xstore = [&](char *buf, const SnRegs &r) {
#pragma unroll
        for (int u = 0; u < SN_RPT; u++) put_rows4(buf, jl + SN_RPP * u, e4, r.xf[u]);
    };
    SnRegs nxt = xload(0), nxt2 = xload(min(1, KT - 1));

    // zero the P tile (4608 B: 64 lanes x 16 B x 4 + the rest; the trash words need no value)
#pragma unroll
    for (int o = lane * 16; o < SN_PT; o += 64 * 16) *reinterpret_cast<uint4 *>(ptile + o) = make_uint4(0u, 0u, 0u, 0u);
    // this wave's entries: lane l holds entry l of row i0 + r (r = 0 .. 31); no match ever for the
    // lanes past a row's end or retired as duplicates (packed = ~0)
    unsigned packed[MA_WROWS];
    float val[MA_WROWS];
    unsigned long long mine = 0ull;                     // key tiles this lane has an entry in
    int longest = 0;
    unsigned char *owner = reinterpret_cast<unsigned char *>(ptile + SN_OWNER);
    float *dsum = reinterpret_cast<float *>(ptile + SN_SUM);
    // (three separate loops: the row bounds from ONE vector load, then all 64 loads of the entries in
    // flight together, then the packing -- written as one loop the compiler waited for each row's
    // scalar bounds and then for its column ids before it issued the next row's: 32 round trips, 35 us)
    const int bound = indptr[min(i0 + min(lane, MA_WROWS), S)];   // lane r: where row i0 + r begins
    int cols[MA_WROWS];
#pragma unroll
    for (int r = 0; r < MA_WROWS; r++) {
        const int st = __builtin_amdgcn_readlane(bound, r);
        const int pc = min(st + lane, nnz - 1);         // (unconditional loads)
        cols[r] = idx_b[pc];
        val[r] = val_b[pc];
    }
#pragma unroll
    for (int r = 0; r < MA_WROWS; r++) {
        const int st = __builtin_amdgcn_readlane(bound, r), en = __builtin_amdgcn_readlane(bound, r + 1);
        longest = max(longest, en - st);
        const bool ok = st + lane < en;
        packed[r] = ok ? ((unsigned)(cols[r] >> 5) << 16) | (unsigned)(r * SN_PLD * 4 + (cols[r] & 31) * 4) : ~0u;
    }
    // duplicates within a row: one lane per column wins owner[col]; four rows per LDS round trip
    // (an owner array each)
    const int own_ld = ((S + 15) / 16) * 16;
#ifndef SN_ABL_NO_DEDUPE
#pragma unroll
    for (int r0 = 0; r0 < MA_WROWS; r0 += SN_NOWN) {
#pragma unroll
        for (int u = 0; u < SN_NOWN; u++)
            if (packed[r0 + u] != ~0u) owner[u * own_ld + cols[r0 + u]] = (unsigned char)lane;
        wave_lds_fence();
        int won[SN_NOWN];
        bool any = false;
#pragma unroll
        for (int u = 0; u < SN_NOWN; u++) {
            won[u] = packed[r0 + u] != ~0u ? (int)owner[u * own_ld + cols[r0 + u]] : lane;
            any |= won[u] != lane;
        }
        if (__ballot(any) != 0ull) {                    // (wave-uniform: rows with duplicates)
#pragma unroll
            for (int u = 0; u < SN_NOWN; u++) {
                dsum[lane] = 0.f;
                wave_lds_fence();
                if (won[u] != lane)
                    __hip_atomic_fetch_add(dsum + won[u], val[r0 + u], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WAVEFRONT);
                wave_lds_fence();
                val[r0 + u] += dsum[lane];
                if (won[u] != lane) packed[r0 + u] = ~0u;
                wave_lds_fence();
            }
        }
        wave_lds_fence();
    }
#endif
#pragma unroll
    for (int r = 0; r < MA_WROWS; r++) mine |= packed[r] != ~0u ? 1ull << (packed[r] >> 16) : 0ull;
    if (longest > 64) {                                 // (wave-uniform: rare)
        for (int r = 0; r < MA_WROWS; r++) {
            const int i = min(i0 + r, S);
            const int st = indptr[i], en = indptr[min(i + 1, S)];
            for (int pos = st + 64 + lane; pos < en; pos += 64) mine |= 1ull << (idx_b[pos] >> 5);
        }
    }
    unsigned long long live = 0ull;                     // ... any lane of the wave has
    for (int t = 0; t < KT; t++)
        if (__ballot((mine >> t) & 1ull) != 0ull) live |= 1ull << t;
    xstore(img, nxt);
    __syncthreads();

    f32x16 yacc[MA_ET];
#pragma unroll
    for (int eh = 0; eh < MA_ET; eh++)
#pragma unroll
        for (int r = 0; r < 16; r++) yacc[eh][r] = 0.f;
    const unsigned trash = (unsigned)(SN_TRASH + 4 * lane);

#ifdef SN_ABL_NO_TILES
    if (S == 12345)
#endif
    for (int t = 0; t < KT; t++) {
        const char *buf = img + (t & 1) * SN_IMG;
        nxt = nxt2;                                     // tile t + 1: requested an iteration ago
        nxt2 = xload(min(t + 2, KT - 1));
        if ((live >> t) & 1ull) {                       // (wave-uniform)
            const unsigned tk = (unsigned)t << 16;
#ifdef SN_ABL_NO_LOOP
            if (tk == 0x7fff0000u)
#endif
#pragma unroll
            for (int r = 0; r < MA_WROWS; r++) {
                // this tile's entries: a = the cell's byte address; the others: a >= 65536 -> the trash word
                const unsigned a = min(packed[r] ^ tk, trash);
                *reinterpret_cast<float *>(ptile + a) = val[r];
            }
            if (longest > 64) {
                wave_lds_fence();
                for (int r = 0; r < MA_WROWS; r++) {
                    const int i = min(i0 + r, S);
                    const int st = indptr[i], en = indptr[min(i + 1, S)];
                    for (int pos = st + 64 + lane; pos < en; pos += 64) {
                        const int col = idx_b[pos];
                        if ((col >> 5) == t)
                            __hip_atomic_fetch_add(reinterpret_cast<float *>(ptile) + r * SN_PLD + (col & 31),
                                                   val_b[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                }
            }
            wave_lds_fence();
            // the tile in the accumulator layout: row c32, keys 32 t + acc_row(r, h) -- four float4;
            // and zero again behind the read
            float p[16];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float4 *cell = reinterpret_cast<float4 *>(ptile) + (c32 * SN_PLD + 8 * g + 4 * h) / 4;
                const float4 v4 = *cell;
                p[4 * g] = v4.x; p[4 * g + 1] = v4.y; p[4 * g + 2] = v4.z; p[4 * g + 3] = v4.w;
                *cell = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            wave_lds_fence();
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const Frag pf = split8(p[8 * s2], p[8 * s2 + 1], p[8 * s2 + 2], p[8 * s2 + 3],
                                       p[8 * s2 + 4], p[8 * s2 + 5], p[8 * s2 + 6], p[8 * s2 + 7]);
#pragma unroll
                for (int eh = 0; eh < MA_ET; eh++)
#ifdef SN_ABL_NO_MMA
                    yacc[eh][s2] += __builtin_bit_cast(float, pf.hi.x ^ pf.lo.y);
#else
                    yacc[eh] = mm<2, 2>(pf, read_cols_tr<2>(buf, buf + MA_RIMG, 32 * eh, lane, s2), yacc[eh]);
#endif
            }
        }
        if (t + 1 < KT) xstore(img + ((t + 1) & 1) * SN_IMG, nxt);
        __syncthreads();
    }
    // register r of lane (c32, h): row i0 + acc_row(r, h), column 32 eh + c32
    float *y_b = y + yv.base;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int i = i0 + acc_row(r, h);
        if (i < S) {
#pragma unroll
            for (int eh = 0; eh < MA_ET; eh++) y_b[(size_t)i * yv.ld + 32 * eh + c32] = yacc[eh][r];
        }
    }
}

// -> SPT_OK (indptr [S + 1] shared by the batch, as everywhere; x, y head layouts as dense_view)
int launch_spmm_mfma(const int32_t *indptr, const int32_t *indices, const float *values, const float *x,
                     float *y, int B, int S, int nnz, int x_heads, int y_heads, hipStream_t s) {
    const int bpb = (S + SN_ROWS - 1) / SN_ROWS;
    if ((long long)B * bpb > 0x7FFFFFFFll) return SPT_EUNSUP;
    const int lds = 2 * SN_IMG + SN_WAVES * sn_wave_bytes(S);
    SPT_HIP_TRY(hipFuncSetAttribute((const void *)spmm_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    lds));
    hipLaunchKernelGGL(spmm_mfma_kernel, dim3((unsigned)(B * bpb)), dim3(SN_THREADS), lds, s, indptr, indices,
                       values, x, y, S, nnz, x_heads, y_heads, bpb);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

// ---- launchers of this head dimension (the d_head 64 unit dispatches to spt::e128's) ----
template <typename T>
static int launch_forward_t(const unsigned long long *masks, const unsigned char *cells,
                            const unsigned char *pool, const unsigned *sat, const T *q, const T *k,
                            const T *v, T *y, float *row_sum, float *bounds, int batch_size, int S,
                            float scale, float clamp, int heads, int y_transposed, hipStream_t s) {
    const int bpb = (S + MA_ROWS - 1) / MA_ROWS;
    const size_t lds = mfma_forward_lds();
    const dim3 grid((unsigned)batch_size * bpb), block(MA_THREADS);
#define SPT_MF(YT)                                                                               \
    do {                                                                                         \
        SPT_HIP_TRY(hipFuncSetAttribute((const void *)attention_mfma_forward_kernel<T, YT>,      \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
        hipLaunchKernelGGL((attention_mfma_forward_kernel<T, YT>), grid, block, lds, s, masks, cells, pool, \
                           sat, q, k, v, y, row_sum, bounds, S, scale, clamp, heads, bpb);       \
    } while (0)
    if (y_transposed) SPT_MF(true); else SPT_MF(false);
#undef SPT_MF
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

template <typename T>
static int launch_backward_t(const unsigned long long *masks, const unsigned char *cells,
                             const unsigned char *cells_t, const unsigned char *pool,
                             const unsigned char *pool_t, const unsigned *sat, const T *q, const T *k,
                             const T *v, const T *y, const T *grad_y, const float *row_sum,
                             const float *bounds, float *delta, T *grad_q, T *grad_k, T *grad_v,
                             int batch_size, int S, float scale, float clamp, int heads,
                             int transposed, hipStream_t s) {
    const int bpb = (S + MA_ROWS - 1) / MA_ROWS;
    const dim3 grid((unsigned)batch_size * bpb), block(MA_THREADS);
    const size_t lds_r = 2 * MA_IMG + (size_t)MA_WAVES * MA_PARK,
                 lds_k = 2 * MK_IMG + (size_t)MA_WAVES * (MA_PARK + 2048);
    // 64 gradient columns per launch (the accumulators of 128 would not fit the registers
    // beside the operands' fragments): d_head 128 runs each kernel twice
#define SPT_MB(GT)                                                                              \
    do {                                                                                        \
        SPT_HIP_TRY(hipFuncSetAttribute(                                                        \
            (const void *)attention_mfma_backward_rows_kernel<T, GT>,                      \
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));                           \
        for (int half = 0; half < MA_ET / MR_QTILES; half++)                                    \
            hipLaunchKernelGGL((attention_mfma_backward_rows_kernel<T, GT>), grid, block, lds_r, \
                               s, masks, cells, pool, sat, q, k, v, grad_y, y, row_sum, bounds, \
                               grad_q, delta, S, scale, clamp, heads, bpb, half);               \
        if (MA_BH == 1) {                                                                       \
            SPT_KEYS(GT, 0, 0);                                                                 \
        } else {                                                                                \
            SPT_KEYS(GT, 1, 0);                                                                 \
            for (int half = 0; half < MA_ET / MK_KTILES; half++) SPT_KEYS(GT, 2, half);         \
        }                                                                                       \
    } while (0)
#define SPT_KEYS(GT, MODE, HALF)                                                                \
    do {                                                                                        \
        SPT_HIP_TRY(hipFuncSetAttribute(                                                        \
            (const void *)attention_mfma_backward_keys_kernel<T, GT, MODE>,                \
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k));                           \
        hipLaunchKernelGGL((attention_mfma_backward_keys_kernel<T, GT, MODE>), grid, block, \
                           lds_k, s, masks, cells_t, pool_t, sat, q, k, v, grad_y, row_sum,     \
                           bounds, delta, grad_k, grad_v, S, scale, clamp, heads, bpb, HALF);   \
    } while (0)
    if (transposed) SPT_MB(true);
    else SPT_MB(false);
#undef SPT_MB
#undef SPT_KEYS
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

// dtype: SPT_F32 (float tensors) or SPT_BF16 (bf16 storage; row_sum and delta stay fp32)
int launch_forward(const unsigned long long *masks, const unsigned char *cells,
                   const unsigned char *pool, const unsigned *sat, int dtype, const void *q,
                   const void *k, const void *v, void *y, float *row_sum, float *bounds,
                   int batch_size, int S, float scale, float clamp, int heads, int y_transposed,
                   hipStream_t s) {
    if (dtype == SPT_BF16)
        return launch_forward_t<bf16_t>(masks, cells, pool, sat, (const bf16_t *)q, (const bf16_t *)k,
                                        (const bf16_t *)v, (bf16_t *)y, row_sum, bounds, batch_size, S,
                                        scale, clamp, heads, y_transposed, s);
    return launch_forward_t<float>(masks, cells, pool, sat, (const float *)q, (const float *)k,
                                   (const float *)v, (float *)y, row_sum, bounds, batch_size, S, scale,
                                   clamp, heads, y_transposed, s);
}

int launch_backward(const unsigned long long *masks, const unsigned char *cells,
                    const unsigned char *cells_t, const unsigned char *pool,
                    const unsigned char *pool_t, const unsigned *sat, int dtype, const void *q,
                    const void *k, const void *v, const void *y, const void *grad_y,
                    const float *row_sum, const float *bounds, float *delta, void *grad_q,
                    void *grad_k, void *grad_v, int batch_size, int S, float scale, float clamp,
                    int heads, int transposed, hipStream_t s) {
    if (dtype == SPT_BF16)
        return launch_backward_t<bf16_t>(masks, cells, cells_t, pool, pool_t, sat, (const bf16_t *)q,
                                         (const bf16_t *)k, (const bf16_t *)v, (const bf16_t *)y,
                                         (const bf16_t *)grad_y, row_sum, bounds, delta,
                                         (bf16_t *)grad_q, (bf16_t *)grad_k, (bf16_t *)grad_v,
                                         batch_size, S, scale, clamp, heads, transposed, s);
    return launch_backward_t<float>(masks, cells, cells_t, pool, pool_t, sat, (const float *)q,
                                    (const float *)k, (const float *)v, (const float *)y,
                                    (const float *)grad_y, row_sum, bounds, delta, (float *)grad_q,
                                    (float *)grad_k, (float *)grad_v, batch_size, S, scale, clamp,
                                    heads, transposed, s);
}

#if MA_E_VALUE == 64
namespace e128 {
int launch_forward(const unsigned long long *masks, const unsigned char *cells,
                   const unsigned char *pool, const unsigned *sat, int dtype, const void *q,
                   const void *k, const void *v, void *y, float *row_sum, float *bounds,
                   int batch_size, int S, float scale, float clamp, int heads, int y_transposed,
                   hipStream_t s);
int launch_backward(const unsigned long long *masks, const unsigned char *cells,
                    const unsigned char *cells_t, const unsigned char *pool,
                    const unsigned char *pool_t, const unsigned *sat, int dtype, const void *q,
                    const void *k, const void *v, const void *y, const void *grad_y,
                    const float *row_sum, const float *bounds, float *delta, void *grad_q,
                    void *grad_k, void *grad_v, int batch_size, int S, float scale, float clamp,
                    int heads, int transposed, hipStream_t s);
}  // namespace e128

namespace e128 {
int launch_spmm_mfma(const int32_t *indptr, const int32_t *indices, const float *values, const float *x,
                     float *y, int B, int S, int nnz, int x_heads, int y_heads, hipStream_t s);
}
// The shapes of the matrix-core spmm (non-transposed); everything else is for the gather form
// (spmm.hip).  All S keys of a row stripe are multiplied: worth it from a density of 1 / 16
// (lookup: 1 / 8); a row's first 64 entries live in registers: mean row length <= 64.
bool spmm_mfma_takes(int B, int S, int E, int nnz) {
    if ((E != 64 && E != 128) || S < 64 || S > MA_MAXNT * MA_KT || B <= 0 || nnz <= 0) return false;
    if ((long long)nnz * 16 < (long long)S * S || (long long)nnz > 64ll * S) return false;
    // OPT-IN (SPT_SPMM_MFMA=1): measured at the configs[1] shape (256 slices x 512 x 64, 64 entries per
    // row; profiles/r04_ops_kernels.txt) this form takes 65.7 us against the gather form's 46.3 -- see
    // the kernel's header for where they go -- so the gather form stays the default.
    const char *on = getenv("SPT_SPMM_MFMA");
    return on && on[0] == '1';
}
int spmm_mfma_launch(const int32_t *indptr, const int32_t *indices, const float *values, const float *x,
                     float *y, int B, int S, int E, int nnz, int x_heads, int y_heads, hipStream_t s) {
    if (!spmm_mfma_takes(B, S, E, nnz)) return SPT_EUNSUP;
    return E == 64 ? launch_spmm_mfma(indptr, indices, values, x, y, B, S, nnz, x_heads, y_heads, s)
                   : e128::launch_spmm_mfma(indptr, indices, values, x, y, B, S, nnz, x_heads, y_heads, s);
}

static bool mfma_shape_ok(int S, int E, int nnz) {
    if ((E != 64 && E != 128) || S <= 0 || nnz <= 0 || nnz % S != 0) return false;
    const int Z = nnz / S;
    return Z <= MA_MAXZ && Z % 4 == 0 && S <= MA_MAXNT * MA_KT;
}
// Workspace: 256-byte header (word 0: layout, word 1: flags -- bit 0: a compact layout was
// given a pattern that repeats a column outside key tile 0) | masks, sat | cells | cells_t
// [| pool | pool_t in the compact layout].
struct TileSet {
    unsigned long long *masks;
    unsigned *sat;
    unsigned char *cells, *cells_t, *pool, *pool_t;
};
constexpr size_t MA_HEADER = 256;
static size_t tile_cells_bytes(int B, int S, int layout) {
    const size_t RT = (S + MA_WROWS - 1) / MA_WROWS;
    return (size_t)B * tri(RT) * (layout == SPT_TILES_COMPACT ? 128 : MA_CELLS);
}
static size_t tile_pool_bytes(int B, int S, int layout) {
    const size_t RT = (S + MA_WROWS - 1) / MA_WROWS;
    return layout == SPT_TILES_COMPACT ? (size_t)B * RT * MA_CELLS : 0;
}
static size_t tile_mask_bytes(int B, int S) {
    const size_t RT = (S + MA_WROWS - 1) / MA_WROWS;
    return (((size_t)B * RT * (2 * sizeof(unsigned long long) + sizeof(unsigned))) + 255) & ~(size_t)255;
}
static TileSet carve_tiles(void *ws, int B, int S, int layout) {
    TileSet t;
    char *p = static_cast<char *>(ws) + MA_HEADER;
    t.masks = reinterpret_cast<unsigned long long *>(p);
    t.sat = reinterpret_cast<unsigned *>(t.masks + 2 * (size_t)B * ((S + MA_WROWS - 1) / MA_WROWS));
    t.cells = reinterpret_cast<unsigned char *>(p + tile_mask_bytes(B, S));
    t.cells_t = t.cells + tile_cells_bytes(B, S, layout);
    t.pool = t.pool_t = nullptr;
    if (layout == SPT_TILES_COMPACT) {
        t.pool = t.cells_t + tile_cells_bytes(B, S, layout);
        t.pool_t = t.pool + tile_pool_bytes(B, S, layout);
    }
    return t;
}
#endif

MA_NS_CLOSE

#if MA_E_VALUE == 64
extern "C" int spt_attention_mfma_supported(int seq_length, int d_head, int nnz) {
    return spt::mfma_shape_ok(seq_length, d_head, nnz) ? 1 : 0;
}

extern "C" int64_t spt_attention_mfma_tiles_bytes(int batch_size, int seq_length, int nnz,
                                                  int layout) {
    using namespace spt;
    if (batch_size <= 0 || !mfma_shape_ok(seq_length, MA_E, nnz)) return 0;
    if (layout != SPT_TILES_FULL && layout != SPT_TILES_COMPACT) return 0;
    return (int64_t)(MA_HEADER + tile_mask_bytes(batch_size, seq_length) +
                     2 * tile_cells_bytes(batch_size, seq_length, layout) +
                     2 * tile_pool_bytes(batch_size, seq_length, layout));
}

extern "C" int spt_attention_mfma_prepare(const int32_t *indices, void *tiles, int batch_size,
                                          int seq_length, int nnz, int layout, void *stream) {
    using namespace spt;
    if (!indices || !tiles || batch_size <= 0) return SPT_EINVAL;
    if (layout != SPT_TILES_FULL && layout != SPT_TILES_COMPACT) return SPT_EINVAL;
    if (!mfma_shape_ok(seq_length, MA_E, nnz)) return SPT_EUNSUP;
    const int S = seq_length, Z = nnz / S, NT = (S + MA_KT - 1) / MA_KT;
    const int RT = (S + MA_WROWS - 1) / MA_WROWS, total = batch_size * RT;
    const TileSet ts = carve_tiles(tiles, batch_size, S, layout);
    const size_t lds = (size_t)MB_WAVES * prepare_lds_per_wave();
    const dim3 grid((total + MB_WAVES - 1) / MB_WAVES), block(MB_WAVES * SPT_WAVE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    SPT_ZERO_WORDS(tiles, MA_HEADER / 4, s);
    SPT_LAUNCH_CHECK();
    // (at S <= 512 -- two chunks of key tiles -- the byte-count kernel is the faster one: 22.8 against
    // 26.0 us at the configs[2] shape; SPT_CELL_TILES_COUNTS / _BITS force either for A/B runs)
    const bool bits = layout == SPT_TILES_COMPACT && !getenv("SPT_CELL_TILES_COUNTS") &&
                      (NT > 2 * MB_CHUNK || getenv("SPT_CELL_TILES_BITS"));
    if (bits) {
        const size_t lds_bits = (size_t)MB_WAVES * cell_bits_lds_per_wave(NT);
        SPT_HIP_TRY(hipFuncSetAttribute((const void *)attention_cell_bits_kernel,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bits));
        hipLaunchKernelGGL(attention_cell_bits_kernel, grid, block, lds_bits, s, indices, ts.masks,
                           ts.cells, ts.cells_t, ts.pool, ts.pool_t, ts.sat, S, Z, NT, RT, total);
    } else if (Z > 255) {
        SPT_HIP_TRY(hipFuncSetAttribute((const void *)attention_cell_tiles_kernel<true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(attention_cell_tiles_kernel<true>, grid, block, lds, s, indices,
                           ts.masks, ts.cells, ts.cells_t, ts.pool, ts.pool_t, ts.sat, S, Z, NT, RT,
                           total);
    } else {
        SPT_HIP_TRY(hipFuncSetAttribute((const void *)attention_cell_tiles_kernel<false>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(attention_cell_tiles_kernel<false>, grid, block, lds, s, indices,
                           ts.masks, ts.cells, ts.cells_t, ts.pool, ts.pool_t, ts.sat, S, Z, NT, RT,
                           total);
    }
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

static int mfma_forward_any(const void *tiles, int layout, int dtype, const void *q,
                            const void *k, const void *v, void *y, float *row_sum, float *bounds,
                            int batch_size, int seq_length, int d_head, int nnz, float scale,
                            float clamp, int heads, int y_transposed, void *stream) {
    using namespace spt;
    if (!tiles || !q || !k || !v || !y || !row_sum) return SPT_EINVAL;
    if (batch_size <= 0 || heads < 0) return SPT_EINVAL;
    if (layout != SPT_TILES_FULL && layout != SPT_TILES_COMPACT) return SPT_EINVAL;
    if (!mfma_shape_ok(seq_length, d_head, nnz)) return SPT_EUNSUP;
    if (heads > 0 && batch_size % heads != 0) return SPT_ESHAPE;
    const TileSet ts = carve_tiles(const_cast<void *>(tiles), batch_size, seq_length, layout);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return d_head == 64
               ? launch_forward(ts.masks, ts.cells, ts.pool, ts.sat, dtype, q, k, v, y, row_sum, bounds,
                                batch_size, seq_length, scale, clamp, heads, y_transposed, s)
               : e128::launch_forward(ts.masks, ts.cells, ts.pool, ts.sat, dtype, q, k, v, y, row_sum,
                                      bounds, batch_size, seq_length, scale, clamp, heads,
                                      y_transposed, s);
}

static int mfma_backward_any(const void *tiles, int layout, int dtype, const void *q,
                             const void *k, const void *v, const void *y, const void *grad_y,
                             const float *row_sum, const float *bounds, float *delta, void *grad_q,
                             void *grad_k, void *grad_v, int batch_size, int seq_length, int d_head,
                             int nnz, float scale, float clamp, int heads, int transposed,
                             void *stream) {
    using namespace spt;
    if (!tiles || !q || !k || !v || !y || !grad_y || !row_sum || !delta || !grad_q || !grad_k ||
        !grad_v)
        return SPT_EINVAL;
    if (batch_size <= 0 || heads < 0) return SPT_EINVAL;
    if (layout != SPT_TILES_FULL && layout != SPT_TILES_COMPACT) return SPT_EINVAL;
    if (!mfma_shape_ok(seq_length, d_head, nnz)) return SPT_EUNSUP;
    if (heads > 0 && batch_size % heads != 0) return SPT_ESHAPE;
    if (transposed && (seq_length & 3)) return SPT_EUNSUP;
    const TileSet ts = carve_tiles(const_cast<void *>(tiles), batch_size, seq_length, layout);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return d_head == 64
               ? launch_backward(ts.masks, ts.cells, ts.cells_t, ts.pool, ts.pool_t, ts.sat, dtype, q,
                                 k, v, y, grad_y, row_sum, bounds, delta, grad_q, grad_k, grad_v,
                                 batch_size, seq_length, scale, clamp, heads, transposed, s)
               : e128::launch_backward(ts.masks, ts.cells, ts.cells_t, ts.pool, ts.pool_t, ts.sat,
                                       dtype, q, k, v, y, grad_y, row_sum, bounds, delta, grad_q,
                                       grad_k, grad_v, batch_size, seq_length, scale, clamp, heads,
                                       transposed, s);
}

extern "C" int spt_attention_mfma_forward(const void *tiles, int layout, const float *q,
                                          const float *k, const float *v, float *y,
                                          float *row_sum, float *bounds, int batch_size,
                                          int seq_length, int d_head, int nnz, float scale,
                                          float clamp, int heads, int y_transposed, void *stream) {
    return mfma_forward_any(tiles, layout, SPT_F32, q, k, v, y, row_sum, bounds, batch_size,
                            seq_length, d_head, nnz, scale, clamp, heads, y_transposed, stream);
}

extern "C" int spt_attention_mfma_bounds_floats(int batch_size) {
    return batch_size > 0 ? batch_size * spt::MA_BOUND_SLOTS * 2 : 0;
}

extern "C" int spt_attention_mfma_backward(const void *tiles, int layout, const float *q,
                                           const float *k, const float *v, const float *y,
                                           const float *grad_y, const float *row_sum,
                                           const float *bounds, float *delta, float *grad_q,
                                           float *grad_k, float *grad_v, int batch_size,
                                           int seq_length, int d_head, int nnz, float scale,
                                           float clamp, int heads, int transposed, void *stream) {
    return mfma_backward_any(tiles, layout, SPT_F32, q, k, v, y, grad_y, row_sum, bounds, delta,
                             grad_q, grad_k, grad_v, batch_size, seq_length, d_head, nnz, scale,
                             clamp, heads, transposed, stream);
}

extern "C" int spt_attention_mfma_forward_bf16(const void *tiles, int layout, const uint16_t *q,
                                               const uint16_t *k, const uint16_t *v, uint16_t *y,
                                               float *row_sum, float *bounds, int batch_size,
                                               int seq_length, int d_head, int nnz, float scale,
                                               float clamp, int heads, int y_transposed,
                                               void *stream) {
    return mfma_forward_any(tiles, layout, SPT_BF16, q, k, v, y, row_sum, bounds, batch_size,
                            seq_length, d_head, nnz, scale, clamp, heads, y_transposed, stream);
}

extern "C" int spt_attention_mfma_backward_bf16(const void *tiles, int layout, const uint16_t *q,
                                                const uint16_t *k, const uint16_t *v,
                                                const uint16_t *y, const uint16_t *grad_y,
                                                const float *row_sum, const float *bounds,
                                                float *delta, uint16_t *grad_q, uint16_t *grad_k,
                                                uint16_t *grad_v, int batch_size, int seq_length,
                                                int d_head, int nnz, float scale, float clamp,
                                                int heads, int transposed, void *stream) {
    return mfma_backward_any(tiles, layout, SPT_BF16, q, k, v, y, grad_y, row_sum, bounds, delta,
                             grad_q, grad_k, grad_v, batch_size, seq_length, d_head, nnz, scale,
                             clamp, heads, transposed, stream);
}
#endif
