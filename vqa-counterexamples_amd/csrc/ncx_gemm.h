// ncx_gemm.h -- segmented fp32 MFMA GEMM engine for gfx950 (MI355X).
//
// One kernel template computes   C[M,N] = sum over "pairs" p of  A_p . B_p^T-ish   where every operand
// is described by an XDesc: a logical row-major matrix x(r, c) that is never materialised:
//
//   X_PLAIN       x(r,c) = base[r*ld + c]
//   X_GATHER      x(r,c) = base[idx[r]*ld + c]                       (feature-table / embedding row gather)
//   X_GATHER_MUL  x(r,c) = base[idx[r]*ld + c] * base[idx2[r]*ld + c] (v_orig * v_other, cx.py:296)
//   X_SOFTMAX     x(r,c) = exp2(base[r*ld + c]*log2(e) - lse2[r])     (softmax(a_knns), cx.py:281;
//                 lse2[r] = log2(sum_c exp(x - max)) + max*log2(e), one FMA + one v_exp_f32 per element)
//
// This is how the reference's torch.cat of ten segments (vqa/models/cx.py:309-320) is consumed without
// building it: the first Linear layer is a CHAIN of (segment of X, column slice of linear_1.weight) pairs
// accumulated into one MFMA accumulator, and its weight gradient is a GROUP of independent
// (dpre^T . segment) products that share the dpre operand.
//
// Operand forms.  An operand tile lives in LDS as a row-major copy of an x sub-block and is either
//   "col-is-k": the reduction index runs along the COLUMNS of x  (tile [rows=out index][32 k], pitch 34)
//   "row-is-k": the reduction index runs along the ROWS of x     (tile [32 k][cols=out index], pitch = 8 mod 32)
// so NT (forward: X.W^T), TN (weight grad: D^T.X) and NN (input grad: D.W) products all use the same
// loader (global loads are always 16 B along the contiguous c axis) and the same conflict-free
// ds_read_b32 fragment reads for v_mfma_f32_16x16x4_f32.
//
// MFMA: __builtin_amdgcn_mfma_f32_16x16x4f32 (exact fp32, 256 FLOP/clk/CU; peak 157.3 TFLOP/s).
// A: lane l holds A[i = l&15][k = l>>4];  B: B[k = l>>4][j = l&15];  C/D: col = l&15, row = 4*(l>>4)+reg.
// 256 threads = 4 waves in a 2x2 layout; wave tile = (BM/2) x (BN/2) in 16x16 blocks.
// Pipeline: register-staged double-buffered LDS, one barrier per 32-deep K-step: loads for step t+1 are
// issued before the MFMAs of step t and written to the other LDS buffer after them.
#pragma once
#include "ncx_common.h"
#include <type_traits>

namespace ncx {

enum XKind : int { X_PLAIN = 0, X_GATHER = 1, X_GATHER_MUL = 2, X_SOFTMAX = 3 };

struct XDesc {
    const float* base;
    const int*   idx;
    const int*   idx2;
    const float* mx;     // X_SOFTMAX: lse2 per row
    const float* inv;    // unused (kept for layout)
    long long    ld;
    int kind;
    int rows;
    int cols;
    int idx_stride;      // idx[r * idx_stride] (1 = dense; K+1 picks column 0 of img_idx[B][K+1])
};

static inline XDesc x_plain(const float* base, long long ld, int rows, int cols) {
    XDesc d{}; d.base = base; d.ld = ld; d.kind = X_PLAIN; d.rows = rows; d.cols = cols; d.idx_stride = 1; return d;
}
static inline XDesc x_gather(const float* table, long long ld, const int* idx, int rows, int cols) {
    XDesc d{}; d.base = table; d.ld = ld; d.idx = idx; d.kind = X_GATHER; d.rows = rows; d.cols = cols; d.idx_stride = 1; return d;
}
static inline XDesc x_gather_strided(const float* table, long long ld, const int* idx, int idx_stride, int rows, int cols) {
    XDesc d = x_gather(table, ld, idx, rows, cols); d.idx_stride = idx_stride; return d;
}
static inline XDesc x_gather_mul(const float* table, long long ld, const int* idx, const int* idx2, int rows, int cols) {
    XDesc d{}; d.base = table; d.ld = ld; d.idx = idx; d.idx2 = idx2; d.kind = X_GATHER_MUL; d.rows = rows; d.cols = cols; d.idx_stride = 1; return d;
}
static inline XDesc x_softmax(const float* logits, long long ld, const float* mx, const float* inv, int rows, int cols) {
    XDesc d{}; d.base = logits; d.ld = ld; d.mx = mx; d.inv = inv; d.kind = X_SOFTMAX; d.rows = rows; d.cols = cols; d.idx_stride = 1; return d;
}

constexpr int NCX_MAX_SEG = 10;
// 64x64 tiles of the row-reduction (TN) form run three workgroups per CU: they serve short reductions (the
// answer_embedding gradient: 1216 tiles x 16 k-steps), where a third resident workgroup hides the per-workgroup prologue
// and epilogue and 768 slots hold the launch in two rounds instead of three (measured 89 -> 76 us).
#ifndef NCX_OCC64_TN
#define NCX_OCC64_TN 3
#endif
constexpr int GEMM_BK = 32;

enum GemmMode : int { MODE_CHAIN = 0, MODE_GROUP = 1 };

// Epilogue: v = acc (+ rowadd[(r/rowdiv)][n]) (+ bias[n]); relu; dropout; gate; store.
struct EpiArgs {
    const float* rowadd; long long ld_rowadd; int rowdiv;
    const float* bias;
    int relu;                    // activation: 0 none, 1 relu, 2 tanh
    int dropout;                 // 0 none, 1 counter-based generator, 2 explicit keep mask
    float drop_p, drop_scale;
    unsigned seed_lo, seed_hi, layer;
    const float* keep_mask; long long ld_mask;
    const float* gate; long long ld_gate; float gate_scale;   // v *= gate[r][n] > 0 ? gate_scale : 0
    // FOLD kernels only (MUTAN fusion, fusion.py:96-115): after every chained pair s,
    //   zacc += (acc + fold_bias[s*N + n]) * fold_mul[(r / fold_div)*ld_fold + s*N + n];  acc = 0;   output = zacc
    const float* fold_mul; const float* fold_bias; long long ld_fold; int fold_div;
    // split output rows: row r = g*b + j goes to out0[b] when j == 0, else to out[(g-1)*b + j-1]   (0: off)
    int rowsplit_g; float* out0; long long ldo0;
};

struct FixupArgs;
struct GemmArgs {
    XDesc a[NCX_MAX_SEG];        // CHAIN: a[s] of pair s;  GROUP: a[i] of problem i
    XDesc b[NCX_MAX_SEG];
    int   klen[NCX_MAX_SEG];     // reduction extent of pair/problem i
    float* out[NCX_MAX_SEG];     // CHAIN: out[0];  GROUP: one per problem
    long long ldo[NCX_MAX_SEG];
    int   n_cols[NCX_MAX_SEG];   // N of problem i (CHAIN: n_cols[0])
    int   split[NCX_MAX_SEG];    // aligned k-chunks per output tile of problem i (CHAIN: split[0]); <= 1: none
    int   tile0[NCX_MAX_SEG + 1];// first linear tile id of problem i           (filled by the launcher)
    int   wg0[NCX_MAX_SEG + 1];  // first workgroup id of problem i = sum tiles*split (filled by the launcher)
    int   mode, nseg, M, pad_;
    int   total_wgs;             // work items (filled by the launcher); the grid may be smaller: persistent loop
    struct FixupArgs* defer_fix; // host only: when set, a split launch hands its fix-up back instead of launching it (merged fix-ups)
    int   pad2_;
    float* slab;                 // [workgroups][BM*BN] partial tiles of the split problems
    EpiArgs epi;
};

// ------------------------------------------------------------------------------------------------
#if defined(__HIPCC__)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Row-is-k LDS pitch.  Lane group lk reads k-rows 8t + 2*lk + e, so the two groups of a 32-lane half sit TWO
// rows apart.  With the interleaved block mapping (below) a lane reads EXT/32 consecutive floats:
//   ds_read_b128 (EXT = 128): 16 lanes cover all 64 banks; the hardware's mixed 16-lane groups stay disjoint
//                 when 2*pitch = 0 mod 64  -> pitch = 128 (no padding)
//   ds_read_b64  (EXT = 64):  a 32-lane half = two 128-byte runs two rows apart -> 2*pitch = 32 mod 64 -> pitch = 80
//   ds_read_b32  (other EXT): 2*pitch = 16 mod 32 -> pitch = 8 mod 32
__host__ __device__ constexpr int pitch_rowk(int c) {
    return c == 128 ? 128 : c == 64 ? 80 : ((c - 8 + 31) / 32) * 32 + 8;
}

// 4 consecutive columns of one row (always a valid row pointer; cols >= 4).  The load is UNCONDITIONAL and
// identical for interior and edge tiles: the 16-byte window is slid left to stay inside [0, cols)
// (a per-lane guard around a load makes hipcc put every load in its own basic block, and two alternative
// load paths merge through register copies that force an early s_waitcnt).  Edge tiles repair the window
// afterwards with fix_window (pure VALU, at LDS-store time).
__device__ __forceinline__ f32x4 load_window(const float* p, int c, int cols) {
    return *(const f32x4u*)(p + min(c, cols - 4));
}
// w = load_window(p, c, cols)  ->  {x[c], x[c+1], x[c+2], x[c+3]} with zeros at and beyond cols
__device__ __forceinline__ f32x4 fix_window(f32x4 w, int c, int cols) {
    const int sh = c - min(c, cols - 4);          // 0 for interior windows
    f32x4 r;
    r[0] = sh == 0 ? w[0] : sh == 1 ? w[1] : sh == 2 ? w[2] : w[3];
    r[1] = sh == 0 ? w[1] : sh == 1 ? w[2] : w[3];
    r[2] = sh == 0 ? w[2] : w[3];
    r[3] = w[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = (c + j < cols) ? r[j] : 0.f;
    return r;
}
// guarded form used by the bandwidth kernels (zero fill, any cols >= 1)
__device__ __forceinline__ f32x4 load4(const float* p, int c, int cols) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c + 3 < cols) {
        v = *(const f32x4u*)(p + c);
    } else {
        if (c     < cols) v[0] = p[c];
        if (c + 1 < cols) v[1] = p[c + 1];
        if (c + 2 < cols) v[2] = p[c + 2];
    }
    return v;
}

// Per-thread loader of one operand tile.
//   COLK  (col-is-k): tile [EXT rows][32 k-cols]; thread owns column quad 4*(tid&7) of rows (tid>>3) + 32*i
//   !COLK (row-is-k): tile [32 k-rows][EXT cols]; thread owns row (tid>>3), column quads 4*(tid&7) + 32*i
// i < NV = EXT/32 either way, so a wave-instruction always covers 8 rows x 128 contiguous bytes.
template <bool COLK, int EXT>
struct OpLoader {
    static constexpr int NV = EXT / 32;
    static constexpr int NR = COLK ? NV : 1;             // distinct rows per thread
    static constexpr int PITCH = COLK ? GEMM_BK + 4 : pitch_rowk(EXT);
    static constexpr int ELEMS = (COLK ? EXT : GEMM_BK) * PITCH;
    static_assert(EXT % 32 == 0, "tile extent");

    f32x4 v0[NV], v1[NV];
    const float* p0[NR];
    const float* p1[NR];
    float mx[NR];                // X_SOFTMAX: base-2 log-sum-exp of the row
    unsigned rowmask;            // bit i: row i of this thread is inside the matrix
    int kind, cols, rows, o0;    // o0: first row (COLK) / first column (!COLK) of the tile in x-space
    int nidx0, nidx1;            // !COLK: gather indices prefetched for the NEXT k-step
    bool full_o;                 // !COLK: the tile's column range is fully inside the matrix
    bool full_k;                 // COLK: set per k-step

    __device__ __forceinline__ void setup(const XDesc& d, int origin, int tid) {
        kind = d.kind; cols = d.cols; rows = d.rows; o0 = origin;
        full_o = origin + EXT <= d.cols;
        if (COLK) {
            rowmask = 0;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = origin + i * 32 + (tid >> 3);
                const int rc = min(r, d.rows - 1);
                rowmask |= (r < d.rows ? 1u : 0u) << i;
                if (kind == X_GATHER) {
                    p0[i] = d.base + (long long)d.idx[(long long)(rc) * d.idx_stride] * d.ld;
                } else if (kind == X_GATHER_MUL) {
                    p0[i] = d.base + (long long)d.idx[(long long)(rc) * d.idx_stride] * d.ld;
                    p1[i] = d.base + (long long)d.idx2[(long long)(rc) * d.idx_stride] * d.ld;
                } else {
                    p0[i] = d.base + (long long)rc * d.ld;
                    if (kind == X_SOFTMAX) { mx[i] = d.mx[rc]; }
                }
            }
        }
    }
    // !COLK: fetch the gather indices of k-row block `k` (consumed by the following issue()).
    __device__ __forceinline__ void prefetch_rows(const XDesc& d, int k, int tid) {
        if (!COLK) {
            const int rc = min(k + (tid >> 3), d.rows - 1);
            if (kind == X_GATHER || kind == X_GATHER_MUL) nidx0 = d.idx[(long long)(rc) * d.idx_stride];
            if (kind == X_GATHER_MUL) nidx1 = d.idx2[(long long)(rc) * d.idx_stride];
        }
    }
    __device__ __forceinline__ void issue(const XDesc& d, int k, int tid) {
        int c0, cstride;
        if (COLK) {
            c0 = k + 4 * (tid & 7); cstride = 0;
            full_k = k + GEMM_BK <= cols;
        } else {
            const int r = k + (tid >> 3);
            const int rc = min(r, rows - 1);
            rowmask = r < rows ? 1u : 0u;
            if (kind == X_GATHER) {
                p0[0] = d.base + (long long)nidx0 * d.ld;
            } else if (kind == X_GATHER_MUL) {
                p0[0] = d.base + (long long)nidx0 * d.ld;
                p1[0] = d.base + (long long)nidx1 * d.ld;
            } else {
                p0[0] = d.base + (long long)rc * d.ld;
                if (kind == X_SOFTMAX) { mx[0] = d.mx[rc]; }
            }
            c0 = o0 + 4 * (tid & 7); cstride = 32;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) v0[i] = load_window(p0[COLK ? i : 0], c0 + cstride * i, cols);
        if (kind == X_GATHER_MUL) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v1[i] = load_window(p1[COLK ? i : 0], c0 + cstride * i, cols);
        }
    }

    template <int KIND, bool FULL>
    __device__ __forceinline__ void store_t(float* lds, int k, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int ri = COLK ? i : 0;
            f32x4 v = v0[i];
            if (KIND == X_GATHER_MUL) {
                v = v * v1[i];
            } else if (KIND == X_SOFTMAX) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[j], 1.44269504088896341f, -mx[ri]));
            }
            if (!FULL) {
                const int c = COLK ? k + 4 * (tid & 7) : o0 + 4 * (tid & 7) + 32 * i;
                v = fix_window(v, c, cols);
            }
            const bool rv = (rowmask >> ri) & 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = rv ? v[j] : 0.f;
            float* dst = COLK ? lds + (i * 32 + (tid >> 3)) * PITCH + 4 * (tid & 7)
                              : lds + (tid >> 3) * PITCH + 4 * (tid & 7) + 32 * i;
            *(f32x4*)dst = v;
        }
    }

    // ---- interior fast path: no bounds handling, kind compile-time, straight-line -------------------
    // KIND: COLK operands have their row pointers resolved in setup(), so X_GATHER == X_PLAIN there.
    template <int KIND, bool MASK = false>
    __device__ __forceinline__ void issue_fast(const XDesc& d, int k, int tid, int idx_ahead = GEMM_BK) {
#ifdef NCX_ABLATE_LOADS          // timing experiment only: keep the registers of the first tile
        return;
#endif
        if (COLK) {
            const int c = k + 4 * (tid & 7);
#pragma unroll
            for (int i = 0; i < NV; ++i) v0[i] = *(const f32x4u*)(p0[i] + c);
            if (KIND == X_GATHER_MUL) {
#pragma unroll
                for (int i = 0; i < NV; ++i) v1[i] = *(const f32x4u*)(p1[i] + c);
            }
        } else {
            const int r = k + (tid >> 3);
            const int c = o0 + 4 * (tid & 7);
            if (KIND == X_GATHER || KIND == X_GATHER_MUL) p0[0] = d.base + (long long)nidx0 * d.ld;
            else p0[0] = d.base + (long long)r * d.ld;
            if (KIND == X_GATHER_MUL) p1[0] = d.base + (long long)nidx1 * d.ld;
            if (KIND == X_SOFTMAX) { mx[0] = d.mx[r]; }
            // MASK (column-edge tile): windows slid left to stay inside the row, repaired in store_fast
#pragma unroll
            for (int i = 0; i < NV; ++i) v0[i] = MASK ? load_window(p0[0], c + 32 * i, cols) : *(const f32x4u*)(p0[0] + c + 32 * i);
            if (KIND == X_GATHER_MUL) {
#pragma unroll
                for (int i = 0; i < NV; ++i) v1[i] = MASK ? load_window(p1[0], c + 32 * i, cols) : *(const f32x4u*)(p1[0] + c + 32 * i);
            }
            // gather indices of this loader's next k-step
            if (KIND == X_GATHER || KIND == X_GATHER_MUL) nidx0 = d.idx[(long long)(min(r + idx_ahead, rows - 1)) * d.idx_stride];
            if (KIND == X_GATHER_MUL) nidx1 = d.idx2[(long long)(min(r + idx_ahead, rows - 1)) * d.idx_stride];
        }
    }
    template <int KIND, bool MASK = false>
    __device__ __forceinline__ void store_fast(float* lds, int tid, int i0 = 0, int i1 = NV) const {
#ifdef NCX_ABLATE_STORES         // timing experiment only
        return;
#endif
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (i < i0 || i >= i1) continue;
            const int ri = COLK ? i : 0;
            f32x4 v = v0[i];
            if (KIND == X_GATHER_MUL) {
                v = v * v1[i];
            } else if (KIND == X_SOFTMAX) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[j], 1.44269504088896341f, -mx[ri]));
            }
            if (!COLK && MASK) v = fix_window(v, o0 + 4 * (tid & 7) + 32 * i, cols);   // column-edge tile
            if (COLK && MASK) {             // rows beyond the matrix (M / N edge tiles): clamped pointers, zeroed here
                const bool rv = (rowmask >> i) & 1u;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = rv ? v[j] : 0.f;
            }
            float* dst = COLK ? lds + (i * 32 + (tid >> 3)) * PITCH + 4 * (tid & 7)
                              : lds + (tid >> 3) * PITCH + 4 * (tid & 7) + 32 * i;
            *(f32x4*)dst = v;
        }
    }
    // after a run of fast steps the generic path needs its bookkeeping back
    __device__ __forceinline__ void resync_after_fast() { rowmask = COLK ? rowmask : 1u; full_k = true; }

    // transform + zero padding + write into the LDS tile; k = k position the registers were issued for
    __device__ __forceinline__ void store(float* lds, int k, int tid) const {
        const bool full = COLK ? full_k : full_o;
        if (full) {
            if (kind == X_GATHER_MUL) store_t<X_GATHER_MUL, true>(lds, k, tid);
            else if (kind == X_SOFTMAX) store_t<X_SOFTMAX, true>(lds, k, tid);
            else store_t<X_PLAIN, true>(lds, k, tid);
        } else {
            if (kind == X_GATHER_MUL) store_t<X_GATHER_MUL, false>(lds, k, tid);
            else if (kind == X_SOFTMAX) store_t<X_SOFTMAX, false>(lds, k, tid);
            else store_t<X_PLAIN, false>(lds, k, tid);
        }
    }
};

// XCD-aware workgroup order inside one problem.  Workgroups are dealt round-robin over the 8 XCDs (id % 8), each
// with a private L2.  The tiles along the SHORT tile dimension ("minor": the 2 N-tiles of a forward row block, the 4
// M-tiles of a weight-gradient column block) all stream the same expensive operand rows (gathered features,
// logits), so they are placed 8 ids apart: same XCD, dispatched together -> one HBM/MALL fetch instead of one
// per sharer.  Within a group of 8 consecutive ids the major tile varies first (same k-chunk z): those share the
// cheap operand (weights / dpre chunk).  Pure permutation: any placement is correct, this one is fast.
struct WgMap {
    int tiles_m, tiles_n, S;
    __host__ __device__ bool minor_n() const { return tiles_n <= tiles_m; }
    __host__ __device__ int mc() const { return minor_n() ? tiles_n : tiles_m; }       // minor count
    __host__ __device__ int jt() const { return minor_n() ? tiles_m : tiles_n; }       // major tiles
    // Split problems whose k-chunk count divides 8 or is a multiple of it use the "chunk per XCD" layout instead:
    // workgroup id -> XCD x = id % 8 owns k-chunks z = x (mod 8): the chunk's slice of the SHARED operand (dpre in the
    // weight-gradient launch: every output tile reads it) stays in that XCD's L2 for the whole launch and is fetched
    // from HBM once, instead of all chunks cycling through every L2 (measured: 781 MB fetched per launch against
    // 430 MB of operands).  Inside an XCD the minor tiles of one major tile still sit next to each other in time.
    // S < 8: 8/S XCDs share a chunk and take major tiles round-robin; ids whose major tile does not exist are padding.
    __host__ __device__ bool zx() const { return S > 1 && (S % 8 == 0 || 8 % S == 0); }
    __host__ __device__ int count() const {
        if (!zx()) return tiles_m * tiles_n * S;
        if (S % 8 == 0) return S * tiles_m * tiles_n;
        const int g = 8 / S;
        return 8 * mc() * ((jt() + g - 1) / g);
    }
    // local workgroup id -> (tm, tn, z); false: padding id (no work)
    __host__ __device__ bool decode(int lw, int& tm, int& tn, int& z) const {
        if (zx()) {
            const int m = mc(), x = lw & 7;
            int j = lw >> 3, mi, major;
            if (S % 8 == 0) { const int c = S / 8; z = x + 8 * (j % c); j /= c; mi = j % m; major = j / m; }
            else { const int g = 8 / S; z = x % S; mi = j % m; major = (j / m) * g + x / S; }
            tm = minor_n() ? major : mi; tn = minor_n() ? mi : major;
            return major < jt();
        }
        const int m = mc(), J = jt() * S, full = (J / 8) * 8;
        int mi, q;
        if (lw < full * m) { const int blk = lw / (8 * m), rem = lw - blk * 8 * m; mi = rem / 8; q = blk * 8 + (rem & 7); }
        else { const int t = lw - full * m, r = J - full; mi = t / r; q = full + t - mi * r; }
        z = q / jt();
        const int major = q - z * jt();
        tm = minor_n() ? major : mi; tn = minor_n() ? mi : major;
        return true;
    }
    // (tm, tn, z) -> local workgroup id (slab slot of the partial tile)
    __host__ __device__ int encode(int tm, int tn, int z) const {
        const int m = mc();
        const int mi = minor_n() ? tn : tm, major = minor_n() ? tm : tn;
        if (zx()) {
            if (S % 8 == 0) { const int c = S / 8; return ((major * m + mi) * c + (z >> 3)) * 8 + (z & 7); }
            const int g = 8 / S;
            return ((major / g) * m + mi) * 8 + (major % g) * S + z;
        }
        const int J = jt() * S, full = (J / 8) * 8;
        const int q = z * jt() + major;
        if (q < full) return (q / 8) * 8 * m + mi * 8 + (q & 7);
        return full * m + mi * (J - full) + (q - full);
    }
};

template <int BM, int BN, bool A_COLK, bool B_COLK>
struct GemmCfg {
    static constexpr int BK = GEMM_BK;
    typedef OpLoader<A_COLK, BM> ALoad;
    typedef OpLoader<B_COLK, BN> BLoad;
    static constexpr int PA = ALoad::PITCH, PB = BLoad::PITCH;
    static constexpr int A_ELEMS = ALoad::ELEMS, B_ELEMS = BLoad::ELEMS;
    static constexpr int LDS_BYTES = 2 * (A_ELEMS + B_ELEMS) * 4;
    static constexpr int WM = BM / 32, WN = BN / 32;   // 16x16 blocks per wave (2x2 waves)
};

__device__ __forceinline__ float apply_epilogue(const EpiArgs& e, float v, int r, int n, int ncols_total) {
    if (e.rowadd) v += e.rowadd[(long long)(r / e.rowdiv) * e.ld_rowadd + n];
    if (e.bias) v += e.bias[n];
    if (e.relu == 1) v = v > 0.f ? v : 0.f;
    else if (e.relu == 2) v = tanhf(v);                      // activation_v / activation_q = tanh (fusion.py:84-93)
    if (e.dropout == 1) {
        v = dropout_keep(e.seed_lo, e.seed_hi, e.layer, (unsigned long long)r * (unsigned)ncols_total + (unsigned)n, e.drop_p)
                ? v * e.drop_scale : 0.f;
    } else if (e.dropout == 2) {
        v = e.keep_mask[(long long)r * e.ld_mask + n] != 0.f ? v * e.drop_scale : 0.f;
    }
    if (e.gate) v = e.gate[(long long)r * e.ld_gate + n] > 0.f ? v * e.gate_scale : 0.f;
    return v;
}

// Big tiles get the whole 512-entry register file (one workgroup per CU): at 2 waves per SIMD the 128x128 and
// 96x128 instantiations spill 90-260 VGPRs.
template <int BM, int BN, bool A_COLK, bool B_COLK, bool FOLD = false>
__global__ __launch_bounds__(256, (BM * BN >= 96 * 128) ? 1 : (BM * BN <= 64 * 64 && !A_COLK && !B_COLK && !FOLD) ? NCX_OCC64_TN : 2) void seg_gemm_kernel(const GemmArgs args) {
    typedef GemmCfg<BM, BN, A_COLK, B_COLK> Cfg;
    constexpr int BK = Cfg::BK, PA = Cfg::PA, PB = Cfg::PB, WM = Cfg::WM, WN = Cfg::WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const lds_a = smem;                               // [2][A_ELEMS]
    float* const lds_b = smem + 2 * Cfg::A_ELEMS;            // [2][B_ELEMS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);

    const int M = args.M;
  // Optional persistent loop (NCX_PERSISTENT=1; default: one workgroup per item): the grid holds at most one workgroup per resident slot (a multiple of 8, so item % 8 -- the XCD
  // affinity WgMap is built on -- is the same for every item of a workgroup); a slot then never idles between the
  // exit of one short workgroup and the dispatch of the next.
  for (int item = blockIdx.x; item < args.total_wgs; item += gridDim.x) {
    // ---- which problem / tile / k-chunk -------------------------------------------------------------
    // Work items of problem p: w = wg0[p] + WgMap::encode(tm, tn, z) (XCD-aware order, see WgMap).
    // (accumulators are declared per item: hoisted out of this loop the 16-vector array of the 128x128 tile was
    // kept in scratch memory -- 32 scratch accesses per k-step, 45 instead of 116 TFLOP/s)
    f32x4 acc[WM][WN];
    f32x4 zacc[FOLD ? WM : 1][FOLD ? WN : 1];
#pragma unroll
    for (int i = 0; i < (FOLD ? WM : 1); ++i)
#pragma unroll
        for (int j = 0; j < (FOLD ? WN : 1); ++j) zacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int prob = 0, lw = item;
    if (args.mode == MODE_GROUP) {
        while (prob + 1 < args.nseg && lw >= args.wg0[prob + 1]) ++prob;
        lw -= args.wg0[prob];
    }
    const int S = args.split[prob] > 1 ? args.split[prob] : 1;
    const int N = args.n_cols[prob];
    const WgMap wmap{(M + BM - 1) / BM, (N + BN - 1) / BN, S};
    int tm, tn, z;
    if (!wmap.decode(lw, tm, tn, z)) continue;            // padding id of the chunk-per-XCD layout
    const int m0 = tm * BM, n0 = tn * BN;
    const int first_seg = args.mode == MODE_GROUP ? prob : 0;
    const int last_seg  = args.mode == MODE_GROUP ? prob + 1 : args.nseg;
    int total_steps = 0;
    for (int s = first_seg; s < last_seg; ++s) total_steps += (args.klen[s] + BK - 1) / BK;
    const int step_begin = (int)((long long)total_steps * z / S);
    const int step_end   = (int)((long long)total_steps * (z + 1) / S);

    // cursor: (segment, k position)
    int seg = first_seg, kpos = 0;
    {
        int skip = step_begin;
        while (seg < last_seg) {
            const int ns = (args.klen[seg] + BK - 1) / BK;
            if (skip < ns) { kpos = skip * BK; break; }
            skip -= ns; ++seg;
        }
    }

    typename Cfg::ALoad la;
    typename Cfg::BLoad lb;
    auto adesc = [&](int s) __attribute__((always_inline)) -> const XDesc& { return args.a[s]; };

#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};


    // Interleaved block mapping for row-is-k operands: MFMA block q of this wave owns tile rows/cols
    // {w0 + W*i + q : i = 0..15} (W = blocks per wave), so ONE ds_read_b128 / _b64 per k-row feeds all W blocks
    // (16 instead of 64 LDS reads per k-step on the 128x128 tile) and a lane ends up with W consecutive output
    // columns.  Col-is-k operands keep contiguous 16-row blocks (one ds_read_b64 per block and k-pair).
    constexpr bool A_IL = !A_COLK && (WM == 4 || WM == 2);
    constexpr bool B_IL = !B_COLK && (WN == 4 || WN == 2);
    auto trow = [&](int qa, int r) __attribute__((always_inline)) { return A_IL ? wm0 + WM * r + qa : wm0 + 16 * qa + r; };   // tile-local row
    auto tcol = [&](int qb, int c) __attribute__((always_inline)) { return B_IL ? wn0 + WN * c + qb : wn0 + 16 * qb + c; };   // tile-local col

    // LDS -> register fragments of sub-step t (8 k-values: two MFMAs per accumulator)
    auto read_frags = [&](const float* pa, const float* pb, int t, f32x2 (&af)[WM], f32x2 (&bf)[WN]) __attribute__((always_inline)) {
        const int kk = t * 8 + 2 * lk;
        if (A_COLK) {
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = *(const f32x2*)(pa + (wm0 + i * 16 + li) * PA + kk);
        } else if (A_IL) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (WM == 4) { const f32x4 v = *(const f32x4*)(pa + (kk + e) * PA + wm0 + 4 * li);
#pragma unroll
                               for (int i = 0; i < WM; ++i) af[i][e] = v[i]; }
                else         { const f32x2 v = *(const f32x2*)(pa + (kk + e) * PA + wm0 + 2 * li);
#pragma unroll
                               for (int i = 0; i < WM; ++i) af[i][e] = v[i & 1]; }
            }
        } else {
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = f32x2{pa[kk * PA + wm0 + i * 16 + li], pa[(kk + 1) * PA + wm0 + i * 16 + li]};
        }
        if (B_COLK) {
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[j] = *(const f32x2*)(pb + (wn0 + j * 16 + li) * PB + kk);
        } else if (B_IL) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (WN == 4) { const f32x4 v = *(const f32x4*)(pb + (kk + e) * PB + wn0 + 4 * li);
#pragma unroll
                               for (int j = 0; j < WN; ++j) bf[j][e] = v[j]; }
                else         { const f32x2 v = *(const f32x2*)(pb + (kk + e) * PB + wn0 + 2 * li);
#pragma unroll
                               for (int j = 0; j < WN; ++j) bf[j][e] = v[j & 1]; }
            }
        } else {
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[j] = f32x2{pb[kk * PB + wn0 + j * 16 + li], pb[(kk + 1) * PB + wn0 + j * 16 + li]};
        }
    };
    auto mfma_frags = [&](const f32x2 (&af)[WM], const f32x2 (&bf)[WN]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
    };
    // k order inside a 32-deep step: MFMA (t, e) takes k = 8t + 2*lk + e from lane group lk, for BOTH operands
    // (any bijection works as long as A and B agree).  A col-is-k tile then serves two MFMAs per ds_read_b64
    // (pitch 36: 36*i mod 64 hits 16 distinct multiples of 4, + 2*lk + e: conflict-free over a 32-lane half).
    auto compute_range = [&](int buf, int t0, int t1) __attribute__((always_inline)) {
        const float* pa = lds_a + buf * Cfg::A_ELEMS;
        const float* pb = lds_b + buf * Cfg::B_ELEMS;
#pragma unroll
        for (int t = t0; t < t1; ++t) {
            f32x2 af[WM], bf[WN];
            read_frags(pa, pb, t, af, bf);
            mfma_frags(af, bf);
        }
    };
    auto compute = [&](int buf) __attribute__((always_inline)) { compute_range(buf, 0, BK / 8); };
    constexpr int NMFMA = 2 * WM * WN;                                          // MFMAs per sub-step

    // One interior k-step with compile-time operand kinds.  Hand-pinned software pipeline (one wave per SIMD
    // has nobody else to hide its stalls):
    //   global loads of tile t+1  |  sub-steps 0..2: fragment reads run one sub-step ahead of their MFMAs  |
    //   sub-step 3's MFMAs interleaved with the transform (exp2 / product) + ds_write of tile t+1  |  barrier
    auto fast_run = [&](auto akind_c, auto bkind_c, auto mask_c, int nfast, int& buf) __attribute__((always_inline)) {
        constexpr int AK = decltype(akind_c)::value, BKD = decltype(bkind_c)::value;
        constexpr bool MK = decltype(mask_c)::value;
        const XDesc& da = adesc(seg);
        const XDesc& db = args.b[seg];
        for (int it = 0; it < nfast; ++it) {
            kpos += BK;
            const float* pa = lds_a + buf * Cfg::A_ELEMS;
            const float* pb = lds_b + buf * Cfg::B_ELEMS;
            f32x2 af0[WM], bf0[WN], af1[WM], bf1[WN];
            read_frags(pa, pb, 0, af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            // sub-step 0: its MFMAs carry the global loads of tile t+1 (one address add + one load per gap) and
            // the fragment reads of sub-step 1 -- with one wave per SIMD only what sits BETWEEN two MFMAs is hidden
            la.template issue_fast<AK, MK>(da, kpos, tid);
            lb.template issue_fast<BKD, MK>(db, kpos, tid);
            read_frags(pa, pb, 1, af1, bf1);
            mfma_frags(af0, bf0);
#pragma unroll
            for (int q = 0; q < NMFMA; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     // VALU (address arithmetic)
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
            }
            __builtin_amdgcn_sched_barrier(0);
            read_frags(pa, pb, 2, af0, bf0);
            mfma_frags(af1, bf1);
#pragma unroll
            for (int q = 0; q < NMFMA; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            read_frags(pa, pb, 3, af1, bf1);
            mfma_frags(af0, bf0);
#pragma unroll
            for (int q = 0; q < NMFMA; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // sub-step 3: transform (exp2 / product) + ds_write of tile t+1 in the MFMA shadows
            la.template store_fast<AK, MK>(lds_a + (buf ^ 1) * Cfg::A_ELEMS, tid);
            lb.template store_fast<BKD, MK>(lds_b + (buf ^ 1) * Cfg::B_ELEMS, tid);
            mfma_frags(af1, bf1);
#pragma unroll
            for (int q = 0; q < NMFMA; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // up to 3 VALU in its shadow
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // up to 1 ds_write
            }
            __syncthreads();
            buf ^= 1;
        }
        la.resync_after_fast(); lb.resync_after_fast();
    };
    // Depth-2 variant for the big tiles (one workgroup per CU, registers to spare): the loads of tile t+2 are
    // issued while tile t is multiplied and tile t+1 (loaded a whole k-step ago, so never waited for) is
    // transformed and written to LDS spread over sub-steps 1 and 2.  Measured by ablation on the depth-1 loop:
    // load waits and the clustered store phase cost ~10 % each.
    auto fast_run2 = [&](auto akind_c, auto bkind_c, auto mask_c, int nfast, int& buf) __attribute__((always_inline)) {
        constexpr int AK = decltype(akind_c)::value, BKD = decltype(bkind_c)::value;
        constexpr bool MK = decltype(mask_c)::value;
        const XDesc& da = adesc(seg);
        const XDesc& db = args.b[seg];
        typename Cfg::ALoad la2 = la;
        typename Cfg::BLoad lb2 = lb;
        la.template issue_fast<AK, MK>(da, kpos + BK, tid, 2 * BK);        // tile t+1 (indices prefetched by the caller)
        lb.template issue_fast<BKD, MK>(db, kpos + BK, tid, 2 * BK);
        la2.prefetch_rows(da, kpos + 2 * BK, tid);
        lb2.prefetch_rows(db, kpos + 2 * BK, tid);
        auto body = [&](auto issue_c, typename Cfg::ALoad& sa, typename Cfg::BLoad& sb, typename Cfg::ALoad& ia, typename Cfg::BLoad& ib) __attribute__((always_inline)) {
            constexpr bool ISSUE = decltype(issue_c)::value;
            kpos += BK;                                                  // kpos = tile being stored (t+1)
            const float* pa = lds_a + buf * Cfg::A_ELEMS;
            const float* pb = lds_b + buf * Cfg::B_ELEMS;
            float* wa = lds_a + (buf ^ 1) * Cfg::A_ELEMS;
            float* wb = lds_b + (buf ^ 1) * Cfg::B_ELEMS;
            f32x2 af0[WM], bf0[WN], af1[WM], bf1[WN];
            constexpr int NVA = Cfg::ALoad::NV, NVB = Cfg::BLoad::NV;
            read_frags(pa, pb, 0, af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            if (ISSUE) {
                ia.template issue_fast<AK, MK>(da, kpos + BK, tid, 2 * BK);
                ib.template issue_fast<BKD, MK>(db, kpos + BK, tid, 2 * BK);
            }
            read_frags(pa, pb, 1, af1, bf1);
            mfma_frags(af0, bf0);
#pragma unroll
            for (int q = 0; q < NMFMA; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            read_frags(pa, pb, 2, af0, bf0);
            sa.template store_fast<AK, MK>(wa, tid, 0, (NVA + 1) / 2);
            sb.template store_fast<BKD, MK>(wb, tid, 0, (NVB + 1) / 2);
            mfma_frags(af1, bf1);
#pragma unroll
            for (int q = 0; q < NMFMA; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            read_frags(pa, pb, 3, af1, bf1);
            sa.template store_fast<AK, MK>(wa, tid, (NVA + 1) / 2, NVA);
            sb.template store_fast<BKD, MK>(wb, tid, (NVB + 1) / 2, NVB);
            mfma_frags(af0, bf0);
#pragma unroll
            for (int q = 0; q < NMFMA; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_frags(af1, bf1);
            __syncthreads();
            buf ^= 1;
        };
        typedef std::true_type T; typedef std::false_type F;
        int it = 0;
        for (; it + 3 <= nfast; it += 2) { body(T{}, la, lb, la2, lb2); body(T{}, la2, lb2, la, lb); }
        if (nfast - it == 2) { body(T{}, la, lb, la2, lb2); body(F{}, la2, lb2, la, lb); }
        else                 { body(F{}, la, lb, la2, lb2); }
        la.resync_after_fast(); lb.resync_after_fast();
        la.prefetch_rows(da, kpos + BK, tid);                            // re-arm the generic path's index prefetch
        lb.prefetch_rows(db, kpos + BK, tid);
    };
    constexpr bool DEPTH2 = BM == 96 && BN == 128;      // 128x128 would need > 512 VGPRs (it spilled: 31 TFLOP/s at B=2048)
    // M / N edge tiles of col-is-k operands run the same fast path with the row mask compiled in (MAIN pays ~4 % for
    // it, so interior tiles get the mask-free variant)
    const bool edge_rows = m0 + BM > M || n0 + BN > N;
    auto fast_dispatch = [&](auto akind_c, auto bkind_c, int nfast, int& buf) __attribute__((always_inline)) {
        if (edge_rows) {
            if constexpr (DEPTH2) fast_run2(akind_c, bkind_c, std::true_type{}, nfast, buf);
            else fast_run(akind_c, bkind_c, std::true_type{}, nfast, buf);
        } else {
            if constexpr (DEPTH2) fast_run2(akind_c, bkind_c, std::false_type{}, nfast, buf);
            else fast_run(akind_c, bkind_c, std::false_type{}, nfast, buf);
        }
    };
    // The fast path only needs full K-steps: M / N edge tiles run its masked variant (col-is-k operands: rows beyond the
    // matrix zeroed; row-is-k operands: column windows clamped and repaired).
    const bool tile_interior = args.pad_ == 0;                                          // pad_ != 0: diagnostics

    // ---- main loop ------------------------------------------------------------------------------
    if (step_begin < step_end) {
        la.setup(adesc(seg), A_COLK ? m0 : m0, tid);
        lb.setup(args.b[seg], B_COLK ? n0 : n0, tid);
        la.prefetch_rows(adesc(seg), kpos, tid);
        lb.prefetch_rows(args.b[seg], kpos, tid);
        la.issue(adesc(seg), kpos, tid);
        lb.issue(args.b[seg], kpos, tid);
        la.prefetch_rows(adesc(seg), kpos + BK, tid);
        lb.prefetch_rows(args.b[seg], kpos + BK, tid);
        la.store(lds_a, kpos, tid);
        lb.store(lds_b, kpos, tid);
        __syncthreads();
        int buf = 0;
        int step = step_begin;
        while (step < step_end) {
            // interior stretch: steps whose NEXT tile is a full tile of the same segment
            int nfast = 0;
            if (tile_interior) {
                const int in_seg = args.klen[seg] / BK - 1 - kpos / BK;      // full tiles after the current one
                nfast = min(step_end - 1 - step, in_seg);
                // row-is-k operands: the reduction rows of every fast tile must exist (klen == rows there)
            }
            if (nfast > 0) {
                const int ak = la.kind == X_GATHER && A_COLK ? (int)X_PLAIN : la.kind;
                const int bk = lb.kind == X_GATHER && B_COLK ? (int)X_PLAIN : lb.kind;
                bool done = true;
                typedef std::integral_constant<int, X_PLAIN> KP;
                typedef std::integral_constant<int, X_GATHER> KG;
                typedef std::integral_constant<int, X_GATHER_MUL> KM;
                typedef std::integral_constant<int, X_SOFTMAX> KS;
                if (bk == X_PLAIN) {
                    if (ak == X_PLAIN) fast_dispatch(KP{}, KP{}, nfast, buf);
                    else if (A_COLK && B_COLK && ak == X_GATHER_MUL) fast_dispatch(KM{}, KP{}, nfast, buf);
                    else if (A_COLK && B_COLK && ak == X_SOFTMAX) fast_dispatch(KS{}, KP{}, nfast, buf);
                    else done = false;
                } else if (!A_COLK && !B_COLK && ak == X_PLAIN) {
                    if (bk == X_GATHER) fast_dispatch(KP{}, KG{}, nfast, buf);
                    else if (bk == X_GATHER_MUL) fast_dispatch(KP{}, KM{}, nfast, buf);
                    else if (bk == X_SOFTMAX) fast_dispatch(KP{}, KS{}, nfast, buf);
                    else done = false;
                } else {
                    done = false;
                }
                if (done) {
                    step += nfast;
                    // re-arm the generic path's index prefetch for the step after the current tile
                    continue;
                }
            }
            const bool has_next = step + 1 < step_end;
            int nkpos = kpos + BK, nseg = seg;
            if (has_next) {
                if (nkpos >= args.klen[seg]) {
                    nseg = seg + 1; nkpos = 0;
                    la.setup(adesc(nseg), m0, tid);
                    lb.setup(args.b[nseg], n0, tid);
                    la.prefetch_rows(adesc(nseg), 0, tid);
                    lb.prefetch_rows(args.b[nseg], 0, tid);
                }
                la.issue(adesc(nseg), nkpos, tid);
                lb.issue(args.b[nseg], nkpos, tid);
                la.prefetch_rows(adesc(nseg), nkpos + BK, tid);
                lb.prefetch_rows(args.b[nseg], nkpos + BK, tid);
            }
            compute(buf);
            if (FOLD && (!has_next || nseg != seg)) {                // pair `seg` is complete: fold it into zacc
                // operands fetched with clamped indices and no branches (guarded loads serialise at L2 latency)
#pragma unroll
                for (int i = 0; i < (FOLD ? WM : 1); ++i) {
                    float fb[WN], fm[WN][4];
#pragma unroll
                    for (int j = 0; j < (FOLD ? WN : 1); ++j) {
                        const long long c = (long long)seg * N + min(n0 + tcol(j, li), N - 1);
                        fb[j] = args.epi.fold_bias[c];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int r = min(m0 + trow(i, lk * 4 + q), M - 1);
                            fm[j][q] = args.epi.fold_mul[(long long)(r / args.epi.fold_div) * args.epi.ld_fold + c];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < (FOLD ? WN : 1); ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            zacc[FOLD ? i : 0][FOLD ? j : 0][q] += (acc[i][j][q] + fb[j]) * fm[j][q];
                            acc[i][j][q] = 0.f;
                        }
                }
            }
            if (has_next) {
                la.store(lds_a + (buf ^ 1) * Cfg::A_ELEMS, nkpos, tid);
                lb.store(lds_b + (buf ^ 1) * Cfg::B_ELEMS, nkpos, tid);
            }
            __syncthreads();
            buf ^= 1; seg = nseg; kpos = nkpos; ++step;
        }
    }

    // ---- epilogue -------------------------------------------------------------------------------
    if (S > 1) {
        // partial tile -> this workgroup's slab slot (whole padded tile: padded operands contribute zeros)
        float* slot = args.slab + (long long)item * (BM * BN);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    slot[trow(i, lk * 4 + q) * BN + tcol(j, li)] = acc[i][j][q];
        __syncthreads();                    // the LDS tiles are reused by the next item
        continue;
    }
    float* out = args.out[args.mode == MODE_GROUP ? prob : 0];
    const long long ldo = args.ldo[args.mode == MODE_GROUP ? prob : 0];
    const EpiArgs& e = args.epi;
    // Per row-block: every operand of the epilogue (row add, bias, dropout mask, gate) is fetched first, with
    // clamped indices and without per-element branches, then the block is finished and stored.  (A guard around
    // each load put it in its own basic block: 48-64 serialised L2 round trips per workgroup.)
    float bias_v[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) bias_v[j] = e.bias ? e.bias[min(n0 + tcol(j, li), N - 1)] : 0.f;
    // 4x4-block tiles: the row-block loop stays rolled (fully unrolled it exceeds the unroller's size limit, which
    // left acc[] dynamically indexed -> the whole accumulator array lived in scratch); the block's accumulators
    // are picked with compile-time-indexed selects instead.
    constexpr bool EPI_ROLL = WM * WN >= 16;
    constexpr int EPI_UNROLL = EPI_ROLL ? 1 : WM;
#pragma unroll EPI_UNROLL
    for (int i = 0; i < WM; ++i) {
        f32x4 accrow[WN];
#pragma unroll
        for (int ii = 0; ii < WM; ++ii)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const f32x4 src = FOLD ? zacc[FOLD ? ii : 0][FOLD ? j : 0] : acc[ii][j];
                if (ii == 0) accrow[j] = src;
                else accrow[j] = (ii == i) ? src : accrow[j];
            }
        float addv[WN][4], gatev[WN][4], maskv[WN][4];
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int nc = min(n0 + tcol(j, li), N - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rc = min(m0 + trow(i, lk * 4 + q), M - 1);
                addv[j][q] = e.rowadd ? e.rowadd[(long long)(rc / e.rowdiv) * e.ld_rowadd + nc] : 0.f;
                gatev[j][q] = e.gate ? e.gate[(long long)rc * e.ld_gate + nc] : 1.f;
                maskv[j][q] = e.dropout == 2 ? e.keep_mask[(long long)rc * e.ld_mask + nc] : 1.f;
            }
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + tcol(j, li);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = m0 + trow(i, lk * 4 + q);
                float v = accrow[j][q] + addv[j][q] + bias_v[j];
                if (e.relu == 1) v = v > 0.f ? v : 0.f;
                else if (e.relu == 2) v = tanhf(v);
                if (e.dropout == 1)
                    v = dropout_keep(e.seed_lo, e.seed_hi, e.layer, (unsigned long long)r * (unsigned)N + (unsigned)n, e.drop_p) ? v * e.drop_scale : 0.f;
                else if (e.dropout == 2)
                    v = maskv[j][q] != 0.f ? v * e.drop_scale : 0.f;
                if (e.gate) v = gatev[j][q] > 0.f ? v * e.gate_scale : 0.f;
                if (r < M && n < N) {
                    const int g = e.rowsplit_g;
                    if (g > 0) {
                        const int b = r / g, jj = r - b * g;
                        if (jj == 0) e.out0[(long long)b * e.ldo0 + n] = v;
                        else out[((long long)b * (g - 1) + jj - 1) * ldo + n] = v;
                    } else {
                        out[(long long)r * ldo + n] = v;
                    }
                }
            }
        }
    }
    __syncthreads();
  }   // persistent loop
}

// Split fix-up: output tile t of problem p = sum of its S partial tiles (slab slots wg0[p] + WgMap::encode(tm,tn,z),
// z ascending: deterministic) + optional bias, scattered to the problem's output with bounds.  Unsplit problems are skipped.
struct FixupArgs {
    float* out[NCX_MAX_SEG]; long long ldo[NCX_MAX_SEG]; int n_cols[NCX_MAX_SEG]; int split[NCX_MAX_SEG];
    int tile0[NCX_MAX_SEG + 1]; int wg0[NCX_MAX_SEG + 1];
    const float* bias;
    const float* slab;
    int mode, nseg, M, act;      // act: 0 none, 1 relu, 2 tanh (applied after the bias)
    int grid_x, valid;           // host: tiles of the launch; valid = a deferred fix-up is pending
};
// One block of the fix-up: `tile_blk` = tile index over the launch's problems, `slice` = which 1024-element slice of the tile
template <int BM, int BN>
__device__ __forceinline__ void split_fixup_body(const FixupArgs& a, int tile_blk, int slice) {
    int tile = tile_blk, prob = 0;
    if (a.mode == MODE_GROUP) {
        while (prob + 1 < a.nseg && tile >= a.tile0[prob + 1]) ++prob;
        tile -= a.tile0[prob];
    }
    const int S = a.split[prob];
    if (S <= 1) return;
    const int N = a.n_cols[prob];
    const int tiles_n = (N + BN - 1) / BN;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    float* out = a.out[prob];
    const long long ldo = a.ldo[prob];
    // one float4 per thread: blockIdx.y picks which 1024-element slice of the tile (BM*BN/1024 slices) -- small
    // problems have few tiles, so the slices are spread over blocks for memory-level parallelism
    const int e = slice * 1024 + threadIdx.x * 4;
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    const WgMap wmap{(a.M + BM - 1) / BM, tiles_n, S};
    // k-chunk order: deterministic.  Eight partial tiles are requested before the first is added (one load per iteration of a
    // runtime-length loop is one dependent round trip per k-chunk: 7 / 14 us for 4-16 chunks)
    for (int z = 0; z < S; z += 8) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = *(const f32x4*)(a.slab + ((long long)a.wg0[prob] + wmap.encode(tile / tiles_n, tile % tiles_n, min(z + j, S - 1))) * (BM * BN) + e);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc4 += z + j < S ? v[j] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int gr = m0 + e / BN, c = e % BN;
    if (gr < a.M) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gc = n0 + c + j;
            if (gc < N) {
                float v = acc4[j] + (a.bias ? a.bias[gc] : 0.f);
                if (a.act == 1) v = v > 0.f ? v : 0.f; else if (a.act == 2) v = tanhf(v);
                out[(long long)gr * ldo + gc] = v;
            }
        }
    }
}
template <int BM, int BN>
__global__ __launch_bounds__(256) void split_fixup_kernel(const FixupArgs a) { split_fixup_body<BM, BN>(a, blockIdx.x, blockIdx.y); }
// Two deferred fix-ups of the same tile shape in one launch (e.g. Gt and Sh: two back-to-back split GEMMs whose reductions are
// ~6 us each, most of it launch and ramp)
template <int BM, int BN>
__global__ __launch_bounds__(256) void split_fixup2_kernel(const FixupArgs a, const FixupArgs b) {
    constexpr int NS = BM * BN / 1024;
    const int id = blockIdx.x, na = a.grid_x * NS;
    if (id < na) split_fixup_body<BM, BN>(a, id / NS, id % NS);
    else         split_fixup_body<BM, BN>(b, (id - na) / NS, (id - na) % NS);
}
template <int BM, int BN>
static inline hipError_t launch_fixup2(const FixupArgs& a, const FixupArgs& b, hipStream_t stream) {
    constexpr int NS = BM * BN / 1024;
    if (a.valid && b.valid) hipLaunchKernelGGL((split_fixup2_kernel<BM, BN>), dim3((a.grid_x + b.grid_x) * NS), dim3(256), 0, stream, a, b);
    else if (a.valid)       hipLaunchKernelGGL((split_fixup_kernel<BM, BN>), dim3(a.grid_x, NS), dim3(256), 0, stream, a);
    else if (b.valid)       hipLaunchKernelGGL((split_fixup_kernel<BM, BN>), dim3(b.grid_x, NS), dim3(256), 0, stream, b);
    return hipGetLastError();
}

// ---- host-side launcher ---------------------------------------------------------------------------
// Workgroups needed by `args` with BM x BN tiles (also fills tile0 / wg0).
static inline long long gemm_layout(GemmArgs& args, int BM, int BN, bool* any_split) {
    const int tiles_m = (args.M + BM - 1) / BM;
    const int np = args.mode == MODE_GROUP ? args.nseg : 1;
    int tiles = 0; long long wgs = 0; bool sp = false;
    for (int i = 0; i < np; ++i) {
        const int tn = (args.n_cols[i] + BN - 1) / BN;
        const int t = tiles_m * tn;
        const int S = args.split[i] > 1 ? args.split[i] : 1;
        args.tile0[i] = tiles; args.wg0[i] = (int)wgs;
        tiles += t; wgs += WgMap{tiles_m, tn, S}.count(); sp |= S > 1;
    }
    args.tile0[np] = tiles; args.wg0[np] = (int)wgs;
    if (any_split) *any_split = sp;
    return wgs;
}

// Resident workgroups per CU of an instantiation (registers / LDS), for the planner; 2 when no device is present.
template <int BM, int BN, bool A_COLK, bool B_COLK, bool FOLD = false>
static inline int seg_gemm_occupancy() {
    typedef GemmCfg<BM, BN, A_COLK, B_COLK> Cfg;
    int n = 0;
    (void)hipFuncSetAttribute((const void*)seg_gemm_kernel<BM, BN, A_COLK, B_COLK, FOLD>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, seg_gemm_kernel<BM, BN, A_COLK, B_COLK, FOLD>, 256, Cfg::LDS_BYTES) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = 2;
    }
    return n;
}

template <int BM, int BN>
static inline hipError_t finish_split_launch(GemmArgs& args, hipStream_t stream);

template <int BM, int BN, bool A_COLK, bool B_COLK, bool FOLD = false>
static inline hipError_t launch_seg_gemm(GemmArgs& args, hipStream_t stream) {
    typedef GemmCfg<BM, BN, A_COLK, B_COLK> Cfg;
    bool any_split = false;
    const long long wgs = gemm_layout(args, BM, BN, &any_split);
    if (wgs == 0) return hipSuccess;
    static DevMask attr_set{0};
    {
        hipError_t e = set_max_lds_once(attr_set, (const void*)seg_gemm_kernel<BM, BN, A_COLK, B_COLK, FOLD>, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    args.total_wgs = (int)wgs;
    long long grid = wgs;
    {   // persistent grid for many-short-workgroup launches: one workgroup per resident slot
        static int slots = 0;
        if (!slots) {
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) { (void)hipGetLastError(); cus = 256; }
            slots = seg_gemm_occupancy<BM, BN, A_COLK, B_COLK, FOLD>() * cus;
            slots -= slots % 8;
        }
        // Measured (grouped dW1, 5856 items on 768 slots): 0.526 ms persistent vs 0.474 ms with one workgroup per
        // item -- the hardware dispatcher balances the unequal items better than a static stride.  Opt-in only.
        const char* pe = hook_env("NCX_PERSISTENT");
        if (pe && atoi(pe) && wgs > 2 * slots) grid = slots;
    }
    hipLaunchKernelGGL((seg_gemm_kernel<BM, BN, A_COLK, B_COLK, FOLD>), dim3((unsigned)grid), dim3(256), Cfg::LDS_BYTES, stream, args);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !any_split) return e;
    return finish_split_launch<BM, BN>(args, stream);
}

// The fix-up of a split launch whose layout (tile0 / wg0) is in `args`: launched here, or handed to the caller (defer_fix).
template <int BM, int BN>
static inline hipError_t finish_split_launch(GemmArgs& args, hipStream_t stream) {
    FixupArgs f{};
    const int np = args.mode == MODE_GROUP ? args.nseg : 1;
    for (int i = 0; i < np; ++i) { f.out[i] = args.out[i]; f.ldo[i] = args.ldo[i]; f.n_cols[i] = args.n_cols[i]; f.split[i] = args.split[i]; }
    for (int i = 0; i <= np; ++i) { f.tile0[i] = args.tile0[i]; f.wg0[i] = args.wg0[i]; }
    f.bias = args.epi.bias; f.slab = args.slab; f.mode = args.mode; f.nseg = args.nseg; f.M = args.M; f.act = args.epi.relu;
    f.grid_x = args.tile0[np]; f.valid = 1;
    if (args.defer_fix) { *args.defer_fix = f; return hipSuccess; }      // the caller launches it (merged with another one)
    hipLaunchKernelGGL((split_fixup_kernel<BM, BN>), dim3(args.tile0[np], BM * BN / 1024), dim3(256), 0, stream, f);
    return hipGetLastError();
}

// Tile geometry of a config, for the planner.
static inline void cfg_tile(int cfg, int& bm, int& bn) {
    bm = (cfg == 1 || cfg == 4) ? 128 : (cfg == 2 || cfg == 3) ? 96 : 64; bn = (cfg == 0 || cfg == 3 || cfg == 4) ? 64 : 128;
}

#endif  // __HIPCC__

}  // namespace ncx
