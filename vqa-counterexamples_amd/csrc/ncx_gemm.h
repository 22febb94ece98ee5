// ncx_gemm.h -- segmented fp32 MFMA GEMM engine for gfx950 (MI355X).
//
// One kernel template computes   C[M,N] = sum over "pairs" p of  A_p . B_p^T-ish   where every operand
// is described by an XDesc: a logical row-major matrix x(r, c) that is never materialised:
//
//   X_PLAIN       x(r,c) = base[r*ld + c]
//   X_GATHER      x(r,c) = base[idx[r]*ld + c]                       (feature-table / embedding row gather)
//   X_GATHER_MUL  x(r,c) = base[idx[r]*ld + c] * base[idx2[r]*ld + c] (v_orig * v_other, cx.py:296)
//   X_SOFTMAX     x(r,c) = exp(base[r*ld + c] - mx[r]) * inv[r]      (softmax(a_knns), cx.py:281)
//
// This is how the reference's torch.cat of ten segments (vqa/models/cx.py:309-320) is consumed without
// building it: the first Linear layer is a CHAIN of (segment of X, column slice of linear_1.weight) pairs
// accumulated into one MFMA accumulator, and its weight gradient is a GROUP of independent
// (dpre^T . segment) products that share the dpre operand.
//
// Operand forms.  An operand tile lives in LDS as a row-major copy of an x sub-block and is either
//   "col-is-k": the reduction index runs along the COLUMNS of x  (tile [rows=out index][32 k], pitch 34)
//   "row-is-k": the reduction index runs along the ROWS of x     (tile [32 k][cols=out index], pitch = 16 mod 32)
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

namespace ncx {

enum XKind : int { X_PLAIN = 0, X_GATHER = 1, X_GATHER_MUL = 2, X_SOFTMAX = 3 };

struct XDesc {
    const float* base;
    const int*   idx;
    const int*   idx2;
    const float* mx;
    const float* inv;
    long long    ld;
    int kind;
    int rows;
    int cols;
    int pad_;
};

static inline XDesc x_plain(const float* base, long long ld, int rows, int cols) {
    XDesc d{}; d.base = base; d.ld = ld; d.kind = X_PLAIN; d.rows = rows; d.cols = cols; return d;
}
static inline XDesc x_gather(const float* table, long long ld, const int* idx, int rows, int cols) {
    XDesc d{}; d.base = table; d.ld = ld; d.idx = idx; d.kind = X_GATHER; d.rows = rows; d.cols = cols; return d;
}
static inline XDesc x_gather_mul(const float* table, long long ld, const int* idx, const int* idx2, int rows, int cols) {
    XDesc d{}; d.base = table; d.ld = ld; d.idx = idx; d.idx2 = idx2; d.kind = X_GATHER_MUL; d.rows = rows; d.cols = cols; return d;
}
static inline XDesc x_softmax(const float* logits, long long ld, const float* mx, const float* inv, int rows, int cols) {
    XDesc d{}; d.base = logits; d.ld = ld; d.mx = mx; d.inv = inv; d.kind = X_SOFTMAX; d.rows = rows; d.cols = cols; return d;
}

constexpr int NCX_MAX_SEG = 6;
constexpr int GEMM_BK = 32;

enum GemmMode : int { MODE_CHAIN = 0, MODE_GROUP = 1 };

// Epilogue: v = acc (+ rowadd[(r/rowdiv)][n]) (+ bias[n]); relu; dropout; gate; store.
struct EpiArgs {
    const float* rowadd; long long ld_rowadd; int rowdiv;
    const float* bias;
    int relu;
    int dropout;                 // 0 none, 1 counter-based generator, 2 explicit keep mask
    float drop_p, drop_scale;
    unsigned seed_lo, seed_hi, layer;
    const float* keep_mask; long long ld_mask;
    const float* gate; long long ld_gate; float gate_scale;   // v *= gate[r][n] > 0 ? gate_scale : 0
};

struct GemmArgs {
    XDesc a[NCX_MAX_SEG];
    XDesc b[NCX_MAX_SEG];
    int   klen[NCX_MAX_SEG];     // reduction extent of pair/problem i
    float* out[NCX_MAX_SEG];     // CHAIN: out[0];  GROUP: one per problem
    long long ldo[NCX_MAX_SEG];
    int   n_cols[NCX_MAX_SEG];   // N of problem i (CHAIN: n_cols[0])
    int   tile0[NCX_MAX_SEG + 1];// GROUP: first linear tile id of problem i
    int   mode, nseg, M, ksplit;
    long long split_stride;      // elements between split-K slabs of out (ksplit > 1)
    EpiArgs epi;
};

// ------------------------------------------------------------------------------------------------
#if defined(__HIPCC__)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr int pitch_rowk(int c) { return ((c - 16 + 31) / 32) * 32 + 16; }  // >= c, = 16 mod 32

struct RowMeta {
    const float* p0;
    const float* p1;
    float mx, inv;
};

__device__ __forceinline__ RowMeta fetch_meta(const XDesc& d, int r) {
    RowMeta m; m.p0 = nullptr; m.p1 = nullptr; m.mx = 0.f; m.inv = 0.f;
    if (r < d.rows) {
        switch (d.kind) {
        case X_PLAIN:      m.p0 = d.base + (long long)r * d.ld; break;
        case X_GATHER:     m.p0 = d.base + (long long)d.idx[r] * d.ld; break;
        case X_GATHER_MUL: m.p0 = d.base + (long long)d.idx[r] * d.ld;
                           m.p1 = d.base + (long long)d.idx2[r] * d.ld; break;
        default:           m.p0 = d.base + (long long)r * d.ld; m.mx = d.mx[r]; m.inv = d.inv[r]; break;
        }
    }
    return m;
}

// 4 consecutive columns c..c+3 of one row, zero beyond `cols` / for an invalid row.
__device__ __forceinline__ f32x4 load4(const float* p, int c, int cols) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (p != nullptr) {
        if (c + 3 < cols) {
            v = *(const f32x4u*)(p + c);
        } else {
            if (c     < cols) v[0] = p[c];
            if (c + 1 < cols) v[1] = p[c + 1];
            if (c + 2 < cols) v[2] = p[c + 2];
        }
    }
    return v;
}

__device__ __forceinline__ f32x4 xform(int kind, const RowMeta& m, f32x4 v0, f32x4 v1, int c, int cols) {
    if (kind == X_GATHER_MUL) {
        return v0 * v1;
    } else if (kind == X_SOFTMAX) {
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            r[j] = (m.p0 != nullptr && c + j < cols) ? __expf(v0[j] - m.mx) * m.inv : 0.f;
        return r;
    }
    return v0;
}

// Thread <-> element mapping of a tile with R rows x C cols (C in {32, 64, 128}); 256 threads.
template <int R, int C>
struct TileMap {
    static constexpr int TPR = C / 4;          // threads per row
    static constexpr int RP  = 256 / TPR;      // rows per pass
    static constexpr int NP  = (R + RP - 1) / RP;
    static_assert(256 % TPR == 0, "tile width");
    static_assert(R % RP == 0, "tile rows");
};

template <int BM, int BN, bool A_COLK, bool B_COLK>
struct GemmCfg {
    static constexpr int BK = GEMM_BK;
    static constexpr int AR = A_COLK ? BM : BK, AC = A_COLK ? BK : BM;   // A tile rows/cols in x-space
    static constexpr int BR = B_COLK ? BN : BK, BC = B_COLK ? BK : BN;
    static constexpr int PA = A_COLK ? BK + 2 : pitch_rowk(BM);
    static constexpr int PB = B_COLK ? BK + 2 : pitch_rowk(BN);
    static constexpr int A_ELEMS = AR * PA, B_ELEMS = BR * PB;
    static constexpr int LDS_BYTES = 2 * (A_ELEMS + B_ELEMS) * 4;
    static constexpr int WM = BM / 32, WN = BN / 32;   // 16x16 blocks per wave (2x2 waves)
    typedef TileMap<AR, AC> AMap;
    typedef TileMap<BR, BC> BMap;
};

__device__ __forceinline__ float apply_epilogue(const EpiArgs& e, float v, int r, int n, int ncols_total) {
    if (e.rowadd) v += e.rowadd[(long long)(r / e.rowdiv) * e.ld_rowadd + n];
    if (e.bias) v += e.bias[n];
    if (e.relu) v = v > 0.f ? v : 0.f;
    if (e.dropout == 1) {
        v = dropout_keep(e.seed_lo, e.seed_hi, e.layer, (unsigned long long)r * (unsigned)ncols_total + (unsigned)n, e.drop_p)
                ? v * e.drop_scale : 0.f;
    } else if (e.dropout == 2) {
        v = e.keep_mask[(long long)r * e.ld_mask + n] != 0.f ? v * e.drop_scale : 0.f;
    }
    if (e.gate) v = e.gate[(long long)r * e.ld_gate + n] > 0.f ? v * e.gate_scale : 0.f;
    return v;
}

template <int BM, int BN, bool A_COLK, bool B_COLK>
__global__ __launch_bounds__(256, 2) void seg_gemm_kernel(const GemmArgs args) {
    typedef GemmCfg<BM, BN, A_COLK, B_COLK> Cfg;
    typedef typename Cfg::AMap AMap;
    typedef typename Cfg::BMap BMap;
    constexpr int BK = Cfg::BK, PA = Cfg::PA, PB = Cfg::PB, WM = Cfg::WM, WN = Cfg::WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const lds_a = smem;                               // [2][A_ELEMS]
    float* const lds_b = smem + 2 * Cfg::A_ELEMS;            // [2][B_ELEMS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);

    // ---- which problem / tile -------------------------------------------------------------------
    int prob = 0, tile = blockIdx.x;
    if (args.mode == MODE_GROUP) {
        while (prob + 1 < args.nseg && tile >= args.tile0[prob + 1]) ++prob;
        tile -= args.tile0[prob];
    }
    const int N = args.n_cols[prob];
    const int M = args.M;
    const int tiles_n = (N + BN - 1) / BN;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

    // ---- K range of this split ------------------------------------------------------------------
    const int first_seg = args.mode == MODE_GROUP ? prob : 0;
    const int last_seg  = args.mode == MODE_GROUP ? prob + 1 : args.nseg;
    int total_steps = 0;
    for (int s = first_seg; s < last_seg; ++s) total_steps += (args.klen[s] + BK - 1) / BK;
    const int z = blockIdx.y;
    const int step_begin = (int)((long long)total_steps * z / args.ksplit);
    const int step_end   = (int)((long long)total_steps * (z + 1) / args.ksplit);

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

    // ---- per-thread tile coordinates ------------------------------------------------------------
    const int a_tr = tid / AMap::TPR, a_tc = (tid % AMap::TPR) * 4;   // row-in-pass, col offset
    const int b_tr = tid / BMap::TPR, b_tc = (tid % BMap::TPR) * 4;

    RowMeta a_meta[AMap::NP], b_meta[BMap::NP];
    f32x4 a_v0[AMap::NP], a_v1[AMap::NP], b_v0[BMap::NP], b_v1[BMap::NP];
    int a_kind = 0, b_kind = 0, a_cols = 0, b_cols = 0;

    auto load_seg_meta = [&](int s) {
        const XDesc& da = args.a[args.mode == MODE_GROUP ? 0 : s];
        const XDesc& db = args.b[s];
        a_kind = da.kind; b_kind = db.kind; a_cols = da.cols; b_cols = db.cols;
        if (A_COLK) {
#pragma unroll
            for (int p = 0; p < AMap::NP; ++p) a_meta[p] = fetch_meta(da, m0 + p * AMap::RP + a_tr);
        }
        if (B_COLK) {
#pragma unroll
            for (int p = 0; p < BMap::NP; ++p) b_meta[p] = fetch_meta(db, n0 + p * BMap::RP + b_tr);
        }
    };

    auto issue_loads = [&](int s, int k) {
        const XDesc& da = args.a[args.mode == MODE_GROUP ? 0 : s];
        const XDesc& db = args.b[s];
        if (!A_COLK) {
#pragma unroll
            for (int p = 0; p < AMap::NP; ++p) a_meta[p] = fetch_meta(da, k + p * AMap::RP + a_tr);
        }
        if (!B_COLK) {
#pragma unroll
            for (int p = 0; p < BMap::NP; ++p) b_meta[p] = fetch_meta(db, k + p * BMap::RP + b_tr);
        }
        const int ac = A_COLK ? k + a_tc : m0 + a_tc;
        const int bc = B_COLK ? k + b_tc : n0 + b_tc;
#pragma unroll
        for (int p = 0; p < AMap::NP; ++p) {
            a_v0[p] = load4(a_meta[p].p0, ac, a_cols);
            if (a_kind == X_GATHER_MUL) a_v1[p] = load4(a_meta[p].p1, ac, a_cols);
        }
#pragma unroll
        for (int p = 0; p < BMap::NP; ++p) {
            b_v0[p] = load4(b_meta[p].p0, bc, b_cols);
            if (b_kind == X_GATHER_MUL) b_v1[p] = load4(b_meta[p].p1, bc, b_cols);
        }
    };

    auto store_lds = [&](int buf, int k) {
        float* la = lds_a + buf * Cfg::A_ELEMS;
        float* lb = lds_b + buf * Cfg::B_ELEMS;
        const int ac = A_COLK ? k + a_tc : m0 + a_tc;
        const int bc = B_COLK ? k + b_tc : n0 + b_tc;
#pragma unroll
        for (int p = 0; p < AMap::NP; ++p) {
            const f32x4 v = xform(a_kind, a_meta[p], a_v0[p], a_v1[p], ac, a_cols);
            float* dst = la + (p * AMap::RP + a_tr) * PA + a_tc;
            if (A_COLK) { *(f32x2*)dst = f32x2{v[0], v[1]}; *(f32x2*)(dst + 2) = f32x2{v[2], v[3]}; }
            else        { *(f32x4*)dst = v; }
        }
#pragma unroll
        for (int p = 0; p < BMap::NP; ++p) {
            const f32x4 v = xform(b_kind, b_meta[p], b_v0[p], b_v1[p], bc, b_cols);
            float* dst = lb + (p * BMap::RP + b_tr) * PB + b_tc;
            if (B_COLK) { *(f32x2*)dst = f32x2{v[0], v[1]}; *(f32x2*)(dst + 2) = f32x2{v[2], v[3]}; }
            else        { *(f32x4*)dst = v; }
        }
    };

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const float* la = lds_a + buf * Cfg::A_ELEMS;
        const float* lb = lds_b + buf * Cfg::B_ELEMS;
#pragma unroll
        for (int k4 = 0; k4 < BK / 4; ++k4) {
            float af[WM], bf[WN];
            const int kk = k4 * 4 + lk;
#pragma unroll
            for (int i = 0; i < WM; ++i)
                af[i] = A_COLK ? la[(wm0 + i * 16 + li) * PA + kk] : la[kk * PA + wm0 + i * 16 + li];
#pragma unroll
            for (int j = 0; j < WN; ++j)
                bf[j] = B_COLK ? lb[(wn0 + j * 16 + li) * PB + kk] : lb[kk * PB + wn0 + j * 16 + li];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- main loop ------------------------------------------------------------------------------
    if (step_begin < step_end) {
        load_seg_meta(seg);
        issue_loads(seg, kpos);
        store_lds(0, kpos);
        __syncthreads();
        int buf = 0;
        for (int step = step_begin; step < step_end; ++step) {
            const bool has_next = step + 1 < step_end;
            int nkpos = kpos + BK, nseg = seg;
            if (has_next) {
                if (nkpos >= args.klen[seg]) { nseg = seg + 1; nkpos = 0; load_seg_meta(nseg); }
                issue_loads(nseg, nkpos);
            }
            compute(buf);
            if (has_next) store_lds(buf ^ 1, nkpos);
            __syncthreads();
            buf ^= 1; seg = nseg; kpos = nkpos;
        }
    }

    // ---- epilogue -------------------------------------------------------------------------------
    float* out = args.out[args.mode == MODE_GROUP ? prob : 0];
    const long long ldo = args.ldo[args.mode == MODE_GROUP ? prob : 0];
    if (args.ksplit > 1) out += (long long)z * args.split_stride;
    const bool plain = args.ksplit > 1;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + wn0 + j * 16 + li;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = m0 + wm0 + i * 16 + lk * 4 + q;
                if (r < M && n < N) {
                    float v = acc[i][j][q];
                    if (!plain) v = apply_epilogue(args.epi, v, r, n, N);
                    out[(long long)r * ldo + n] = v;
                }
            }
        }
    }
}

// ---- host-side launcher ---------------------------------------------------------------------------
template <int BM, int BN, bool A_COLK, bool B_COLK>
static inline hipError_t launch_seg_gemm(GemmArgs& args, hipStream_t stream) {
    typedef GemmCfg<BM, BN, A_COLK, B_COLK> Cfg;
    int tiles = 0;
    const int tiles_m = (args.M + BM - 1) / BM;
    if (args.mode == MODE_GROUP) {
        for (int s = 0; s < args.nseg; ++s) {
            args.tile0[s] = tiles;
            tiles += tiles_m * ((args.n_cols[s] + BN - 1) / BN);
        }
        args.tile0[args.nseg] = tiles;
    } else {
        tiles = tiles_m * ((args.n_cols[0] + BN - 1) / BN);
    }
    if (tiles == 0) return hipSuccess;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)seg_gemm_kernel<BM, BN, A_COLK, B_COLK>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (args.ksplit < 1) args.ksplit = 1;
    dim3 grid(tiles, args.ksplit, 1);
    hipLaunchKernelGGL((seg_gemm_kernel<BM, BN, A_COLK, B_COLK>), grid, dim3(256), Cfg::LDS_BYTES, stream, args);
    return hipGetLastError();
}

#endif  // __HIPCC__

}  // namespace ncx
