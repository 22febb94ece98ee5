// ncx_internal.h -- declarations shared by the translation units of libneuralcx_hip.so.
#pragma once
#include "ncx_common.h"
#include "ncx_gemm.h"

namespace ncx {

enum TileCfg : int { CFG_64x64 = 0, CFG_128x128 = 1, CFG_96x128 = 2, CFG_96x64 = 3, CFG_128x64 = 4 };
enum GemmForm : int { FORM_NT = 0, FORM_TN = 1, FORM_NN = 2 };

int run_gemm_nt(GemmArgs& a, int cfg, hipStream_t s);
int run_fixup2(const FixupArgs& a, const FixupArgs& b, int cfg, hipStream_t s);
int run_gemm_nt_fold(GemmArgs& a, hipStream_t s);
int run_gemm_tn(GemmArgs& a, int cfg, hipStream_t s);
int run_gemm_nn(GemmArgs& a, int cfg, hipStream_t s);
int occupancy_nt(int cfg);
int occupancy_tn(int cfg);
int occupancy_nn(int cfg);

// Column offsets of the reference's concat (vqa/models/cx.py:309-320) == layout of linear_1.weight.
struct SegOffsets {
    int v_orig, v_other, v_mult, v_dist, v_rank, q_emb, z_orig, z_other, a_gt, a_other, din;
};
static inline SegOffsets seg_offsets(const ncx_dims& d) {
    SegOffsets o; int c = 0;
    o.v_orig = c;  c += d.dv;
    o.v_other = c; c += d.dv;
    o.v_mult = c;  c += d.dv;
    o.v_dist = c;  c += 1;
    o.v_rank = c;  c += d.K;
    o.q_emb = c;   c += d.dq;
    o.z_orig = c;  c += d.dz;
    o.z_other = c; c += d.dz;
    o.a_gt = c;    c += d.da;
    o.a_other = c; c += d.da;
    o.din = c;
    return o;
}

// Split-K plan of one GEMM: tile config + number of K splits (slabs reduced by k_slab_reduce).
struct GemmPlan { int cfg; int split; };   // split: aligned k-chunks per output tile (1 = none)
GemmPlan plan_gemm(int form, long long M, long long N, long long ksteps, bool allow_96);
int num_cus();

// Workspace partition (byte offsets from a 256-byte aligned base).  Saved-for-backward part first.
struct WsLayout {
    size_t idx_k, idx_o, idx_ob;        // int32 [M], [M], [B]: feature-table rows per logical row
    size_t mx, inv;                     // [M] softmax stats of a_knns rows
    size_t misc;                        // [M][K+1]: v_dist column + v_rank block (cx.py:299-307)
    size_t gt;                          // Gt[H][A] = W1[:, a_other] . E^T  (re-associated K3)
    size_t sh;                          // Sh[B][H]: per-triplet shared part of linear_1 (+ bias)
    size_t h[3];                        // post-dropout activations of layers 1..L, [M][H]
    size_t dpre[2];                     // [M][H] ping-pong of pre-activation gradients
    size_t dsh;                         // [B][H]
    size_t dgt;                         // [H][A]
    size_t dagt;                        // dGgt [H][A]  (bf16 variant only; the fp32 path keeps it transposed, below)
    size_t dgtT;                        // fp32 path: dGt^T [A][Hp4] then dGgt^T [A][Hp4] (contiguous: the all-reduce bucket under DP); Hp4 = H up to a multiple of 4
    size_t dgtT2;                       // private copy of dGt^T with zero rows up to a multiple of 32 (A operand of the dW1[:, a_other] launch, ncx_dwtn.hip)
    size_t w1aT;                        // fp32 path: W1[:, a_other]^T then W1[:, a_gt]^T, [da][Hp32] each (zero padded)
    size_t partial;                     // column-sum partials [NCX_COLSUM_CHUNKS][H] x 2 + scalars
    size_t slab;                        // split-K slabs (max over all uses)
    size_t slab_bytes;
    size_t slab2, slab2_bytes;          // slab of the GEMMs that run on the internal side stream
    size_t xc, wc, dpre_bf, bf_slab;    // NCX_F_BF16: packed bf16 candidate rows / weights / dpre, k-chunk slabs of dWc
    size_t bf_emb;                      // NCX_F_BF16: bf16 images of E, E^T, [W1ak; W1agt], [dGt; dGgt]
    size_t km_slab;                     // k-chunk slabs of the fused v_other / v_mult weight gradient (ncx_dwkm.hip)
    size_t mslab, mslab_bytes;          // [split][M][N] partial sums of the fused forward kernel's split problems (max over its uses)
    size_t wpad;                        // zero-padded copies of the weight slices whose width is not a multiple of 32 (ncx_main.h)
    int ldm;                            // leading dimension of misc: K + 1 rounded up to a multiple of 4 (zero padded)
    int ldgt;                           // leading dimension of Gt: A rounded up to a multiple of 32 (zero padded); bf16 variant: A
    size_t total;
};
constexpr int NCX_COLSUM_CHUNKS = 256;
constexpr int NCX_PRELUDE_WAVES = 1024;       // waves of k_bwd_prelude = rows of its partial sums (>= NCX_COLSUM_CHUNKS: same buffer)

// ncx_main.hip: the fused forward kernel of the Linear layers (segments of never-materialised operands chained into one
// accumulator; Sh / bias / ReLU / Dropout in the epilogue)
enum MainKind : int { MK_PLAIN = 0, MK_GATHER = 1, MK_GATHER_MUL = 2, MK_SOFTMAX = 3,
                      // the v_other and v_orig * v_other segments as ONE pass over the v_k rows against per-triplet effective
                      // weights W_k + diag(v_o[b]) W_m (the forward twin of the weight gradient's per-triplet fold); first
                      // segment only, K == 24, 48 x 64 tiles
                      MK_VFOLD = 4 };
constexpr int MAIN_MAX_SEG = 5;
struct MainSeg {
    const float* a;      // operand rows: x(r, c) = a[row(r) * lda + c], row(r) = r or idx[r]
    const int*   idx;    // MK_GATHER / MK_GATHER_MUL: table row of logical row r
    const int*   idx2;   // MK_GATHER_MUL: x(r, c) = a[idx[r]][c] * a[idx2[r]][c]        (v_other * v_orig, cx.py:296)
    const float* lse;    // MK_SOFTMAX: x(r, c) = exp2(a[r][c] * log2(e) - lse[r])        (softmax(a_knns), cx.py:281)
    const float* b;      // weight slice: w(n, c) = b[n * ldb + c]
    const float* b2;     // MK_VFOLD: the second weight slice (linear_1.weight[:, v_mult]), same ldb
    long long lda, ldb;
    int kind, klen;      // klen: reduction extent (columns of the segment), >= 4
};
struct MainArgs {
    MainSeg seg[MAIN_MAX_SEG]; int nseg, M, N, split; float* out; long long ldo; EpiArgs epi;
    float* slab;                     // split > 1: [split][M][N] partial sums
    // dist_out != NULL (sequence G, X, P, ...; split == 1; v rows a multiple of 32 wide): the pairwise distance
    // ||v_o - v_k + 1e-6|| (cx.py:300) is accumulated while the v_o * v_k segment streams both rows, patched into column 0 of
    // the following (dist | rank) segment's first tile and stored to dist_out[r * ld_dist] for the backward pass -- k_prep then
    // does not have to read the 200 MB of feature rows a second time
    float* dist_out; long long ld_dist;
    unsigned long long* stamps;      // diagnostics (tools/mb/mb_main.hip): 16 words per workgroup of s_memtime / s_memrealtime stamps; NULL in the library
    int x6;                          // NCX_F_X6: the plain / softmax segments of the 192 x 64 fold form on the bf16 matrix path with three-plane operands
};
int main_forward(MainArgs& a, hipStream_t s);
// k-split of a problem for the fused forward kernel (48 x 128 tiles): 1 when its tiles already give every CU a workgroup,
// else as many k-chunks as keep all workgroups resident at once (two per CU), at least 8 k-steps each.  T: k-steps of the whole chain.
int main_split(long long M, long long N, long long T);
int main_fold_rows_eff(long long M, long long N, int rowdiv);   // ... the form main_forward takes (hooks, K = 48 rule)
int main_fold_rows(long long M, long long N);      // tile rows of the per-triplet fold: 192 (one 8-wave workgroup per CU), 96 (four triplets per workgroup) or 48

// ncx_dwkm.hip: d linear_1.weight[:, v_other] and [:, v_mult] in one MFMA pass with a per-triplet fold
constexpr int DW_KM_SPLIT = 8;
bool dw_km_supported(const ncx_dims& d);
size_t dw_km_slab_bytes(const ncx_dims& d);
int dw_km(const ncx_dims& d, const float* dpre, const float* feats, const int* idx_k, const int* idx_o, float* slab,
          float* g_vother, float* g_vmult, long long din, hipStream_t s, bool finish = true);
int dw_km_finish(const ncx_dims& d, const float* slab, float* g_vother, float* g_vmult, long long din, const FixupArgs* fix, int fix_cfg,
                 hipStream_t s);
// ncx_dwtn.hip: every other row-reduction product of linear_1's weight gradient (dGt, z_other, dist | rank, the per-triplet shared
// segments) as one balanced launch of 8-wave workgroups on 256 x 64 tiles
constexpr int TN8_MAX_PROB = 8;
struct Tn8Prob {             // out[h][n] = sum_r A[r][h] * x(r, n)
    const float* A;          // [rows][H]: dpre (rows = B K) or dSh (rows = B)
    const float* X; long long ldx;   // operand rows, x(r, n) = X[row(r) * ldx + n]
    const float* lse;        // non-NULL: x = exp2(X log2e - lse[r])   (softmax(a_knns), cx.py:281)
    int gsel;                // row(r): 0 = r; 1 = idx_ob[r] (feature-table row of triplet r's original image); 2 = answer_aids[r]   (rows == B)
    int rows, N;             // rows % 32 == 0; N % 4 == 0 (columns the operand rows hold)
    float* out; long long ldo; int n_valid;      // columns written: n < n_valid <= N
    int rows_valid;          // 0: = rows.  Else operand rows >= rows_valid do not exist (their index is clamped) and A's rows there are ZERO
    float* outT; float* outT2; long long ldT; int padT2;   // optional transposed copies out^T[n][h] (ld ldT); outT2 also gets padT2 zero rows after n_valid
};
bool dw_tn8_supported(const ncx_dims& d);
bool dw_tn8_shapes_ok(const ncx_dims& d);          // ... ignoring the bf16 flag (the variant's fp32 shared segments)
size_t dw_tn8_slab_bytes(const ncx_dims& d);
// problems [0, n_al) (n_al <= 1): tile-aligned row chunks, one per workgroup; [n_al, np): laid end to end and cut into equal ranges.
// do_al / do_rest: which part this call launches (the chunking of each part does not depend on the other: phased backward)
int dw_tn8(const ncx_dims& d, const Tn8Prob* probs, int np, int n_al, bool do_al, bool do_rest, const int* idx_ob, const int* aid,
           float* slab, size_t slab_bytes, hipStream_t s);
// ncx_mutan.hip: the rank-R fusion of the MUTAN producer as ONE product per question against an effective weight tile built on the fly
bool mutan_fold_supported(const ncx_dims& d, const ncx_mutan_params& m);
int mutan_fold(const ncx_dims& d, const ncx_mutan_params& m, const float* xv, const float* hq, float* z_orig, float* z_knns, hipStream_t s);
WsLayout ws_layout(const ncx_dims& d);
__host__ __device__ static inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }
// The fused forward kernel (ncx_main.h) takes operands whose widths are multiples of 4 (16-byte windows, no straddling);
// other shapes (only the ragged test shapes: every real width is a multiple of 8) run on the generic engine.
// ... and whose extents are below 4 GiB (its loads are buffer loads with 32-bit byte offsets: the feature table, the candidate logits, the activations).
static inline bool main_fwd_extents_ok(const ncx_dims& d) {
    const long long lim = (1ll << 32) - 65536, M = (long long)d.B * d.K, wa = d.A > d.da ? d.A : d.da;
    return (long long)d.n_img * d.dv * 4 < lim && M * wa * 4 < lim && M * (long long)d.H * 4 < lim && M * (long long)d.dz * 4 < lim && (long long)d.A * wa * 4 < lim &&
           (long long)d.H * (2ll * d.dv + d.dq + 2ll * d.dz + 2ll * d.da + d.K + 2) * 4 < lim;
}
static inline bool main_fwd_dims_ok(const ncx_dims& d) {
    return d.dv % 4 == 0 && d.dz % 4 == 0 && ((d.flags & NCX_F_A_EMB) ? d.A : d.da) % 4 == 0 && !(d.flags & NCX_F_BF16) && main_fwd_extents_ok(d);
}
static inline bool hidden_fwd_dims_ok(const ncx_dims& d) { return d.H % 4 == 0 && main_fwd_extents_ok(d); }
// private flag bit (never set by callers: check_dims rejects it): the pairwise distance is computed inside the fused forward
// kernel, k_prep leaves the feature rows alone
constexpr uint32_t NCX_F_PRIV_DIST_IN_MAIN = 1u << 30;


}  // namespace ncx
