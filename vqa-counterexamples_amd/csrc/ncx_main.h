// ncx_main.h -- the fused forward kernel of the scoring MLP's Linear layers (gfx950, fp32 MFMA).
//
//   h[r, n] = epilogue( sum over segments s of  X_s[r, :] . W_s[n, :]^T  (+ Sh[r / K, n]) (+ bias[n]) )
//
// linear_1 of NeuralModel.forward (vqa/models/cx.py:309-322) is the chain of the five per-candidate segments
// [v_k | v_o*v_k | dist,rank | z_k | softmax(a_k)] against [W1 column slices | Gt]; the torch.cat is never built: every
// segment is a (never materialised operand, weight slice) pair accumulated into the same MFMA accumulators, the
// per-triplet shared part Sh[b] enters in the epilogue together with ReLU and Dropout (cx.py:322).  Hidden layers
// (linear_2/3, cx.py:323-326) are the same kernel with one plain segment and a bias.
//
// Structure (measured with tools/mb/mb_gemm.hip and tools/mb/mb_main.hip at the configs[1] shape, 41 GF):  one 96x128
// workgroup per CU with the whole register file ran 340-355 us; TWO OR MORE independent workgroups per CU (48x128, 96x64 or
// 64x64 tiles) 310-320 us (130 TFLOP/s): a workgroup parked at its barrier or waiting for its first fragment reads leaves the
// matrix pipes to the other one.  256 threads = 4 waves, 32-deep k-steps, double-buffered LDS (pitch 36: conflict-free
// ds_read_b64 fragment reads, two MFMAs per read), global loads issued one or two k-steps before their LDS store, and a
// ROTATED loop: the MFMAs of a step's last sub-step are issued after the barrier, so they cover the first fragment reads of
// the next tile.
//
// The k-step is ONE basic block, always (in-kernel stamps showed branches inside it costing 15 %: hipcc cannot interleave
// MFMAs with loads / LDS stores across them).  What makes that possible:
//   * ragged reduction extents need no code: the WEIGHT side of every segment is zero-padded to a multiple of 32 columns
//     (ncx_api.hip packs padded copies of the slices that need it; Gt is allocated padded), the OPERAND side only has to
//     be a multiple of 4 wide: its 16-byte loads are windows slid left to stay inside the row, so a window beyond the
//     extent re-reads valid (finite) columns that meet zero weights;
//   * rows beyond the matrix / the tile are clamped loads into spare LDS rows, never predicated;
//   * a segment's last step (nothing left to store) is its own instantiation; it loads the NEXT segment's first tiles, so a
//     segment switch costs a barrier, not a round trip to memory.  The sequence of operand kinds is a template parameter
//     (five sequences exist: linear_1 with / without the v_mult and a_emb lesions, and the hidden layers), so every such
//     hand-over is compiled for its two kinds and the register allocator sees exactly what is live where.
#pragma once
#include "ncx_internal.h"
#include <type_traits>

namespace ncx {

constexpr int MF_BK = 32, MF_P = MF_BK + 4, MF_T = 256;

// loop kinds: how a tile's registers become LDS values (the row gather only changes the pointer set-up)
enum { LK_PLAIN = 0, LK_MUL = 1, LK_SOFTMAX = 2 };
template <int V> struct IntC { static constexpr int value = V; };

// BM x BN tile, T threads = T / 64 waves as WGM x WGN, DEPTH register sets of global loads in flight, OCC = waves per SIMD the register
// budget is sized for (workgroups per CU x T / 256)
// X6 (NCX_F_X6; not the default): the plain / softmax segments run on the bf16 matrix path with three-plane fp32-grade operands (see run_seg)
constexpr int MF6_P = 80;                                            // bytes per 32-deep row of one bf16 plane (64 + 16: conflict-free ds_read_b128)
template <int BM_, int BN_, int WGM_, int WGN_, int DEPTH_, int OCC_, int T_ = MF_T, bool X6_ = false>
struct MainCfg {
    static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_, DEPTH = DEPTH_, OCC = OCC_, T = T_;
    static constexpr bool X6 = X6_;
    static constexpr int RP = T / 8;                                // tile rows per loader pass (8 threads x 16 bytes per 32-float row)
    static constexpr int BM_LDS = (BM + RP - 1) / RP * RP;        // A rows held in LDS (a multiple of the loader pass)
    static constexpr int LDS6 = 2 * 3 * (BM_LDS + BN) * MF6_P;      // X6: [2 buffers][3 planes][A rows | W rows][MF6_P]
    static constexpr int LDS_FOLD6 = 2 * (3 * 208 * MF6_P + 8 * 32 * 4 + 128 * MF_P * 4);   // X6 fold: [2][three planes of 208 v_k rows | 8 v_o rows fp32 | W_k, W_m 128 rows fp32]
    static constexpr int LDS32 = 2 * (BM_LDS + BN) * MF_P * 4;
    static constexpr int LDS = !X6_ ? LDS32 : (LDS_FOLD6 > LDS6 ? (LDS_FOLD6 > LDS32 ? LDS_FOLD6 : LDS32) : (LDS6 > LDS32 ? LDS6 : LDS32));
    static constexpr int LDS_FOLD = BM % 96 == 0 ? 2 * (32 * (T / 64) + 128) * MF_P * 4   // MK_VFOLD, one triplet per wave: A 32 rows per wave, W_k | W_m 2 x 64 rows
                                                 : 2 * (80 + 2 * 64) * MF_P * 4;          // MK_VFOLD on 48-row tiles: A 80 rows, two effective weight tiles
};

typedef const __attribute__((address_space(1))) float* gfptr;      // global address space: global_load, never flat_load
typedef const __attribute__((address_space(1))) int* giptr;
typedef const __attribute__((address_space(1))) f32x4u* gf4ptr;

template <class CFG, bool DIST, int K0, int K1 = -1, int K2 = -1, int K3 = -1, int K4 = -1>
__global__ __launch_bounds__(CFG::T, CFG::OCC) void k_main_fwd(const MainArgs args) {
    constexpr bool VFOLD = K0 == MK_VFOLD;
    constexpr int DIST_SEG = VFOLD ? 1 : 2;              // the (dist | rank) segment: right after the one that streams v_o and v_k
    static_assert(!DIST || (VFOLD ? K1 == MK_PLAIN : (K1 == MK_GATHER_MUL && K2 == MK_PLAIN)), "DIST: v_o, v_k segment followed by the dist | rank segment");
    static_assert(!VFOLD || (CFG::BM == 48 && CFG::BN == 64 && CFG::WGM == 1 && CFG::WGN == 4 && CFG::DEPTH == 2) ||
                            (CFG::BM == 96 && CFG::BN == 64 && CFG::WGM == 2 && CFG::WGN == 2 && CFG::DEPTH == 2) ||
                            (CFG::BM == 192 && CFG::BN == 64 && CFG::T == 512 && CFG::WGM == 4 && CFG::WGN == 2 && CFG::DEPTH == 2), "fold: 48 x 64, 96 x 64 or (8 waves) 192 x 64 tiles");
    static_assert(!(DIST && CFG::X6), "the X6 form leaves the distance to k_prep");
    constexpr int KS[6] = {K0, K1, K2, K3, K4, -1};
    constexpr int NSEG = K1 < 0 ? 1 : K2 < 0 ? 2 : K3 < 0 ? 3 : K4 < 0 ? 4 : 5;
    constexpr int BM = CFG::BM, BN = CFG::BN, BK = MF_BK, P = MF_P, DEPTH = CFG::DEPTH, BML = CFG::BM_LDS;
    constexpr int WTM = BM / CFG::WGM, WTN = BN / CFG::WGN, WM = WTM / 16, WN = WTN / 16, NSUB = BK / 8, NMF = 2 * WM * WN;
    constexpr int MT = CFG::T, RP = CFG::RP;              // threads per workgroup; tile rows per loader pass
    static_assert(CFG::WGM * CFG::WGN == MT / 64 && WTM % 16 == 0 && WTN % 16 == 0 && BN % RP == 0 && NSUB == 4, "tile");
    static_assert(MT == 256 || !VFOLD || BM % 96 == 0, "the 48-row fold is a 256-thread form");
    // f32x4 per thread and k-step: thread t owns column quad t & 7 of tile rows (t >> 3) + RP i
    constexpr int NA = BML / RP, NB = BN / RP;
    extern __shared__ __attribute__((aligned(16))) float mf_smem[];
    float* const lds_a = mf_smem;                       // [2][BML][P]
    float* const lds_b = mf_smem + 2 * BML * P;         // [2][BN][P]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = (wave / CFG::WGN) * WTM, wn0 = (wave % CFG::WGN) * WTN;
    const int M = args.M, N = args.N;
    // XCD-aware order: workgroup ids are dealt round-robin over the 8 XCDs (id % 8); the column tiles of one row tile stream
    // the same operand rows (gathered features, logits), so they get consecutive ids ON ONE XCD: one HBM fetch, L2 hits after
    // Split-K (args.split = S > 1: problems with too few tiles for 256 CUs -- small batches, the weight-only products): S
    // workgroups share a tile, each takes a contiguous range of the k-steps of the WHOLE segment chain and stores its raw
    // accumulators to slab[z][M][N]; k_main_fixup sums the S slabs in fixed order and applies the epilogue.
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int S = args.split > 1 ? args.split : 1;
    // (row tile, k-chunk) units are dealt over the XCDs; the column tiles of a unit are consecutive ids on one XCD
    const int id = blockIdx.x, xcd = id & 7, local = id >> 3;
    const int tn = local % tiles_n, unit = (local / tiles_n) * 8 + xcd;
    if (unit >= tiles_m * S) return;
    const int tm = unit / S, z = unit - tm * S;
    const int m0 = tm * BM, n0 = tn * BN;
    const int quad = tid & 7, trow = tid >> 3;
    unsigned long long* const stamps = args.stamps ? args.stamps + (size_t)blockIdx.x * 16 : nullptr;
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if (stamps && tid == 0) stamps[slot] = __builtin_readcyclecounter();
    };
    if (stamps && tid == 0) { stamps[14] = __builtin_amdgcn_s_memrealtime(); stamps[13] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20); }   // XCC_ID
    stamp(0);

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // afB / bfB always hold the fragments of a sub-step whose MFMAs have not been issued yet (zeros at the start)
    f32x2 afA[WM], bfA[WN], afB[WM], bfB[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) afB[i] = f32x2{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < WN; ++j) bfB[j] = f32x2{0.f, 0.f};

    auto read_frags = [&](int buf, int s, f32x2 (&af)[WM], f32x2 (&bf)[WN]) __attribute__((always_inline)) {
        const float* a = lds_a + buf * BML * P + (wm0 + li) * P + 8 * s + 2 * lk;
        const float* b = lds_b + buf * BN * P + (wn0 + li) * P + 8 * s + 2 * lk;
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = *(const f32x2*)(a + i * 16 * P);
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[j] = *(const f32x2*)(b + j * 16 * P);
    };
    // k order inside a 32-deep step: MFMA (s, e) takes k = 8 s + 2 lk + e from lane group lk, for BOTH operands
    auto mfma = [&](const f32x2 (&af)[WM], const f32x2 (&bf)[WN]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
    };

    // ---- X6: operands as three bf16 planes (x = x1 + x2 + x3 exactly, by truncation, cut when the tile is stored to LDS), six products per
    // 16 x 16 x 32 block on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: what is dropped is 2^-24 relative.  A sub-step is one 16-row block
    // row of the wave (6 WN MFMAs); the last block row of a step runs after the barrier, over the first reads of the next buffer.
    constexpr bool X6 = CFG::X6;
    static_assert(!X6 || (VFOLD && BM == 192 && WM == 3 && WN == 2 && DEPTH == 2), "X6: the 8-wave 192 x 64 fold form only");
    typedef __bf16 mbf16x8 __attribute__((ext_vector_type(8)));
    typedef unsigned int mu32x2 __attribute__((ext_vector_type(2)));
    constexpr int P6 = MF6_P, A6_PL = BML * P6, PL6 = (BML + BN) * P6, BUF6 = 3 * PL6;
    unsigned char* const lds6 = (unsigned char*)mf_smem;
    mbf16x8 xa[2][3], xb[2][X6 ? WN : 1][3];             // [alternating set][plane]; [parity of the step][column block][plane]
    auto split_store6 = [&](f32x4 v, unsigned char* base) __attribute__((always_inline)) {
        unsigned p1[4], p2[4], p3[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xj = v[j];               // (a scalar copy first: __builtin_bit_cast applied to the vector element itself reads element 0 -- hipcc 7.2)
            p1[j] = __builtin_bit_cast(unsigned, xj) & 0xFFFF0000u;
            const float r1 = xj - __builtin_bit_cast(float, p1[j]);
            p2[j] = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
            const float r2 = r1 - __builtin_bit_cast(float, p2[j]);
            p3[j] = __builtin_bit_cast(unsigned, r2);
        }
        const mu32x2 w1 = {__builtin_amdgcn_perm(p1[1], p1[0], 0x07060302u), __builtin_amdgcn_perm(p1[3], p1[2], 0x07060302u)};
        const mu32x2 w2 = {__builtin_amdgcn_perm(p2[1], p2[0], 0x07060302u), __builtin_amdgcn_perm(p2[3], p2[2], 0x07060302u)};
        const mu32x2 w3 = {__builtin_amdgcn_perm(p3[1], p3[0], 0x07060302u), __builtin_amdgcn_perm(p3[3], p3[2], 0x07060302u)};
        *(mu32x2*)(base) = w1; *(mu32x2*)(base + PL6) = w2; *(mu32x2*)(base + 2 * PL6) = w3;
    };
    auto read_a6 = [&](int buf, int i, mbf16x8 (&f)[3]) __attribute__((always_inline)) {
        const unsigned char* a = lds6 + buf * BUF6 + (wm0 + 16 * i + li) * P6 + 16 * lk;
#pragma unroll
        for (int p = 0; p < 3; ++p) f[p] = *(const mbf16x8*)(a + p * PL6);
    };
    auto read_b6 = [&](int buf, mbf16x8 (&f)[X6 ? WN : 1][3]) __attribute__((always_inline)) {
        const unsigned char* b = lds6 + buf * BUF6 + A6_PL + (wn0 + li) * P6 + 16 * lk;
#pragma unroll
        for (int j = 0; j < (X6 ? WN : 1); ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) f[j][p] = *(const mbf16x8*)(b + j * 16 * P6 + p * PL6);
    };
    // the 6 WN MFMAs of block row i; the column blocks alternate (a dependent MFMA is WN issues away); small terms first
    auto mfma6 = [&](const mbf16x8 (&a)[3], const mbf16x8 (&b)[X6 ? WN : 1][3], f32x4 (&c)[WN]) __attribute__((always_inline)) {
        if constexpr (X6) {
#pragma unroll
            for (int j = 0; j < WN; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[j][2], c[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < WN; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[j][1], c[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < WN; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[j][0], c[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < WN; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[j][1], c[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < WN; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[j][0], c[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < WN; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[j][0], c[j], 0, 0, 0);
        }
    };

    // this workgroup's range [g0, g1) of the chain's k-steps
    int nsteps[MAIN_MAX_SEG], total_steps = 0;
#pragma unroll
    for (int i = 0; i < MAIN_MAX_SEG; ++i) { nsteps[i] = i < NSEG ? (args.seg[i].klen + BK - 1) / BK : 0; total_steps += nsteps[i]; }
    const int g0 = (int)((long long)total_steps * z / S), g1 = (int)((long long)total_steps * (z + 1) / S);

    // Global-load register sets and operand pointers of the segment being loaded (shared by all segments).
    f32x4 va[DEPTH][NA], vm[DEPTH][NA], vb[DEPTH][NB];
    // Operand loads are buffer loads: descriptor (the segment's base) + a 32-bit per-lane byte offset; the weight side adds the k-step's offset in a scalar
    // register (no vector instruction per load), the operand side its slid window (vector instructions are not hidden under fp32 MFMAs: DESIGN S5d).
    // Every caller keeps operand extents below 4 GiB (main_fwd_extents_ok).
    typedef unsigned int bu32x4 __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t rsPA = __builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0x00020000), rsPB = rsPA;
    unsigned roA[NA], roM[NA], roB[NB]; float lse[NA];
    float dacc[NA];                                      // DIST: sum (v_o - v_k + 1e-6)^2 over this thread's column quads, per A row
    // (one-triplet-per-wave fold forms: the finished distances of the tile's rows wait here, behind the LDS of the segments that follow and of the layout conversion)
    constexpr int FOLD_DIST_OFF = 2 * (BML + BN) * P > 32 * (MT / 64) * 68 ? 2 * (BML + BN) * P : 32 * (MT / 64) * 68;
#pragma unroll
    for (int i = 0; i < NA; ++i) dacc[i] = 0.f;
    auto setup = [&](auto kind_c, const MainSeg& g) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_c)::value;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = min(m0 + min(trow + RP * i, BM - 1), M - 1);          // (rows beyond the tile / the matrix: clamped, never stored)
            constexpr bool GAT = KIND == MK_GATHER || KIND == MK_GATHER_MUL;
            const unsigned row = GAT ? (unsigned)((giptr)g.idx)[r] : (unsigned)r;
            roA[i] = row * (unsigned)(g.lda * 4);
            roM[i] = KIND == MK_GATHER_MUL ? (unsigned)((giptr)g.idx2)[r] * (unsigned)(g.lda * 4) : roA[i];
            lse[i] = KIND == MK_SOFTMAX ? ((gfptr)g.lse)[r] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) roB[i] = (unsigned)min(n0 + trow + RP * i, N - 1) * (unsigned)(g.ldb * 4) + 16u * quad;
        rsPA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.a), 0, 0xFFFFFFF0u, 0x00020000);
        rsPB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.b), 0, 0xFFFFFFF0u, 0x00020000);
    };
    // tile t of a segment -> register set S.  Operand side: 16-byte windows slid left to stay inside [0, klen) (klen % 4 == 0:
    // a window is either the true one or entirely beyond the extent, where the zero-padded weights null it); weight side:
    // padded rows, read as they are.  Tiles beyond the segment re-load its last tile and are never stored.
    auto issue = [&](auto set_c, auto mul_c, int klen, int nst, int t) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        const int tk4 = min(t, nst - 1) * (BK * 4);                       // uniform
        const unsigned ca4 = (unsigned)min(tk4 + 16 * quad, (klen - 4) * 4);
#pragma unroll
        for (int i = 0; i < NA; ++i) va[S][i] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsPA, roA[i] + ca4, 0, 0));
        if (decltype(mul_c)::value) {
#pragma unroll
            for (int i = 0; i < NA; ++i) vm[S][i] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsPA, roM[i] + ca4, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) vb[S][i] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsPB, roB[i], tk4, 0));
    };
    typedef IntC<0> S0; typedef IntC<DEPTH - 1> S1;
    typedef std::true_type Tt; typedef std::false_type Ff;

    // ---- one segment: its k-steps through the load pipeline ----------------------------------------------------------
    // On entry tiles t0 .. t0+DEPTH-1 of the segment are in flight in register sets 0 .. DEPTH-1 and `pa/pm/pb/lse` are its
    // pointers.  [t0, t1): the segment's steps that fall into this workgroup's range; pf_next: the next segment follows.
    auto run_seg = [&](auto idx_c, const int t0, const int t1, const bool pf_next) __attribute__((always_inline)) {
        constexpr int I = decltype(idx_c)::value;
        constexpr int KIND = KS[I], NKIND = KS[I + 1];                      // NKIND < 0: last segment
        constexpr int LKIND = KIND == MK_GATHER_MUL ? LK_MUL : KIND == MK_SOFTMAX ? LK_SOFTMAX : LK_PLAIN;
        typedef std::integral_constant<bool, LKIND == LK_MUL> MULC;
        const MainSeg& sg = args.seg[I];
        const int klen = sg.klen;
        const int nst = (klen + BK - 1) / BK;
        float lse_c[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) lse_c[i] = lse[i];
        // transform + LDS store of the tile in register set S; items [h0, h1).  PATCH (DIST, first tile of the segment after
        // v_o * v_k): column 0 <- the distance just accumulated
        auto stash = [&](auto set_c, auto patch_c, int buf, int h0, int h1) __attribute__((always_inline)) {
            constexpr int S = decltype(set_c)::value;
            constexpr bool PATCH = decltype(patch_c)::value;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (i < h0 || i >= h1) continue;
                f32x4 v = va[S][i];
                if (DIST && LKIND == LK_MUL) {
                    const f32x4 df = vm[S][i] - va[S][i] + 1e-6f;            // v_o - v_k + eps (cx.py:300)
                    dacc[i] += (df[0] * df[0] + df[1] * df[1]) + (df[2] * df[2] + df[3] * df[3]);
                }
                if (PATCH) v[0] = quad == 0 ? dacc[i] : v[0];
                if (LKIND == LK_MUL) v = v * vm[S][i];
                if (LKIND == LK_SOFTMAX) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[j], 1.44269504088896341f, -lse_c[i]));
                }
                if constexpr (X6) split_store6(v, lds6 + buf * BUF6 + (trow + RP * i) * P6 + 8 * quad);
                else
                *(f32x4*)(lds_a + buf * BML * P + (trow + RP * i) * P + 4 * quad) = v;       // (rows >= BM: spare LDS rows, never read)
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (i + NA < h0 || i + NA >= h1) continue;
                if constexpr (X6) split_store6(vb[S][i], lds6 + buf * BUF6 + A6_PL + (trow + RP * i) * P6 + 8 * quad);
                else
                *(f32x4*)(lds_b + buf * BN * P + (trow + RP * i) * P + 4 * quad) = vb[S][i];
            }
        };
        // prologue: tile t0 -> LDS buffer 0 (DEPTH 2: tile t0+2 into the freed set)
        if constexpr (DIST && I == DIST_SEG) {                  // the 8 threads of a row hold its partial sums: reduce, sqrt, keep, store
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if constexpr (VFOLD && BM % 96 == 0) {
                    dacc[i] = mf_smem[FOLD_DIST_OFF + min(trow + RP * i, BM - 1)];      // finished by run_vfold4 (another thread layout): through LDS
                } else {
                float d2 = dacc[i];
                d2 += __shfl_xor(d2, 1, 64); d2 += __shfl_xor(d2, 2, 64); d2 += __shfl_xor(d2, 4, 64);
                dacc[i] = sqrtf(d2);
                }
                const int r = m0 + trow + RP * i;
                if (quad == 0 && tn == 0 && trow + RP * i < BM && r < M) args.dist_out[(long long)r * args.ld_dist] = dacc[i];
            }
            stash(S0{}, Tt{}, 0, 0, NA + NB);
        } else
        stash(S0{}, Ff{}, 0, 0, NA + NB);
        issue(S0{}, MULC{}, klen, nst, t0 + DEPTH);
        __syncthreads();

        // step t (parity PAR = t & 1): LDS buffer PAR holds tile t; DEPTH 2: set PAR^1 holds tile t+1, set PAR tile t+2;
        // DEPTH 1: the one set holds tile t+1 and is re-issued for tile t+2 once stored.
        // LAST: nothing to store; the free register sets take the next segment's first tiles instead.
        auto step = [&](auto par_c, auto last_c, int t) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_c)::value;
            constexpr bool LAST = decltype(last_c)::value;
            typedef IntC<DEPTH == 2 ? (PAR ^ 1) : 0> SS;
            if constexpr (X6) {
                // sets alternate with the step's parity (three block rows per step): xa[PAR ^ 1] comes in holding block row 2 of the previous step
                read_b6(PAR, xb[PAR]);
                read_a6(PAR, 0, xa[PAR]);
                mfma6(xa[PAR ^ 1], xb[PAR ^ 1], acc[2]);
#pragma unroll
                for (int q = 0; q < 6 * WN; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_barrier(0);
                if (LAST && NKIND >= 0 && pf_next) {
                    const MainSeg& nx = args.seg[I + 1 < MAIN_MAX_SEG ? I + 1 : I];
                    setup(IntC<NKIND>{}, nx);
                    const int kn = nx.klen, nn = (kn + BK - 1) / BK;
                    typedef std::integral_constant<bool, NKIND == MK_GATHER_MUL> NM;
                    issue(S0{}, NM{}, kn, nn, 0);
                    issue(S1{}, NM{}, kn, nn, 1);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    read_a6(PAR, i + 1, xa[PAR ^ 1 ^ i]);
                    if (!LAST) {
                        if (i == 0) stash(SS{}, Ff{}, PAR ^ 1, 0, (NA + NB) / 2);
                        if (i == 1) { stash(SS{}, Ff{}, PAR ^ 1, (NA + NB) / 2, NA + NB); issue(SS{}, MULC{}, klen, nst, t + 1 + DEPTH); }
                    }
                    mfma6(xa[PAR ^ i], xb[PAR], acc[i]);
#pragma unroll
                    for (int q = 0; q < 6 * WN; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (LAST) {                              // nothing is carried over a segment switch: block row 2 now, zeros into the set the next segment's first step multiplies first
                    mfma6(xa[PAR], xb[PAR], acc[2]);
                    const mbf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int p = 0; p < 3; ++p) { xa[0][p] = z8; xa[1][p] = z8; }
                }
                __syncthreads();
                return;
            }
            read_frags(PAR, 0, afA, bfA);
            mfma(afB, bfB);                                              // (t-1, last sub-step): covers the reads above
#pragma unroll
            for (int q = 0; q < NMF; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            if (LAST && NKIND >= 0 && pf_next) {
                const MainSeg& nx = args.seg[I + 1 < MAIN_MAX_SEG ? I + 1 : I];
                setup(IntC<NKIND>{}, nx);
                const int kn = nx.klen, nn = (kn + BK - 1) / BK;
                typedef std::integral_constant<bool, NKIND == MK_GATHER_MUL> NM;
                issue(S0{}, NM{}, kn, nn, 0);
                if (DEPTH == 2) issue(S1{}, NM{}, kn, nn, 1);
            }
#pragma unroll
            for (int s = 0; s < NSUB - 1; ++s) {
                auto& afc = (s & 1) ? afB : afA; auto& bfc = (s & 1) ? bfB : bfA;
                auto& afn = (s & 1) ? afA : afB; auto& bfn = (s & 1) ? bfA : bfB;
                read_frags(PAR, s + 1, afn, bfn);
                if (!LAST) {
                    if (s == 0) stash(SS{}, Ff{}, PAR ^ 1, 0, (NA + NB) / 2);
                    if (s == 1) stash(SS{}, Ff{}, PAR ^ 1, (NA + NB) / 2, NA + NB);
                    if (s == 2) issue(SS{}, MULC{}, klen, nst, t + 1 + DEPTH);
                }
                mfma(afc, bfc);
#pragma unroll
                for (int q = 0; q < NMF; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        };
        typedef IntC<0> P0; typedef IntC<1> P1;
        int t = t0;                                                                               // (parity = (t - t0) & 1)
        for (; t + 2 < t1; t += 2) { step(P0{}, Ff{}, t); step(P1{}, Ff{}, t + 1); }              // t + 1 <= t1 - 2
        if (t + 1 < t1) { step(P0{}, Ff{}, t); step(P1{}, Tt{}, t + 1); }
        else step(P0{}, Tt{}, t);
    };

    // ---- MK_VFOLD: v_k . (W_k + diag(v_o[b]) W_m)^T for the two triplets of the tile ------------------------------------------
    // (v_o * v_k) . W_m^T = v_k . (diag(v_o) W_m)^T and the 24 candidate rows of a triplet share v_o, so the two 2048-deep
    // segments become ONE pass over the v_k rows against a per-triplet effective weight tile built on the vector ALU at LDS-store
    // time (2 x 64 x 32 FMAs per k-step).  MFMA row blocks must be triplet-pure: each triplet's 24 rows sit in 32 LDS rows
    // (8 zero rows), i.e. 4 block rows instead of the 2 x 3 of the two plain segments: 2/3 of their MFMA work.  Waves 2 x 2:
    // wave (wr, wc) = triplet wr x column half wc.  The accumulators of this phase live in the padded row layout; they pass
    // through LDS once into the compact 48-row layout of the segments that follow.
    auto run_vfold = [&](const MainSeg& sg, const bool pf_next) __attribute__((always_inline)) {
        constexpr int AR = 80;                              // A rows in LDS: 2 x 32 (24 + 8 zero rows) + 16 dump rows for the loader's idle items
        float* const fa = mf_smem;                          // [2][AR][P]
        float* const fb = mf_smem + 2 * AR * P;             // [2][2 triplets][64][P]
        const int klen = sg.klen, nst = klen / BK;
        const int wr = wave >> 1, wc = wave & 1;
        gfptr pk[2], pwk[2], pwm[2], po[2];
        int lrow[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rho = trow + 32 * i;                                        // tile row 0 .. 63 (>= 48: idle)
            const int r = min(m0 + min(rho, BM - 1), M - 1);
            pk[i] = (gfptr)sg.a + (long long)((giptr)sg.idx)[r] * sg.lda;
            lrow[i] = rho < 24 ? rho : rho < 48 ? rho + 8 : 64 + (rho - 48);
            const int n = min(n0 + trow + 32 * i, N - 1);
            pwk[i] = (gfptr)sg.b + (long long)n * sg.ldb;
            pwm[i] = (gfptr)sg.b2 + (long long)n * sg.ldb;
            po[i] = (gfptr)sg.a + (long long)((giptr)sg.idx2)[min(m0 + 24 * i, M - 1)] * sg.lda;      // v_o of triplet i
        }
        const int t_of0 = trow < 24 ? 0 : 1;                                       // triplet of this thread's first A item (the second: 1)
        f32x4 vk[2][2], vo[2][2], vwk[2][2], vwm[2][2];
        auto vissue = [&](auto set_c, int t) __attribute__((always_inline)) {
            constexpr int S = decltype(set_c)::value;
            const int c = min(t, nst - 1) * BK + 4 * quad;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#ifndef NCX_ABL_FOLD       // (tools/mb/mb_fold.hip timing ablations, results wrong: 1 no W_m loads, 2 no v_o loads, 3 no weight loads, 4 no v_k loads)
                vk[S][i] = *(gf4ptr)(pk[i] + c); vo[S][i] = *(gf4ptr)(po[i] + c); vwk[S][i] = *(gf4ptr)(pwk[i] + c); vwm[S][i] = *(gf4ptr)(pwm[i] + c);
#else
                if (NCX_ABL_FOLD != 4 || t == 0) vk[S][i] = *(gf4ptr)(pk[i] + c);
                if (NCX_ABL_FOLD != 2 || t == 0) vo[S][i] = *(gf4ptr)(po[i] + c);
                if (NCX_ABL_FOLD != 3 || t == 0) vwk[S][i] = *(gf4ptr)(pwk[i] + c);
                if ((NCX_ABL_FOLD != 3 && NCX_ABL_FOLD != 1) || t == 0) vwm[S][i] = *(gf4ptr)(pwm[i] + c);
#endif
            }
        };
        // part 0: the v_k rows + triplet 0's effective weights; part 1: triplet 1's
        auto vstash = [&](auto set_c, int buf, int part) __attribute__((always_inline)) {
            constexpr int S = decltype(set_c)::value;
            if (part == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (DIST) {
                        const f32x4 ov = i == 0 ? (t_of0 == 0 ? vo[S][0] : vo[S][1]) : vo[S][1];
                        const f32x4 df = ov - vk[S][i] + 1e-6f;                    // v_o - v_k + eps (cx.py:300)
                        dacc[i] += (df[0] * df[0] + df[1] * df[1]) + (df[2] * df[2] + df[3] * df[3]);
                    }
                    *(f32x4*)(fa + buf * AR * P + lrow[i] * P + 4 * quad) = vk[S][i];
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = __builtin_fmaf(vo[S][part][e], vwm[S][j][e], vwk[S][j][e]);
                *(f32x4*)(fb + ((buf * 2 + part) * 64 + trow + 32 * j) * P + 4 * quad) = w;
            }
        };
        // the zero rows of both A buffers (rows 24..31 and 56..63): 2 x 16 rows x 8 quads = 256 quads
        {
            const int b = tid >> 7, rr = (tid >> 3) & 15, zr = rr < 8 ? 24 + rr : 48 + rr;
            *(f32x4*)(fa + b * AR * P + zr * P + 4 * quad) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 acc4[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x2 a0[2], b0[2], a1[2], b1[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { a1[i] = f32x2{0.f, 0.f}; b1[i] = f32x2{0.f, 0.f}; }
        auto vread = [&](int buf, int s, f32x2 (&af)[2], f32x2 (&bf)[2]) __attribute__((always_inline)) {
            const float* a = fa + buf * AR * P + (32 * wr + li) * P + 8 * s + 2 * lk;
            const float* b = fb + ((buf * 2 + wr) * 64 + 32 * wc + li) * P + 8 * s + 2 * lk;
#pragma unroll
            for (int i = 0; i < 2; ++i) { af[i] = *(const f32x2*)(a + i * 16 * P); bf[i] = *(const f32x2*)(b + i * 16 * P); }
        };
        auto vmfma = [&](const f32x2 (&af)[2], const f32x2 (&bf)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc4[i][j], 0, 0, 0);
        };
        typedef IntC<0> V0; typedef IntC<1> V1;
        vissue(V0{}, 0);
        vissue(V1{}, 1);
        vstash(V0{}, 0, 0); vstash(V0{}, 0, 1);
        vissue(V0{}, 2);
        __syncthreads();
        auto vstep = [&](auto par_c, auto last_c, int t) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_c)::value;
            constexpr bool LAST = decltype(last_c)::value;
            typedef IntC<PAR ^ 1> SS;
            vread(PAR, 0, a0, b0);
            vmfma(a1, b1);                                               // (t-1, last sub-step)
#pragma unroll
            for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            if (LAST && pf_next) {                                       // the next (plain) segment's first tiles
                const MainSeg& nx = args.seg[1];
                setup(IntC<K1 < 0 ? 0 : K1>{}, nx);
                const int kn = nx.klen, nn = (kn + BK - 1) / BK;
                issue(S0{}, Ff{}, kn, nn, 0);
                issue(S1{}, Ff{}, kn, nn, 1);
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                auto& afc = (s & 1) ? a1 : a0; auto& bfc = (s & 1) ? b1 : b0;
                auto& afn = (s & 1) ? a0 : a1; auto& bfn = (s & 1) ? b0 : b1;
                vread(PAR, s + 1, afn, bfn);
                if (!LAST) {
                    if (s == 0) vstash(SS{}, PAR ^ 1, 0);
                    if (s == 1) vstash(SS{}, PAR ^ 1, 1);
                    if (s == 2) vissue(SS{}, t + 3);
                }
                vmfma(afc, bfc);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        };
        int t = 0;
        for (; t + 2 < nst; t += 2) { vstep(V0{}, Ff{}, t); vstep(V1{}, Ff{}, t + 1); }
        if (t + 1 < nst) { vstep(V0{}, Ff{}, t); vstep(V1{}, Tt{}, t + 1); }
        else vstep(V0{}, Tt{}, t);
        vmfma(a1, b1);                                                   // the last sub-step
        // padded (wave = triplet x column half) -> compact (wave = 16 columns, all 48 rows) accumulator layout, through LDS
        constexpr int CP = 68;
        float* const fc = mf_smem;                                       // [64 padded rows][CP]   (every fragment read is complete: last barrier)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) fc[(32 * wr + 16 * i + 4 * lk + q) * CP + 32 * wc + 16 * j + li] = acc4[i][j][q];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rho = wm0 + 16 * i + 4 * lk + q;                   // compact tile row
                acc[i][0][q] = fc[(rho < 24 ? rho : rho + 8) * CP + wn0 + li];
            }
        __syncthreads();
    };

    // ---- MK_VFOLD on 96 x 64 tiles: FOUR triplets per workgroup, wave w = triplet w x all 64 columns --------------------------------
    // The 48-row form materialises an effective weight tile per triplet in LDS and moves 32 KB from L2 per workgroup and k-step for
    // 32 MFMAs per wave (8.7 flop per byte: the phase is bound by operand delivery, tools/mb/mb_fold.hip).  Here W_k and W_m go to
    // LDS once for four triplets, v_o[w] rides in row 24 of triplet w's 32-row A block (rows 24..31 are padding: their outputs are
    // dropped), and the effective-weight fragment is ONE fma per MFMA on the way from LDS to the matrix core:
    //     b = fma(v_o[w][k], W_m[n][k], W_k[n][k])        (the same expression, hence the same bits, as the 48-row form)
    // 28.8 KB per workgroup and k-step for 64 MFMAs per wave, 512 workgroups = one round at two per CU.
    auto run_vfold4 = [&](const MainSeg& sg, const bool pf_next) __attribute__((always_inline)) {
      if constexpr (VFOLD && BM % 96 == 0) {
        constexpr int AR = 32 * (MT / 64), BR = 128;        // one 32-row A block per wave (= per triplet at K = 24)
        constexpr int NAI = AR / RP, NBI = BR / RP, BH = 64 / RP;     // loader items per thread: A rows, W rows (BH of them W_k, BH W_m)
        static_assert(NAI == 4 && BH >= 1, "fold loader");
        float* const fa = mf_smem;                          // [2][AR][P]   triplet w: rows 32 w .. 32 w + 23 = v_k, row 32 w + 24 = v_o[w]
        float* const fb = mf_smem + 2 * AR * P;             // [2][BR][P]   rows 0 .. 63 = W_k, 64 .. 127 = W_m (tile columns n0 ..)
        const int klen = sg.klen, nst = klen / BK;
        // kr = 24: four triplets, 32-row blocks (24 v_k rows, v_o, 7 spare); kr = 48 (K = 48): two triplets, 64-row blocks (48 v_k rows,
        // v_o, 15 spare) -- 4 MFMA row blocks per 48 rows against the 6 of the two plain segments, like 4 per 24 against 2 x 3
        const int kr = args.epi.rowdiv, blk = kr == 24 ? 32 : 64;
        // Operand loads as buffer loads: descriptor (uniform base) + a 32-bit per-lane byte offset that never changes (row x pitch + quad) + a uniform
        // 32-bit step offset in a scalar register -- NO vector instruction per load (the 64-bit `pointer + column` add per load was 6 of this loop's ~38
        // vector instructions per k-step, and those are not hidden under fp32 MFMAs: DESIGN S5d).  Host-checked: every operand extent < 4 GiB.
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.a), 0, 0xFFFFFFF0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.b), 0, 0xFFFFFFF0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.b2), 0, 0xFFFFFFF0u, 0x00020000);
        unsigned voA[NAI], voB[NBI];
        // loader item i of this thread: A row 32 wave + (lane >> 3) + 8 i -- a wave loads the rows of ITS 32-row block, so at K = 24 item 3 (block rows 24 ..
        // 31: v_o, every one of them) gives every lane the v_o quad of the columns its v_k items cover: the pairwise distance needs no exchange
        const int arow4 = 32 * wave + (lane >> 3);
#pragma unroll
        for (int i = 0; i < NAI; ++i) {
            const int lr = arow4 + 8 * i, tr = lr / blk, j = lr - tr * blk;     // LDS row -> (triplet of the tile, row of its block)
            const int r = min(m0 + kr * tr + (j < kr ? j : 0), M - 1);
            voA[i] = (unsigned)(j < kr ? ((giptr)sg.idx)[r] : ((giptr)sg.idx2)[r]) * (unsigned)(sg.lda * 4) + 16u * quad;
        }
#pragma unroll
        for (int i = 0; i < NBI; ++i) {                     // ... and W row trow + RP i of [W_k (64 rows) ; W_m (64 rows)]
            const int n = min(n0 + trow + RP * (i % BH), N - 1);
            voB[i] = (unsigned)n * (unsigned)(sg.ldb * 4) + 16u * quad;
        }
        f32x4 va4[2][NAI], vb4[2][NBI];
        float dsum[3] = {0.f, 0.f, 0.f};
        typedef unsigned int bu32x4 __attribute__((ext_vector_type(4)));
        auto vissue = [&](auto set_c, int t) __attribute__((always_inline)) {
            constexpr int SS_ = decltype(set_c)::value;
            const int so = min(t, nst - 1) * (BK * 4);            // uniform byte offset of the k-step
#pragma unroll
            for (int i = 0; i < NAI; ++i) va4[SS_][i] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsA, voA[i], so, 0));
#pragma unroll
            for (int i = 0; i < NBI; ++i) vb4[SS_][i] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(i < BH ? rsK : rsM, voB[i], so, 0));
        };
        auto vstash = [&](auto set_c, int buf, int part) __attribute__((always_inline)) {      // part 0: the A rows, part 1: W_k | W_m
            constexpr int SS_ = decltype(set_c)::value;
            if (part == 0) {
#pragma unroll
                for (int i = 0; i < NAI; ++i) *(f32x4*)(fa + buf * AR * P + (arow4 + 8 * i) * P + 4 * quad) = va4[SS_][i];
            } else {
#pragma unroll
                for (int i = 0; i < NBI; ++i) *(f32x4*)(fb + buf * BR * P + (trow + RP * i) * P + 4 * quad) = vb4[SS_][i];
            }
        };
        // DIST: sum (v_o - v_k + 1e-6)^2 over this thread's column quad (cx.py:300) for row (lane >> 3) + 8 i of the wave's triplet; one item per sub-step
        // (all three in the sub-step that stores the rows put 26 vector operations behind its last MFMA: 5 790 cycles per k-step against 5 085)
        auto vdist = [&](auto set_c, int i0, int i1) __attribute__((always_inline)) {
            constexpr int SS_ = decltype(set_c)::value;
            if constexpr (DIST) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (i < i0 || i >= i1) continue;
                    const f32x4 df = va4[SS_][3] - va4[SS_][i] + 1e-6f;
                    dsum[i] += (df[0] * df[0] + df[1] * df[1]) + (df[2] * df[2] + df[3] * df[3]);
                }
            }
        };
        f32x4 acc4[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        struct Frag { f32x2 a[2], wk[4], wm[4], vo; };
        Frag x0, x1;
#pragma unroll
        for (int i = 0; i < 2; ++i) x1.a[i] = f32x2{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) { x1.wk[j] = f32x2{0.f, 0.f}; x1.wm[j] = f32x2{0.f, 0.f}; }
        x1.vo = f32x2{0.f, 0.f};
        const int vo_row = kr == 24 ? 32 * wave + 24 : 64 * (wave >> 1) + 48;       // wave w: LDS rows 32 w .. 32 w + 31; its triplet's v_o row
        auto fread = [&](int buf, int s, Frag& x) __attribute__((always_inline)) {
            const float* a = fa + buf * AR * P + (32 * wave + li) * P + 8 * s + 2 * lk;
            const float* b = fb + buf * BR * P + li * P + 8 * s + 2 * lk;
#pragma unroll
            for (int i = 0; i < 2; ++i) x.a[i] = *(const f32x2*)(a + i * 16 * P);
            x.vo = *(const f32x2*)(fa + buf * AR * P + vo_row * P + 8 * s + 2 * lk);
#pragma unroll
            for (int j = 0; j < 4; ++j) { x.wk[j] = *(const f32x2*)(b + j * 16 * P); x.wm[j] = *(const f32x2*)(b + (64 + j * 16) * P); }
        };
        auto fmfma = [&](const Frag& x) __attribute__((always_inline)) {
            // (the two k of a fragment pair as ONE v_pk_fma_f32: vector instructions are not free under fp32 MFMAs -- each costs the matrix pipe ~10 cycles,
            // round 4 -- and this loop had 32 of them per k-step; same fma per element, same bits)
            f32x2 bw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bw[j] = __builtin_elementwise_fma(x.vo, x.wm[j], x.wk[j]);      // (hipcc splits most of these back into two v_fma_f32; as four v_pk_fma_f32 in an asm block: 4 871 cycles per k-step against 5 085 without the wait states an MFMA reading them needs -- wrong results -- and 5 530 with them)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.a[i][e], bw[j][e], acc4[i][j], 0, 0, 0);
        };
        typedef IntC<0> V0; typedef IntC<1> V1;
        vissue(V0{}, 0);
        vissue(V1{}, 1);
        vstash(V0{}, 0, 0); vstash(V0{}, 0, 1); vdist(V0{}, 0, 3);
        vissue(V0{}, 2);
        __syncthreads();
        auto vstep = [&](auto par_c, auto last_c, int t) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_c)::value;
            constexpr bool LAST = decltype(last_c)::value;
            typedef IntC<PAR ^ 1> SS;
            fread(PAR, 0, x0);
            fmfma(x1);                                                   // (t-1, last sub-step)
#pragma unroll
            for (int q = 0; q < 16; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            if (LAST && pf_next) {                                       // the next (plain) segment's first tiles
                const MainSeg& nx = args.seg[1];
                setup(IntC<K1 < 0 ? 0 : K1>{}, nx);
                const int kn = nx.klen, nn = (kn + BK - 1) / BK;
                issue(S0{}, Ff{}, kn, nn, 0);
                issue(S1{}, Ff{}, kn, nn, 1);
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                Frag& cur = (s & 1) ? x1 : x0;
                Frag& nxt = (s & 1) ? x0 : x1;
                fread(PAR, s + 1, nxt);
                if (!LAST) {
                    if (s == 0) { vstash(SS{}, PAR ^ 1, 0); vdist(SS{}, 0, 1); }
                    if (s == 1) { vstash(SS{}, PAR ^ 1, 1); vdist(SS{}, 1, 2); }
                    if (s == 2) { vdist(SS{}, 2, 3); vissue(SS{}, t + 3); }
                }
                fmfma(cur);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, DIST ? 4 : 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        };
        int t = 0;
        for (; t + 2 < nst; t += 2) { vstep(V0{}, Ff{}, t); vstep(V1{}, Ff{}, t + 1); }
        if (t + 1 < nst) { vstep(V0{}, Ff{}, t); vstep(V1{}, Tt{}, t + 1); }
        else vstep(V0{}, Tt{}, t);
        fmfma(x1);                                                       // the last sub-step
        if constexpr (DIST) {                                            // the 8 threads of a row hold its partial sums: reduce, sqrt, park in LDS for the next segment's loader
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float d2 = dsum[i];
                d2 += __shfl_xor(d2, 1, 64); d2 += __shfl_xor(d2, 2, 64); d2 += __shfl_xor(d2, 4, 64);
                if (quad == 0) mf_smem[FOLD_DIST_OFF + 24 * wave + (lane >> 3) + 8 * i] = sqrtf(d2);      // compact tile row (K = 24)
            }
        }
        // padded (wave = triplet, 32 rows x 64 columns) -> compact (the 2 x 2 wave layout of the segments that follow), through LDS
        constexpr int CP = 68;
        float* const fc = mf_smem;                                       // [128 padded rows][CP]   (every fragment read is complete: last barrier)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) fc[(32 * wave + 16 * i + 4 * lk + q) * CP + 16 * j + li] = acc4[i][j][q];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rho = wm0 + 16 * i + 4 * lk + q;               // compact tile row 0 .. 95 = triplet rho / kr, candidate rho % kr
                    acc[i][j][q] = fc[(blk * (rho / kr) + rho % kr) * CP + wn0 + 16 * j + li];
                }
        __syncthreads();
      }
    };

    // ---- MK_VFOLD under X6 (192 x 64 tiles, K = 24: eight triplets, wave w = triplet w x all 64 columns) --------------------------------
    // As run_vfold4, on the bf16 matrix path: the v_k rows are cut into three bf16 planes when they are stored to LDS (24 compact rows per
    // triplet; a wave's second block row runs 8 rows into its neighbour's -- those outputs are dropped); W_k, W_m and v_o stay fp32 in LDS, and
    // the effective-weight fragment of a column block is built per wave on the vector ALU,
    //     b = fma(v_o[w][k], W_m[n][k], W_k[n][k])     (the expression of the fp32 forms)
    // then cut into its three planes in registers: 8 fma + 44 split / pack operations per lane and column block, under the 12 MFMAs of the
    // block before it.  The last column block's MFMAs of a step run after the barrier, over the first reads and the first build of the next.
    auto run_vfold6 = [&](const MainSeg& sg, const bool pf_next) __attribute__((always_inline)) {
      if constexpr (X6) {
        constexpr int ARC = 208;                                   // A rows per plane: 192 + the 8 the last wave's second block row reads (+ 8 spare)
        constexpr int A_PLX = ARC * P6, VO_OFF = 3 * A_PLX, W_OFF = VO_OFF + 8 * 32 * 4, BUFX = W_OFF + 128 * P * 4;
        static_assert(2 * BUFX <= CFG::LDS, "fold tile (X6)");
        static_assert(RP == 64, "fold loader (X6)");
        const int klen = sg.klen, nst = klen / BK;
        // loader items of thread (trow, quad): 0 .. 2 = v_k row trow + 64 i of the tile (triplet rho / 24, candidate rho % 24); ONE float of v_o (triplet
        // (tid >> 5) & 7, column tid & 31: the 8 x 32 floats of the step twice over -- 2 KB of returning loads instead of the 8 KB of a quad per thread);
        // 4 / 5 = row trow of W_k / W_m
        // (buffer loads: descriptor + constant lane offset + scalar step offset, no vector instruction per load -- as in run_vfold4)
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.a), 0, 0xFFFFFFF0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.b), 0, 0xFFFFFFF0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.b2), 0, 0xFFFFFFF0u, 0x00020000);
        typedef unsigned int bu32x4 __attribute__((ext_vector_type(4)));
        unsigned voA[3], voB;
#pragma unroll
        for (int i = 0; i < 3; ++i) voA[i] = (unsigned)((giptr)sg.idx)[min(m0 + trow + 64 * i, M - 1)] * (unsigned)(sg.lda * 4) + 16u * quad;
        const int vo_t = (tid >> 5) & 7, vo_c = tid & 31;
        const unsigned voV = (unsigned)((giptr)sg.idx2)[min(m0 + 24 * vo_t, M - 1)] * (unsigned)(sg.lda * 4) + 4u * vo_c;
        voB = (unsigned)min(n0 + trow, N - 1) * (unsigned)(sg.ldb * 4) + 16u * quad;
        f32x4 va4[2][3], vb4[2][2];
        float vv[2];
        auto vissue = [&](auto set_c, int t) __attribute__((always_inline)) {
            constexpr int SS_ = decltype(set_c)::value;
            const int so = min(t, nst - 1) * (BK * 4);
#pragma unroll
            for (int i = 0; i < 3; ++i) va4[SS_][i] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsA, voA[i], so, 0));
            vv[SS_] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsA, voV, so, 0));
            vb4[SS_][0] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsK, voB, so, 0));
            vb4[SS_][1] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsM, voB, so, 0));
        };
        auto split3 = [&](const float (&x)[4], unsigned (&w1)[2], unsigned (&w2)[2], unsigned (&w3)[2]) __attribute__((always_inline)) {
            unsigned p1[4], p2[4], p3[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p1[j] = __builtin_bit_cast(unsigned, x[j]) & 0xFFFF0000u;
                const float r1 = x[j] - __builtin_bit_cast(float, p1[j]);
                p2[j] = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
                const float r2 = r1 - __builtin_bit_cast(float, p2[j]);
                p3[j] = __builtin_bit_cast(unsigned, r2);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                w1[h] = __builtin_amdgcn_perm(p1[2 * h + 1], p1[2 * h], 0x07060302u);
                w2[h] = __builtin_amdgcn_perm(p2[2 * h + 1], p2[2 * h], 0x07060302u);
                w3[h] = __builtin_amdgcn_perm(p3[2 * h + 1], p3[2 * h], 0x07060302u);
            }
        };
        auto vstash = [&](auto set_c, int buf, int part) __attribute__((always_inline)) {      // part 0: v_k items 0, 1; part 1: v_k item 2 + v_o; part 2: W_k | W_m
            constexpr int SS_ = decltype(set_c)::value;
            unsigned char* const bb = lds6 + buf * BUFX;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if ((i < 2) != (part == 0) || part == 2) continue;
                const f32x4 v = va4[SS_][i];
                const float x[4] = {v[0], v[1], v[2], v[3]};
                unsigned w1[2], w2[2], w3[2];
                split3(x, w1, w2, w3);
                unsigned char* d = bb + (trow + 64 * i) * P6 + 8 * quad;
                *(mu32x2*)(d) = mu32x2{w1[0], w1[1]}; *(mu32x2*)(d + A_PLX) = mu32x2{w2[0], w2[1]}; *(mu32x2*)(d + 2 * A_PLX) = mu32x2{w3[0], w3[1]};
            }
            if (part == 1) *(float*)(bb + VO_OFF + (vo_t * 32 + vo_c) * 4) = vv[SS_];
            if (part == 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i) *(f32x4*)(bb + W_OFF + ((trow + 64 * i) * P + 4 * quad) * 4) = vb4[SS_][i];
            }
        };
        f32x4 acc4[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        struct Raw { f32x4 wk[2], wm[2]; };                          // W_k / W_m [n = 16 j + li][k = 8 lk .. 8 lk + 7] of one column block
        const mbf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
        mbf16x8 af[2][2][3];                                        // [set][block row][plane]: A fragments of a step (set = parity of the step)
        mbf16x8 bq[2][3];                                           // effective-weight planes of a column block, two in flight
        f32x4 vo[2];                                                // v_o[w][8 lk .. 8 lk + 7]
#pragma unroll
        for (int p = 0; p < 3; ++p) { af[1][0][p] = z8; af[1][1][p] = z8; bq[1][p] = z8; }
        auto read_af = [&](int buf, mbf16x8 (&f)[2][3]) __attribute__((always_inline)) {
            const unsigned char* a = lds6 + buf * BUFX + (24 * wave + li) * P6 + 16 * lk;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) f[i][p] = *(const mbf16x8*)(a + i * 16 * P6 + p * A_PLX);
        };
        auto read_vo = [&](int buf) __attribute__((always_inline)) {
            const unsigned char* v = lds6 + buf * BUFX + VO_OFF + (wave * 32 + 8 * lk) * 4;
            vo[0] = *(const f32x4*)(v); vo[1] = *(const f32x4*)(v + 16);
        };
        auto read_w = [&](int buf, int j, Raw& r) __attribute__((always_inline)) {
            const unsigned char* b = lds6 + buf * BUFX + W_OFF + ((16 * j + li) * P + 8 * lk) * 4;
            r.wk[0] = *(const f32x4*)(b); r.wk[1] = *(const f32x4*)(b + 16);
            r.wm[0] = *(const f32x4*)(b + 64 * P * 4); r.wm[1] = *(const f32x4*)(b + 64 * P * 4 + 16);
        };
        auto build = [&](const Raw& r, mbf16x8 (&o)[3]) __attribute__((always_inline)) {
            unsigned w1[4], w2[4], w3[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float v_ = vo[h][e], m_ = r.wm[h][e], k_ = r.wk[h][e]; x[e] = __builtin_fmaf(v_, m_, k_); }
                unsigned a1[2], a2[2], a3[2];
                split3(x, a1, a2, a3);
                w1[2 * h] = a1[0]; w1[2 * h + 1] = a1[1]; w2[2 * h] = a2[0]; w2[2 * h + 1] = a2[1]; w3[2 * h] = a3[0]; w3[2 * h + 1] = a3[1];
            }
            typedef unsigned int mu32x4 __attribute__((ext_vector_type(4)));
            o[0] = __builtin_bit_cast(mbf16x8, mu32x4{w1[0], w1[1], w1[2], w1[3]});
            o[1] = __builtin_bit_cast(mbf16x8, mu32x4{w2[0], w2[1], w2[2], w2[3]});
            o[2] = __builtin_bit_cast(mbf16x8, mu32x4{w3[0], w3[1], w3[2], w3[3]});
        };
        // the 12 MFMAs of column block j: the two block rows alternate; small terms first
        auto fmfma = [&](const mbf16x8 (&a)[2][3], const mbf16x8 (&b)[3], int j) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[2], acc4[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[1], acc4[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][2], b[0], acc4[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[1], acc4[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[0], acc4[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[0], acc4[i][j], 0, 0, 0);
        };
        auto pin6 = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        typedef IntC<0> V0; typedef IntC<1> V1;
        vissue(V0{}, 0);
        vissue(V1{}, 1);
        vstash(V0{}, 0, 0); vstash(V0{}, 0, 1); vstash(V0{}, 0, 2);
        vissue(V0{}, 2);
        __syncthreads();
        // step t (parity PAR): af[PAR ^ 1] / bq[1] come in holding the A fragments and column block 3 of step t - 1 (zeros at the start)
        auto vstep = [&](auto par_c, auto last_c, int t) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_c)::value;
            constexpr bool LAST = decltype(last_c)::value;
            typedef IntC<PAR ^ 1> SS;
            Raw r0, r1;
            read_vo(PAR);
            read_w(PAR, 0, r0);
            read_af(PAR, af[PAR]);
            read_w(PAR, 1, r1);
            fmfma(af[PAR ^ 1], bq[1], 3);                                // (t - 1, column block 3)
            build(r0, bq[0]);
            pin6();
            if (LAST && pf_next) {                                       // the next (plain) segment's first tiles
                const MainSeg& nx = args.seg[1];
                setup(IntC<K1 < 0 ? 0 : K1>{}, nx);
                const int kn = nx.klen, nn = (kn + BK - 1) / BK;
                issue(S0{}, Ff{}, kn, nn, 0);
                issue(S1{}, Ff{}, kn, nn, 1);
            }
            // column block 0 under the build of 1, 1 under 2, 2 under 3
            read_w(PAR, 2, r0);
            if (!LAST) vstash(SS{}, PAR ^ 1, 0);
            fmfma(af[PAR], bq[0], 0);
            build(r1, bq[1]);
            pin6();
            read_w(PAR, 3, r1);
            if (!LAST) vstash(SS{}, PAR ^ 1, 1);
            fmfma(af[PAR], bq[1], 1);
            build(r0, bq[0]);
            pin6();
            if (!LAST) { vstash(SS{}, PAR ^ 1, 2); vissue(SS{}, t + 3); }
            fmfma(af[PAR], bq[0], 2);
            build(r1, bq[1]);
            pin6();
            __syncthreads();
        };
        int t = 0;
        for (; t + 2 < nst; t += 2) { vstep(V0{}, Ff{}, t); vstep(V1{}, Ff{}, t + 1); }
        if (t + 1 < nst) { vstep(V0{}, Ff{}, t); vstep(V1{}, Tt{}, t + 1); fmfma(af[1], bq[1], 3); }
        else { vstep(V0{}, Tt{}, t); fmfma(af[0], bq[1], 3); }
        // padded (wave = triplet, 32 rows x 64 columns) -> compact (the 4 x 2 wave layout of the segments that follow), through LDS
        constexpr int CP = 68;
        float* const fc = mf_smem;                                       // [8 x 32 padded rows][CP]   (every fragment read is complete: last barrier)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) fc[(32 * wave + 16 * i + 4 * lk + q) * CP + 16 * j + li] = acc4[i][j][q];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rho = wm0 + 16 * i + 4 * lk + q;               // compact tile row 0 .. 191 = triplet rho / 24, candidate rho % 24
                    acc[i][j][q] = fc[(32 * (rho / 24) + rho % 24) * CP + wn0 + 16 * j + li];
                }
        __syncthreads();
      }
    };

    // segment I runs its steps [lo, hi) = [g0, g1) intersected with the segment; the first one that has any loads its own
    // first tiles, every later one finds them loaded by its predecessor's last step
    bool started = false;
    int base = 0;
    auto seg_pass = [&](auto idx_c) __attribute__((always_inline)) {
        constexpr int I = decltype(idx_c)::value;
        const int nst = nsteps[I];
        const int lo = min(max(g0 - base, 0), nst), hi = min(max(g1 - base, 0), nst);
        base += nst;
        if (lo < hi) {
            if (!started) {
                const MainSeg& g = args.seg[I];
                setup(IntC<KS[I]>{}, g);
                typedef std::integral_constant<bool, KS[I] == MK_GATHER_MUL> MI;
                issue(S0{}, MI{}, g.klen, nst, lo);
                if (DEPTH == 2) issue(S1{}, MI{}, g.klen, nst, lo + 1);
                started = true;
            }
            run_seg(idx_c, lo, hi, g1 > base);           // (g1 > base: the range continues into the next segment)
        }
        stamp(1 + I);
    };
    if constexpr (VFOLD) {
        if constexpr (X6) {
            run_vfold6(args.seg[0], NSEG > 1);
            const mbf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};       // (here, not at the top: 72 registers of zeros would live through the fold)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    xa[q][p] = z8;
#pragma unroll
                    for (int j = 0; j < WN; ++j) xb[q][j][p] = z8;
                }
        }
        else if constexpr (BM % 96 == 0) run_vfold4(args.seg[0], NSEG > 1);
        else run_vfold(args.seg[0], NSEG > 1);
        base = nsteps[0]; started = true;
        stamp(1);
    } else {
        seg_pass(IntC<0>{});
    }
    if constexpr (NSEG > 1) seg_pass(IntC<1>{});
    if constexpr (NSEG > 2) seg_pass(IntC<2>{});
    if constexpr (NSEG > 3) seg_pass(IntC<3>{});
    if constexpr (NSEG > 4) seg_pass(IntC<4>{});
    if constexpr (!X6) mfma(afB, bfB);                   // the last sub-step of the last segment (X6: every segment finishes its own last block row)

    // ---- epilogue: + Sh[r / K] (+ bias), ReLU, Dropout, store ---------------------------------------------------------
    // Every operand is fetched behind one uniform test per kind (a test per element puts each load in its own basic block:
    // 24 serialised round trips); r / K of a lane's 4 consecutive rows comes from ONE division.
    // The finished tile goes to memory through LDS.  The accumulator layout gives every lane 4 rows x 1 column per block, so an
    // in-register epilogue is WM * WN * 4 unrolled copies of (row add, bias, ReLU, dropout hash, scalar store): 6-12 KB of code
    // that each workgroup runs once -- in-kernel stamps put it at 13-17k cycles, mostly instruction fetch.  Instead the raw
    // accumulators are staged in LDS (WM * WN * 4 ds_write_b32) and ONE rolled loop over 16-byte row pieces applies the
    // epilogue with 16-byte operand loads and stores whole rows.
    constexpr int SP = BN + 4;
    float* const stage = mf_smem;                        // [BM][SP]  (every fragment read is complete: the last step's barrier)
    constexpr int QR = BN / 4;                            // 16-byte pieces per tile row
    const EpiArgs& e = args.epi;
    float* const dst = S > 1 ? args.slab + (long long)z * M * N : args.out;      // S > 1: raw partial sums; k_main_fixup runs the epilogue
    const long long ldd = S > 1 ? (long long)N : args.ldo;
    const bool plain = S > 1 || !(e.rowadd || e.bias || e.relu || e.dropout || e.gate);
    // (the row-add operand of piece f + 256 is fetched while piece f is finished: rolled, but never waiting on its own load)
    auto item = [&](int f, int& r, int& n, int& nl) __attribute__((always_inline)) {
        const int rr = f / QR, cq = f - rr * QR;
        r = m0 + rr; n = n0 + 4 * cq;
        nl = min(n, N - 4 >= 0 ? N - 4 : 0);                               // operand window slid left at the right edge (N >= 4)
        return rr * SP + 4 * cq;
    };
    auto win = [&](const float* p, int nl, int sh) __attribute__((always_inline)) {   // p[n .. n+3] (zero beyond N) from a 16-byte load at nl
        const f32x4 w = *(const f32x4u*)(p + nl);
        f32x4 o;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) o[jj] = jj + sh < 4 ? (sh == 0 ? w[jj] : sh == 1 ? w[(jj + 1) & 3] : sh == 2 ? w[(jj + 2) & 3] : w[(jj + 3) & 3]) : 0.f;
        return o;
    };
    const bool use_add = !plain && e.rowadd;
    const int rdiv = use_add ? e.rowdiv : 1;
    // The row-add operand (Sh[r / K]) of piece f + 3 x MT is requested while piece f is finished: a FIFO of three (round 4).  With one
    // piece of lookahead (rounds 2-3) every iteration of this rolled loop waited out most of an L2 round trip for the load the iteration
    // before it had issued: in-kernel stamps put the epilogue at ~25 k cycles for six pieces per thread, ~10 us of a 270 us workgroup.
    auto fetch_add = [&](int f) __attribute__((always_inline)) -> f32x4 {
        int r2, n2, nl2; item(f, r2, n2, nl2);
        const int n2c = min(n2, N - 1), nl2c = min(nl2, n2c);
        return win(e.rowadd + (long long)(min(r2, M - 1) / rdiv) * e.ld_rowadd, nl2c, n2c - nl2c);
    };
    constexpr int LAST_F = BM * QR - 1;
    f32x4 add0 = {0.f, 0.f, 0.f, 0.f}, add1 = add0, add2 = add0;
    if (use_add) { add0 = fetch_add(min(tid, LAST_F)); add1 = fetch_add(min(tid + MT, LAST_F)); add2 = fetch_add(min(tid + 2 * MT, LAST_F)); }
    // (the three requests above are in flight while the accumulators go through LDS)
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) stage[(wm0 + 16 * i + 4 * lk + q) * SP + wn0 + 16 * j + li] = acc[i][j][q];
    __syncthreads();
#pragma unroll 1
    for (int f = tid; f < BM * QR; f += MT) {
        int r, n, nl;
        const int so = item(f, r, n, nl);
        const f32x4 addv = add0;
        add0 = add1; add1 = add2;
        if (use_add) add2 = fetch_add(min(f + 3 * MT, LAST_F));
        if (r >= M || n >= N) continue;
        f32x4 v = *(const f32x4*)(stage + so);
        const int sh = n - nl;
        if (!plain) {
            f32x4 add = addv, msk = {1.f, 1.f, 1.f, 1.f}, gte = {1.f, 1.f, 1.f, 1.f};
            if (e.bias) add += win(e.bias, nl, sh);
            if (e.dropout == 2) msk = win(e.keep_mask + (long long)r * e.ld_mask, nl, sh);
            if (e.gate) gte = win(e.gate + (long long)r * e.ld_gate, nl, sh);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float x = v[jj] + add[jj];
                if (e.relu == 1) x = x > 0.f ? x : 0.f;
                else if (e.relu == 2) x = tanhf(x);
                if (e.dropout == 1)
                    x = dropout_keep(e.seed_lo, e.seed_hi, e.layer, (unsigned long long)r * (unsigned)N + (unsigned)(n + jj), e.drop_p) ? x * e.drop_scale : 0.f;
                else if (e.dropout == 2)
                    x = msk[jj] != 0.f ? x * e.drop_scale : 0.f;
                if (e.gate) x = gte[jj] > 0.f ? x * e.gate_scale : 0.f;
                v[jj] = x;
            }
        }
        float* o = dst + (long long)r * ldd + n;
        if (n + 3 < N) *(f32x4u*)o = v;
        else { o[0] = v[0]; if (n + 1 < N) o[1] = v[1]; if (n + 2 < N) o[2] = v[2]; }
    }
    stamp(8);
    if (stamps && tid == 0) stamps[15] = __builtin_amdgcn_s_memrealtime();
}

// out[r][n] = epilogue(sum_z slab[z][r][n]): 4 consecutive n per thread (N % 4 != 0: scalar tail), z ascending (deterministic)
__global__ __launch_bounds__(256) void k_main_fixup(const MainArgs a) {
    const int M = a.M, N = a.N, S = a.split;
    const int nq = (N + 3) / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * nq) return;
    const int r = (int)(i / nq), n0 = (int)(i - (long long)r * nq) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const long long mn = (long long)M * N, off = (long long)r * N + n0;
    for (int z = 0; z < S; ++z)
#pragma unroll
        for (int j = 0; j < 4; ++j) if (n0 + j < N) v[j] += a.slab[z * mn + off + j];
    const EpiArgs& e = a.epi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + j;
        if (n >= N) break;
        a.out[(long long)r * a.ldo + n] = apply_epilogue(e, v[j], r, n, N);
    }
}

// Operand requirements of the kernel (see the header comment): klen >= 4 and a multiple of 4; weight rows readable (zero
// padded) up to the next multiple of 32 columns.
static inline bool main_fwd_operand_ok(int klen) { return klen >= 4 && klen % 4 == 0; }

template <int K0, int... REST> struct FirstKind { static constexpr int value = K0; };

template <class CFG, bool DIST, int... KINDS>
static inline int launch_main_fwd_seq(MainArgs& a, hipStream_t s) {
    constexpr int lds = FirstKind<KINDS...>::value == MK_VFOLD ? (CFG::LDS_FOLD > CFG::LDS ? CFG::LDS_FOLD : CFG::LDS) : CFG::LDS;
    static DevMask attr{0};
    NCX_HIP_TRY(set_max_lds_once(attr, (const void*)k_main_fwd<CFG, DIST, KINDS...>, lds));
    const int tiles_m = (a.M + CFG::BM - 1) / CFG::BM, tiles_n = (a.N + CFG::BN - 1) / CFG::BN;
    const int S = a.split > 1 ? a.split : 1;
    const int grid = ((tiles_m * S + 7) / 8) * 8 * tiles_n;
    hipLaunchKernelGGL((k_main_fwd<CFG, DIST, KINDS...>), dim3(grid), dim3(CFG::T), lds, s, a);
    NCX_HIP_TRY(hipGetLastError());
    if (S > 1) {
        const long long nthreads = (long long)a.M * ((a.N + 3) / 4);
        hipLaunchKernelGGL(k_main_fixup, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, a);
        NCX_HIP_TRY(hipGetLastError());
    }
    return NCX_OK;
}

// The five operand-kind sequences the library uses (anything else: NCX_E_FLAGS).
template <class CFG>
static inline int launch_main_fwd(MainArgs& a, hipStream_t s) {
    if (a.nseg < 1 || a.nseg > MAIN_MAX_SEG || a.M < 1 || a.N < 1) return NCX_E_DIMS;
    for (int i = 0; i < a.nseg; ++i) if (!main_fwd_operand_ok(a.seg[i].klen)) return NCX_E_DIMS;
    if (a.epi.rowadd && a.epi.rowdiv < 3) return NCX_E_DIMS;                        // (the epilogue's row -> triplet map assumes K >= 3)
    if (a.split > 1 && !a.slab) return NCX_E_WORKSPACE;
    auto is = [&](std::initializer_list<int> ks) { if ((int)ks.size() != a.nseg) return false; int i = 0; for (int k : ks) if (a.seg[i++].kind != k) return false; return true; };
    constexpr int G = MK_GATHER, X = MK_GATHER_MUL, P = MK_PLAIN, S = MK_SOFTMAX;
    constexpr int V = MK_VFOLD;
    if (a.seg[0].kind == V) {                // the per-triplet fold of the two v segments: its own tile shape
        if constexpr (CFG::BM == 48 && CFG::BN == 64 && CFG::WGM == 1 && CFG::DEPTH == 2) {
            if (a.split > 1 || a.seg[0].klen % MF_BK || a.seg[0].klen < 2 * MF_BK || !a.epi.rowadd || a.epi.rowdiv != 24) return NCX_E_FLAGS;
            if (a.dist_out) {
                if (is({V, P, P, S})) return launch_main_fwd_seq<CFG, true, V, P, P, S>(a, s);
                if (is({V, P, P, P})) return launch_main_fwd_seq<CFG, true, V, P, P, P>(a, s);
            } else {
                if (is({V, P, P, S})) return launch_main_fwd_seq<CFG, false, V, P, P, S>(a, s);
                if (is({V, P, P, P})) return launch_main_fwd_seq<CFG, false, V, P, P, P>(a, s);
            }
        }
        if constexpr (CFG::BM % 96 == 0 && CFG::BN == 64 && CFG::WGM * CFG::WGN == CFG::T / 64 && CFG::DEPTH == 2) {      // one triplet per wave (run_vfold4): 96 x 64 / 4 waves, 192 x 64 / 8 waves
            if (a.split > 1 || a.seg[0].klen % MF_BK || a.seg[0].klen < 2 * MF_BK || !a.epi.rowadd || (a.epi.rowdiv != 24 && a.epi.rowdiv != 48))
                return NCX_E_FLAGS;
            if (a.dist_out) {                        // the pairwise distance inside the fold: a triplet per wave (K = 24), fp32 forms
                if constexpr (CFG::X6) return NCX_E_FLAGS;
                else {
                    if (a.epi.rowdiv != 24) return NCX_E_FLAGS;
                    if (is({V, P, P, S})) return launch_main_fwd_seq<CFG, true, V, P, P, S>(a, s);
                    if (is({V, P, P, P})) return launch_main_fwd_seq<CFG, true, V, P, P, P>(a, s);
                    return NCX_E_FLAGS;
                }
            }
            if (is({V, P, P, S})) return launch_main_fwd_seq<CFG, false, V, P, P, S>(a, s);
            if (is({V, P, P, P})) return launch_main_fwd_seq<CFG, false, V, P, P, P>(a, s);
        }
        return NCX_E_FLAGS;
    }
    if constexpr (CFG::X6) return NCX_E_FLAGS;       // (the X6 form exists for the fold sequences only)
    else {
    if (a.dist_out) {                        // (in-kernel pairwise distance: whole k range in one workgroup, no slid windows in the v rows)
        if (a.split > 1 || a.nseg < 3 || a.seg[1].klen % MF_BK) return NCX_E_FLAGS;
        if (is({G, X, P, P, S})) return launch_main_fwd_seq<CFG, true, G, X, P, P, S>(a, s);
        if (is({G, X, P, P, P})) return launch_main_fwd_seq<CFG, true, G, X, P, P, P>(a, s);
        return NCX_E_FLAGS;
    }
    if (is({G, X, P, P, S})) return launch_main_fwd_seq<CFG, false, G, X, P, P, S>(a, s);
    if (is({P}))             return launch_main_fwd_seq<CFG, false, P>(a, s);
    if (is({G}))             return launch_main_fwd_seq<CFG, false, G>(a, s);          // (the MUTAN producer's x_v: one gathered segment)
    if (is({G, P, P, S}))    return launch_main_fwd_seq<CFG, false, G, P, P, S>(a, s);
    if (is({G, X, P, P, P})) return launch_main_fwd_seq<CFG, false, G, X, P, P, P>(a, s);
    if (is({G, P, P, P}))    return launch_main_fwd_seq<CFG, false, G, P, P, P>(a, s);
    if (is({P, P}))          return launch_main_fwd_seq<CFG, false, P, P>(a, s);
    return NCX_E_FLAGS;
    }
}

}  // namespace ncx
