// ncx_dwtn.hip -- the row-reduction (TN) weight-gradient products of linear_1 that are NOT the per-triplet fold (ncx_dwkm.hip), as ONE
// balanced launch of 8-wave workgroups (round 4):
//
//   dGt[h][a]             = sum_r dpre[r][h] * softmax(a_knns)[r][a]          (M = B K rows; what the answer_embedding gradient waits for)
//   dW1[:, z_other][h][c]  = sum_r dpre[r][h] * z_knns[r][c]
//   dW1[:, dist|rank][h][c] = sum_r dpre[r][h] * misc[r][c]
//   dW1[:, shared][h][c]   = sum_b dSh[b][h] * [v_o | q | z_o | E[aid]][b][c]   (B rows)
//
// Reference: the gradients autograd produces for vqa/models/cx.py:309-322 (same products the generic engine's grouped launch computed).
//
// Why a kernel of its own.  On the generic engine (128 x 64 tiles, 4 waves, two workgroups per CU) the dGt problem alone ran at 0.64 of
// the fp32-MFMA peak and the short problems behind it added 61 us for 4.3 GF (tools/exp_dw1c_split.py): 175 us for the launch.  Here
//   * a workgroup is 8 waves on a 256 x 64 tile = ALL of H (one workgroup per CU): every operand row is fetched, transformed (softmax)
//     and stored to LDS once per 256 output rows instead of once per 128; 64 MFMAs per wave between two barriers; two register sets of
//     global loads in flight (a load has 1.5 k-steps to land), the LDS stores of the next tile and the loads of the one after spread
//     under the MFMAs of the current one, the last sub-step's MFMAs issued after the barrier (tools/mb/mb_tn.hip: dGt alone
//     125 -> 109 us; LDS-DMA staging measured there too: slower, 121 us);
//   * the launch is BALANCED: workgroup w owns one aligned chunk of a dGt tile (S row chunks per tile, tiles x S = the grid) and then the
//     range [w R, (w + 1) R) of the k-steps of all other tiles laid end to end (unaligned: a range may cover the tail of one tile and
//     the head of the next; every piece goes to its own slab slot, the reduction that follows sums a tile's pieces in ascending
//     workgroup order: deterministic).  The operand loader follows the workgroup's step sequence across piece boundaries (its
//     descriptors come from a table in LDS, one step ahead), so a switch costs an accumulator flush, not a pipeline restart.
//   * The chunking of the dGt part does not depend on whether the rest is launched with it, and vice versa: the phased backward
//     (ncx_backward_phase 5 | 2 | 4) is bit-identical to the whole one.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "ncx_internal.h"
#include "ncx_dwred.h"

namespace ncx {

constexpr int TN8_PA = 256, TN8_PB = 80;                       // LDS pitches (floats): ds_read_b128 / ds_read_b64 fragment reads without conflicts
constexpr int TN8_MAX_SEG = 16;                                // pieces per workgroup (host-checked)
constexpr int TN8_SEG_WORDS = 16;

typedef const __attribute__((address_space(1))) float* tn_gfptr;
typedef const __attribute__((address_space(1))) f32x4u* tn_gf4ptr;

struct Tn8Args {
    Tn8Prob p[TN8_MAX_PROB];
    int np, n_al;                     // problems [0, n_al) are "aligned" (n_al <= 1), [n_al, np) make up the rest sequence
    int H, tiles_m, B;
    int S, al_wgs, al_tiles_n;        // aligned problem: S row chunks per tile, al_wgs = tiles x S workgroups
    int R;                            // rest k-steps per workgroup
    int do_al, do_rest;
    int rest_tiles[TN8_MAX_PROB], rest_steps[TN8_MAX_PROB], rest_tile0[TN8_MAX_PROB];   // per problem: tiles (row tiles x column tiles), k-steps per tile, first global rest tile
    int rest_pre[TN8_MAX_PROB + 1];   // prefix sums of tiles x steps over the rest problems (rest_pre[np] = all rest k-steps)
    const int* idx_ob; const int* aid;
    float* slab;                      // [slot][256][64]
    int n_slots;
};

// LDS image of one piece of a workgroup's step sequence (16 dwords)
struct Tn8Seg {
    unsigned long long a_ptr;         // A + row-tile offset (floats are added per step)
    unsigned long long x_ptr;         // X
    unsigned long long l_ptr;         // lse (or any readable address when soft == 0)
    int ldx;                          // floats
    int soft;                         // 1: x = exp2(X log2e - lse[r])
    int gsel;                         // 0: row r; 1 / 2: row idx table 0 / 1 in LDS
    int n0, ncl;                      // first column of the tile; N - 4 (the 16-byte windows are clamped into the row)
    int t0;                           // first k-step of the piece
    int vbeg, vend;                   // the piece's range in the workgroup's virtual step sequence
    int slot;                         // slab slot of the partial tile
    int rv;                           // operand rows that exist (X rows beyond: index clamped; A is zero there)
};
static_assert(sizeof(Tn8Seg) == TN8_SEG_WORDS * 4, "segment record");

// this workgroup's pieces (thread 0): the aligned chunk, then its range of the rest sequence
__device__ __forceinline__ void tn8_pieces(const Tn8Args& a, int w, Tn8Seg* segs, int* nseg_out, int* vtot_out) {
    constexpr int BM = TN8_BM, BN = TN8_BN, BK = TN8_BK;
    int n = 0, v = 0;
    auto put = [&](int prob, int tile, int t0, int t1, int slot) {
        const Tn8Prob& p = a.p[prob];
        const int tm = tile % a.tiles_m, tn = tile / a.tiles_m;
        Tn8Seg sg;
        sg.a_ptr = (unsigned long long)(uintptr_t)(p.A + (long long)tm * BM);
        sg.x_ptr = (unsigned long long)(uintptr_t)p.X;
        sg.l_ptr = (unsigned long long)(uintptr_t)(p.lse ? p.lse : p.A);
        sg.ldx = (int)p.ldx; sg.soft = p.lse ? 1 : 0; sg.gsel = p.gsel;
        sg.n0 = tn * BN; sg.ncl = p.N - 4; sg.t0 = t0; sg.vbeg = v; sg.vend = v + (t1 - t0); sg.slot = slot; sg.rv = p.rows_valid > 0 ? p.rows_valid : p.rows;
        if (n < TN8_MAX_SEG) segs[n++] = sg;
        v += t1 - t0;
    };
    if (a.do_al && w < a.al_wgs) {
        const int tile = w / a.S, z = w - tile * a.S;
        const int steps = a.p[0].rows / BK;
        const int g0 = (int)((long long)steps * z / a.S), g1 = (int)((long long)steps * (z + 1) / a.S);
        if (g1 > g0) put(0, tile, g0, g1, w);
    }
    if (a.do_rest) {
        long long g = (long long)w * a.R;
        const long long gend = min(g + a.R, (long long)a.rest_pre[a.np]);
        int pr = a.n_al;
        while (g < gend) {
            while (pr + 1 < a.np && g >= a.rest_pre[pr + 1]) ++pr;
            const int local = (int)(g - a.rest_pre[pr]);
            const int tile = local / a.rest_steps[pr], t0 = local - tile * a.rest_steps[pr];
            const int t1 = (int)min((long long)a.rest_steps[pr], t0 + (gend - g));
            put(pr, tile, t0, t1, a.al_wgs + a.rest_tile0[pr] + tile + w);
            g += t1 - t0;
        }
    }
    *nseg_out = n; *vtot_out = v;
}

// SPLIT: only the aligned problem is a softmax operand (what backward_impl launches): the transform of a tile is then known from its position in the
// workgroup's sequence -- exp2 for the aligned chunk, nothing for the rest -- and the per-element select (a compare + four v_cndmask per k-step, and the unused
// fma / exp2 of plain pieces) leaves the step.
template <bool SPLIT>
__global__ __launch_bounds__(TN8_T, 1) void k_dw_tn8(const Tn8Args a) {
    constexpr int T = TN8_T, BM = TN8_BM, BN = TN8_BN, BK = TN8_BK, PA = TN8_PA, PB = TN8_PB;
    constexpr int WM = 4, WN = 2, NA = 4, NMF = 2 * WM * WN;
    extern __shared__ __attribute__((aligned(16))) float tn_smem[];
    float* const lds_a = tn_smem;                               // [2][32][PA]
    float* const lds_b = tn_smem + 2 * BK * PA;                 // [2][32][PB]
    Tn8Seg* const segs = (Tn8Seg*)(tn_smem + 2 * BK * (PA + PB));          // [TN8_MAX_SEG]
    int* const lds_idx = (int*)(segs + TN8_MAX_SEG);            // [2][B + 32]: idx_ob, answer ids (row gathers of the per-triplet problems; 32 spare entries: the loader reads one step ahead)
    const int IB = a.B + 32;
    __shared__ int s_nseg, s_vtot;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = 64 * (wave >> 1), wn0 = 32 * (wave & 1);
    const int w = blockIdx.x;

    if (tid == 0) tn8_pieces(a, w, segs, &s_nseg, &s_vtot);
    if (a.do_rest) {                                            // row-gather tables of the per-triplet problems
        for (int i = tid; i < IB; i += T) { const int ii = min(i, a.B - 1); lds_idx[i] = a.idx_ob ? a.idx_ob[ii] : 0; lds_idx[IB + i] = a.aid ? a.aid[ii] : 0; }
    }
    __syncthreads();
    const int nseg = s_nseg, V = s_vtot;
    if (V == 0) return;

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- loader: follows the virtual step sequence; its piece record travels one issue ahead -------------------------------------------
    const int arow = tid >> 6, aq = tid & 63;                   // A item i: tile row arow + 8 i, 16-byte quad aq
    const int brow = tid >> 4, bq = tid & 15;                   // X item: tile row brow, quad bq
    f32x4 va[2][NA], vb[2];
    float vl[2], vs[2];                                         // lse of the X item's row; 1 = softmax piece
    // Vector instructions are not hidden under fp32 MFMAs (DESIGN S5d: each costs the matrix pipe ~10 cycles), and the loader that recomputed every address from
    // the piece record cost ~25 of them per k-step.  So an issue is `uniform base + 32-bit lane offset`, the bases advanced by scalar arithmetic (issue_fast, the
    // only loader code inside a step); whenever the NEXT issue enters a new piece, needs the row clamp (the last rows of a padded problem) or runs past the
    // sequence, loader_switch() re-derives the state from the piece record BETWEEN two steps (a uniform branch in the loop, not in the step).  Same addresses,
    // same values as before: results are bit-identical.
    int lq = 0, l_vend = 0;                                     // uniform: piece of the last switch; first virtual step the advanced state is NOT valid for
    // (buffer loads: descriptor = the piece's operand base, voffset = the lane's constant offset, soffset = the step's uniform offset: no vector instruction per load)
    // (the one-float lse load keeps its pointer: a third descriptor did not stay in scalar registers and hipcc wrapped that load in a waterfall loop)
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0x00020000), rsX = rsA;
    const char* lB = nullptr;
    unsigned aO = 0, xO = 0;                                    // uniform byte offsets of the next issue
    unsigned aS = 0, xS = 0, lS = 0, ldx4 = 0;                  // uniform strides per k-step (bytes); row pitch of the gathered operand (bytes)
    int gi = 0, gS = 0;                                         // per-lane index into the gather tables for the issue after next; its uniform stride (0: no gather)
    float vsP = 0.f;
    unsigned offA[NA], offX = 0, offL = 0;                      // per-lane byte offsets
#pragma unroll
    for (int i = 0; i < NA; ++i) offA[i] = (unsigned)(((arow + 8 * i) * a.H + 4 * aq) * 4);
    int rgn = 0;                                                // gathered row of the next issue
    auto rfl = [&](int x) __attribute__((always_inline)) -> int { return __builtin_amdgcn_readfirstlane(x); };
    auto rfl64 = [&](unsigned long long x) __attribute__((always_inline)) -> unsigned long long {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x);
    };
    auto loader_switch = [&](int v) {                // make the state valid for the issue of virtual step v
        const int vv = min(v, V - 1);                            // (virtual steps beyond the sequence re-load the last tile; never consumed)
        int q = lq;
        while (q + 1 < nseg && vv >= segs[q].vend) ++q;
        q = rfl(q);
        const Tn8Seg sg = segs[q];
        const unsigned long long a_ptr = rfl64(sg.a_ptr), x_ptr = rfl64(sg.x_ptr), l_ptr = rfl64(sg.l_ptr);
        const int ldx = rfl(sg.ldx), soft = rfl(sg.soft), gsel = rfl(sg.gsel), n0 = rfl(sg.n0), ncl = rfl(sg.ncl), t0 = rfl(sg.t0), vbeg = rfl(sg.vbeg),
                  vend = rfl(sg.vend), rv = rfl(sg.rv);
        const int t = t0 + (vv - vbeg);
        const long long r0 = (long long)t * BK;
        const bool clamp = !gsel && r0 + BK > rv;                // some row of THIS step lies beyond the rows that exist: per-lane row clamp, one step
        const int col = min(n0 + 4 * bq, ncl), r = (int)r0 + brow;
        lq = q; vsP = soft ? 1.f : 0.f; ldx4 = gsel ? (unsigned)ldx * 4u : 0u;             // (ldx4 = 0: the gathered-row term of the X offset vanishes)
        // (operand extents are below 4 GiB: host-checked)
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)a_ptr, 0, 0xFFFFFFF0u, 0x00020000);
        rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)x_ptr, 0, 0xFFFFFFF0u, 0x00020000);
        aO = (unsigned)(r0 * a.H * 4);
        if (gsel)       { xO = 0u;                         offX = (unsigned)(col * 4); }
        else if (clamp) { xO = 0u;                         offX = (unsigned)(((long long)min(r, rv - 1) * ldx + col) * 4); }
        else            { xO = (unsigned)(r0 * ldx * 4);   offX = (unsigned)((brow * ldx + col) * 4); }
        if (!soft)      { lB = (const char*)(uintptr_t)l_ptr;            offL = 0u; }
        else if (clamp) { lB = (const char*)(uintptr_t)l_ptr;            offL = (unsigned)(min(r, rv - 1) * 4); }
        else            { lB = (const char*)(uintptr_t)l_ptr + r0 * 4;   offL = (unsigned)(brow * 4); }
        if (v >= V - 1) {                                        // the last tile of the sequence, and every issue after it: stay
            aS = 0; xS = 0; lS = 0; l_vend = 0x7fffffff;
        } else {
            aS = (unsigned)(BK * a.H * 4); xS = (gsel || clamp) ? 0u : (unsigned)(BK * ldx * 4); lS = (soft && !clamp) ? (unsigned)(BK * 4) : 0u;
            // valid until the piece ends, the first step that needs the clamp, or the last step of the sequence (which switches to "stay")
            int ve = min(vend, V - 1);
            if (!gsel && (long long)(t0 + (vend - vbeg)) * BK > (long long)rv) ve = min(ve, vbeg + (rv / BK - t0));      // (step rv / BK is the first with a row beyond rv)
            l_vend = clamp ? v + 1 : max(ve, v + 1);
        }
        // (the "stay" state re-reads the same table entry: an index that kept advancing past the 32 spare entries named rows that do not exist)
        gi = gsel ? (gsel == 2 ? IB : 0) + min((int)r0 + brow, a.B - 1) : 0; gS = (gsel && v < V - 1) ? BK : 0;
        rgn = lds_idx[gi]; gi += gS;
    };
    auto issue_fast = [&](auto set_c) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        typedef unsigned int bu32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < NA; ++i) va[S][i] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsA, offA[i], aO, 0));
        const unsigned xo = __umul24((unsigned)rgn, ldx4) + offX;
        vb[S] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsX, xo, xO, 0));
        vl[S] = *(tn_gfptr)(lB + offL);
        vs[S] = vsP;
        aO += aS; xO += xS; lB += lS;
        rgn = lds_idx[gi]; gi += gS;                             // (always executed; used only by gathered pieces, whose tables have 32 spare entries)
    };
    auto issue = [&](auto set_c, int v) __attribute__((always_inline)) {        // (prologue)
        if (v >= l_vend) loader_switch(v);
        issue_fast(set_c);
    };
    // MODE 0: select per element by the set's softmax flag; 1: the tile is a softmax operand; 2: it is a plain one
    auto stash = [&](auto set_c, auto mode_c, int buf, int h0, int h1) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value, MODE = decltype(mode_c)::value;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i < h0 || i >= h1) continue;
            *(f32x4*)(lds_a + buf * BK * PA + (arow + 8 * i) * PA + 4 * aq) = va[S][i];
        }
        if (NA >= h0 && NA < h1) {
            f32x4 v = vb[S], e;
            if (MODE != 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[j], 1.44269504088896341f, -vl[S]));
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = MODE == 1 ? e[j] : (vs[S] != 0.f ? e[j] : v[j]);
            }
            *(f32x4*)(lds_b + buf * BK * PB + brow * PB + 4 * bq) = v;
        }
    };
    f32x4 afA[2], afB[2];                 // [e]: the 4 interleaved A blocks of a 16-byte read
    f32x2 bfA[2], bfB[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) { afB[e] = f32x4{0.f, 0.f, 0.f, 0.f}; bfB[e] = f32x2{0.f, 0.f}; }
    // Interleaved block mapping: MFMA block q of a wave owns tile rows {wm0 + 4 i + q} (columns {wn0 + 2 j + q}), so one ds_read_b128
    // (ds_read_b64) per k-row feeds all four (two) blocks.  k order inside a step: MFMA (s, e) takes k = 8 s + 2 lk + e from lane group lk.
    auto read_frags = [&](int buf, int s, f32x4 (&af)[2], f32x2 (&bf)[2]) __attribute__((always_inline)) {
        const int kk = 8 * s + 2 * lk;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            af[e] = *(const f32x4*)(lds_a + buf * BK * PA + (kk + e) * PA + wm0 + 4 * li);
            bf[e] = *(const f32x2*)(lds_b + buf * BK * PB + (kk + e) * PB + wn0 + 2 * li);
        }
    };
    auto mfma = [&](const f32x4 (&af)[2], const f32x2 (&bf)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e][i], bf[e][j], acc[i][j], 0, 0, 0);
    };
    typedef std::integral_constant<int, 0> S0; typedef std::integral_constant<int, 1> S1;
    issue(S0{}, 0);
    issue(S1{}, 1);
    stash(S0{}, std::integral_constant<int, 0>{}, 0, 0, NA + 1);
    issue(S0{}, 2);
    __syncthreads();
    // the piece being multiplied
    int qc = 0, vend_c = segs[0].vend, slot_c = segs[0].slot;
    auto flush = [&]() __attribute__((always_inline)) {          // partial tile -> its slab slot; lane holds, per (block q, reg), 2 consecutive columns
        float* const slot = a.slab + (long long)slot_c * (BM * BN);
#pragma unroll
        for (int q = 0; q < WM; ++q)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int row = wm0 + 4 * (4 * lk + rg) + q;
                const f32x2 v = {acc[q][0][rg], acc[q][1][rg]};
                *(f32x2*)(slot + row * BN + wn0 + 2 * li) = v;
                acc[q][0][rg] = 0.f; acc[q][1][rg] = 0.f;
            }
    };
    auto step = [&](auto par_c, auto mode_c, int v) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        typedef decltype(mode_c) MD;
        typedef std::integral_constant<int, PAR ^ 1> SS;
        read_frags(PAR, 0, afA, bfA);
        mfma(afB, bfB);                                  // (v - 1, last sub-step; zeros after a flush): covers the reads above
#pragma unroll
        for (int q = 0; q < NMF; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            auto& afc = (s & 1) ? afB : afA; auto& bfc = (s & 1) ? bfB : bfA;
            auto& afn = (s & 1) ? afA : afB; auto& bfn = (s & 1) ? bfA : bfB;
            read_frags(PAR, s + 1, afn, bfn);
            if (s == 0) stash(SS{}, MD{}, PAR ^ 1, 0, 2);
            if (s == 1) stash(SS{}, MD{}, PAR ^ 1, 2, NA + 1);
            if (s == 2) issue_fast(SS{});                     // (virtual step v + 3: the loop made the state valid for it)
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
        if (v + 1 == vend_c) {                           // the piece ends here (uniform): last sub-step, partial tile out, next piece
            mfma(afB, bfB);
#pragma unroll
            for (int e = 0; e < 2; ++e) { afB[e] = f32x4{0.f, 0.f, 0.f, 0.f}; bfB[e] = f32x2{0.f, 0.f}; }
            flush();
            ++qc;
            if (qc < nseg) { vend_c = segs[qc].vend; slot_c = segs[qc].slot; }
        }
    };
    typedef std::integral_constant<int, 0> M0; typedef std::integral_constant<int, 1> M1; typedef std::integral_constant<int, 2> M2;
    auto pair = [&](auto mode_c, int v) __attribute__((always_inline)) {
        if (v + 3 >= l_vend) loader_switch(v + 3);
        step(S0{}, mode_c, v);
        if (v + 4 >= l_vend) loader_switch(v + 4);
        step(S1{}, mode_c, v + 1);
    };
    int v = 0;
    if constexpr (SPLIT) {
        // step v stores tile v + 1: a softmax operand while v + 1 < nal (the aligned chunk's steps), a plain one from v = nal - 1 on; the loops hand over
        // at even v (the LDS buffer parity), the 0 or 2 steps in between select per element
        const int nal = (a.do_al && w < a.al_wgs && segs[0].slot == w) ? segs[0].vend : 0;       // (the aligned chunk's slab slot is the workgroup id: no rest piece has it)
        const int nA = nal > 1 ? (nal - 1) & ~1 : 0, nB = nal & ~1;
        for (; v + 1 < V && v < nA; v += 2) pair(M1{}, v);
        for (; v + 1 < V && v < nB; v += 2) pair(M0{}, v);
        for (; v + 1 < V; v += 2) pair(M2{}, v);
        if (v < V) { if (v + 3 >= l_vend) loader_switch(v + 3); step(S0{}, M0{}, v); }
    } else {
        for (; v + 1 < V; v += 2) pair(M0{}, v);
        if (v < V) { if (v + 3 >= l_vend) loader_switch(v + 3); step(S0{}, M0{}, v); }
    }
}


// ---- the same launch on the bf16 matrix path with fp32-grade operands (NCX_F_X6; not the default) -----------------------------------------
// Every operand element is cut into three bf16 values by truncation, x = x1 + x2 + x3 EXACTLY (x1 = x & 0xFFFF0000, x2 = (x - x1) & 0xFFFF0000,
// x3 = x - x1 - x2; every residual is exact in fp32), when it is stored to LDS (once per element, as in the fp32 kernel); the six products
// a1 x1 + (a1 x2 + a2 x1) + (a1 x3 + a2 x2 + a3 x1) run on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (products of bf16 values are exact in
// fp32): what is dropped is 2^-24 relative, the rounding error of one fp32 operation.  6 MFMAs of 16 cycles replace 8 of 32 per 16 x 16 x 32
// block.  LDS holds three bf16 planes per operand in the global [k][m] layout; fragments come from ds_read_b64_tr_b16 (hardware transpose);
// row pitches = 32 mod 256 bytes.  Measured on the dGt problem alone (tools/mb/mb_tn.hip): 80 us against 109, same error against fp64 (1.0e-9 on
// values of 1.9e-3); the kernel is then bound by LDS + global-load return traffic (~3 000 cycles per k-step against 1 536 of MFMA), not by the
// matrix cores.  Pieces, slab slots and the reduction are those of k_dw_tn8; the block mapping inside a wave is the plain one (block (i, j) =
// rows wm0 + 16 i .., columns wn0 + 16 j ..), so the summation order differs from the fp32 kernel's: results agree to fp32 rounding, not bitwise.
typedef __bf16 tn_bf16x8 __attribute__((ext_vector_type(8)));
typedef short tn_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int tn_u32x2 __attribute__((ext_vector_type(2)));
constexpr int TN6_PA = 544, TN6_PB = 160;                                    // bytes per k-row of one plane (512 + 32, 128 + 32)
constexpr int TN6_PL = TN8_BK * (TN6_PA + TN6_PB), TN6_BUF = 3 * TN6_PL;    // one plane (A rows | X rows), one buffer

template <bool SPLIT>
__global__ __launch_bounds__(TN8_T, 1) void k_dw_tn8_x6(const Tn8Args a) {
    constexpr int T = TN8_T, BM = TN8_BM, BN = TN8_BN, BK = TN8_BK, PA = TN6_PA, PB = TN6_PB, PL = TN6_PL, BUF = TN6_BUF, A_PL = BK * PA;
    constexpr int NA = 4;
    extern __shared__ __attribute__((aligned(1024))) unsigned char tn6_smem[];
    Tn8Seg* const segs = (Tn8Seg*)(tn6_smem + 2 * BUF);          // [TN8_MAX_SEG]
    int* const lds_idx = (int*)(segs + TN8_MAX_SEG);            // [2][B + 32]
    const int IB = a.B + 32;
    __shared__ int s_nseg, s_vtot;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = 64 * (wave >> 1), wn0 = 32 * (wave & 1);
    const int w = blockIdx.x;
    if (tid == 0) tn8_pieces(a, w, segs, &s_nseg, &s_vtot);
    if (a.do_rest) {
        for (int i = tid; i < IB; i += T) { const int ii = min(i, a.B - 1); lds_idx[i] = a.idx_ob ? a.idx_ob[ii] : 0; lds_idx[IB + i] = a.aid ? a.aid[ii] : 0; }
    }
    __syncthreads();
    const int nseg = s_nseg, V = s_vtot;
    if (V == 0) return;

    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- loader: as in k_dw_tn8 ---------------------------------------------------------------------------------------------------------
    const int arow = tid >> 6, aq = tid & 63;
    const int brow = tid >> 4, bq = tid & 15;
    f32x4 va[2][NA], vb[2];
    float vl[2], vs[2];
    // The loader of k_dw_tn8 (this kernel is bound by its vector instructions and its LDS / load-return traffic: the loader that recomputed every address from
    // the piece record cost ~25 vector instructions per k-step).  So an issue is `uniform base + 32-bit lane offset`, the bases advanced by scalar arithmetic (issue_fast, the
    // only loader code inside a step); whenever the NEXT issue enters a new piece, needs the row clamp (the last rows of a padded problem) or runs past the
    // sequence, loader_switch() re-derives the state from the piece record BETWEEN two steps (a uniform branch in the loop, not in the step).  Same addresses,
    // same values as before: results are bit-identical.
    int lq = 0, l_vend = 0;                                     // uniform: piece of the last switch; first virtual step the advanced state is NOT valid for
    // (buffer loads: descriptor = the piece's operand base, voffset = the lane's constant offset, soffset = the step's uniform offset: no vector instruction per load)
    // (the one-float lse load keeps its pointer: a third descriptor did not stay in scalar registers and hipcc wrapped that load in a waterfall loop)
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0x00020000), rsX = rsA;
    const char* lB = nullptr;
    unsigned aO = 0, xO = 0;                                    // uniform byte offsets of the next issue
    unsigned aS = 0, xS = 0, lS = 0, ldx4 = 0;                  // uniform strides per k-step (bytes); row pitch of the gathered operand (bytes)
    int gi = 0, gS = 0;                                         // per-lane index into the gather tables for the issue after next; its uniform stride (0: no gather)
    float vsP = 0.f;
    unsigned offA[NA], offX = 0, offL = 0;                      // per-lane byte offsets
#pragma unroll
    for (int i = 0; i < NA; ++i) offA[i] = (unsigned)(((arow + 8 * i) * a.H + 4 * aq) * 4);
    int rgn = 0;                                                // gathered row of the next issue
    auto rfl = [&](int x) __attribute__((always_inline)) -> int { return __builtin_amdgcn_readfirstlane(x); };
    auto rfl64 = [&](unsigned long long x) __attribute__((always_inline)) -> unsigned long long {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x);
    };
    auto loader_switch = [&](int v) {                // make the state valid for the issue of virtual step v
        const int vv = min(v, V - 1);                            // (virtual steps beyond the sequence re-load the last tile; never consumed)
        int q = lq;
        while (q + 1 < nseg && vv >= segs[q].vend) ++q;
        q = rfl(q);
        const Tn8Seg sg = segs[q];
        const unsigned long long a_ptr = rfl64(sg.a_ptr), x_ptr = rfl64(sg.x_ptr), l_ptr = rfl64(sg.l_ptr);
        const int ldx = rfl(sg.ldx), soft = rfl(sg.soft), gsel = rfl(sg.gsel), n0 = rfl(sg.n0), ncl = rfl(sg.ncl), t0 = rfl(sg.t0), vbeg = rfl(sg.vbeg),
                  vend = rfl(sg.vend), rv = rfl(sg.rv);
        const int t = t0 + (vv - vbeg);
        const long long r0 = (long long)t * BK;
        const bool clamp = !gsel && r0 + BK > rv;                // some row of THIS step lies beyond the rows that exist: per-lane row clamp, one step
        const int col = min(n0 + 4 * bq, ncl), r = (int)r0 + brow;
        lq = q; vsP = soft ? 1.f : 0.f; ldx4 = gsel ? (unsigned)ldx * 4u : 0u;             // (ldx4 = 0: the gathered-row term of the X offset vanishes)
        // (operand extents are below 4 GiB: host-checked)
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)a_ptr, 0, 0xFFFFFFF0u, 0x00020000);
        rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)x_ptr, 0, 0xFFFFFFF0u, 0x00020000);
        aO = (unsigned)(r0 * a.H * 4);
        if (gsel)       { xO = 0u;                         offX = (unsigned)(col * 4); }
        else if (clamp) { xO = 0u;                         offX = (unsigned)(((long long)min(r, rv - 1) * ldx + col) * 4); }
        else            { xO = (unsigned)(r0 * ldx * 4);   offX = (unsigned)((brow * ldx + col) * 4); }
        if (!soft)      { lB = (const char*)(uintptr_t)l_ptr;            offL = 0u; }
        else if (clamp) { lB = (const char*)(uintptr_t)l_ptr;            offL = (unsigned)(min(r, rv - 1) * 4); }
        else            { lB = (const char*)(uintptr_t)l_ptr + r0 * 4;   offL = (unsigned)(brow * 4); }
        if (v >= V - 1) {                                        // the last tile of the sequence, and every issue after it: stay
            aS = 0; xS = 0; lS = 0; l_vend = 0x7fffffff;
        } else {
            aS = (unsigned)(BK * a.H * 4); xS = (gsel || clamp) ? 0u : (unsigned)(BK * ldx * 4); lS = (soft && !clamp) ? (unsigned)(BK * 4) : 0u;
            // valid until the piece ends, the first step that needs the clamp, or the last step of the sequence (which switches to "stay")
            int ve = min(vend, V - 1);
            if (!gsel && (long long)(t0 + (vend - vbeg)) * BK > (long long)rv) ve = min(ve, vbeg + (rv / BK - t0));      // (step rv / BK is the first with a row beyond rv)
            l_vend = clamp ? v + 1 : max(ve, v + 1);
        }
        // (the "stay" state re-reads the same table entry: an index that kept advancing past the 32 spare entries named rows that do not exist)
        gi = gsel ? (gsel == 2 ? IB : 0) + min((int)r0 + brow, a.B - 1) : 0; gS = (gsel && v < V - 1) ? BK : 0;
        rgn = lds_idx[gi]; gi += gS;
    };
    auto issue_fast = [&](auto set_c) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        typedef unsigned int bu32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < NA; ++i) va[S][i] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsA, offA[i], aO, 0));
        const unsigned xo = __umul24((unsigned)rgn, ldx4) + offX;
        vb[S] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsX, xo, xO, 0));
        vl[S] = *(tn_gfptr)(lB + offL);
        vs[S] = vsP;
        aO += aS; xO += xS; lB += lS;
        rgn = lds_idx[gi]; gi += gS;                             // (always executed; used only by gathered pieces, whose tables have 32 spare entries)
    };
    auto issue = [&](auto set_c, int v) __attribute__((always_inline)) {        // (prologue)
        if (v >= l_vend) loader_switch(v);
        issue_fast(set_c);
    };
    // x -> three bf16 planes (four values: one 8-byte store per plane)
    auto split_store = [&](f32x4 v, unsigned char* base) __attribute__((always_inline)) {
        unsigned p1[4], p2[4], p3[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xj = v[j];               // (a scalar copy first: __builtin_bit_cast applied to the vector element itself reads element 0 -- hipcc 7.2)
            p1[j] = __builtin_bit_cast(unsigned, xj) & 0xFFFF0000u;
            const float r1 = xj - __builtin_bit_cast(float, p1[j]);
            p2[j] = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
            const float r2 = r1 - __builtin_bit_cast(float, p2[j]);
            p3[j] = __builtin_bit_cast(unsigned, r2);            // (at most 8 significant bits are left: the high half holds them all)
        }
        // v_perm_b32: the high halves of two dwords -> one dword (the even element in the low half)
        const tn_u32x2 w1 = {__builtin_amdgcn_perm(p1[1], p1[0], 0x07060302u), __builtin_amdgcn_perm(p1[3], p1[2], 0x07060302u)};
        const tn_u32x2 w2 = {__builtin_amdgcn_perm(p2[1], p2[0], 0x07060302u), __builtin_amdgcn_perm(p2[3], p2[2], 0x07060302u)};
        const tn_u32x2 w3 = {__builtin_amdgcn_perm(p3[1], p3[0], 0x07060302u), __builtin_amdgcn_perm(p3[3], p3[2], 0x07060302u)};
        *(tn_u32x2*)(base) = w1; *(tn_u32x2*)(base + PL) = w2; *(tn_u32x2*)(base + 2 * PL) = w3;
    };
    auto stash = [&](auto set_c, auto mode_c, int buf, int h0, int h1) __attribute__((always_inline)) {      // MODE: as k_dw_tn8 (0 select, 1 softmax, 2 plain)
        constexpr int S = decltype(set_c)::value, MODE = decltype(mode_c)::value;
        unsigned char* const base = tn6_smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i < h0 || i >= h1) continue;
            split_store(va[S][i], base + (arow + 8 * i) * PA + aq * 8);
        }
        if (NA >= h0 && NA < h1) {
            f32x4 v = vb[S], e;
            if (MODE != 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float xj = v[j]; e[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(xj, 1.44269504088896341f, -vl[S])); }
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float xj = v[j], ej = e[j]; v[j] = MODE == 1 ? ej : (vs[S] != 0.f ? ej : xj); }
            }
            split_store(v, base + A_PL + brow * PB + bq * 8);
        }
    };
    typedef __attribute__((address_space(3))) tn_s16x4* lds_s16x4;
    const int tq = li >> 2, tp = li & 3;
    const int fragA = (4 * lk + tq) * PA + (wm0 + 4 * tp) * 2, fragB = A_PL + (4 * lk + tq) * PB + (wn0 + 4 * tp) * 2;
    auto read_a = [&](int buf, int i, tn_bf16x8 (&af)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned char* ta = tn6_smem + buf * BUF + p * PL + fragA + i * 32;
            const tn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ta));
            const tn_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ta + 16 * PA));
            af[p] = __builtin_bit_cast(tn_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    auto read_b = [&](int buf, tn_bf16x8 (&bf)[3][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned char* tb = tn6_smem + buf * BUF + p * PL + fragB + j * 32;
                const tn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tb));
                const tn_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tb + 16 * PB));
                bf[p][j] = __builtin_bit_cast(tn_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
    };
    // the 12 MFMAs of block row i: the two column blocks alternate (a dependent MFMA is two issues away); small terms first
    auto mfma12 = [&](const tn_bf16x8 (&af)[3], const tn_bf16x8 (&bf)[3][2], f32x4 (&c)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[2][j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[1][j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bf[0][j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[1][j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[0][j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[0][j], c[j], 0, 0, 0);
    };
    auto pin = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> S0; typedef std::integral_constant<int, 1> S1;
    tn_bf16x8 afA[3], afB[3], bfr[2][3][2];
    const tn_bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        afB[p] = zero8;
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[1][p][j] = zero8;
    }
    issue(S0{}, 0);
    issue(S1{}, 1);
    stash(S0{}, std::integral_constant<int, 0>{}, 0, 0, NA + 1);
    issue(S0{}, 2);
    __syncthreads();
    int qc = 0, vend_c = segs[0].vend, slot_c = segs[0].slot;
    auto flush = [&]() __attribute__((always_inline)) {          // partial tile -> its slab slot (lane: row 4 lk + rg of block row i, column li of block column j)
        float* const slot = a.slab + (long long)slot_c * (BM * BN);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int row = wm0 + 16 * i + 4 * lk + rg;
#pragma unroll
                for (int j = 0; j < 2; ++j) { slot[row * BN + wn0 + 16 * j + li] = acc[i][j][rg]; acc[i][j][rg] = 0.f; }
            }
    };
    auto step = [&](auto par_c, auto mode_c, int v) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        typedef decltype(mode_c) MD;
        typedef std::integral_constant<int, PAR ^ 1> SS;
        // block row 3 of step v - 1 (zeros at the start and after a flush) over the first reads of this buffer
        read_b(PAR, bfr[PAR]);
        read_a(PAR, 0, afA);
        mfma12(afB, bfr[PAR ^ 1], acc[3]);
#pragma unroll
        for (int q = 0; q < 12; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
        __builtin_amdgcn_sched_barrier(0);
        read_a(PAR, 1, afB); stash(SS{}, MD{}, PAR ^ 1, 0, 2); mfma12(afA, bfr[PAR], acc[0]); pin();
        read_a(PAR, 2, afA); stash(SS{}, MD{}, PAR ^ 1, 2, NA + 1); mfma12(afB, bfr[PAR], acc[1]); pin();
        read_a(PAR, 3, afB); issue_fast(SS{}); mfma12(afA, bfr[PAR], acc[2]); pin();       // (virtual step v + 3: the loop made the state valid for it)
        __syncthreads();
        if (v + 1 == vend_c) {                           // the piece ends here (uniform): block row 3, partial tile out, next piece
            mfma12(afB, bfr[PAR], acc[3]);
#pragma unroll
            for (int p = 0; p < 3; ++p) afB[p] = zero8;
            flush();
            ++qc;
            if (qc < nseg) { vend_c = segs[qc].vend; slot_c = segs[qc].slot; }
        }
    };
    typedef std::integral_constant<int, 0> M0; typedef std::integral_constant<int, 1> M1; typedef std::integral_constant<int, 2> M2;
    auto pair = [&](auto mode_c, int v) __attribute__((always_inline)) {
        if (v + 3 >= l_vend) loader_switch(v + 3);
        step(S0{}, mode_c, v);
        if (v + 4 >= l_vend) loader_switch(v + 4);
        step(S1{}, mode_c, v + 1);
    };
    int v = 0;
    if constexpr (SPLIT) {
        const int nal = (a.do_al && w < a.al_wgs && segs[0].slot == w) ? segs[0].vend : 0;
        const int nA = nal > 1 ? (nal - 1) & ~1 : 0, nB = nal & ~1;
        for (; v + 1 < V && v < nA; v += 2) pair(M1{}, v);
        for (; v + 1 < V && v < nB; v += 2) pair(M0{}, v);
        for (; v + 1 < V; v += 2) pair(M2{}, v);
        if (v < V) { if (v + 3 >= l_vend) loader_switch(v + 3); step(S0{}, M0{}, v); }
    } else {
        for (; v + 1 < V; v += 2) pair(M0{}, v);
        if (v < V) { if (v + 3 >= l_vend) loader_switch(v + 3); step(S0{}, M0{}, v); }
    }
}


// ---- host ------------------------------------------------------------------------------------------------------------------------------
static inline int tn8_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// NCX_F_X6: the launch runs on the bf16 matrix path with three-plane operands (k_dw_tn8_x6); the row-gather tables share LDS with six planes
constexpr int TN6_MAX_B = 2048;
bool dw_tn8_x6(const ncx_dims& d) { return (d.flags & NCX_F_X6) && d.B <= TN6_MAX_B && !hook_env("NCX_NO_X6"); }
bool dw_tn8_supported(const ncx_dims& d) { return !(d.flags & NCX_F_BF16) && dw_tn8_shapes_ok(d); }
// the bf16 variant keeps the per-triplet shared segments of linear_1 in fp32: their weight gradient takes this kernel too (rest sequence only)
bool dw_tn8_shapes_ok(const ncx_dims& d) {
    if (hook_env("NCX_NO_TN8")) return false;
    const long long M = (long long)d.B * d.K;
    // all of H in 256-row tiles; whole 32-row k-steps for both reduction extents; 16-byte operand windows; the per-triplet row-gather
    // tables fit in LDS; enough k-steps per chunk to be worth a pipeline (below that the generic engine's plans are as good)
    if (d.H % TN8_BM != 0 || d.B % TN8_BK != 0 || M % TN8_BK != 0 || d.B > 4096 || d.B < 128) return false;
    if (d.dv % 4 || d.dq % 4 || d.dz % 4 || d.da % 4 || d.A % 4) return false;
    // the loader's buffer loads take 32-bit byte offsets: every operand (the feature table, the logits, dpre) below 4 GiB
    const long long lim = (1ll << 32) - 65536;
    if ((long long)d.n_img * d.dv * 4 >= lim || M * (long long)(d.A > d.da ? d.A : d.da) * 4 >= lim || M * (long long)d.H * 4 >= lim || (long long)d.A * d.da * 4 >= lim) return false;
    return true;
}

struct Tn8Plan { int grid, S, al_wgs, al_tiles, R, n_slots, rest_tiles_total; };
static Tn8Plan tn8_plan(const ncx_dims& d, const Tn8Prob* p, int np, int n_al, int* rest_tiles, int* rest_steps, int* rest_tile0, int* rest_pre) {
    Tn8Plan pl{};
    const int cus = num_cus(), tiles_m = d.H / TN8_BM;
    pl.grid = cus;
    if (n_al) {
        pl.al_tiles = tiles_m * tn8_cdiv(p[0].N, TN8_BN);
        const int steps = p[0].rows / TN8_BK;
        int S = cus / pl.al_tiles; if (S < 1) S = 1;
        while (S > 1 && steps / S < 8) --S;                        // at least 8 k-steps per chunk
        pl.S = S; pl.al_wgs = pl.al_tiles * S;
        if (pl.al_wgs > pl.grid) pl.grid = pl.al_wgs;
    } else { pl.S = 1; pl.al_wgs = 0; pl.al_tiles = 0; }
    long long g = 0; int t0 = 0;
    for (int i = 0; i < TN8_MAX_PROB; ++i) { rest_tiles[i] = 0; rest_steps[i] = 1; rest_tile0[i] = 0; rest_pre[i] = 0; }
    for (int i = n_al; i < np; ++i) {
        rest_tiles[i] = tiles_m * tn8_cdiv(p[i].N, TN8_BN); rest_steps[i] = p[i].rows / TN8_BK; rest_tile0[i] = t0; rest_pre[i] = (int)g;
        t0 += rest_tiles[i]; g += (long long)rest_tiles[i] * rest_steps[i];
    }
    for (int i = np; i <= TN8_MAX_PROB; ++i) rest_pre[i] = (int)g;
    for (int i = 0; i < n_al; ++i) rest_pre[i] = 0;
    pl.rest_tiles_total = t0;
    pl.R = g > 0 ? tn8_cdiv(g, pl.grid) : 1;
    pl.n_slots = pl.al_wgs + t0 + pl.grid;
    return pl;
}

// Slab bytes for the problems backward_impl hands to this kernel (worst case over the lesion flags: the a_other column block joins
// the rest sequence when the answer-embedding segment is lesioned)
size_t dw_tn8_slab_bytes(const ncx_dims& d) {
    if (!dw_tn8_shapes_ok(d)) return 0;
    const int cus = num_cus(), tiles_m = d.H / TN8_BM;
    const bool aemb = d.flags & NCX_F_A_EMB, bf16 = d.flags & NCX_F_BF16;
    long long al_wgs = 0;
    if (aemb && !bf16) { const int t = tiles_m * tn8_cdiv(d.A, TN8_BN); int S = cus / t; if (S < 1) S = 1; al_wgs = (long long)t * S; }
    long long rest_tiles = 0;
    const int cols[7] = {(aemb || bf16) ? 0 : d.da, bf16 ? 0 : d.dz, bf16 ? 0 : pad_to(d.K + 1, 4), d.dv, d.dq, d.dz, d.da};
    for (int i = 0; i < 7; ++i) if (cols[i]) rest_tiles += (long long)tiles_m * tn8_cdiv(cols[i], TN8_BN);
    const long long grid = al_wgs > cus ? al_wgs : cus;
    return (size_t)(al_wgs + rest_tiles + grid) * TN8_BM * TN8_BN * 4;
}

static int tn8_fill(const ncx_dims& d, const Tn8Prob* probs, int np, int n_al, bool do_al, bool do_rest, Tn8Args& a, Tn8ReduceArgs& r,
                    float* slab, size_t slab_bytes, Tn8Plan& pl) {
    if (np < 1 || np > TN8_MAX_PROB || n_al < 0 || n_al > 1 || n_al > np) return NCX_E_DIMS;
    for (int i = 0; i < np; ++i) {
        if (probs[i].rows % TN8_BK || probs[i].N < 4 || probs[i].N % 4 || probs[i].ldx % 4) return NCX_E_DIMS;
        if (probs[i].gsel && probs[i].rows != d.B) return NCX_E_DIMS;
        a.p[i] = probs[i]; r.p[i] = probs[i];
    }
    pl = tn8_plan(d, probs, np, n_al, a.rest_tiles, a.rest_steps, a.rest_tile0, a.rest_pre);
    if ((size_t)pl.n_slots * TN8_BM * TN8_BN * 4 > slab_bytes) return NCX_E_WORKSPACE;
    // pieces per workgroup: the aligned chunk + every tile boundary its rest range can cross
    int min_steps = 1 << 30;
    for (int i = n_al; i < np; ++i) min_steps = a.rest_steps[i] < min_steps ? a.rest_steps[i] : min_steps;
    if (np > n_al && 1 + pl.R / min_steps + 2 > TN8_MAX_SEG) return NCX_E_DIMS;
    a.np = np; a.n_al = n_al; a.H = d.H; a.tiles_m = d.H / TN8_BM; a.B = d.B;
    a.S = pl.S; a.al_wgs = pl.al_wgs; a.al_tiles_n = n_al ? tn8_cdiv(probs[0].N, TN8_BN) : 0; a.R = pl.R;
    a.do_al = do_al && n_al; a.do_rest = do_rest && np > n_al;
    a.slab = slab; a.n_slots = pl.n_slots;
    r.np = np; r.n_al = n_al; r.tiles_m = a.tiles_m; r.S = pl.S; r.al_wgs = pl.al_wgs; r.R = pl.R; r.do_al = a.do_al; r.do_rest = a.do_rest;
    for (int i = 0; i < TN8_MAX_PROB; ++i) { r.rest_tiles[i] = a.rest_tiles[i]; r.rest_steps[i] = a.rest_steps[i]; r.rest_tile0[i] = a.rest_tile0[i]; }
    for (int i = 0; i <= TN8_MAX_PROB; ++i) r.rest_pre[i] = a.rest_pre[i];
    r.al_tiles = pl.al_tiles; r.slab = slab;
    r.n_tiles_total = (a.do_al ? pl.al_tiles : 0) + (a.do_rest ? pl.rest_tiles_total : 0);
    return NCX_OK;
}

// Launch the products; the fixed-order sums are handed back in `red` (n_tiles_total == 0: nothing to reduce)
int dw_tn8_products(const ncx_dims& d, const Tn8Prob* probs, int np, int n_al, bool do_al, bool do_rest, const int* idx_ob, const int* aid,
                    float* slab, size_t slab_bytes, Tn8ReduceArgs* red, hipStream_t s) {
    Tn8Args a{}; Tn8ReduceArgs r{}; Tn8Plan pl;
    int rc = tn8_fill(d, probs, np, n_al, do_al, do_rest, a, r, slab, slab_bytes, pl);
    if (rc) return rc;
    *red = r;
    if (!a.do_al && !a.do_rest) { red->n_tiles_total = 0; return NCX_OK; }
    a.idx_ob = idx_ob; a.aid = aid;
    if (dw_tn8_x6(d)) {
        const int lds = 2 * TN6_BUF + TN8_MAX_SEG * (int)sizeof(Tn8Seg) + 2 * (d.B + 32) * 4;
        static DevMask attr6{0};
        bool split6 = !hook_env("NCX_TN8_NO_SPLIT");
        for (int i = 0; i < np; ++i) split6 = split6 && ((probs[i].lse != nullptr) == (i < n_al));
        static DevMask attr6s{0};
        if (split6) {
            NCX_HIP_TRY(set_max_lds_once(attr6s, (const void*)k_dw_tn8_x6<true>, 2 * TN6_BUF + TN8_MAX_SEG * (int)sizeof(Tn8Seg) + 2 * (TN6_MAX_B + 32) * 4));
            hipLaunchKernelGGL(k_dw_tn8_x6<true>, dim3(pl.grid), dim3(TN8_T), lds, s, a);
        } else {
            NCX_HIP_TRY(set_max_lds_once(attr6, (const void*)k_dw_tn8_x6<false>, 2 * TN6_BUF + TN8_MAX_SEG * (int)sizeof(Tn8Seg) + 2 * (TN6_MAX_B + 32) * 4));
            hipLaunchKernelGGL(k_dw_tn8_x6<false>, dim3(pl.grid), dim3(TN8_T), lds, s, a);
        }
        NCX_HIP_TRY(hipGetLastError());
        return NCX_OK;
    }
    const int lds = 2 * TN8_BK * (TN8_PA + TN8_PB) * 4 + TN8_MAX_SEG * (int)sizeof(Tn8Seg) + 2 * (d.B + 32) * 4;
    const int lds_max = 2 * TN8_BK * (TN8_PA + TN8_PB) * 4 + TN8_MAX_SEG * (int)sizeof(Tn8Seg) + 2 * (4096 + 32) * 4;
    bool split = !hook_env("NCX_TN8_NO_SPLIT");                  // the rest sequence holds plain operands only, the aligned problem (if any) the softmax one
    for (int i = 0; i < np; ++i) split = split && ((probs[i].lse != nullptr) == (i < n_al));
    static DevMask attr{0}, attr_s{0};
    if (split) {
        NCX_HIP_TRY(set_max_lds_once(attr_s, (const void*)k_dw_tn8<true>, lds_max));
        hipLaunchKernelGGL(k_dw_tn8<true>, dim3(pl.grid), dim3(TN8_T), lds, s, a);
    } else {
        NCX_HIP_TRY(set_max_lds_once(attr, (const void*)k_dw_tn8<false>, lds_max));
        hipLaunchKernelGGL(k_dw_tn8<false>, dim3(pl.grid), dim3(TN8_T), lds, s, a);
    }
    NCX_HIP_TRY(hipGetLastError());
    return NCX_OK;
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_dw_reduce_km_tn8(const KmReduceArgs km, const Tn8ReduceArgs tn) {
    if ((int)blockIdx.x < km.nblk) { km_reduce_body<VEC>(km, blockIdx.x); return; }
    tn8_reduce_body(tn, blockIdx.x - km.nblk);
}

int dw_reduce_km_tn8(const KmReduceArgs* km, bool km_vec, const Tn8ReduceArgs* tn, hipStream_t s) {
    KmReduceArgs k0{}; Tn8ReduceArgs t0{};
    if (km) k0 = *km;                                      // (nblk == 0: no fold slab)
    if (tn) t0 = *tn;
    const int nb = k0.nblk + t0.n_tiles_total * 8;
    if (nb == 0) return NCX_OK;
    if (km_vec) hipLaunchKernelGGL(k_dw_reduce_km_tn8<true>, dim3(nb), dim3(256), 0, s, k0, t0);
    else        hipLaunchKernelGGL(k_dw_reduce_km_tn8<false>, dim3(nb), dim3(256), 0, s, k0, t0);
    NCX_HIP_TRY(hipGetLastError());
    return NCX_OK;
}

int dw_tn8(const ncx_dims& d, const Tn8Prob* probs, int np, int n_al, bool do_al, bool do_rest, const int* idx_ob, const int* aid,
           float* slab, size_t slab_bytes, hipStream_t s) {
    Tn8ReduceArgs r{};
    int rc = dw_tn8_products(d, probs, np, n_al, do_al, do_rest, idx_ob, aid, slab, slab_bytes, &r, s);
    if (rc) return rc;
    return dw_reduce_km_tn8(nullptr, true, &r, s);
}

}  // namespace ncx
