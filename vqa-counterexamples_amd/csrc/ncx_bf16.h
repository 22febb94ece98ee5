// ncx_bf16.h -- the "bf16 weights" variant of the two dominant GEMMs (BASELINE configs[4]; flag NCX_F_BF16).
//
// With bf16 operands the MFMA rate is 16x the fp32-input rate, so the first-layer GEMMs stop being compute-bound and
// the fp32 design's "never materialise the concat" rule flips: k_prep packs every candidate row ONCE as bf16
//     Xc[r] = [ v_k | v_o * v_k | dist, one-hot rank | z_k | softmax(a_k) ]        (M x Kc, Kc padded to 128)
// and the per-step weight pack is  Wc = bf16([ W1 slices in the same column order | Gt ])  (H x Kc), after which
//     forward   h1   = epilogue(Sh[b] + Xc . Wc^T)          plain bf16 NT GEMM  (v_mfma_f32_16x16x32_bf16, fp32 accumulate)
//     backward  dWc  = bf16(dpre1)^T . Xc                   plain bf16 TN GEMM  (operands read with ds_read_b64_tr_b16)
// Everything else of the step (shared segments, Gt, hidden layers, dE, loss, Adam on fp32 master weights) is the fp32
// path unchanged.  The reference has no reduced-precision mode: the parity oracle is the same restatement with the
// operands of these two products rounded to bf16 (round-to-nearest-even) and fp32 accumulation.
#pragma once
#include "ncx_common.h"
#include "ncx_gemm.h"

namespace ncx {

typedef unsigned short u16;

// Column layout of a packed row; every segment starts on a multiple of 8 columns (16-byte vector stores in k_prep),
// the gaps are zero in Xc and Wc.
struct Bf16Cols { int c_vk, c_vm, c_misc, c_z, c_p, raw, kc; };
static inline Bf16Cols bf16_cols(const ncx_dims& d) {
    Bf16Cols c;
    auto up8 = [](int x) { return (x + 7) / 8 * 8; };
    c.c_vk = 0; c.c_vm = up8(d.dv); c.c_misc = c.c_vm + up8(d.dv); c.c_z = c.c_misc + up8(d.K + 1); c.c_p = c.c_z + up8(d.dz);
    c.raw = c.c_p + d.A;
    c.kc = (c.raw + 127) / 128 * 128;          // whole 128-column tiles of the weight-gradient GEMM, 64-deep K-steps of the forward
    return c;
}
constexpr int BF16_SPLIT = 8;                 // k-chunks of the weight-gradient GEMM: one per XCD

// bytes of the bf16 workspace regions
static inline size_t bf16_xc_bytes(const ncx_dims& d) { return (size_t)d.B * d.K * bf16_cols(d).kc * 2; }
static inline size_t bf16_wc_bytes(const ncx_dims& d) { return (size_t)((d.H + 127) / 128 * 128) * bf16_cols(d).kc * 2; }
static inline size_t bf16_dpre_bytes(const ncx_dims& d) { return (size_t)d.B * d.K * ((d.H + 127) / 128 * 128) * 2; }
static inline size_t bf16_slab_bytes(const ncx_dims& d) { return (size_t)BF16_SPLIT * d.H * bf16_cols(d).kc * 4; }

// The answer-embedding products of the variant (Gt forward; dW1[:, a_other] and dE backward) take bf16 copies of the
// weights too: E [A][dap], E^T [da][Ap], [W1ak; W1agt] [2H][dap] and the bf16 image of [dGt; dGgt] [2H][Ap]
// (dap / Ap: da / A rounded up to 128).
struct Bf16Emb { u16* e_bf; u16* et_bf; u16* w1a_bf; u16* dg_bf; int dap, Ap; };
static inline int bf16_up128(int x) { return (x + 127) / 128 * 128; }
static inline size_t bf16_emb_bytes(const ncx_dims& d) {
    const size_t dap = bf16_up128(d.da), Ap = bf16_up128(d.A);
    return ((size_t)d.A * dap + (size_t)d.da * Ap + (size_t)2 * d.H * dap + (size_t)2 * d.H * Ap) * 2 + 4 * 256;
}
static inline Bf16Emb bf16_emb_layout(const ncx_dims& d, char* base) {
    Bf16Emb m; m.dap = bf16_up128(d.da); m.Ap = bf16_up128(d.A);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base + off; off = (off + bytes + 255) / 256 * 256; return (u16*)p; };
    m.e_bf = take((size_t)d.A * m.dap * 2); m.et_bf = take((size_t)d.da * m.Ap * 2);
    m.w1a_bf = take((size_t)2 * d.H * m.dap * 2); m.dg_bf = take((size_t)2 * d.H * m.Ap * 2);
    return m;
}
int bf16_pack_embedding(const ncx_dims& d, const float* E, const float* w1, const Bf16Emb& m, hipStream_t s);
int bf16_gt(const ncx_dims& d, const Bf16Emb& m, float* gt, hipStream_t s);
int bf16_dw1ak(const ncx_dims& d, const Bf16Emb& m, const float* dgt_dagt, float* g_w1, hipStream_t s);
int bf16_de(const ncx_dims& d, const Bf16Emb& m, const float* dgt_dagt, float* g_E, hipStream_t s);

// Wc = bf16([W1 candidate-segment columns | Gt]), zero in the padding columns and rows (rows padded to 128)
int bf16_pack_wc(const ncx_dims& d, const float* w1, const float* gt, u16* wc, hipStream_t s);
// h1 = epilogue(Xc . Wc^T): out [M, H] fp32
int bf16_main_forward(const ncx_dims& d, const u16* xc, const u16* wc, const EpiArgs& epi, float* h1, hipStream_t s);
// dpre -> bf16 (rows padded to the 128-column tile), dWc = dpre_bf^T . Xc in BF16_SPLIT k-chunks, reduced in fixed order
// and scattered to d linear_1.weight (candidate columns) and dGt
int bf16_dw1c(const ncx_dims& d, const float* dpre, u16* dpre_bf, const u16* xc, float* slab, float* g_w1, float* dgt,
              hipStream_t s);

}  // namespace ncx
