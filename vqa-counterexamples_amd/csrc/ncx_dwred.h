// ncx_dwred.h -- the fixed-order reductions of the two weight-gradient kernels of linear_1 (ncx_dwkm.hip: k-chunk slabs of the
// per-triplet fold; ncx_dwtn.hip: partial tiles of the 8-wave row-reduction launch), as device bodies so that ONE launch can run both.
#pragma once
#include "ncx_internal.h"

namespace ncx {

constexpr int TN8_BM = 256, TN8_BN = 64, TN8_BK = 32, TN8_T = 512;

// Fixed-order sum of the k-chunk partials.  VEC: 4 consecutive columns per thread (dv % 4 == 0); all chunk loads are issued
// before the first add (DW_KM_SPLIT is a compile-time bound: a runtime-length loop made every chunk a dependent round trip).
struct KmReduceArgs { const float* slab; int nz, H, dv; long long din; float* g_vother; float* g_vmult; int nblk; };
template <bool VEC>
__device__ __forceinline__ void km_reduce_body(const KmReduceArgs& r, int blk) {
    constexpr int W = VEC ? 4 : 1;
    typedef float vec __attribute__((ext_vector_type(VEC ? 4 : 1)));
    const long long i = ((long long)blk * 256 + threadIdx.x) * W, n = (long long)r.H * r.dv;
    if (i >= n) return;
    const int h = (int)(i / r.dv), c = (int)(i - (long long)h * r.dv);
    vec vk[DW_KM_SPLIT], vm[DW_KM_SPLIT];
#pragma unroll
    for (int z = 0; z < DW_KM_SPLIT; ++z) {
        const int zz = z < r.nz ? z : r.nz - 1;
        vk[z] = *(const vec*)(r.slab + ((long long)zz * 2 + 0) * n + i);
        vm[z] = *(const vec*)(r.slab + ((long long)zz * 2 + 1) * n + i);
    }
    vec sk = vk[0], sm = vm[0];
#pragma unroll
    for (int z = 1; z < DW_KM_SPLIT; ++z) { const vec zero = {}; sk += z < r.nz ? vk[z] : zero; sm += z < r.nz ? vm[z] : zero; }
#pragma unroll
    for (int j = 0; j < W; ++j) {                        // (rows of linear_1.weight are din floats apart: not 16-byte aligned in general)
        r.g_vother[(long long)h * r.din + c + j] = sk[j];
        r.g_vmult[(long long)h * r.din + c + j] = sm[j];
    }
}

// ---- fixed-order sums of the partial tiles -------------------------------------------------------------------------------------------------
// One block per (tile, 32-row group): out[h][n0 + c] = sum over the tile's slots, aligned tiles z = 0 .. S-1, rest tiles in ascending
// workgroup order (the workgroups whose range [w R, (w + 1) R) meets the tile's k-steps).
struct Tn8ReduceArgs {
    Tn8Prob p[TN8_MAX_PROB];
    int np, n_al, tiles_m, S, al_wgs, R, do_al, do_rest;
    int rest_tiles[TN8_MAX_PROB], rest_steps[TN8_MAX_PROB], rest_tile0[TN8_MAX_PROB], rest_pre[TN8_MAX_PROB + 1];
    int al_tiles, n_tiles_total;      // blocks: (al_tiles if do_al) + (rest tiles if do_rest), x 8 row groups
    const float* slab;
};
__device__ __forceinline__ void tn8_reduce_body(const Tn8ReduceArgs& r, int blk) {
    int tile_blk = blk >> 3;
    const int rgp = blk & 7;                                    // 32-row group of the 256-row tile
    int prob, tile, s_first, s_count, s_stride = 1;
    if (r.do_al && tile_blk < r.al_tiles) {
        prob = 0; tile = tile_blk; s_first = tile * r.S; s_count = r.S;
    } else {
        if (r.do_al) tile_blk -= r.al_tiles;
        prob = r.n_al;
        while (prob + 1 < r.np && tile_blk >= r.rest_tile0[prob + 1]) ++prob;
        tile = tile_blk - r.rest_tile0[prob];
        const long long g0 = (long long)r.rest_pre[prob] + (long long)tile * r.rest_steps[prob], g1 = g0 + r.rest_steps[prob];
        const int w0 = (int)(g0 / r.R), w1 = (int)((g1 - 1) / r.R);
        s_first = r.al_wgs + r.rest_tile0[prob] + tile + w0; s_count = w1 - w0 + 1;
    }
    const Tn8Prob& p = r.p[prob];
    const int tm = tile % r.tiles_m, tn = tile / r.tiles_m;
    __shared__ float tn8_tt[32][TN8_BN + 1];                     // the block's 32 x 64 sums, for the transposed copies
    // 32 rows x 64 columns = 512 float4: two per thread
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int f = threadIdx.x + 256 * it, row = rgp * 32 + (f >> 4), c = 4 * (f & 15);
        const float* src = r.slab + (long long)s_first * (TN8_BM * TN8_BN) + row * TN8_BN + c;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < s_count; z += 8) {                   // eight partials requested before the first add
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const f32x4*)(src + (long long)min(z + j, s_count - 1) * s_stride * (TN8_BM * TN8_BN));
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += z + j < s_count ? v[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int h = tm * TN8_BM + row, n = tn * TN8_BN + c;
        float* o = p.out + (long long)h * p.ldo + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (n + j < p.n_valid) o[j] = sum[j];      // (rows of linear_1.weight are din floats apart: not 16-byte aligned in general)
        if (p.outT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) tn8_tt[f >> 4][c + j] = sum[j];
        }
    }
    if (p.outT) {            // out^T[n][h]: thread = (column n = tid >> 2, 8 consecutive h): 32-byte pieces of 128-byte row segments
        __syncthreads();
        const int nl = threadIdx.x >> 2, hq = (threadIdx.x & 3) * 8;
        const int n = tn * TN8_BN + nl, h0 = tm * TN8_BM + rgp * 32 + hq;
        if (n < p.n_valid + p.padT2) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = n < p.n_valid ? tn8_tt[hq + j][nl] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (n < p.n_valid) p.outT[(long long)n * p.ldT + h0 + j] = v[j];
                if (p.outT2) p.outT2[(long long)n * p.ldT + h0 + j] = v[j];
            }
        }
    }
}

// ncx_dwtn.hip: products launched, reduction handed back (merged with the per-triplet fold's by dw_reduce_km_tn8)
int dw_tn8_products(const ncx_dims& d, const Tn8Prob* probs, int np, int n_al, bool do_al, bool do_rest, const int* idx_ob, const int* aid,
                    float* slab, size_t slab_bytes, Tn8ReduceArgs* red, hipStream_t s);
// one launch: the fixed-order sums of the fold kernel's k-chunks (km may be null) and of the TN launch's partial tiles (tn may be null)
int dw_reduce_km_tn8(const KmReduceArgs* km, bool km_vec, const Tn8ReduceArgs* tn, hipStream_t s);
// ncx_dwkm.hip: the reduction arguments of the fold kernel's slab (what dw_km_finish would launch)
KmReduceArgs dw_km_reduce_args(const ncx_dims& d, const float* slab, float* g_vother, float* g_vmult, long long din, bool* vec);

}  // namespace ncx
