// ncx_dwkm.hip -- weight gradient of the v_other and v_orig*v_other segments of linear_1 in ONE MFMA pass.
//
//   dW1[:, v_other][h][c] = sum_r dpre[r][h] * v_k[r][c]
//   dW1[:, v_mult ][h][c] = sum_r dpre[r][h] * v_o[b(r)][c] * v_k[r][c]  =  sum_b v_o[b][c] * T_b[h][c],   T_b = sum_k dpre[(b,k)][h] v_k[(b,k)][c]
//
// The K candidate rows of a triplet share v_o, so a reduction step that is exactly one triplet (K rows = K/4
// v_mfma_f32_16x16x4_f32 steps) yields T_b in a scratch accumulator, which is then folded into BOTH outputs on the vector
// ALU: acc_k += T_b, acc_m += v_o[b] (.) T_b.  Half the MFMA work of treating the two segments as separate GEMM problems
// (2 x 12.9 GF at configs[1]).  Reference: the same gradients autograd produces for cx.py:296,309-322.
//
// 128 (h) x 64 (c) output tile pair per workgroup, 256 threads = 2 x 2 waves (64 x 32 each), split over the triplets in
// S chunks (chunk z on XCD z: its dpre rows stay in that L2), partial tiles to a slab, fixed-order reduction afterwards.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ncx_internal.h"

namespace ncx {

constexpr int KM_BM = 128, KM_BN = 64, KM_PA = KM_BM + 16, KM_PB = KM_BN + 16;   // LDS pitches = 16 mod 32: ds_read_b32 at its 2-cycle floor

// K = rows per reduction step (a whole triplet, or a 24-row part of one: the rows of a step share v_o); Kc = candidates
// per triplet, a multiple of K.
template <int K>
__global__ __launch_bounds__(256, 2) void k_dw_km(const float* __restrict__ dpre, int H, const float* __restrict__ feats, int dv,
                                                  const int* __restrict__ idx_k, const int* __restrict__ idx_o, int B, int Kc,
                                                  int chunk, int tiles_m, int S, float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) float km_smem[];
    constexpr int a_elems = K * KM_PA, b_elems = K * KM_PB;
    float* const lds_a = km_smem;                       // [2][K][KM_PA]
    float* const lds_b = km_smem + 2 * a_elems;         // [2][K][KM_PB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 32;
    const int z = blockIdx.x % S, t = blockIdx.x / S;
    const int tm = t % tiles_m, tn = t / tiles_m;
    const int m0 = tm * KM_BM, n0 = tn * KM_BN;
    const int spt = Kc / K;                              // reduction steps per triplet
    const int t0 = z * chunk, t1 = min(t0 + chunk, B);   // triplets of this chunk
    if (t0 >= B) return;
    const int b0 = t0 * spt, b1 = t1 * spt;              // reduction steps of this chunk (step s: rows s*K .. s*K+K-1)

    constexpr int WM = 4, WN = 2;
    f32x4 acc_k[WM][WN], acc_m[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) { acc_k[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_m[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // loader roles: A tile [K][128]: float4 f = tid + 256 i -> row f / 32, quad f % 32;
    //               B tile [K][64]:  float4 f = tid + 256 i -> row f / 16, quad f % 16.
    // Two register sets: the loads of triplet b+2 are issued before the MFMAs of triplet b, so every load has two
    // reduction steps to land; every load / LDS store is unconditional (triplet index clamped, a surplus step is folded
    // with weight 0): a branch around the loads makes the compiler drain vmcnt first.
    constexpr int NA = (K * 32 + 255) / 256, NB = (K * 16 + 255) / 256;
    const bool edge_a = m0 + KM_BM > H, edge_b = n0 + KM_BN > dv;       // (uniform per workgroup)
    f32x4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
    float vo0[WN], vo1[WN];
    auto issue = [&](f32x4 (&ra)[NA], f32x4 (&rb)[NB], float (&vo)[WN], int b) __attribute__((always_inline)) {
        const float keep = b < b1 ? 1.f : 0.f;
        b = min(b, b1 - 1);
        const long long r0 = (long long)b * K;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + 256 * i, row = min(f >> 5, K - 1), c = m0 + 4 * (f & 31);
            ra[i] = load_window(dpre + (r0 + row) * H, c, H);            // unconditional 16-byte window (repaired in stash on edge tiles)
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + 256 * i, row = min(f >> 4, K - 1), c = n0 + 4 * (f & 15);
            rb[i] = load_window(feats + (long long)idx_k[r0 + row] * dv, c, dv);
        }
        const float* vor = feats + (long long)idx_o[r0] * dv;              // (idx_o is per row: any row of the step names its triplet's v_o)
#pragma unroll
        for (int j = 0; j < WN; ++j) vo[j] = vor[min(n0 + wn0 + 16 * j + li, dv - 1)];
        return keep;
    };
    auto stash = [&](const f32x4 (&ra)[NA], const f32x4 (&rb)[NB], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + 256 * i;
            const f32x4 v = edge_a ? fix_window(ra[i], m0 + 4 * (f & 31), H) : ra[i];
            if ((f >> 5) < K) *(f32x4*)(lds_a + buf * a_elems + (f >> 5) * KM_PA + 4 * (f & 31)) = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + 256 * i;
            const f32x4 v = edge_b ? fix_window(rb[i], n0 + 4 * (f & 15), dv) : rb[i];
            if ((f >> 4) < K) *(f32x4*)(lds_b + buf * b_elems + (f >> 4) * KM_PB + 4 * (f & 15)) = v;
        }
    };
    constexpr int nk4 = K / 4;
    auto compute_fold = [&](int buf, const float (&vo)[WN], float keep) __attribute__((always_inline)) {
        const float* pa = lds_a + buf * a_elems + lk * KM_PA + wm0 + li;
        const float* pb = lds_b + buf * b_elems + lk * KM_PB + wn0 + li;
        f32x4 tt[WM][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) tt[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < nk4; ++s4) {
            float af[WM], bf[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = pa[s4 * 4 * KM_PA + 16 * i];
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[j] = pb[s4 * 4 * KM_PB + 16 * j];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) tt[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], tt[i][j], 0, 0, 0);
        }
        float vm[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) vm[j] = vo[j] * keep;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc_k[i][j][q] = __builtin_fmaf(tt[i][j][q], keep, acc_k[i][j][q]);
                    acc_m[i][j][q] = __builtin_fmaf(tt[i][j][q], vm[j], acc_m[i][j][q]);
                }
    };

    float k0 = issue(ra0, rb0, vo0, b0);
    float k1 = issue(ra1, rb1, vo1, b0 + 1);
    stash(ra0, rb0, 0);
    float vc[WN], kc = k0;                               // v_o and weight of the triplet whose tile sits in LDS[0]
#pragma unroll
    for (int j = 0; j < WN; ++j) vc[j] = vo0[j];
    __syncthreads();
    for (int b = b0; b < b1; b += 2) {
        // LDS[0] = triplet b (vc, kc); set 1 = triplet b+1 in flight
        k0 = issue(ra0, rb0, vo0, b + 2);
        compute_fold(0, vc, kc);
        stash(ra1, rb1, 1);
        float vn[WN]; float kn = k1;
#pragma unroll
        for (int j = 0; j < WN; ++j) vn[j] = vo1[j];
        __syncthreads();
        // LDS[1] = triplet b+1 (vn, kn); set 0 = triplet b+2 in flight
        k1 = issue(ra1, rb1, vo1, b + 3);
        compute_fold(1, vn, kn);
        stash(ra0, rb0, 0);
        kc = k0;
#pragma unroll
        for (int j = 0; j < WN; ++j) vc[j] = vo0[j];
        __syncthreads();
    }
    float* dk = slab + ((long long)z * 2 + 0) * H * dv;
    float* dm = slab + ((long long)z * 2 + 1) * H * dv;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + wm0 + 16 * i + 4 * lk + q;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = n0 + wn0 + 16 * j + li;
                if (m < H && n < dv) { dk[(long long)m * dv + n] = acc_k[i][j][q]; dm[(long long)m * dv + n] = acc_m[i][j][q]; }
            }
        }
}

__global__ __launch_bounds__(256) void k_dw_km_reduce(const float* __restrict__ slab, int nz, int H, int dv, long long din,
                                                      float* __restrict__ g_vother, float* __restrict__ g_vmult) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)H * dv) return;
    const int h = (int)(i / dv), c = (int)(i - (long long)h * dv);
    float sk = 0.f, sm = 0.f;
    for (int z = 0; z < nz; ++z) {
        sk += slab[((long long)z * 2 + 0) * H * dv + i];
        sm += slab[((long long)z * 2 + 1) * H * dv + i];
    }
    g_vother[(long long)h * din + c] = sk;
    g_vmult[(long long)h * din + c] = sm;
}

bool dw_km_supported(const ncx_dims& d) {
    // candidates per triplet a multiple of 24 (the reference's knn_size 24, configs[4]'s 48: two 24-row steps per triplet);
    // rows need 4 columns for the 16-byte windows
    // ... and at least 256 triplets: below that the step is a latency chain and the grouped GEMM path measured faster
    // (B = 32 / 64 / 128: 0.489 / 0.493 / 0.553 ms against 0.52 / 0.53 / 0.58 with this kernel)
    return (d.flags & NCX_F_V_MULT) && d.K % 24 == 0 && d.H >= 4 && d.dv >= 4 && (d.B >= 256 || hook_env("NCX_KM_FORCE"));
}
size_t dw_km_slab_bytes(const ncx_dims& d) { return dw_km_supported(d) ? (size_t)DW_KM_SPLIT * 2 * d.H * d.dv * 4 : 0; }

int dw_km(const ncx_dims& d, const float* dpre, const float* feats, const int* idx_k, const int* idx_o, float* slab,
          float* g_vother, float* g_vmult, long long din, hipStream_t s) {
    // up to 8 k-chunks (one per XCD), at least 16 triplets each: tiny batches are not worth 8 partial tiles
    const int S = d.B / 16 >= DW_KM_SPLIT ? DW_KM_SPLIT : (d.B / 16 >= 1 ? d.B / 16 : 1);
    const int chunk = (d.B + S - 1) / S, nz = (d.B + chunk - 1) / chunk;
    const int tiles_m = (d.H + KM_BM - 1) / KM_BM, tiles_n = (d.dv + KM_BN - 1) / KM_BN;
    constexpr int R = 24;
    const int lds = 2 * R * (KM_PA + KM_PB) * 4;
    static bool attr = false;
    if (!attr) { NCX_HIP_TRY(hipFuncSetAttribute((const void*)k_dw_km<R>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr = true; }
    hipLaunchKernelGGL(k_dw_km<R>, dim3(tiles_m * tiles_n * S), dim3(256), lds, s, dpre, d.H, feats, d.dv, idx_k, idx_o, d.B, d.K, chunk,
                       tiles_m, S, slab);
    NCX_HIP_TRY(hipGetLastError());
    const long long n = (long long)d.H * d.dv;
    hipLaunchKernelGGL(k_dw_km_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)slab, nz, d.H, d.dv, din, g_vother,
                       g_vmult);
    NCX_HIP_TRY(hipGetLastError());
    return NCX_OK;
}

}  // namespace ncx
