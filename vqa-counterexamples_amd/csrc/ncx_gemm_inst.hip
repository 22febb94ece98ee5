// Explicit instantiations of the segmented GEMM engine (one translation unit per product form keeps
// the parallel build short).  FORM is defined by the Makefile: 0 = NT, 1 = TN, 2 = NN.
#include "ncx_gemm.h"
#include "ncx_internal.h"

namespace ncx {

#if NCX_FORM == 0
// C[M,N] = sum_p X_p[M,k] . W_p[N,k]^T      (forward layers; both operands col-is-k)
int run_gemm_nt(GemmArgs& a, int cfg, hipStream_t s) {
    switch (cfg) {
    case CFG_128x128: return (int)launch_seg_gemm<128, 128, true, true>(a, s);
    case CFG_96x128:  return (int)launch_seg_gemm<96, 128, true, true>(a, s);
    case CFG_96x64:   return (int)launch_seg_gemm<96, 64, true, true>(a, s);
    default:          return (int)launch_seg_gemm<64, 64, true, true>(a, s);
    }
}
// chained pairs folded into a second accumulator (MUTAN fusion); 64x64 tiles: M = B*(K+1) rows x N = dim_mm
int run_gemm_nt_fold(GemmArgs& a, hipStream_t s) { return (int)launch_seg_gemm<64, 64, true, true, true>(a, s); }

// two deferred split fix-ups of one tile shape in one launch (either may be invalid = its GEMM was not split)
int run_fixup2(const FixupArgs& a, const FixupArgs& b, int cfg, hipStream_t s) {
    switch (cfg) {
    case CFG_128x128: return (int)launch_fixup2<128, 128>(a, b, s);
    case CFG_96x128:  return (int)launch_fixup2<96, 128>(a, b, s);
    case CFG_96x64:   return (int)launch_fixup2<96, 64>(a, b, s);
    case CFG_128x64:  return (int)launch_fixup2<128, 64>(a, b, s);
    default:          return (int)launch_fixup2<64, 64>(a, b, s);
    }
}

int occupancy_nt(int cfg) {
    static int occ[4] = {0, 0, 0, 0};
    if (!occ[cfg & 3]) {
        switch (cfg) {
        case CFG_128x128: occ[1] = seg_gemm_occupancy<128, 128, true, true>(); break;
        case CFG_96x128:  occ[2] = seg_gemm_occupancy<96, 128, true, true>(); break;
        case CFG_96x64:   occ[3] = seg_gemm_occupancy<96, 64, true, true>(); break;
        default:          occ[0] = seg_gemm_occupancy<64, 64, true, true>(); break;
        }
    }
    return occ[cfg & 3];
}
#elif NCX_FORM == 1
// C[M,N] = D[k,M]^T . X[k,N]                (weight gradients; both operands row-is-k)
int run_gemm_tn(GemmArgs& a, int cfg, hipStream_t s) {
    switch (cfg) {
    case CFG_128x128: return (int)launch_seg_gemm<128, 128, false, false>(a, s);
    case CFG_128x64:  return (int)launch_seg_gemm<128, 64, false, false>(a, s);
    default:          return (int)launch_seg_gemm<64, 64, false, false>(a, s);
    }
}
int occupancy_tn(int cfg) {
    static int occ[3] = {0, 0, 0};
    const int i = cfg == CFG_128x128 ? 1 : cfg == CFG_128x64 ? 2 : 0;
    if (!occ[i]) occ[i] = i == 1 ? seg_gemm_occupancy<128, 128, false, false>()
                        : i == 2 ? seg_gemm_occupancy<128, 64, false, false>() : seg_gemm_occupancy<64, 64, false, false>();
    return occ[i];
}
#else
// C[M,N] = D[M,k] . W[k,N]                  (input gradients; A col-is-k, B row-is-k)
int run_gemm_nn(GemmArgs& a, int cfg, hipStream_t s) {
    switch (cfg) {
    case CFG_128x128: return (int)launch_seg_gemm<128, 128, true, false>(a, s);
    default:          return (int)launch_seg_gemm<64, 64, true, false>(a, s);
    }
}
int occupancy_nn(int cfg) {
    static int occ[2] = {0, 0};
    const int i = cfg == CFG_128x128;
    if (!occ[i]) occ[i] = i ? seg_gemm_occupancy<128, 128, true, false>() : seg_gemm_occupancy<64, 64, true, false>();
    return occ[i];
}
#endif

}  // namespace ncx
