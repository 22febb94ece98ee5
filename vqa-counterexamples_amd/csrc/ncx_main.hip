// ncx_main.hip -- instantiation of the fused forward kernel of the Linear layers (ncx_main.h) for libneuralcx_hip.so.
#include "ncx_main.h"

namespace ncx {

typedef MainCfg<48, 128, 1, 4, 2, 2> MainCfg0;      // two workgroups per CU, 48 x 128 tiles (2 triplets at K = 24), loads two k-steps ahead
typedef MainCfg<96, 64, 2, 2, 1, 2> MainCfg1;
typedef MainCfg<96, 128, 2, 2, 2, 1> MainCfg2;      // one workgroup per CU with the whole register file
typedef MainCfg<64, 64, 2, 2, 2, 3> MainCfg3;       // short chains (the answer-embedding gradient: 16 k-steps): three small workgroups per CU
typedef MainCfg<48, 64, 1, 4, 2, 2> MainCfgFold;    // MK_VFOLD sequences: 48 x 64 tiles (two triplets), two workgroups per CU
typedef MainCfg<96, 64, 2, 2, 2, 2> MainCfgFold4;   // MK_VFOLD sequences: 96 x 64 tiles (four triplets share the W_k | W_m tiles), two workgroups per CU

// Measured inside the training step at configs[1] (B = 512: 256 tiles of 96 x 128): 338 us with one 96 x 128 workgroup per CU,
// 346 / 348 us with two 48 x 128 / 96 x 64 workgroups per CU (on back-to-back launches of the kernel alone the order is the
// other way round: the step leaves the caches in another state).  The big tile needs enough tiles to fill the chip; smaller
// batches (data-parallel shards: 64 triplets per GPU) take the 48-row tile: twice the workgroups.
int main_forward(MainArgs& a, hipStream_t s) {
    if (a.nseg > 0 && a.seg[0].kind == MK_VFOLD) {
        // four triplets per workgroup when that still gives (nearly) two workgroups per CU; else two (twice the workgroups).
        // Measured at configs[1] (tools/mb/mb_fold.hip, bit-identical outputs): 293 us against 318.
        bool four = main_fold_rows(a.M, a.N) == 96 || a.epi.rowdiv == 48;      // (K = 48 exists on 96-row tiles only)
        if (const char* f4 = hook_env("NCX_FOLD4")) four = atoi(f4) != 0;
        return four ? launch_main_fwd<MainCfgFold4>(a, s) : launch_main_fwd<MainCfgFold>(a, s);
    }
    long long T = 0;
    for (int i = 0; i < a.nseg; ++i) T += (a.seg[i].klen + MF_BK - 1) / MF_BK;
    if (a.split <= 1 && T <= 32 && (long long)((a.M + 63) / 64) * ((a.N + 63) / 64) >= 2 * num_cus() && !hook_env("NCX_MAIN_CFG"))
        return launch_main_fwd<MainCfg3>(a, s);          // a workgroup that short spends as long starting and storing as multiplying: more of them per CU
    const long long tiles96 = (long long)((a.M + 95) / 96) * ((a.N + 127) / 128);
    int cfg = (a.split <= 1 && tiles96 * 10 >= (long long)num_cus() * 9) ? 2 : 0;
    if (const char* e = hook_env("NCX_MAIN_CFG")) cfg = atoi(e);       // experiment hook (NCX_EXPERIMENT=1)
    if (cfg == 1) return launch_main_fwd<MainCfg1>(a, s);
    if (cfg == 2) return launch_main_fwd<MainCfg2>(a, s);
    return launch_main_fwd<MainCfg0>(a, s);
}

int main_fold_rows(long long M, long long N) {
    const long long wg96 = ((M + 95) / 96) * ((N + 63) / 64);
    return wg96 * 4 >= (long long)num_cus() * 2 * 3 ? 96 : 48;          // >= 3/4 of the two-per-CU slots
}

int main_split(long long M, long long N, long long T) {
    const long long tiles = ((M + 47) / 48) * ((N + 127) / 128), cus = num_cus();
    if (tiles >= cus) return 1;
    long long S = 2 * cus / tiles;                  // all workgroups resident at once (two per CU): one round, no tail
    if (S > T / 8) S = T / 8;
    if (const char* e = hook_env("NCX_MAIN_SPLIT")) S = atoll(e);      // experiment hook (NCX_EXPERIMENT=1)
    return (int)(S < 1 ? 1 : S);
}

}  // namespace ncx
