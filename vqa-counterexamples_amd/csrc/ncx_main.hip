// ncx_main.hip -- instantiation of the fused forward kernel of the Linear layers (ncx_main.h) for libneuralcx_hip.so.
#include "ncx_main.h"

namespace ncx {

typedef MainCfg<48, 128, 1, 4, 2, 2> MainCfg0;      // two workgroups per CU, 48 x 128 tiles (2 triplets at K = 24), loads two k-steps ahead
typedef MainCfg<96, 64, 2, 2, 1, 2> MainCfg1;
typedef MainCfg<96, 128, 2, 2, 2, 1> MainCfg2;      // one workgroup per CU with the whole register file

int main_forward(MainArgs& a, hipStream_t s) {
    int cfg = 0;
    if (const char* e = hook_env("NCX_MAIN_CFG")) cfg = atoi(e);       // experiment hook (NCX_EXPERIMENT=1)
    if (cfg == 1) return launch_main_fwd<MainCfg1>(a, s);
    if (cfg == 2) return launch_main_fwd<MainCfg2>(a, s);
    return launch_main_fwd<MainCfg0>(a, s);
}

}  // namespace ncx
