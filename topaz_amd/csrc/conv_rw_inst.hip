// instantiations of the weights-resident 3x3 32 -> 32 kernel (conv_rw.h): dilation 1 / 2 / 4 x plain / residual / residual + eval-BN
#include "conv_rw.h"

namespace tpz {

template <int DIL>
static hipError_t launch_rw_epi(const SplitArgs& a, int epi, int workgroups, hipStream_t s) {
    if (epi == EPI_PLAIN) return launch_rw_cfg<DIL, EPI_PLAIN>(a, workgroups, s);
    if (epi == EPI_RES) return launch_rw_cfg<DIL, EPI_RES>(a, workgroups, s);
    if (epi == EPI_RES_POST) return launch_rw_cfg<DIL, EPI_RES_POST>(a, workgroups, s);
    return hipErrorInvalidValue;
}

hipError_t launch_conv_rw(const SplitArgs& a, int dil, int epi, int workgroups, hipStream_t s) {
    if (dil == 1) return launch_rw_epi<1>(a, epi, workgroups, s);
    if (dil == 2) return launch_rw_epi<2>(a, epi, workgroups, s);
    if (dil == 4) return launch_rw_epi<4>(a, epi, workgroups, s);
    return hipErrorInvalidValue;
}

}  // namespace tpz
