// 3-D instantiations for the filled 3-D scoring networks (ResNet8 / ResNet16 with dims = 3,
// topaz/model/features/resnet.py:53-63,111-123,280-339; `topaz extract --dims 3`): 7^3 single-channel stems,
// dilated 3^3 ResidA convs with the residual / eval-BN epilogues, 1^3 projections and the dilated 5^3 last conv
// with the fused 1x1x1 head.  Tiles are polyphase in y and z (rows and planes strided by the dilation), so the
// LDS halo stays K - 1 rows / planes at every dilation.  A stage is one tap row (RPS = 1) for the 5^3 kernels and
// one kz plane of taps (RPS = K) for the 3^3 ones.
#include "conv_registry.h"

#define TPZ_CONV3D_RESID(K, D, MT, TD, TH, TW, KG, RPS)                                 \
    TPZ_CONV3D_EPI(K, D, MT, TD, TH, TW, KG, RPS, false, ::tpz::EPI_PLAIN)              \
    TPZ_CONV3D_EPI(K, D, MT, TD, TH, TW, KG, RPS, false, ::tpz::EPI_RES)                \
    TPZ_CONV3D_EPI(K, D, MT, TD, TH, TW, KG, RPS, false, ::tpz::EPI_RES_POST)

//          K  D  MT  TD TH  TW KG RPS CIN1
TPZ_CONV3D(7, 1, 32, 4, 4, 32, 1, 7, true)
TPZ_CONV3D(7, 1, 64, 2, 4, 32, 1, 7, true)
//               K  D  MT  TD TH TW KG RPS
TPZ_CONV3D_RESID(3, 1, 32, 2, 4, 32, 1, 3)
TPZ_CONV3D_RESID(3, 1, 64, 2, 4, 32, 1, 3)
TPZ_CONV3D_RESID(3, 2, 32, 2, 4, 32, 1, 3)
TPZ_CONV3D_RESID(3, 2, 64, 2, 4, 32, 1, 3)
TPZ_CONV3D_RESID(3, 2, 128, 2, 4, 32, 1, 3)
TPZ_CONV3D_RESID(3, 4, 32, 2, 4, 32, 1, 3)
TPZ_CONV3D_RESID(3, 4, 64, 2, 4, 32, 1, 3)
TPZ_CONV3D_RESID(3, 4, 128, 2, 4, 32, 1, 3)
TPZ_CONV3D_RESID(3, 8, 64, 2, 4, 32, 1, 3)
TPZ_CONV3D_RESID(3, 8, 128, 2, 4, 32, 1, 3)
// projections
TPZ_CONV3D(1, 1, 64, 2, 4, 32, 2, 1, false)
TPZ_CONV3D(1, 1, 128, 2, 4, 32, 4, 1, false)
// last feature conv 5^3 at dilation 4 with the fused head (2u -> 4u; Cout = 256 loops two co-groups in the kernel)
TPZ_CONV3D_EPI(5, 4, 128, 2, 4, 32, 1, 1, false, ::tpz::EPI_HEAD)
TPZ_CONV3D_EPI(5, 4, 64, 2, 4, 32, 1, 1, false, ::tpz::EPI_HEAD)
TPZ_CONV3D_EPI(5, 4, 32, 2, 4, 32, 1, 1, false, ::tpz::EPI_HEAD)
