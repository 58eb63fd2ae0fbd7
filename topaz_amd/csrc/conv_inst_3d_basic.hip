// 3-D instantiations for filled 3-D BasicConv stacks (topaz/model/features/basic.py:12-111 with dims = 3 -- conv31 / conv63 /
// conv127 trained with `--dims 3`): dilated 5^3 convs at the cumulative strides 2, 4, 8, 16 (PReLU, eval-BN folded), the last
// one with the fused 1x1x1 head.  The 7^3 stem and the 5^3 d4 head kernels are those of conv_inst_3d_score.hip.
// conv31 (dilations 1, 2, 4) and conv63 (1, 2, 4, 8) are covered; a 3-D conv127 would need the 5^3 kernel at dilation 16,
// whose x halo (64 columns on either side of every row of every plane) does not fit three LDS buffers -- the runtime says so.
#include "conv_registry.h"
//             K  D   MT  TD TH  TW KG RPS CIN1  EPI
TPZ_CONV3D_EPI(5, 2,  32, 2, 4, 32, 1, 1, false, ::tpz::EPI_PLAIN)
TPZ_CONV3D_EPI(5, 2,  64, 2, 4, 32, 1, 1, false, ::tpz::EPI_PLAIN)
TPZ_CONV3D_EPI(5, 4,  32, 2, 4, 32, 1, 1, false, ::tpz::EPI_PLAIN)
TPZ_CONV3D_EPI(5, 4,  64, 2, 4, 32, 1, 1, false, ::tpz::EPI_PLAIN)
TPZ_CONV3D_EPI(5, 8,  32, 1, 4, 32, 1, 1, false, ::tpz::EPI_PLAIN)
TPZ_CONV3D_EPI(5, 8,  64, 1, 4, 32, 1, 1, false, ::tpz::EPI_PLAIN)
TPZ_CONV3D_EPI(5, 8,  32, 1, 4, 32, 1, 1, false, ::tpz::EPI_HEAD)
TPZ_CONV3D_EPI(5, 8,  64, 1, 4, 32, 1, 1, false, ::tpz::EPI_HEAD)
