// 3-D instantiations for filled 3-D BasicConv stacks (topaz/model/features/basic.py:12-111 with dims = 3 -- conv31 / conv63 /
// conv127 trained with `--dims 3`): dilated 5^3 convs at the cumulative strides 2, 4, 8, 16 (PReLU, eval-BN folded), the last
// one with the fused 1x1x1 head.  The 7^3 stem and the 5^3 d4 head kernels are those of conv_inst_3d_score.hip.
// conv31 (dilations 1, 2, 4), conv63 (1, 2, 4, 8) and conv127 (1, 2, 4, 8, 16).  The last layer of a 3-D conv127 is the 5^3 kernel
// at dilation 16 with the fused head: its x halo is 64 columns on either side of every row of every plane, so the tile that
// still fits three LDS buffers is 2 x 2 x 16 outputs (a 6 x 6 x 84 input box per channel: ~47x the outputs) -- slow, and the
// reason the 2xf16 path has no tile for it; the layer runs here, on the fp32 matrix cores, instead of being refused.
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
TPZ_CONV3D_EPI(5, 16, 32, 2, 2, 16, 1, 1, false, ::tpz::EPI_HEAD)
TPZ_CONV3D_EPI(5, 16, 64, 2, 2, 16, 1, 1, false, ::tpz::EPI_HEAD)
