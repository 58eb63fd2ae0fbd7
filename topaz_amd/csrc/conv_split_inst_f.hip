#include "conv_split_registry.h"
// conv127/63/31 (basic.py:47-63): 5x5 at dilations 2, 4, 8, 16, 32 units, the last one with the fused head
//         K  D   MT  TH  TW  CC  EPI
TPZ_SPLIT4(5, 2, 32, 8, 32, 2, ::tpz::EPI_PLAIN)
TPZ_SPLIT(5, 4,  32, 16, 32, 2, ::tpz::EPI_PLAIN)
TPZ_SPLIT(5, 8,  32, 16, 32, 1, ::tpz::EPI_PLAIN)
TPZ_SPLIT(5, 4,  32, 16, 32, 2, ::tpz::EPI_HEAD)
TPZ_SPLIT(5, 8,  32, 16, 32, 1, ::tpz::EPI_HEAD)
TPZ_SPLIT(5, 16, 32, 16, 32, 1, ::tpz::EPI_HEAD)
// ResNet heads at 32 and 64 units: 5x5 d4 64->128 (one co-group) is served by the MT = 128 head kernel of inst_a
// column kernels (k x 1 taps, rt_load.hip prepare_split): 1-channel stems with their kx taps as input channels
// (7x7 / 7x7x7 / 11x11 at 32 / 48 / 64 outputs), 1-output-channel last convs with their kx taps as output channels
//             K  D  MT  TH TW  CC  EPI
TPZ_SPLIT4_COL(7,  1, 32, 8, 32, 1, ::tpz::EPI_PLAIN)
TPZ_SPLIT4_COL(7,  1, 48, 8, 32, 1, ::tpz::EPI_PLAIN)
TPZ_SPLIT4_COL(7,  1, 64, 8, 32, 1, ::tpz::EPI_PLAIN)
TPZ_SPLIT4_COL(11, 1, 48, 8, 32, 1, ::tpz::EPI_PLAIN)
TPZ_SPLIT4_COL(5,  1, 16, 8, 32, 2, ::tpz::EPI_PLAIN_F32)
TPZ_SPLIT4_COL(3,  1, 16, 8, 32, 2, ::tpz::EPI_PLAIN_F32)
