#include "conv_split_registry.h"
// conv127/63/31 (basic.py:47-63): 5x5 at dilations 2, 4, 8, 16, 32 units, the last one with the fused head
// (round 6, tools/split_ablate c127, profiles/r06_c127_stages.txt: the d2 / d4 tiles -- the ones with a continuous slot stream,
// CC = 2 -- with FOUR steps per stage, i.e. a workgroup barrier every fourth step, and d2 on the 8-wave 16 x 32 tile instead of the
// 4-wave 8 x 32 one: persistent form 0.57 - 0.60 ms per 2048^2 layer against 0.63 - 0.68; the d8 / d16 tiles are CC = 1 -- their
// input tile of two cells would not fit twice -- and keep one step per stage)
//           K  D   MT  TH  TW  CC  S  EPI
TPZ_SPLIT_S(5, 2,  32, 16, 32, 2, 4, ::tpz::EPI_PLAIN)
TPZ_SPLIT_S(5, 4,  32, 16, 32, 2, 4, ::tpz::EPI_PLAIN)
//         K  D   MT  TH  TW  CC  EPI
TPZ_SPLIT(5, 8,  32, 16, 32, 1, ::tpz::EPI_PLAIN)
TPZ_SPLIT_S(5, 4,  32, 16, 32, 2, 4, ::tpz::EPI_HEAD)
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
