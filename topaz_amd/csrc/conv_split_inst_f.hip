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
