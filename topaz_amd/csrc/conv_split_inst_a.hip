// 2xf16-split kernels of the ResNet8 / ResNet16 scoring stacks at 64 base units (conv_split.h)
#include "conv_split_registry.h"
//         K  D  MT   TH  TW  CC  EPI
TPZ_SPLIT(5, 4, 128, 16, 32, 2, ::tpz::EPI_HEAD)
TPZ_SPLIT(5, 4, 128, 16, 32, 2, ::tpz::EPI_PLAIN)
TPZ_SPLIT4(1, 1, 128, 8, 16, 4, ::tpz::EPI_PLAIN)      // 4 waves, two workgroups per CU: one stores while the other loads (+2.5 %)
