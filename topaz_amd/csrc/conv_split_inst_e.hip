// 2xf16-split kernels for the 32-unit scoring networks (the pretrained resnet8_u32 / resnet16_u32 detectors and
// conv127/63/31): 32- and 64-channel ResidA layers, the 128-channel head, the dilated 5x5 BasicConv stack.
#include "conv_split_registry.h"
//               K  D  MT  TH  TW  CC
TPZ_SPLIT4_RESID_S(3, 1, 32, 8, 32, 2, 2)
TPZ_SPLIT4_RESID(3, 2, 32, 8, 32, 2)
TPZ_SPLIT4_RESID(3, 4, 32, 8, 32, 2)
TPZ_SPLIT_RESID_S(3, 8, 64, 16, 32, 2, 2)
TPZ_SPLIT4(1, 1, 64, 8, 16, 4, ::tpz::EPI_PLAIN)
