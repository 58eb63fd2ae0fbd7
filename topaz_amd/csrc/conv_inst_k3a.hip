// 3x3 conv instantiations, dilation 1 and 2 (ResNet16 blocks, U-Net encoder/decoder, ResidA conv0).
#include "conv_registry.h"
//          K  D  MT   TH  TW  KG RPS CIN1
TPZ_CONV2D_RESID(3, 1, 32, 16, 32, 1, 3)
TPZ_CONV2D_RESID(3, 1, 48, 16, 32, 1, 3)
TPZ_CONV2D_RESID(3, 1, 64, 16, 32, 1, 3)
TPZ_CONV2D_RESID(3, 1, 96, 8, 32, 1, 3)
TPZ_CONV2D_RESID(3, 1, 128, 8, 32, 1, 3)
TPZ_CONV2D_RESID(3, 2, 32, 16, 32, 1, 3)
TPZ_CONV2D_RESID(3, 2, 64, 16, 32, 1, 3)
TPZ_CONV2D_RESID(3, 2, 128, 8, 32, 1, 3)
