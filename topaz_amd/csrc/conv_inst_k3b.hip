// 3x3 conv instantiations, dilation 4 and 8 (ResidA conv0/conv1 of the filled ResNet8/16).
#include "conv_registry.h"
TPZ_CONV2D_RESID(3, 4, 32, 16, 32, 1, 3)
TPZ_CONV2D_RESID(3, 4, 64, 16, 32, 1, 3)
TPZ_CONV2D_RESID(3, 4, 128, 8, 32, 1, 3)
TPZ_CONV2D_RESID(3, 8, 64, 16, 32, 1, 3)
TPZ_CONV2D_RESID(3, 8, 128, 8, 32, 1, 3)
