// 5x5 conv instantiations: last block of the filled ResNets (d=4), U-Net dec1 (d=1).
#include "conv_registry.h"
TPZ_CONV2D_HEAD(5, 1, 32, 16, 32, 1, 5)
TPZ_CONV2D_HEAD(5, 1, 64, 16, 32, 1, 1)
TPZ_CONV2D_HEAD(5, 4, 32, 16, 32, 1, 5)
TPZ_CONV2D_HEAD(5, 4, 64, 16, 32, 1, 1)
TPZ_CONV2D_HEAD(5, 4, 128, 8, 32, 1, 1)
