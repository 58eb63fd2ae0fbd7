// 5x5 conv instantiations: last block of the filled ResNets (d=4), U-Net dec1 (d=1).
#include "conv_registry.h"
TPZ_CONV2D(5, 1, 32, 16, 32, 1, 5, false)
TPZ_CONV2D(5, 1, 64, 16, 32, 1, 1, false)
TPZ_CONV2D(5, 4, 32, 16, 32, 1, 5, false)
TPZ_CONV2D(5, 4, 64, 16, 32, 1, 1, false)
TPZ_CONV2D(5, 4, 128, 8, 32, 1, 1, false)
