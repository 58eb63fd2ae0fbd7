// 5x5 conv instantiations for the filled Conv127/63/31 stacks (d = 2, 8, 16).
#include "conv_registry.h"
TPZ_CONV2D_HEAD(5, 2, 32, 16, 32, 1, 5)
TPZ_CONV2D_HEAD(5, 2, 64, 16, 32, 1, 1)
TPZ_CONV2D_HEAD(5, 8, 32, 16, 32, 1, 5)
TPZ_CONV2D_HEAD(5, 8, 64, 16, 32, 1, 1)
TPZ_CONV2D_HEAD(5, 16, 32, 16, 32, 1, 5)
TPZ_CONV2D_HEAD(5, 16, 64, 16, 32, 1, 1)
