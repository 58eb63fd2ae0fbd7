#include "conv_split_registry.h"
//               K  D  MT  TH  TW  CC
TPZ_SPLIT_RESID(3, 1, 64, 16, 32, 2)
TPZ_SPLIT_RESID(3, 2, 64, 16, 32, 2)
TPZ_SPLIT_RESID(3, 4, 64, 16, 32, 2)
