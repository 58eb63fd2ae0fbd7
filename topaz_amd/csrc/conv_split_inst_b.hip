#include "conv_split_registry.h"
//               K  D  MT   TH  TW  CC
TPZ_SPLIT_RESID(3, 4, 128, 16, 32, 2)
TPZ_SPLIT_RESID(3, 8, 128, 16, 32, 2)
TPZ_SPLIT_RESID(3, 2, 128, 16, 32, 2)
