#include "conv_split_registry.h"
//               K  D  MT   TH  TW  CC
TPZ_SPLIT_RESID(3, 4, 128, 16, 32, 2)
TPZ_SPLIT_RESID(3, 8, 128, 16, 32, 2)
TPZ_SPLIT_RESID(3, 2, 128, 16, 32, 2)
// U-Net dec1.0 (5x5 over a 2x-upsampled source) as one sub-pixel conv: 3x3, 4 x 64 virtual output channels
TPZ_SPLIT(3, 1, 128, 16, 32, 2, ::tpz::EPI_RES)
TPZ_SPLIT(3, 1, 128, 16, 32, 2, ::tpz::EPI_PLAIN)   // ... with the 1-channel skip source folded in as 4 more channels
