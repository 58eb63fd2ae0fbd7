#include "conv_split_registry.h"
// (S = steps per stage: one workgroup barrier every S steps; +2 ... 10 % measured in the same process, tools/split_ablate stages)
//               K  D  MT   TH  TW  CC [S]
TPZ_SPLIT_RESID_S(3, 4, 128, 16, 32, 2, 2)
TPZ_SPLIT_RESID(3, 8, 128, 16, 32, 2)
TPZ_SPLIT_RESID_S(3, 2, 128, 16, 32, 2, 2)
// U-Net dec1.0 (5x5 over a 2x-upsampled source) as one sub-pixel conv: 3x3, 4 x 64 virtual output channels
TPZ_SPLIT4(3, 1, 128, 8, 32, 2, ::tpz::EPI_RES)
TPZ_SPLIT4(3, 1, 128, 8, 32, 2, ::tpz::EPI_PLAIN)   // ... with the 1-channel skip source folded in as 4 more channels
// one step per stage, plain epilogue: the form that takes a folded 1x1 projection (ResidA blocks that change width)
TPZ_SPLIT(3, 4, 128, 16, 32, 2, ::tpz::EPI_PLAIN)
TPZ_SPLIT(3, 2, 128, 16, 32, 2, ::tpz::EPI_PLAIN)
