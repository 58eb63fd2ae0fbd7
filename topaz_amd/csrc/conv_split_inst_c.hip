#include "conv_split_registry.h"
// 64-channel ResidA layers.  d = 1 (also the U-Net dec1.0 parity kernels, which are launched in batches): 4-wave workgroups,
// two per CU.  d = 2 and d = 4: 16x48 pixels on 8 waves -- 6 pixel fragments per wave instead of 4, so that a K step's fixed
// cost (DMA issue, barrier, plan) is spread over 72 MFMAs per wave instead of 48: +5 ... 11 % against the 8x32 / 16x32 tiles
// (tools/split_ablate r4b, profiles/r04_tile_experiments.txt); one step per stage, which is also the form that takes a folded
// 1x1 projection.
//                K  D  MT  TH  TW  CC
TPZ_SPLIT4_RESID(3, 1, 64, 8,  32, 2)
TPZ_SPLIT_RESID(3, 2, 64, 16, 48, 2)
TPZ_SPLIT_RESID(3, 4, 64, 16, 48, 2)
