#include "conv_split_registry.h"
// 64-channel ResidA layers.  d = 1 (also the U-Net dec1.0 parity kernels): 4-wave workgroups, two per CU;
// d = 2 as well since the epilogue / DMA-issue rewrite; d = 4: 8-wave (its halo does not fit two workgroups' LDS)
//                K  D  MT  TH  TW  CC
TPZ_SPLIT4_RESID(3, 1, 64, 8,  32, 2)
TPZ_SPLIT4_RESID(3, 2, 64, 8,  32, 2)          // 4 waves, two workgroups per CU: -3 ... -5 % against the 8-wave S=2 tile (split_ablate d2)
TPZ_SPLIT_RESID_S(3, 4, 64, 16, 32, 2, 2)
TPZ_SPLIT(3, 4, 64, 16, 32, 2, ::tpz::EPI_PLAIN)        // (one step per stage: takes a folded 1x1 projection)
