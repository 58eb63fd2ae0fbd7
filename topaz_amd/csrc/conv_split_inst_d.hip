// 2xf16-split kernels of the 2-D U-Net denoisers at 48 base filters (topaz/denoising/models.py:74-175):
// encoder / decoder 3x3 convs, the per-parity kernels of the decoders' first convs (rt_load.hip
// prepare_phases: 2-tap kernels for k = 3, 3-tap for k = 5, added in place onto the skip-source part),
// and dec1.2, whose consumer (the 1-channel last conv) reads fp32.
#include "conv_split_registry.h"
//         K  D  MT  TH  TW  CC  EPI
TPZ_SPLIT4_S(3, 1, 48, 8, 32, 2, 2, ::tpz::EPI_PLAIN)
TPZ_SPLIT4(3, 1, 96, 8, 32, 2, ::tpz::EPI_PLAIN)
TPZ_SPLIT4(2, 1, 96, 8, 32, 2, ::tpz::EPI_RES)
TPZ_SPLIT4_S(5, 1, 32, 8, 32, 2, 2, ::tpz::EPI_PLAIN_F32)
TPZ_SPLIT4_S(5, 1, 32, 8, 32, 2, 2, ::tpz::EPI_PLAIN)
// UDenoiseNet3D (plane-stacked 3-D, conv_split.h): dec1.0 parity kernels (2 taps per axis, 64 outputs) and dec1.2
// storing fp32 for the 1-channel last conv; the 48 / 96-channel 3x3(x3) and 2x2(x2) kernels above are shared
TPZ_SPLIT4(2, 1, 64, 8, 32, 2, ::tpz::EPI_RES)
TPZ_SPLIT4(2, 1, 64, 8, 32, 2, ::tpz::EPI_PLAIN)      // ... with the 1-channel skip source folded in as one more cell
TPZ_SPLIT4(3, 1, 32, 8, 32, 2, ::tpz::EPI_PLAIN_F32)
// encoder convs with the 2x2 max-pool fused into the epilogue (EPI_POOL): 3x3 48->48 and, as column kernels, the stems
TPZ_SPLIT4_S(3, 1, 48, 8, 32, 2, 2, ::tpz::EPI_POOL)
TPZ_SPLIT4_COL(11, 1, 48, 8, 32, 1, ::tpz::EPI_POOL)
TPZ_SPLIT4_COL(7, 1, 48, 8, 32, 1, ::tpz::EPI_POOL)
