// 3-D instantiations: UDenoiseNet3D (topaz/denoising/models.py:452-564) -- 7^3 single-channel stem,
// 3^3 encoder / decoder convs.  A stage is one kz plane of taps (RPS = K tap rows).
#include "conv_registry.h"
//          K  D  MT  TD TH  TW KG RPS CIN1
TPZ_CONV3D(3, 1, 16, 4, 4, 32, 1, 3, false)
TPZ_CONV3D(3, 1, 32, 2, 4, 32, 1, 3, false)
TPZ_CONV3D(3, 1, 48, 2, 4, 32, 1, 3, false)
TPZ_CONV3D(3, 1, 64, 2, 4, 32, 1, 3, false)
TPZ_CONV3D(3, 1, 96, 2, 4, 32, 1, 3, false)
TPZ_CONV3D(7, 1, 16, 4, 4, 32, 1, 7, true)
TPZ_CONV3D(7, 1, 48, 4, 4, 32, 1, 7, true)
// stems that store split f16 cells for the plane-stacked 2xf16 kernels (conv_split.h): UDenoiseNet3D enc1 ...
TPZ_CONV3D_EPI(7, 1, 48, 4, 4, 32, 1, 7, true, ::tpz::EPI_SPLIT)
