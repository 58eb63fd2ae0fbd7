// 2xf16-split kernels of the fully convolutional denoiser `fcnn` (DenoiseNet2(64, width = 11), topaz/denoising/models.py:52-66,
// 597-598): 11x11 1->64 stem and 64->1 last conv as 11 x 1 column kernels, and the 11x11 64->64 body -- 16.5 of the network's
// 17.15 TFLOP per 4096^2 micrograph.  The body's tile is 64 channels x 16x64 pixels on 8 waves: a wave holds 4 x 8 accumulator
// fragments (the transpose of the 128-channel tile's 8 x 4: the same 24 LDS reads per 96 MFMAs), one cell (8 channels) x 121
// taps = 33 steps per chunk, 264 steps per tile.
#include "conv_split_registry.h"
//         K   D  MT  TH  TW  CC  EPI
TPZ_SPLIT(11, 1, 64, 16, 64, 1, ::tpz::EPI_PLAIN)
TPZ_SPLIT4_COL(11, 1, 64, 8, 32, 1, ::tpz::EPI_PLAIN)
TPZ_SPLIT4_COL(11, 1, 16, 8, 32, 1, ::tpz::EPI_PLAIN_F32)      // (two cells per chunk would not leave room for two workgroups per CU)
// 5 x 1 column stems (ResNet6's 5x5 1 -> units first conv, resnet.py:254-277): the kx taps as the 8 channels of one cell
TPZ_SPLIT4_COL(5, 1, 32, 8, 32, 1, ::tpz::EPI_PLAIN)
TPZ_SPLIT4_COL(5, 1, 64, 8, 32, 1, ::tpz::EPI_PLAIN)
// 3 x 1 column stems (a 3x3 first conv of a user-trained stack)
TPZ_SPLIT4_COL(3, 1, 32, 8, 32, 1, ::tpz::EPI_PLAIN)
TPZ_SPLIT4_COL(3, 1, 64, 8, 32, 1, ::tpz::EPI_PLAIN)
