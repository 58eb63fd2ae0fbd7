// 1x1 projections (ResidA.proj) and the single-input-channel stems (k-group = 4 kx taps).
#include "conv_registry.h"
TPZ_CONV2D(1, 1, 64, 16, 32, 2, 1, false)
TPZ_CONV2D(1, 1, 128, 8, 32, 4, 1, false)
// 5x5 single-channel stems: ResNet6 (resnet.py:264)
TPZ_CONV2D(5, 1, 32, 16, 32, 1, 5, true)
TPZ_CONV2D(5, 1, 64, 16, 32, 1, 5, true)
TPZ_CONV2D(7, 1, 32, 16, 32, 1, 7, true)
TPZ_CONV2D(7, 1, 48, 16, 32, 1, 7, true)
TPZ_CONV2D(7, 1, 64, 16, 32, 1, 7, true)
// the same stems storing split f16 cells: first layer of a network that continues on the 2xf16 path (conv_split.h)
TPZ_CONV2D_EPI(7, 1, 32, 16, 32, 1, 7, true, ::tpz::EPI_SPLIT)
TPZ_CONV2D_EPI(7, 1, 64, 16, 32, 1, 7, true, ::tpz::EPI_SPLIT)
TPZ_CONV2D_EPI(11, 1, 48, 16, 32, 1, 11, true, ::tpz::EPI_SPLIT)   // U-Net enc1
TPZ_CONV2D(11, 1, 48, 16, 32, 1, 11, true)
TPZ_CONV2D(11, 1, 64, 16, 32, 1, 11, true)
// FCNN (DenoiseNet2, denoising/models.py:52-66): 11x11 64 -> 64, one tap row (11 k-steps) per stage
TPZ_CONV2D(11, 1, 64, 16, 32, 1, 1, false)
