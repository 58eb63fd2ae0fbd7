// Kernels of the "phase" path: a convolution whose first source is an exactly 2x nearest-upsampled tensor
// (U-Net decoders, denoising/models.py:140-171) is computed per output phase on the LOW-resolution source with
// pre-summed weights (k = 3 -> 2 taps per axis, k = 5 -> 3 taps), written (plain) to the strided output positions; the skip-source part then runs
// over the full-resolution grid and adds itself in place through the residual epilogue (rt_exec.hip, run_conv_phases).  K = 2 kernels, and the 1-channel skip-source stems.
#include "conv_registry.h"
//              K  D  MT  TH  TW  KG RPS CIN1   EPI
TPZ_CONV2D(2, 1, 32, 16, 32, 2, 2, false)
TPZ_CONV2D(2, 1, 64, 16, 32, 2, 2, false)
TPZ_CONV2D(2, 1, 96, 8,  32, 2, 2, false)
TPZ_CONV2D_EPI(5, 1, 64, 16, 32, 1, 5, true, ::tpz::EPI_RES)
TPZ_CONV2D_EPI(5, 1, 64, 16, 32, 1, 5, true, ::tpz::EPI_SPLIT)     // the same skip-source part, feeding the 2xf16 phases
TPZ_CONV2D_EPI(3, 1, 64, 16, 32, 1, 3, true, ::tpz::EPI_RES)
//              K  D  MT  TD TH  TW  KG RPS CIN1   EPI
TPZ_CONV3D(2, 1, 16, 4, 4, 32, 1, 4, false)
TPZ_CONV3D(2, 1, 64, 2, 4, 32, 1, 4, false)
TPZ_CONV3D(2, 1, 96, 2, 4, 32, 1, 4, false)
TPZ_CONV3D_EPI(3, 1, 64, 2, 4, 32, 1, 9, true, ::tpz::EPI_RES)
TPZ_CONV3D_EPI(3, 1, 64, 2, 4, 32, 1, 9, true, ::tpz::EPI_SPLIT)    // ... and its dec1.0 skip part
TPZ_CONV3D_EPI(3, 1, 16, 4, 4, 32, 1, 3, false, ::tpz::EPI_RES)
TPZ_CONV3D_EPI(3, 1, 96, 2, 4, 32, 1, 3, false, ::tpz::EPI_RES)
