// Registry of compiled conv_mfma_kernel instantiations.  Each conv_inst_*.hip translation
// unit registers its kernels at load time; the runtime picks one by (dims, K, dil, MT, CIN1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "conv_mfma.h"

namespace tpz {

struct ConvKernelInfo {
    int dims, K, D, MT, cin1, epi;
    int TD, TH, TW, KG, RPS, KP, KXG, NCH, SPG, STEPS, W_STAGE, W_CHUNK, lds_bytes;
    hipError_t (*launch)(const ConvArgs&, dim3 grid, hipStream_t);
    char name[160];           // template parameters as text (profiler key; matches the rocprofv3 kernel name)
};

void register_conv(const ConvKernelInfo& info);
const ConvKernelInfo* find_conv(int dims, int K, int D, int MT, bool cin1, int epi);

template <class C, int EPI>
hipError_t launch_conv_cfg(const ConvArgs& a, dim3 grid, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<C, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_mfma_kernel<C, EPI>), grid, dim3(256), C::LDS_BYTES, s, a);
    return hipGetLastError();
}

template <class C, int EPI>
struct ConvRegistrar {
    ConvRegistrar() {
        ConvKernelInfo i;
        i.dims = C::DIMS; i.K = C::K; i.D = C::D; i.MT = C::MT; i.cin1 = C::CIN1 ? 1 : 0; i.epi = EPI;
        i.TD = C::TD; i.TH = C::TH; i.TW = C::TW; i.KG = C::KG; i.RPS = C::RPS; i.KP = C::KP; i.KXG = C::KXG;
        i.NCH = C::NCH; i.SPG = C::SPG; i.STEPS = C::STEPS; i.W_STAGE = C::W_STAGE; i.W_CHUNK = C::W_CHUNK;
        i.lds_bytes = C::LDS_BYTES;
        i.launch = &launch_conv_cfg<C, EPI>;
        snprintf(i.name, sizeof i.name, "conv_mfma_kernel<K=%d,D=%d,MT=%d,TD=%d,TH=%d,TW=%d,KG=%d,RPS=%d,CIN1=%d,DIMS=%d,EPI=%d>",
                 i.K, i.D, i.MT, i.TD, i.TH, i.TW, i.KG, i.RPS, i.cin1, i.dims, i.epi);
        register_conv(i);
    }
};

#define TPZ_CAT2(a, b) a##b
#define TPZ_CAT(a, b) TPZ_CAT2(a, b)
// 2-D kernels: DIMS=2, TD=1
#define TPZ_CONV2D_EPI(K, D, MT, TH, TW, KG, RPS, CIN1, EPI) \
    static ::tpz::ConvRegistrar<::tpz::ConvCfg<K, D, MT, 1, TH, TW, KG, RPS, CIN1, 2>, EPI> TPZ_CAT(tpz_reg_, __COUNTER__);
#define TPZ_CONV2D(K, D, MT, TH, TW, KG, RPS, CIN1) TPZ_CONV2D_EPI(K, D, MT, TH, TW, KG, RPS, CIN1, ::tpz::EPI_PLAIN)
// ResidA conv1 layers (always 3x3): plain + residual + residual/eval-BN epilogues
#define TPZ_CONV2D_RESID(K, D, MT, TH, TW, KG, RPS)                                    \
    TPZ_CONV2D_EPI(K, D, MT, TH, TW, KG, RPS, false, ::tpz::EPI_PLAIN)                 \
    TPZ_CONV2D_EPI(K, D, MT, TH, TW, KG, RPS, false, ::tpz::EPI_RES)                   \
    TPZ_CONV2D_EPI(K, D, MT, TH, TW, KG, RPS, false, ::tpz::EPI_RES_POST)
// last feature conv of a scoring net (always 5x5): plain + fused 1x1 head
#define TPZ_CONV2D_HEAD(K, D, MT, TH, TW, KG, RPS)                                     \
    TPZ_CONV2D_EPI(K, D, MT, TH, TW, KG, RPS, false, ::tpz::EPI_PLAIN)                 \
    TPZ_CONV2D_EPI(K, D, MT, TH, TW, KG, RPS, false, ::tpz::EPI_HEAD)
#define TPZ_CONV3D_EPI(K, D, MT, TD, TH, TW, KG, RPS, CIN1, EPI) \
    static ::tpz::ConvRegistrar<::tpz::ConvCfg<K, D, MT, TD, TH, TW, KG, RPS, CIN1, 3>, EPI> TPZ_CAT(tpz_reg_, __COUNTER__);
#define TPZ_CONV3D(K, D, MT, TD, TH, TW, KG, RPS, CIN1) TPZ_CONV3D_EPI(K, D, MT, TD, TH, TW, KG, RPS, CIN1, ::tpz::EPI_PLAIN)

}  // namespace tpz
