// Registry of compiled conv_split_kernel instantiations (conv_split_inst_*.hip register at load time).
#pragma once
#include <hip/hip_runtime.h>
#include "conv_split.h"
#include "conv_registry.h"

namespace tpz {

struct SplitKernelInfo {
    int K, D, MT, epi;
    int KX, TH, TW, CC, WAVES, NSTEP, W_STEP_BYTES, lds_bytes, SPS;
    int cont, Q;                               // continuous slot stream (SplitCfg::CONT): Q slots per chunk
    SplitSlot (*slot)(int step, int kb);
    SplitSlot (*cont_slot)(int q);
    // steps of one tile's K loop over `cells` (virtual) cells
    int stages(int cells) const { return cont ? ((Q / CC) * cells + 3) / 4 : (cells + CC - 1) / CC * NSTEP; }
    hipError_t (*launch)(const SplitArgs&, dim3 grid, hipStream_t);
    // the same layer of n <= SPLIT_MULTI_MAX independent images in one grid (conv_split_multi_kernel); list[i].n_tiles = the
    // workgroups of entry i's own grid.  nullptr: this configuration launches one image at a time
    hipError_t (*launch_multi)(const SplitArgs* const* list, int n, hipStream_t);
    void (*make_plan)(const SplitPlanKey&, std::vector<SplitStep>&);     // the K-loop schedule the kernel reads (SplitArgs::plan)
    char name[160];
};

void register_split(const SplitKernelInfo& info);
const SplitKernelInfo* find_split(int K, int D, int MT, int epi, int KX = 0, int sps = 0);    // KX = 0: square (KX == K); sps = 0: any

// six instantiations per configuration: the plain single-source 2-D conv (MODE 0: lean scalar code), the same with
// persistent workgroups that prefetch their next tile (MODE 4; a.n_tiles > 0 selects it, the grid is then (workgroups, 1, 1);
// not for the fused head), 2-D with a second source (MODE 1), plane-stacked 3-D with one source (MODE 2), with two whose
// chunks never mix (MODE 11: source-major cell order) and with two in general (MODE 3).  It is the COMBINATION of the last two that is expensive per step (lanes of one DMA round may straddle cells of
// both tensors): 520 - 630 instructions in the K-loop body against 260 - 390 in the others.
template <class C, int EPI>
hipError_t launch_split_cfg(const SplitArgs& a, dim3 grid, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel<C, EPI, 0, 0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel<C, EPI, 0, 1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel<C, EPI, 0, 2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel<C, EPI, 0, 11>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel<C, EPI, 0, 3>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if constexpr (EPI != EPI_HEAD)
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel<C, EPI, 0, 4>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const bool plain = !a.in2 && a.KZ <= 1 && a.Din <= 1;
    if (a.n_tiles > 0) {
        if constexpr (EPI != EPI_HEAD) {
            if (!plain) return hipErrorInvalidValue;
            hipLaunchKernelGGL((conv_split_kernel<C, EPI, 0, 4>), grid, dim3(C::THREADS), C::LDS_BYTES, s, a);
        } else {
            return hipErrorInvalidValue;
        }
    } else if (plain) {
        hipLaunchKernelGGL((conv_split_kernel<C, EPI, 0, 0>), grid, dim3(C::THREADS), C::LDS_BYTES, s, a);
    } else if (a.KZ <= 1 && a.Din <= 1) {
        // a second source in 2-D (fused upsample + concat, sub-pixel skip cell, folded projection): without the plane-stacked
        // addressing the K loop is 290 - 390 instructions instead of 520 - 630
        hipLaunchKernelGGL((conv_split_kernel<C, EPI, 0, 1>), grid, dim3(C::THREADS), C::LDS_BYTES, s, a);
    } else if (!a.in2) {
        // plane-stacked 3-D, one source (most layers of the 3-D U-Net, every layer of the 3-D scoring networks)
        hipLaunchKernelGGL((conv_split_kernel<C, EPI, 0, 2>), grid, dim3(C::THREADS), C::LDS_BYTES, s, a);
    } else if (a.vol_srcmajor) {
        // plane-stacked 3-D, two sources, no chunk mixes them (the 3-D sub-pixel decoder with its space-to-depth skip cell)
        hipLaunchKernelGGL((conv_split_kernel<C, EPI, 0, 11>), grid, dim3(C::THREADS), C::LDS_BYTES, s, a);
    } else {
        hipLaunchKernelGGL((conv_split_kernel<C, EPI, 0, 3>), grid, dim3(C::THREADS), C::LDS_BYTES, s, a);
    }
    return hipGetLastError();
}

// which instantiation (MODE) a launch takes: 0 plain 2-D, 1 2-D with a second source, 2 / 11 / 3 plane-stacked 3-D with one
// source / two sources chunk-uniform / two sources in general
inline int split_mode_of(const SplitArgs& a) {
    if (!a.in2 && a.KZ <= 1 && a.Din <= 1) return 0;
    if (a.KZ <= 1 && a.Din <= 1) return 1;
    if (!a.in2) return 2;
    return a.vol_srcmajor ? 11 : 3;
}

// batched launches exist for the 4-wave tiles (every layer of the U-Nets), modes 0 / 1 / 2 / 11
template <class C, int EPI>
hipError_t launch_split_multi_cfg(const SplitArgs* const* list, int n, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_multi_kernel<C, EPI, 0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_multi_kernel<C, EPI, 1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_multi_kernel<C, EPI, 2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_multi_kernel<C, EPI, 11>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (n < 1 || n > SPLIT_MULTI_MAX) return hipErrorInvalidValue;
    SplitMulti m;
    m.n = (unsigned)n;
    unsigned total = 0;
    const int mode = split_mode_of(*list[0]);
    for (int i = 0; i < n; ++i) {
        if (split_mode_of(*list[i]) != mode || list[i]->n_tiles < 1) return hipErrorInvalidValue;
        m.first[i] = total;
        m.a[i] = *list[i];
        total = (total + (unsigned)list[i]->n_tiles + 7u) & ~7u;
    }
    for (int i = n; i <= SPLIT_MULTI_MAX; ++i) m.first[i] = total;
    const dim3 grid(total, 1, 1);
    if (mode == 0) hipLaunchKernelGGL((conv_split_multi_kernel<C, EPI, 0>), grid, dim3(C::THREADS), C::LDS_BYTES, s, m);
    else if (mode == 1) hipLaunchKernelGGL((conv_split_multi_kernel<C, EPI, 1>), grid, dim3(C::THREADS), C::LDS_BYTES, s, m);
    else if (mode == 2) hipLaunchKernelGGL((conv_split_multi_kernel<C, EPI, 2>), grid, dim3(C::THREADS), C::LDS_BYTES, s, m);
    else if (mode == 11) hipLaunchKernelGGL((conv_split_multi_kernel<C, EPI, 11>), grid, dim3(C::THREADS), C::LDS_BYTES, s, m);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

template <class C>
SplitSlot split_slot_of(int step, int kb) { return C::slot(step, kb); }
template <class C>
SplitSlot split_cont_slot_of(int q) { return C::cont_slot(q); }

template <class C, int EPI>
struct SplitRegistrar {
    SplitRegistrar() {
        SplitKernelInfo i;
        i.K = C::K; i.KX = C::KX; i.D = C::D; i.MT = C::MT; i.epi = EPI;
        i.TH = C::TH; i.TW = C::TW; i.CC = C::CC; i.WAVES = C::WAVES; i.NSTEP = C::NSTEP; i.W_STEP_BYTES = C::W_STEP_BYTES;
        i.lds_bytes = C::LDS_BYTES;
        i.SPS = C::SPS;
        i.slot = &split_slot_of<C>;
        i.cont = C::CONT ? 1 : 0; i.Q = C::Q;
        i.cont_slot = &split_cont_slot_of<C>;
        i.launch = &launch_split_cfg<C, EPI>;
        i.launch_multi = nullptr;
        if constexpr (C::WAVES == 4 && EPI != EPI_HEAD) i.launch_multi = &launch_split_multi_cfg<C, EPI>;
        i.make_plan = &split_make_plan<C>;
        if (C::SPS == 1)
            snprintf(i.name, sizeof i.name, "conv_split_kernel<K=%dx%d,D=%d,MT=%d,TH=%d,TW=%d,CC=%d,W=%d,EPI=%d>", i.K, i.KX,
                     i.D, i.MT, i.TH, i.TW, i.CC, i.WAVES, i.epi);
        else
            snprintf(i.name, sizeof i.name, "conv_split_kernel<K=%dx%d,D=%d,MT=%d,TH=%d,TW=%d,CC=%d,W=%d,S=%d,EPI=%d>", i.K,
                     i.KX, i.D, i.MT, i.TH, i.TW, i.CC, i.WAVES, i.SPS, i.epi);
        register_split(i);
    }
};

#define TPZ_SPLIT(K, D, MT, TH, TW, CC, EPI) \
    static ::tpz::SplitRegistrar<::tpz::SplitCfg<K, D, MT, TH, TW, CC>, EPI> TPZ_CAT(tpz_sreg_, __COUNTER__);
// S steps per stage (barrier every S steps): narrow channel tiles
#define TPZ_SPLIT_S(K, D, MT, TH, TW, CC, S, EPI) \
    static ::tpz::SplitRegistrar<::tpz::SplitCfg<K, D, MT, TH, TW, CC, 8, K, S>, EPI> TPZ_CAT(tpz_sreg_, __COUNTER__);
#define TPZ_SPLIT4_S(K, D, MT, TH, TW, CC, S, EPI) \
    static ::tpz::SplitRegistrar<::tpz::SplitCfg<K, D, MT, TH, TW, CC, 4, K, S>, EPI> TPZ_CAT(tpz_sreg_, __COUNTER__);
// column kernels (K x 1 taps), 4-wave workgroups
#define TPZ_SPLIT4_COL(K, D, MT, TH, TW, CC, EPI) \
    static ::tpz::SplitRegistrar<::tpz::SplitCfg<K, D, MT, TH, TW, CC, 4, 1>, EPI> TPZ_CAT(tpz_sreg_, __COUNTER__);
// 4-wave workgroups, two per CU
#define TPZ_SPLIT4(K, D, MT, TH, TW, CC, EPI) \
    static ::tpz::SplitRegistrar<::tpz::SplitCfg<K, D, MT, TH, TW, CC, 4>, EPI> TPZ_CAT(tpz_sreg_, __COUNTER__);
#define TPZ_SPLIT_RESID_S(K, D, MT, TH, TW, CC, S)        \
    TPZ_SPLIT_S(K, D, MT, TH, TW, CC, S, ::tpz::EPI_PLAIN) \
    TPZ_SPLIT_S(K, D, MT, TH, TW, CC, S, ::tpz::EPI_RES)   \
    TPZ_SPLIT_S(K, D, MT, TH, TW, CC, S, ::tpz::EPI_RES_POST)
#define TPZ_SPLIT4_RESID(K, D, MT, TH, TW, CC)        \
    TPZ_SPLIT4(K, D, MT, TH, TW, CC, ::tpz::EPI_PLAIN) \
    TPZ_SPLIT4(K, D, MT, TH, TW, CC, ::tpz::EPI_RES)   \
    TPZ_SPLIT4(K, D, MT, TH, TW, CC, ::tpz::EPI_RES_POST)
#define TPZ_SPLIT4_RESID_S(K, D, MT, TH, TW, CC, S)        \
    TPZ_SPLIT4_S(K, D, MT, TH, TW, CC, S, ::tpz::EPI_PLAIN) \
    TPZ_SPLIT4_S(K, D, MT, TH, TW, CC, S, ::tpz::EPI_RES)   \
    TPZ_SPLIT4_S(K, D, MT, TH, TW, CC, S, ::tpz::EPI_RES_POST)
// ResidA layers: plain (conv0), residual and residual + eval-BN (conv1)
#define TPZ_SPLIT_RESID(K, D, MT, TH, TW, CC)        \
    TPZ_SPLIT(K, D, MT, TH, TW, CC, ::tpz::EPI_PLAIN) \
    TPZ_SPLIT(K, D, MT, TH, TW, CC, ::tpz::EPI_RES)   \
    TPZ_SPLIT(K, D, MT, TH, TW, CC, ::tpz::EPI_RES_POST)

}  // namespace tpz
