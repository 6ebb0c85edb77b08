// mod_tu.hip -- the general voice-chain kernel's instantiations and launcher (mod_kernels.h).
#include "launch.h"
#include "mod_kernels.h"

namespace klmi {

hipError_t upload_time_table_mod(const double *t)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(c_time), t, sizeof(double) * kTimeSlots);
}

template <int MODE, bool HAS_FX>
static hipError_t launch(bool env, bool filt, hipStream_t st, const BankView &view, int first,
                         int count, int block_base, uint32_t clock_base, int n_blocks,
                         int span_total, float *partial, void *out)
{
    static std::atomic<uint64_t> configured{0};   // per instantiation
    const size_t smem = tile_bytes<MODE>() + (HAS_FX ? fx_strip_bytes<MODE>() : 0);
    auto kernel = voice_mod_kernel<MODE, HAS_FX>;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kernel), smem, configured);
    if (e != hipSuccess)
        return e;
    const int groups = (count + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(kernel, dim3(groups), dim3(kGroup), smem, st, view, (int)env, (int)filt, first,
                       count, block_base, clock_base, n_blocks, span_total, partial, out);
    return hipGetLastError();
}

hipError_t launch_mod_kernel(bool env, bool filt, int mode, hipStream_t st, const BankView &view,
                             int first, int count, int block_base, uint32_t clock_base,
                             int n_blocks, int span_total, float *partial, void *out)
{
#define KLMI_MODE(M)                                                                         \
    case M:                                                                                  \
        return view.n_fx > 0                                                                 \
                   ? launch<M, true>(env, filt, st, view, first, count, block_base, clock_base,  \
                                     n_blocks, span_total, partial, out)                     \
                   : launch<M, false>(env, filt, st, view, first, count, block_base, clock_base, \
                                      n_blocks, span_total, partial, out)
    switch (mode) {
        KLMI_MODE(OUT_MIX_MONO);
        KLMI_MODE(OUT_MIX_STEREO);
        KLMI_MODE(OUT_VOICES_F32);
        KLMI_MODE(OUT_VOICES_F64);
        KLMI_MODE(OUT_BUS_F64);
    default:
        return hipErrorInvalidValue;
    }
#undef KLMI_MODE
}

}  // namespace klmi
