// fm_tu.hip -- the fast voice kernel with an FmOscillator per voice (bank_kernels.h,
// voice_kernel<..., FM = true>): instantiations and launcher.  Ungated chains without effect
// stages; everything else with a modulation input runs in voice_mod_kernel (mod_tu.hip).
#include "launch.h"

namespace klmi {

hipError_t upload_time_table_fm(const double *t)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(c_time), t, sizeof(double) * kTimeSlots);
}

template <bool ENV, bool FILT, int MODE>
static hipError_t launch(hipStream_t st, const BankView &view, int first, int count, int block_base,
                         uint32_t clock_base, int n_blocks, int span_total, float *partial,
                         void *out)
{
    static std::atomic<uint64_t> configured{0};   // per instantiation
    const size_t smem = tile_bytes<MODE>();
    auto kernel = voice_kernel<ENV, FILT, MODE, false, true>;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kernel), smem, configured);
    if (e != hipSuccess)
        return e;
    const int groups = (count + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(kernel, dim3(groups), dim3(kGroup), smem, st, view, first, count,
                       block_base, clock_base, n_blocks, span_total, partial, out);
    return hipGetLastError();
}

template <int MODE>
static hipError_t by_chain(bool env, bool filt, hipStream_t st, const BankView &view, int first,
                           int count, int block_base, uint32_t clock_base, int n_blocks,
                           int span_total, float *partial, void *out)
{
    if (env && filt)
        return launch<true, true, MODE>(st, view, first, count, block_base, clock_base, n_blocks,
                                        span_total, partial, out);
    if (env)
        return launch<true, false, MODE>(st, view, first, count, block_base, clock_base, n_blocks,
                                         span_total, partial, out);
    if (filt)
        return launch<false, true, MODE>(st, view, first, count, block_base, clock_base, n_blocks,
                                         span_total, partial, out);
    return launch<false, false, MODE>(st, view, first, count, block_base, clock_base, n_blocks,
                                      span_total, partial, out);
}

hipError_t launch_fm_kernel(bool env, bool filt, int mode, hipStream_t st, const BankView &view,
                            int first, int count, int block_base, uint32_t clock_base,
                            int n_blocks, int span_total, float *partial, void *out)
{
#define KLMI_MODE(M)                                                                          \
    case M:                                                                                   \
        return by_chain<M>(env, filt, st, view, first, count, block_base, clock_base, n_blocks, \
                           span_total, partial, out)
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
