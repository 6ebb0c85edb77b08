// voice1_tu.hip -- the fast voice kernel for plain ungated chains with the register budget of
// ONE wave per SIMD (bank_kernels.h, voice_kernel<..., OCC = 1>): what a bank of up to one
// workgroup per compute unit runs (65,536 voices on 256 CUs: BASELINE configs[2]).
#include "launch.h"

namespace klmi {

hipError_t upload_time_table_voice1(const double *t)
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
    auto kernel = voice_kernel<ENV, FILT, MODE, false, MT_NONE, 1>;
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

hipError_t launch_voice_kernel_one_wave(bool env, bool filt, int mode, hipStream_t st,
                                        const BankView &view, int first, int count, int block_base,
                                        uint32_t clock_base, int n_blocks, int span_total,
                                        float *partial, void *out)
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
