// fm_tu.hip -- the fast voice kernel with an audio-rate modulation input per voice
// (bank_kernels.h, voice_kernel<..., MOD>): FmOscillator, Lfo(256) -> Oscillator.frequency,
// Lfo(256) -> PwmOscillator.dutyCycle.  Instantiations and launcher.  Ungated chains without
// effect stages, and gated FM voices under a mixer (synthesizers); everything else with a modulation input runs in voice_mod_kernel (mod_tu.hip).
// Built three times, one object per modulation target (Makefile: -DKLMI_TU_MOD=1, 2, 3 ->
// fm_tu.o, vib_tu.o, pwm_tu.o), so that the 44 instantiations compile side by side; the
// dispatching entry points live in the MT_FM object.
#include "launch.h"

#ifndef KLMI_TU_MOD
#define KLMI_TU_MOD 1
#endif
#define KLMI_CAT2(a, b) a##b
#define KLMI_CAT(a, b) KLMI_CAT2(a, b)

namespace klmi {

constexpr int kTuMod = KLMI_TU_MOD;

// this object's copy of the time table / its launcher, named after the target
hipError_t KLMI_CAT(upload_time_table_modfast_, KLMI_TU_MOD)(const double *t)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(c_time), t, sizeof(double) * kTimeSlots);
}

template <bool ENV, bool FILT, int MODE, int MOD>
static hipError_t launch(hipStream_t st, const BankView &view, int first, int count, int block_base,
                         uint32_t clock_base, int n_blocks, int span_total, float *partial,
                         void *out)
{
    static std::atomic<uint64_t> configured{0};   // per instantiation
    const size_t smem = tile_bytes<MODE>();
    auto kernel = voice_kernel<ENV, FILT, MODE, false, MOD>;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kernel), smem, configured);
    if (e != hipSuccess)
        return e;
    const int groups = (count + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(kernel, dim3(groups), dim3(kGroup), smem, st, view, first, count,
                       block_base, clock_base, n_blocks, span_total, partial, out);
    return hipGetLastError();
}

template <int MODE, int MOD>
static hipError_t by_chain(bool env, bool filt, hipStream_t st, const BankView &view, int first,
                           int count, int block_base, uint32_t clock_base, int n_blocks,
                           int span_total, float *partial, void *out)
{
    if (env && filt)
        return launch<true, true, MODE, MOD>(st, view, first, count, block_base, clock_base, n_blocks,
                                        span_total, partial, out);
    if (env)
        return launch<true, false, MODE, MOD>(st, view, first, count, block_base, clock_base, n_blocks,
                                         span_total, partial, out);
    if (filt)
        return launch<false, true, MODE, MOD>(st, view, first, count, block_base, clock_base, n_blocks,
                                         span_total, partial, out);
    return launch<false, false, MODE, MOD>(st, view, first, count, block_base, clock_base, n_blocks,
                                      span_total, partial, out);
}

// by output mode; FM in every mode, the two Lfo targets in the modes a mixer or a reverb bank
// asks for
hipError_t KLMI_CAT(launch_modfast_, KLMI_TU_MOD)(bool env, bool filt, int mode, hipStream_t st,
                                                  const BankView &view, int first, int count,
                                                  int block_base, uint32_t clock_base, int n_blocks,
                                                  int span_total, float *partial, void *out)
{
#define KLMI_MODE(M)                                                                          \
    case M:                                                                                   \
        return by_chain<M, kTuMod>(env, filt, st, view, first, count, block_base, clock_base, \
                                   n_blocks, span_total, partial, out)
    switch (mode) {
        KLMI_MODE(OUT_MIX_MONO);
        KLMI_MODE(OUT_MIX_STEREO);
        KLMI_MODE(OUT_VOICES_F64);
#if KLMI_TU_MOD == 1
        KLMI_MODE(OUT_VOICES_F32);
        KLMI_MODE(OUT_BUS_F64);
#endif
    default:
        return hipErrorInvalidValue;
    }
#undef KLMI_MODE
}

#if KLMI_TU_MOD == 1
// synthesizer voices with an FmOscillator: the gated form (oscillator x envelope under a mixer)
template <int MODE>
static hipError_t launch_gated_fm(hipStream_t st, const BankView &view, int first, int count,
                                  int block_base, uint32_t clock_base, int n_blocks, int span_total,
                                  float *partial, void *out)
{
    static std::atomic<uint64_t> configured{0};   // per instantiation
    const size_t smem = tile_bytes<MODE>();
    auto kernel = voice_kernel<true, false, MODE, true, MT_FM>;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kernel), smem, configured);
    if (e != hipSuccess)
        return e;
    const int groups = (count + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(kernel, dim3(groups), dim3(kGroup), smem, st, view, first, count,
                       block_base, clock_base, n_blocks, span_total, partial, out);
    return hipGetLastError();
}

hipError_t upload_time_table_modfast_2(const double *t);
hipError_t upload_time_table_modfast_3(const double *t);
hipError_t launch_modfast_2(bool, bool, int, hipStream_t, const BankView &, int, int, int, uint32_t,
                            int, int, float *, void *);
hipError_t launch_modfast_3(bool, bool, int, hipStream_t, const BankView &, int, int, int, uint32_t,
                            int, int, float *, void *);

hipError_t upload_time_table_fm(const double *t)
{
    hipError_t e = upload_time_table_modfast_1(t);
    if (e == hipSuccess)
        e = upload_time_table_modfast_2(t);
    if (e == hipSuccess)
        e = upload_time_table_modfast_3(t);
    return e;
}

bool mod_fast_form(const BankView &view, bool plain_chain, bool env, bool filt, int mode)
{
    if (!plain_chain)
        return false;
    if (view.gated || view.poly != 0)   // synthesizer banks: FM voices without a Filter, mixed down
        return view.gated && view.mod_target == MT_FM && env && !filt && view.n_fx == 0 &&
               view.lfo_par == nullptr && (mode == OUT_MIX_MONO || mode == OUT_MIX_STEREO);
    if (view.mod_target == MT_FM)
        return true;
    return (view.mod_target == MT_FREQUENCY || view.mod_target == MT_DUTY) &&
           (mode == OUT_MIX_MONO || mode == OUT_MIX_STEREO || mode == OUT_VOICES_F64);
}

hipError_t launch_fm_kernel(bool env, bool filt, int mode, hipStream_t st, const BankView &view,
                            int first, int count, int block_base, uint32_t clock_base,
                            int n_blocks, int span_total, float *partial, void *out)
{
    if (view.gated)
        return mode == OUT_MIX_STEREO
                   ? launch_gated_fm<OUT_MIX_STEREO>(st, view, first, count, block_base, clock_base,
                                                     n_blocks, span_total, partial, out)
                   : launch_gated_fm<OUT_MIX_MONO>(st, view, first, count, block_base, clock_base,
                                                   n_blocks, span_total, partial, out);
    auto target = view.mod_target == MT_FM          ? launch_modfast_1
                  : view.mod_target == MT_FREQUENCY ? launch_modfast_2
                                                    : launch_modfast_3;
    return target(env, filt, mode, st, view, first, count, block_base, clock_base, n_blocks,
                  span_total, partial, out);
}
#endif

}  // namespace klmi
