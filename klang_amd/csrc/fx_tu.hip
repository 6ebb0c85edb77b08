// fx_tu.hip -- the voice kernel with a compile-time effect program (fx_kernels.h):
// the programs it is built for, the check that maps a bank's stage list onto one of them,
// and the launcher.  Everything else with effect stages runs in voice_mod_kernel (mod_tu.hip).
#include "launch.h"
#include "fx_kernels.h"

namespace klmi {

hipError_t upload_time_table_fx(const double *t)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(c_time), t, sizeof(double) * kTimeSlots);
}

// Tremolo / Delay / Gain behind the optional Filter, in the orders the reference's patches
// use them (examples/tempo_aware_effects.py, arpeggiator_demo.py, aeolian_arp.py)
#define KLMI_FX_PROGRAMS(X)                  \
    X(fx_program(FX_TREMOLO))                \
    X(fx_program(FX_DELAY))                  \
    X(fx_program(FX_GAIN))                   \
    X(fx_program(FX_TREMOLO, FX_GAIN))       \
    X(fx_program(FX_DELAY, FX_GAIN))         \
    X(fx_program(FX_TREMOLO, FX_DELAY))      \
    X(fx_program(FX_DELAY, FX_TREMOLO))

uint32_t fx_program_of(const BankView &view, bool filt, int mode)
{
    if (view.mod_target != MT_NONE || view.gated || view.poly != 0 || view.fx_block_major)
        return 0;
    if (mode != OUT_MIX_MONO && mode != OUT_MIX_STEREO && mode != OUT_VOICES_F64)
        return 0;
    const int first = filt ? 1 : 0;
    const int stages = view.n_fx - first;
    if (stages < 1 || stages > 3 || (filt && view.fx_kind[0] != FX_FILTER))
        return 0;
    uint32_t prog = 0;
    for (int s = 0; s < stages; ++s)
        prog |= (uint32_t)view.fx_kind[first + s] << (4 * s);
#define X(P) \
    if (prog == (P)) \
        return prog;
    KLMI_FX_PROGRAMS(X)
#undef X
    return 0;
}

template <bool ENV, bool FILT, int MODE, uint32_t PROG>
static hipError_t launch(hipStream_t st, const BankView &view, int first, int count, int block_base,
                         uint32_t clock_base, int n_blocks, int span_total, float *partial,
                         void *out)
{
    static std::atomic<uint64_t> configured{0};   // per instantiation
    const size_t smem = tile_bytes<MODE>();
    auto kernel = voice_fx_kernel<ENV, FILT, MODE, PROG>;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kernel), smem, configured);
    if (e != hipSuccess)
        return e;
    const int groups = (count + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(kernel, dim3(groups), dim3(kGroup), smem, st, view, first, count,
                       block_base, clock_base, n_blocks, span_total, partial, out);
    return hipGetLastError();
}

template <int MODE, uint32_t PROG>
static hipError_t by_chain(bool env, bool filt, hipStream_t st, const BankView &view, int first,
                           int count, int block_base, uint32_t clock_base, int n_blocks,
                           int span_total, float *partial, void *out)
{
    if (env && filt)
        return launch<true, true, MODE, PROG>(st, view, first, count, block_base, clock_base,
                                              n_blocks, span_total, partial, out);
    if (env)
        return launch<true, false, MODE, PROG>(st, view, first, count, block_base, clock_base,
                                               n_blocks, span_total, partial, out);
    if (filt)
        return launch<false, true, MODE, PROG>(st, view, first, count, block_base, clock_base,
                                               n_blocks, span_total, partial, out);
    return launch<false, false, MODE, PROG>(st, view, first, count, block_base, clock_base,
                                            n_blocks, span_total, partial, out);
}

template <int MODE>
static hipError_t by_program(uint32_t prog, bool env, bool filt, hipStream_t st,
                             const BankView &view, int first, int count, int block_base,
                             uint32_t clock_base, int n_blocks, int span_total, float *partial,
                             void *out)
{
#define X(P)                                                                                  \
    if (prog == (P))                                                                          \
        return by_chain<MODE, (P)>(env, filt, st, view, first, count, block_base, clock_base, \
                                   n_blocks, span_total, partial, out);
    KLMI_FX_PROGRAMS(X)
#undef X
    return hipErrorInvalidValue;
}

hipError_t launch_fx_kernel(uint32_t prog, bool env, bool filt, int mode, hipStream_t st,
                            const BankView &view, int first, int count, int block_base,
                            uint32_t clock_base, int n_blocks, int span_total, float *partial,
                            void *out)
{
    switch (mode) {
    case OUT_MIX_MONO:
        return by_program<OUT_MIX_MONO>(prog, env, filt, st, view, first, count, block_base,
                                        clock_base, n_blocks, span_total, partial, out);
    case OUT_MIX_STEREO:
        return by_program<OUT_MIX_STEREO>(prog, env, filt, st, view, first, count, block_base,
                                          clock_base, n_blocks, span_total, partial, out);
    case OUT_VOICES_F64:
        return by_program<OUT_VOICES_F64>(prog, env, filt, st, view, first, count, block_base,
                                          clock_base, n_blocks, span_total, partial, out);
    default:
        return hipErrorInvalidValue;
    }
}

}  // namespace klmi
