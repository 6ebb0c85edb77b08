// fx_tu.hip -- the voice kernel with a compile-time effect program (fx_kernels.h):
// the programs it is built for, the check that maps a bank's stage list onto one of them,
// and the launcher.  Everything else with effect stages runs in voice_mod_kernel (mod_tu.hip).
#include "launch.h"
#include "fx_kernels.h"

namespace klmi {

#ifndef KLMI_FX_HALF
#define KLMI_FX_HALF 0
#endif
#if KLMI_FX_HALF == 0
hipError_t upload_time_table_fx(const double *t)
#elif KLMI_FX_HALF == 1
hipError_t upload_time_table_fx_b(const double *t)
#else
hipError_t upload_time_table_fx_c(const double *t)
#endif
{
    return hipMemcpyToSymbol(HIP_SYMBOL(c_time), t, sizeof(double) * kTimeSlots);
}

// Tremolo / Delay / element-wise stages behind the optional Filter, in the orders the
// reference's patches use them (examples/tempo_aware_effects.py, arpeggiator_demo.py,
// aeolian_arp.py) and the short all-element-wise chains (Gain -> Bitcrusher -> Subsampler ...).
// FX_ELEM: the slot that stands for any element-wise kind (fx_kernels.h).  List C: a
// RingModulator with element-wise stages around it.  The list is split over three objects
// (KLMI_FX_HALF 0 / 1 / 2) that compile side by side.
#define KLMI_FX_PROGRAMS_A(X)                \
    X(fx_program(FX_TREMOLO))                \
    X(fx_program(FX_DELAY))                  \
    X(fx_program(FX_GAIN))                   \
    X(fx_program(FX_TREMOLO, FX_GAIN))       \
    X(fx_program(FX_DELAY, FX_GAIN))         \
    X(fx_program(FX_TREMOLO, FX_DELAY))      \
    X(fx_program(FX_DELAY, FX_TREMOLO))
#define KLMI_FX_PROGRAMS_B(X)                \
    X(fx_program(FX_ELEM))                   \
    X(fx_program(FX_ELEM, FX_ELEM))          \
    X(fx_program(FX_ELEM, FX_ELEM, FX_ELEM)) \
    X(fx_program(FX_ELEM, FX_TREMOLO))       \
    X(fx_program(FX_TREMOLO, FX_ELEM))       \
    X(fx_program(FX_ELEM, FX_DELAY))         \
    X(fx_program(FX_DELAY, FX_ELEM))         \
    X(fx_program(FX_ELEM, FX_DELAY, FX_ELEM))
#define KLMI_FX_PROGRAMS_C(X)                \
    X(fx_program(FX_RINGMOD))                \
    X(fx_program(FX_RINGMOD, FX_ELEM))       \
    X(fx_program(FX_ELEM, FX_RINGMOD))       \
    X(fx_program(FX_RINGMOD, FX_ELEM, FX_ELEM))
#if KLMI_FX_HALF == 0
#define KLMI_FX_PROGRAMS_HERE(X) KLMI_FX_PROGRAMS_A(X)
#define KLMI_FX_LAUNCH launch_fx_kernel_a
#elif KLMI_FX_HALF == 1
#define KLMI_FX_PROGRAMS_HERE(X) KLMI_FX_PROGRAMS_B(X)
#define KLMI_FX_LAUNCH launch_fx_kernel_b
#else
#define KLMI_FX_PROGRAMS_HERE(X) KLMI_FX_PROGRAMS_C(X)
#define KLMI_FX_LAUNCH launch_fx_kernel_c
#endif

static bool fx_elementwise(int kind)
{
    return kind == FX_GAIN || kind == FX_TRANSFORM || kind == FX_TANH || kind == FX_OCTAVE ||
           kind == FX_SUBSAMPLE || kind == FX_BITCRUSH;
}

#if KLMI_FX_HALF == 0
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
    // the stage list as it is (plain Gain slots) ...
    uint32_t prog = 0, generic = 0;
    for (int s = 0; s < stages; ++s) {
        const int kind = view.fx_kind[first + s];
        prog |= (uint32_t)kind << (4 * s);
        generic |= (uint32_t)(fx_elementwise(kind) ? FX_ELEM : kind) << (4 * s);
    }
#define X(P) \
    if (prog == (P)) \
        return prog;
    KLMI_FX_PROGRAMS_A(X)
#undef X
    // ... or with every element-wise stage in the generic slot
#define X(P) \
    if (generic == (P)) \
        return generic;
    KLMI_FX_PROGRAMS_B(X)
    KLMI_FX_PROGRAMS_C(X)
#undef X
    return 0;
}
#endif  // KLMI_FX_HALF == 0

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
    KLMI_FX_PROGRAMS_HERE(X)
#undef X
    return hipErrorInvalidValue;
}

hipError_t KLMI_FX_LAUNCH(uint32_t prog, bool env, bool filt, int mode, hipStream_t st,
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

#if KLMI_FX_HALF == 0
hipError_t launch_fx_kernel_b(uint32_t prog, bool env, bool filt, int mode, hipStream_t st,
                              const BankView &view, int first, int count, int block_base,
                              uint32_t clock_base, int n_blocks, int span_total, float *partial,
                              void *out);
hipError_t launch_fx_kernel_c(uint32_t prog, bool env, bool filt, int mode, hipStream_t st,
                              const BankView &view, int first, int count, int block_base,
                              uint32_t clock_base, int n_blocks, int span_total, float *partial,
                              void *out);

hipError_t launch_fx_kernel(uint32_t prog, bool env, bool filt, int mode, hipStream_t st,
                            const BankView &view, int first, int count, int block_base,
                            uint32_t clock_base, int n_blocks, int span_total, float *partial,
                            void *out)
{
#define X(P) \
    if (prog == (P)) \
        return launch_fx_kernel_a(prog, env, filt, mode, st, view, first, count, block_base, \
                                  clock_base, n_blocks, span_total, partial, out);
    KLMI_FX_PROGRAMS_A(X)
#undef X
#define X(P) \
    if (prog == (P)) \
        return launch_fx_kernel_c(prog, env, filt, mode, st, view, first, count, block_base, \
                                  clock_base, n_blocks, span_total, partial, out);
    KLMI_FX_PROGRAMS_C(X)
#undef X
    return launch_fx_kernel_b(prog, env, filt, mode, st, view, first, count, block_base, clock_base,
                              n_blocks, span_total, partial, out);
}
#endif

}  // namespace klmi
