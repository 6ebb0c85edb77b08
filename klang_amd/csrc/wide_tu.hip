// wide_tu.hip -- instantiations and launcher of voice_wide_kernel (wide_kernels.h).
#include "launch.h"
#include "wide_kernels.h"

#include <cstdlib>

namespace klmi {

int wide_lanes_log2(const BankView &view, int count, int mode, int n_cus)
{
    static const int forced = [] {
        const char *e = getenv("KLMI_WIDE_LANES_LOG2");   // tuning knob: 0 switches it off
        return e ? atoi(e) : -1;
    }();
    const bool plain_chain = view.n_fx == 0 || (view.n_fx == 1 && view.fx_kind[0] == FX_FILTER);
    const bool mode_ok = mode == OUT_MIX_MONO || mode == OUT_MIX_STEREO || mode == OUT_VOICES_F32 ||
                         mode == OUT_VOICES_F64;
    if (view.mod_target != MT_NONE || !plain_chain || view.gated || !mode_ok || count <= 0)
        return 0;
    // the widest group that still leaves at most 1.25 waves per SIMD
    const long budget = (long)n_cus * 4 * 5 / 4;
    int best = 0;
    // (groups of 2 and 4 lanes measured slower than one lane per voice: the replicated chain
    // outweighs the extra waves)
    for (int l = 3; l <= kWideMaxLanesLog2; ++l) {
        const long waves = ((long)count + (64 >> l) - 1) / (64 >> l);
        if (waves <= budget)
            best = l;
    }
    if (forced >= 0)
        best = forced > kWideMaxLanesLog2 ? kWideMaxLanesLog2 : forced;
    return best;
}

int voice_partial_rows(const BankView &view, int count, int mode, int n_cus)
{
    const int l = wide_lanes_log2(view, count, mode, n_cus);
    const int per_wave = 64 >> l;
    const int waves = (count + per_wave - 1) / per_wave;
    return (waves + kWavesPerGroup - 1) / kWavesPerGroup * kWavesPerGroup;
}

template <bool ENV, bool FILT, int MODE>
static hipError_t launch(hipStream_t st, const BankView &view, int first, int count, int block_base,
                         uint32_t clock_base, int n_blocks, int span_total, int lanes_log2,
                         float *partial, void *out)
{
    static std::atomic<uint64_t> configured{0};   // per instantiation
    const size_t smem = wide_lds_bytes<MODE>();
    auto kernel = voice_wide_kernel<ENV, FILT, MODE>;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kernel), smem, configured);
    if (e != hipSuccess)
        return e;
    const int per_wave = 64 >> lanes_log2;
    const int waves = (count + per_wave - 1) / per_wave;
    const int groups = (waves + kWavesPerGroup - 1) / kWavesPerGroup;
    hipLaunchKernelGGL(kernel, dim3(groups), dim3(kGroup), smem, st, view, first, count, block_base,
                       clock_base, n_blocks, span_total, lanes_log2, partial, out);
    return hipGetLastError();
}

template <int MODE>
static hipError_t by_chain(bool env, bool filt, hipStream_t st, const BankView &view, int first,
                           int count, int block_base, uint32_t clock_base, int n_blocks,
                           int span_total, int lanes_log2, float *partial, void *out)
{
    if (env && filt)
        return launch<true, true, MODE>(st, view, first, count, block_base, clock_base, n_blocks,
                                        span_total, lanes_log2, partial, out);
    if (env)
        return launch<true, false, MODE>(st, view, first, count, block_base, clock_base, n_blocks,
                                         span_total, lanes_log2, partial, out);
    if (filt)
        return launch<false, true, MODE>(st, view, first, count, block_base, clock_base, n_blocks,
                                         span_total, lanes_log2, partial, out);
    return launch<false, false, MODE>(st, view, first, count, block_base, clock_base, n_blocks,
                                      span_total, lanes_log2, partial, out);
}

hipError_t launch_wide_kernel(bool env, bool filt, int mode, hipStream_t st, const BankView &view,
                              int first, int count, int block_base, uint32_t clock_base,
                              int n_blocks, int span_total, int lanes_log2, float *partial,
                              void *out)
{
    switch (mode) {
    case OUT_MIX_MONO:
        return by_chain<OUT_MIX_MONO>(env, filt, st, view, first, count, block_base, clock_base,
                                      n_blocks, span_total, lanes_log2, partial, out);
    case OUT_MIX_STEREO:
        return by_chain<OUT_MIX_STEREO>(env, filt, st, view, first, count, block_base, clock_base,
                                        n_blocks, span_total, lanes_log2, partial, out);
    case OUT_VOICES_F32:
        return by_chain<OUT_VOICES_F32>(env, filt, st, view, first, count, block_base, clock_base,
                                        n_blocks, span_total, lanes_log2, partial, out);
    case OUT_VOICES_F64:
        return by_chain<OUT_VOICES_F64>(env, filt, st, view, first, count, block_base, clock_base,
                                        n_blocks, span_total, lanes_log2, partial, out);
    default:
        return hipErrorInvalidValue;
    }
}

}  // namespace klmi
