// pipe_tu.hip -- instantiations and launcher of voice_pipe_kernel (pipe_kernels.h).
#include "launch.h"
#include "pipe_kernels.h"

#include <cstdlib>

namespace klmi {

int pipe_voices_log2(const BankView &view, bool filt, int count, int mode, int n_cus)
{
    const int forced = [] {   // read per call: the parity tests switch it inside one process
        const char *e = getenv("KLMI_PIPE_VOICES_LOG2");   // tuning knob: -1 switches it off
        return e ? atoi(e) : -2;
    }();
    const bool plain_chain = view.n_fx == 0 || (view.n_fx == 1 && view.fx_kind[0] == FX_FILTER);
    const bool mode_ok = mode == OUT_MIX_MONO || mode == OUT_MIX_STEREO || mode == OUT_VOICES_F32 ||
                         mode == OUT_VOICES_F64;
    if (view.mod_target != MT_NONE || !plain_chain || view.gated || !mode_ok || count <= 0 ||
        forced == -1)
        return -1;
    // Measured on MI355X (256 CUs): against one lane per voice the pipeline wins
    // up to about 18,000 voices for chains with a biquad (its serial chain is the longer
    // one) and 24,000 without; beyond that the one-lane kernel has enough waves.
    const int per_cu = [] {
        const char *e = getenv("KLMI_PIPE_GROUPS_PER_CU");   // tuning knob
        const int v = e ? atoi(e) : 0;
        return v >= 1 && v <= 64 ? v : 3;
    }();
    if (forced < 0 && (long)count > (filt ? 72L : 96L) * n_cus)
        return -1;
    // the fewest voices per workgroup that keep the grid within three workgroups per CU
    const long budget = (long)per_cu * n_cus;
    for (int l = 0; (1 << l) <= kPipeMaxVoices; ++l) {
        if (forced >= 0 && l != forced && forced <= 6)
            continue;
        if (((long)count + (1 << l) - 1) >> l <= budget || forced >= 0)
            return l;
    }
    return -1;
}

// The grid for `count` voices in workgroups of 2^l: ceil(count / 2^l) workgroups -- or, when
// that is more than one per compute unit and not a whole number per compute unit, the next
// multiple of n_cus, the first *n_big of them with 2^l voices and the others with half as
// many, so that every compute unit carries the same number of voices.
int pipe_groups(int count, int l, int n_cus, int *n_big)
{
    const bool split = [] {
        const char *e = getenv("KLMI_PIPE_SPLIT");   // tuning knob: 0 = equal workgroups
        return !(e && e[0] == '0');
    }();
    const long groups = ((long)count + (1L << l) - 1) >> l;
    *n_big = (int)groups;
    if (!split || l < 1 || n_cus < 1 || groups <= n_cus || groups % n_cus == 0)
        return (int)groups;
    const long total = (groups + n_cus - 1) / n_cus * n_cus;
    const long halves = ((long)count + (1L << (l - 1)) - 1) >> (l - 1);
    const long a = halves - total, b = 2 * total - halves;
    if (a < 0 || b <= 0)
        return (int)groups;
    *n_big = (int)a;
    return (int)total;
}

int voice_partial_rows(const BankView &view, bool filt, int count, int mode, int n_cus)
{
    const int l = pipe_voices_log2(view, filt, count, mode, n_cus);
    if (l >= 0) {
        int n_big = 0;
        return pipe_groups(count, l, n_cus, &n_big);      // one row per workgroup
    }
    return (count + kGroup - 1) / kGroup * kWavesPerGroup;   // one row per wave of 64 voices
}

template <bool ENV, bool FILT, int MODE>
static hipError_t launch(hipStream_t st, const BankView &view, int first, int count, int block_base,
                         uint32_t clock_base, int n_blocks, int span_total, int voices_log2, int n_cus,
                         float *partial, void *out)
{
    static std::atomic<uint64_t> configured{0};   // per instantiation
    const size_t smem = pipe_lds_bytes(1 << voices_log2);
    auto kernel = voice_pipe_kernel<ENV, FILT, MODE>;
    // sized for the largest group: the attribute is set once per device
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kernel),
                                     pipe_lds_bytes(kPipeMaxVoices), configured);
    if (e != hipSuccess)
        return e;
    int n_big = 0;
    const int groups = pipe_groups(count, voices_log2, n_cus, &n_big);
    hipLaunchKernelGGL(kernel, dim3(groups), dim3(kGroup), smem, st, view, first, count, block_base,
                       clock_base, n_blocks, span_total, voices_log2, n_big, partial, out);
    return hipGetLastError();
}

template <int MODE>
static hipError_t by_chain(bool env, bool filt, hipStream_t st, const BankView &view, int first,
                           int count, int block_base, uint32_t clock_base, int n_blocks,
                           int span_total, int voices_log2, int n_cus, float *partial, void *out)
{
    if (env && filt)
        return launch<true, true, MODE>(st, view, first, count, block_base, clock_base, n_blocks,
                                        span_total, voices_log2, n_cus, partial, out);
    if (env)
        return launch<true, false, MODE>(st, view, first, count, block_base, clock_base, n_blocks,
                                         span_total, voices_log2, n_cus, partial, out);
    if (filt)
        return launch<false, true, MODE>(st, view, first, count, block_base, clock_base, n_blocks,
                                         span_total, voices_log2, n_cus, partial, out);
    return launch<false, false, MODE>(st, view, first, count, block_base, clock_base, n_blocks,
                                      span_total, voices_log2, n_cus, partial, out);
}

hipError_t launch_pipe_kernel(bool env, bool filt, int mode, hipStream_t st, const BankView &view,
                              int first, int count, int block_base, uint32_t clock_base,
                              int n_blocks, int span_total, int voices_log2, int n_cus, float *partial,
                              void *out)
{
    switch (mode) {
    case OUT_MIX_MONO:
        return by_chain<OUT_MIX_MONO>(env, filt, st, view, first, count, block_base, clock_base,
                                      n_blocks, span_total, voices_log2, n_cus, partial, out);
    case OUT_MIX_STEREO:
        return by_chain<OUT_MIX_STEREO>(env, filt, st, view, first, count, block_base, clock_base,
                                        n_blocks, span_total, voices_log2, n_cus, partial, out);
    case OUT_VOICES_F32:
        return by_chain<OUT_VOICES_F32>(env, filt, st, view, first, count, block_base, clock_base,
                                        n_blocks, span_total, voices_log2, n_cus, partial, out);
    case OUT_VOICES_F64:
        return by_chain<OUT_VOICES_F64>(env, filt, st, view, first, count, block_base, clock_base,
                                        n_blocks, span_total, voices_log2, n_cus, partial, out);
    default:
        return hipErrorInvalidValue;
    }
}

}  // namespace klmi
