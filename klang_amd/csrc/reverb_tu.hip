// reverb_tu.hip -- reverb_voice_kernel (reverb_voice_kernels.h): Voice -> Reverb -> Mixer in
// one kernel, instantiated per ring storage type, channel count, envelope and voices per wave.
#include "launch.h"
#include "reverb_voice_kernels.h"

namespace klmi {

template <typename RingT, bool STEREO, bool ENV, int VPW>
static hipError_t launch(hipStream_t st, const ReverbState &r, const BankView *view, int block_base,
                         uint32_t clock_base, int n_blocks, float *partial, size_t row_stride)
{
    const int waves = (r.n + VPW - 1) / VPW;
    const int groups = (waves + kReverbWaves - 1) / kReverbWaves;
    hipLaunchKernelGGL((reverb_voice_kernel<RingT, STEREO, ENV, VPW>), dim3(groups),
                       dim3(kReverbThreads), 0, st, r, view, block_base, clock_base, n_blocks,
                       partial, row_stride);
    return hipGetLastError();
}

template <typename RingT, bool STEREO, bool ENV>
static hipError_t by_vpw(int vpw, hipStream_t st, const ReverbState &r, const BankView *view,
                         int block_base, uint32_t clock_base, int n_blocks, float *partial,
                         size_t row_stride)
{
    switch (vpw) {
    case 1: return launch<RingT, STEREO, ENV, 1>(st, r, view, block_base, clock_base, n_blocks, partial, row_stride);
    case 2: return launch<RingT, STEREO, ENV, 2>(st, r, view, block_base, clock_base, n_blocks, partial, row_stride);
    case 4: return launch<RingT, STEREO, ENV, 4>(st, r, view, block_base, clock_base, n_blocks, partial, row_stride);
    default: return hipErrorInvalidValue;
    }
}

template <typename RingT>
static hipError_t by_shape(bool stereo, bool env, int vpw, hipStream_t st, const ReverbState &r,
                           const BankView *view, int block_base, uint32_t clock_base, int n_blocks,
                           float *partial, size_t row_stride)
{
    if (stereo && env)
        return by_vpw<RingT, true, true>(vpw, st, r, view, block_base, clock_base, n_blocks, partial, row_stride);
    if (stereo)
        return by_vpw<RingT, true, false>(vpw, st, r, view, block_base, clock_base, n_blocks, partial, row_stride);
    if (env)
        return by_vpw<RingT, false, true>(vpw, st, r, view, block_base, clock_base, n_blocks, partial, row_stride);
    return by_vpw<RingT, false, false>(vpw, st, r, view, block_base, clock_base, n_blocks, partial, row_stride);
}

int reverb_voice_groups(int n_voices, int vpw)
{
    const int waves = (n_voices + vpw - 1) / vpw;
    return (waves + kReverbWaves - 1) / kReverbWaves;
}

hipError_t launch_reverb_voice(bool rings_f32, bool stereo, bool env, int vpw, hipStream_t st,
                               const ReverbState &r, const BankView *view, int block_base,
                               uint32_t clock_base, int n_blocks, float *partial, size_t row_stride)
{
    if (rings_f32)
        return by_shape<float>(stereo, env, vpw, st, r, view, block_base, clock_base, n_blocks, partial, row_stride);
    return by_shape<double>(stereo, env, vpw, st, r, view, block_base, clock_base, n_blocks, partial, row_stride);
}

}  // namespace klmi
