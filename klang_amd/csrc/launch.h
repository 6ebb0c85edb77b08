// launch.h -- host-side entry points of the voice kernel families.  Each family is a
// translation unit of its own -- voice_tu.hip / voice1_tu.hip: voice_kernel in its two
// register budgets; fm_tu.hip (three objects): its modulated forms; fx_tu.hip: voice_fx_kernel;
// mod_tu.hip: voice_mod_kernel; pipe_tu.hip: voice_pipe_kernel; reverb_tu.hip:
// reverb_voice_kernel -- so that the library builds
// in parallel and an edit to the host code does not recompile them.  Each object holds its
// own copy of the constant time table (upload_time_table_*).
#pragma once
#include "bank_kernels.h"

#include <atomic>

namespace klmi {

// The voice kernels take more than 64 KiB of dynamic LDS, which has to be allowed per
// kernel and device.  `done` holds one bit per device id below 64 (set after the first
// success; racing threads at worst set the attribute twice); higher ids set it on every launch.
inline hipError_t allow_dynamic_lds(const void *kernel, size_t bytes, std::atomic<uint64_t> &done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return e;
    const uint64_t bit = dev >= 0 && dev < 64 ? (uint64_t)1 << dev : 0;
    if (bit && (done.load(std::memory_order_acquire) & bit))
        return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && bit)
        done.fetch_or(bit, std::memory_order_release);
    return e;
}

// t[n] = DT * n into the unit's own constant table (kTimeSlots doubles)
hipError_t upload_time_table_voice(const double *t);
hipError_t upload_time_table_voice1(const double *t);
hipError_t upload_time_table_mod(const double *t);
hipError_t upload_time_table_fm(const double *t);
hipError_t upload_time_table_fx(const double *t);
hipError_t upload_time_table_fx_b(const double *t);
hipError_t upload_time_table_fx_c(const double *t);

// grid = ceil(count / 256) workgroups; `mode` is an OutMode
hipError_t launch_voice_kernel(bool env, bool filt, bool gated, int mode, hipStream_t st,
                               const BankView &view, int first, int count, int block_base, uint32_t clock_base,
                               int n_blocks, int span_total, float *partial, void *out);
// the same for plain ungated chains, built for one wave per SIMD (voice1_tu.hip): grids of at
// most one workgroup per compute unit
hipError_t launch_voice_kernel_one_wave(bool env, bool filt, int mode, hipStream_t st,
                                        const BankView &view, int first, int count, int block_base,
                                        uint32_t clock_base, int n_blocks, int span_total,
                                        float *partial, void *out);
// voice_kernel<..., MOD> (fm_tu.hip): voices with an audio-rate modulation input (FmOscillator,
// Lfo -> frequency, Lfo -> duty cycle), ungated, plain [Filter] chain -- or gated FM voices
// without a Filter under a mixer (synthesizers); mod_fast_form says
// whether the bank and output mode have an instantiation
bool mod_fast_form(const BankView &view, bool plain_chain, bool env, bool filt, int mode);
hipError_t launch_fm_kernel(bool env, bool filt, int mode, hipStream_t st, const BankView &view,
                            int first, int count, int block_base, uint32_t clock_base,
                            int n_blocks, int span_total, float *partial, void *out);
// voice_fx_kernel (fx_tu.hip): the compiled-in effect program a bank's stage list matches
// (0: none -- the general kernel interprets it), and its launcher
uint32_t fx_program_of(const BankView &view, bool filt, int mode);
hipError_t launch_fx_kernel(uint32_t prog, bool env, bool filt, int mode, hipStream_t st,
                            const BankView &view, int first, int count, int block_base,
                            uint32_t clock_base, int n_blocks, int span_total, float *partial,
                            void *out);
hipError_t launch_mod_kernel(bool env, bool filt, int mode, hipStream_t st, const BankView &view,
                             int first, int count, int block_base, uint32_t clock_base,
                             int n_blocks, int span_total, float *partial, void *out);

// Banks that do not fill the chip (pipe_kernels.h): log2 of the voices a workgroup of the
// pipelined kernel owns, -1 when the chain, the mode or the size asks for the
// one-lane-per-voice kernels; the launcher; and the number of partial-bus rows a mix launch
// of `count` voices writes whichever kernel takes it.
int pipe_voices_log2(const BankView &view, bool filt, int count, int mode, int n_cus);
int voice_partial_rows(const BankView &view, bool filt, int count, int mode, int n_cus);
hipError_t launch_pipe_kernel(bool env, bool filt, int mode, hipStream_t st, const BankView &view,
                              int first, int count, int block_base, uint32_t clock_base,
                              int n_blocks, int span_total, int voices_log2, int n_cus,
                              float *partial, void *out);

// reverb_voice_kernel (reverb_tu.hip): Voice -> Reverb -> Mixer fused, for chains without
// filter, stages or modulation and the default ten combs of >= 256 samples; vpw in {1, 2, 4}.
// One partial row per workgroup: reverb_voice_groups(n, vpw) rows.  dev_view: the bank's view
// in DEVICE memory (the kernel reads it there).
struct ReverbState;
int reverb_voice_groups(int n_voices, int vpw);
hipError_t launch_reverb_voice(bool rings_f32, bool stereo, bool env, int vpw, hipStream_t st,
                               const ReverbState &r, const BankView *dev_view, int block_base,
                               uint32_t clock_base, int n_blocks, float *partial, size_t row_stride);

}  // namespace klmi
