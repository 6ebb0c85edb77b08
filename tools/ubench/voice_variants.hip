// Stand-alone timing harness for the fused voice kernel (experiments only):
// builds ONE instantiation of voice_kernel so that a variant compiles in seconds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I klang_amd/csrc \
//         [-D...] tools/ubench/voice_variants.hip -o tools/ubench/bin/vv_<name>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "bank_kernels.h"

using namespace klmi;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class T> T *dev(const std::vector<T> &h)
{
    T *p = nullptr;
    if (hipMalloc(&p, sizeof(T) * h.size()) != hipSuccess) exit(2);
    hipMemcpy(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice);
    return p;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 65536;
    const int n_blocks = argc > 2 ? atoi(argv[2]) : 256;
    const int wave_fn = argc > 3 ? atoi(argv[3]) : W_SAWTOOTH;
    srand(7);
    auto u = [] { return rand() / (double)RAND_MAX; };
    std::vector<double> freq(n), amp(n, 0.7), duty(n, 0.5), phase(n), envv(n), z(n, 0.), gain(n, 1.);
    std::vector<double> envc(7 * (size_t)n);
    std::vector<int32_t> wave(n, wave_fn), crow(n, 0);
    std::vector<uint32_t> flags(n, 0), state(n), evoff(n + 1, 0), cursor(n, 0);
    for (int i = 0; i < n; ++i) {
        freq[i] = 50. * pow(2., 7. * u());
        phase[i] = 6.28 * u();
        envv[i] = 0.9;
        const double c = exp(-1. / (44100. * (0.5 + u())));
        envc[0 * (size_t)n + i] = 0.999, envc[1 * (size_t)n + i] = 0.0012;
        envc[2 * (size_t)n + i] = c, envc[3 * (size_t)n + i] = (0.5 - 0.001) * (1. - c);
        envc[4 * (size_t)n + i] = c, envc[5 * (size_t)n + i] = -0.001 * (1. - c);
        envc[6 * (size_t)n + i] = 0.5;
        state[i] = (uint32_t)S_DECAY | 0x100u;
    }
    std::vector<double> table = {0.02, 0.04, 0.02, -1.56, 0.64};
    double t[kTimeSlots];
    for (int i = 0; i < kTimeSlots; ++i) t[i] = (1. / 44100.) * i;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(c_time), t, sizeof t));

    BankView bk{};
    bk.n = n;
    bk.duty = dev(duty), bk.wave = dev(wave), bk.env_coef = dev(envc), bk.env_flags = dev(flags);
    bk.coef_table = dev(table), bk.gain_l = dev(gain), bk.gain_r = dev(gain);
    bk.phase = dev(phase), bk.env_value = dev(envv), bk.z0 = dev(z), bk.z1 = dev(z);
    bk.env_state = dev(state), bk.frequency_rw = dev(freq), bk.amplitude_rw = dev(amp);
    bk.coef_index_rw = dev(crow), bk.ev_offset = dev(evoff), bk.ev_cursor = dev(cursor);
    const int groups = (n + kGroup - 1) / kGroup;
    const int rows = groups * kWavesPerGroup;
    float *partial;
    CK(hipMalloc(&partial, sizeof(float) * (size_t)rows * n_blocks * kBlock));
    using TT = TileTraits<OUT_MIX_MONO>;
    const size_t smem = sizeof(float) * kWavesPerGroup * TT::chunk * TT::stride;
    auto kernel = voice_kernel<true, true, OUT_MIX_MONO>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(groups), dim3(kGroup), smem, 0, bk, 0, n, 0, 0u, n_blocks, 0, partial, nullptr);
        hipEventRecord(e1);
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    std::vector<float> h(kBlock);
    hipMemcpy(h.data(), partial + (size_t)(n_blocks - 1) * kBlock, sizeof(float) * kBlock, hipMemcpyDeviceToHost);
    double cs = 0;
    for (float x : h) cs += x;
    const double vs = (double)n * n_blocks * kBlock;
    printf("%s n=%d blocks=%d wave=%d: %.3f ms  %.3e voice-samples/s  %.1f cycles@2.4GHz/wave-sample  (checksum %.6g)\n",
           argv[0], n, n_blocks, wave_fn, best, vs / (best * 1e-3), best * 1e-3 * 2.4e9 / (vs / 64 / 1024 * ((n + 65535) / 65536 > 0 ? 1 : 1)) , cs);
    return 0;
}
