// klang_mi.hip -- host side of libklang_mi.so: the C-ABI declared in
// include/klang_mi.h over the gfx950 kernels in bank_kernels.h / aux_kernels.h /
// reverb_kernels.h.  HIP only; no CPU fallback.
#include "../../include/klang_mi.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "aux_kernels.h"
#include "bank_kernels.h"
#include "bounce_kernels.h"
#include "bus_kernels.h"
#include "launch.h"
#include "reverb_kernels.h"

using namespace klmi;

// ---------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------
static thread_local std::string g_error;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(KLMI_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                              \
    } while (0)

#define TRY(expr)          \
    do {                   \
        int rc_ = (expr);  \
        if (rc_ != 0)      \
            return rc_;    \
    } while (0)

extern "C" const char *klmi_last_error(void) { return g_error.c_str(); }
extern "C" const char *klmi_version(void) { return "klang_mi 0.3 (gfx950)"; }

// ---------------------------------------------------------------------------------
// host-side design helpers
// ---------------------------------------------------------------------------------
namespace {

const double kSampleRate = KLMI_SAMPLING_RATE;

// effects.py:260-265: np.logspace(log2(20), log2(20000), 1000, base=2)
void make_grid(double *freqs)
{
    const double start = std::log2(20.), stop = std::log2(20000.);
    const double step = (stop - start) / (double)(KLMI_GRID_POINTS - 1);
    for (int i = 0; i < KLMI_GRID_POINTS; ++i) {
        double y = (double)i * step + start;
        if (i == KLMI_GRID_POINTS - 1)
            y = stop;
        freqs[i] = std::pow(2., y);
    }
}

// scipy.signal.butter(2, wn, 'lowpass') (effects.py:266-269): Butterworth
// prototype, pre-warp, bilinear transform, in closed form.
void butter2(double wn, double b[3], double a[3])
{
    const double K = std::tan(kPi * wn / 2.);
    const double K2 = K * K, r2 = std::sqrt(2.);
    const double norm = 1. / (1. + r2 * K + K2);
    b[0] = K2 * norm;
    b[1] = 2. * b[0];
    b[2] = b[0];
    a[0] = 1.;
    a[1] = 2. * (K2 - 1.) * norm;
    a[2] = (1. - r2 * K + K2) * norm;
}

// effects.py:271-274: searchsorted (left) + clip
int grid_index(const double *freqs, double f)
{
    const double *p = std::lower_bound(freqs, freqs + KLMI_GRID_POINTS, f);
    return (int)std::min<ptrdiff_t>(p - freqs, KLMI_GRID_POINTS - 1);
}

// _envelope.c:112-119
double env_coefficient(double rate, double overshoot)
{
    if (rate <= 0.)
        return 0.;
    return std::exp(-std::log((1.0 + overshoot) / overshoot) / rate);
}

// _envelope.c:187-198
void env_coefficients(double attack, double decay, double sustain, double release, double dt,
                      double overshoot, double out[6])
{
    out[0] = env_coefficient(attack / dt, overshoot);
    out[1] = (1. + overshoot) * (1. - out[0]);
    out[2] = env_coefficient(decay / dt, overshoot);
    out[3] = (sustain - overshoot) * (1. - out[2]);
    out[4] = env_coefficient(release / dt, overshoot);
    out[5] = (0. - overshoot) * (1. - out[4]);
}

double clip(double v, double lo, double hi) { return std::fmin(std::fmax(v, lo), hi); }

double pan_law_exponent(double law, double center)  // panning.py:27-31
{
    return (law - 20. * std::log10(2.)) / (2. * 20. * std::log10(center));
}

void panning(double level, int mode, int has_law, double law, double lr[2])  // panning.py:35-81
{
    level = clip(level, -1., 1.);
    const double angle = kTau / 8 * (1. + level);
    double l, r;
    if (mode == 0) {
        l = .5 * (1. - level);
        r = .5 * (1. + level);
    } else if (mode == 1) {
        l = std::cos(angle);
        r = std::sin(angle);
    } else {
        l = std::sqrt(.5 * (1. - level) * std::cos(angle));
        r = std::sqrt(.5 * (1. + level) * std::sin(angle));
    }
    if (has_law) {
        const double center = mode == 0 ? .5 : mode == 1 ? std::cos(kTau / 8)
                                                         : std::sqrt(.5 * std::cos(kTau / 8));
        const double alpha = pan_law_exponent(law, center);
        l = std::pow(l, alpha);
        r = std::pow(r, alpha);
    }
    lr[0] = l;
    lr[1] = r;
}

bool is_prime(int v)
{
    if (v < 2)
        return false;
    for (int d = 2; (long)d * d <= v; ++d)
        if (v % d == 0)
            return false;
    return true;
}

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count)
    {
        release();
        if (count == 0)
            count = 1;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e == hipSuccess)
            n = count;
        return e;
    }
    hipError_t reserve(size_t count)
    {
        return count <= n ? hipSuccess : alloc(count);
    }
};

}  // namespace

// ---------------------------------------------------------------------------------
// scratch of klmi_bank_render_device_noalloc, the entry point that allocates nothing:
// the per-wave partial buses and the slice sums of the bus fold live here instead of in
// hipMalloc'ed buffers.  Every GPU the library runs on holds its own copy of the array (at
// its own address) until the process ends.  In host code `scratch` is only a stand-in for
// the symbol: klmi_ctx_create asks the device for the address (hipGetSymbolAddress, once
// per context, with that device current) and everything below uses that pointer.
// ---------------------------------------------------------------------------------
constexpr size_t kScratchBytes = KLMI_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

// ---------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------
struct klmi_ctx {
    unsigned char *scratch_ptr = nullptr;  // device address of `scratch` on this context's GPU
    int device = 0;
    hipStream_t stream = nullptr;
    int live_objects = 0;            // banks / batches created on this context
    int n_cus = 256;                 // compute units of the device
    std::vector<double> grid;        // 1000 frequencies
    std::vector<double> table;       // [1000][5] b0 b1 b2 a1 a2
    DevBuf<double> d_table, d_grid;
};

static int use(klmi_ctx *ctx)
{
    if (!ctx)
        return fail(KLMI_E_ARG, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return 0;
}

extern "C" int klmi_ctx_create(int device, void *stream, klmi_ctx **out)
{
    if (!out)
        return fail(KLMI_E_ARG, "klmi_ctx_create: out is null");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return fail(KLMI_E_DEVICE, "klmi_ctx_create: no HIP device (this library has no CPU path)");
    if (device < 0 || device >= n_dev)
        return fail(KLMI_E_ARG, "klmi_ctx_create: device %d out of range (0..%d)", device, n_dev - 1);
    HIP_TRY(hipSetDevice(device));
    klmi_ctx *ctx = new klmi_ctx;
    ctx->device = device;
    // NULL is the device's default (null) stream, as everywhere in HIP: it is also what
    // torch.cuda.current_stream().cuda_stream is (0) unless a side stream is current, so
    // launches stay ordered with the caller's own work on that stream.
    ctx->stream = (hipStream_t)stream;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            ctx->n_cus = prop.multiProcessorCount;
    }
    ctx->grid.resize(KLMI_GRID_POINTS);
    make_grid(ctx->grid.data());
    ctx->table.resize((size_t)KLMI_GRID_POINTS * 5);
    for (int i = 0; i < KLMI_GRID_POINTS; ++i) {
        double b[3], a[3];
        butter2(ctx->grid[i] / (KLMI_SAMPLING_RATE / 2), b, a);  // helpers.py:15 NYQUIST (int)
        double *row = &ctx->table[(size_t)i * 5];
        row[0] = b[0], row[1] = b[1], row[2] = b[2], row[3] = a[1], row[4] = a[2];
        if (b[1] != 2. * b[0] || b[2] != b[0]) {  // the kernel's biquad relies on it
            delete ctx;
            return fail(KLMI_E_STATE, "filter table row %d is not b0*[1,2,1]", i);
        }
    }
    double t[kTimeSlots];
    for (int n = 0; n < kTimeSlots; ++n)
        t[n] = kDt * (double)n;  // helpers.py:30  dt * np.arange(length)
    hipError_t e = ctx->d_table.alloc(ctx->table.size());
    if (e == hipSuccess)
        e = ctx->d_grid.alloc(ctx->grid.size());
    if (e == hipSuccess)
        e = hipMemcpy(ctx->d_grid.p, ctx->grid.data(), ctx->grid.size() * sizeof(double),
                      hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(ctx->d_table.p, ctx->table.data(), ctx->table.size() * sizeof(double),
                      hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpyToSymbol(HIP_SYMBOL(c_time), t, sizeof t);   // this unit's (bounce kernels)
    if (e == hipSuccess)
        e = upload_time_table_voice(t);
    if (e == hipSuccess)
        e = upload_time_table_voice1(t);
    if (e == hipSuccess)
        e = upload_time_table_mod(t);
    if (e == hipSuccess)
        e = upload_time_table_fm(t);
    if (e == hipSuccess)
        e = upload_time_table_fx(t);
    if (e == hipSuccess)
        e = upload_time_table_fx_b(t);
    if (e == hipSuccess)
        e = upload_time_table_fx_c(t);
    if (e == hipSuccess) {
        void *p = nullptr;  // this device's copy of the scratch array (the device is current)
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        ctx->scratch_ptr = static_cast<unsigned char *>(p);
        if (e == hipSuccess && !p)
            e = hipErrorInvalidSymbol;
    }
    if (e != hipSuccess) {
        delete ctx;
        return fail(KLMI_E_DEVICE, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return 0;
}

extern "C" int klmi_ctx_destroy(klmi_ctx *ctx)
{
    if (!ctx)
        return 0;
    if (ctx->live_objects != 0)
        return fail(KLMI_E_STATE, "klmi_ctx_destroy: %d object(s) created on this context are "
                                  "still alive; destroy them first", ctx->live_objects);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    delete ctx;
    return 0;
}

extern "C" int klmi_ctx_synchronize(klmi_ctx *ctx)
{
    TRY(use(ctx));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// What the memory system of this GPU gives a plain streaming kernel (bus_kernels.h):
// `reps` launches over `bytes` per buffer, timed with HIP events on the context's stream.
extern "C" int klmi_ctx_stream_rate(klmi_ctx *ctx, int kind, size_t bytes, int reps,
                                    double *gb_per_s)
{
    TRY(use(ctx));
    if (kind < 0 || kind > 3 || bytes < 4096 || bytes > ((size_t)64 << 30) || reps < 1 || !gb_per_s)
        return fail(KLMI_E_ARG, "klmi_ctx_stream_rate: kind 0..3, bytes >= 4096, reps >= 1");
    const size_t n4 = bytes / sizeof(float4);
    DevBuf<float4> a, b;
    hipError_t e = a.alloc(n4);
    if (e == hipSuccess && kind == 0)
        e = b.alloc(n4);
    if (e != hipSuccess)
        return fail(KLMI_E_MEMORY, "klmi_ctx_stream_rate: %s", hipGetErrorString(e));
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemsetAsync(a.p, 0, n4 * sizeof(float4), st));
    if (b.p)
        HIP_TRY(hipMemsetAsync(b.p, 0, n4 * sizeof(float4), st));
    hipEvent_t t0, t1;
    HIP_TRY(hipEventCreate(&t0));
    HIP_TRY(hipEventCreate(&t1));
    const dim3 grid((unsigned)((n4 + 255) / 256)), block(256);
    auto launch = [&] {
        const float4 *src = kind == 0 ? b.p : a.p;
        switch (kind) {
        case 0: hipLaunchKernelGGL(stream_probe_kernel<0>, grid, block, 0, st, src, a.p, n4, 1.f); break;
        case 1: hipLaunchKernelGGL(stream_probe_kernel<1>, grid, block, 0, st, src, a.p, n4, 1.f); break;
        case 2: hipLaunchKernelGGL(stream_probe_kernel<2>, grid, block, 0, st, src, a.p, n4, 1.f); break;
        default: hipLaunchKernelGGL(stream_probe_kernel<3>, grid, block, 0, st, src, a.p, n4, 1.f); break;
        }
    };
    launch();   // warm-up
    (void)hipEventRecord(t0, st);
    for (int r = 0; r < reps; ++r)
        launch();
    (void)hipEventRecord(t1, st);
    e = hipGetLastError();
    if (e == hipSuccess)
        e = hipEventSynchronize(t1);
    float ms = 0.f;
    if (e == hipSuccess)
        e = hipEventElapsedTime(&ms, t0, t1);
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    if (e != hipSuccess)
        return fail(KLMI_E_DEVICE, "klmi_ctx_stream_rate: %s", hipGetErrorString(e));
    const double moved = (double)(n4 * sizeof(float4)) * (kind <= 1 ? 2. : 1.) * reps;
    *gb_per_s = moved / ((double)ms * 1e-3) / 1e9;
    return 0;
}

extern "C" int klmi_filter_grid(double *freqs)
{
    if (!freqs)
        return fail(KLMI_E_ARG, "klmi_filter_grid: null output");
    make_grid(freqs);
    return 0;
}

extern "C" int klmi_filter_grid_index(double frequency)
{
    // built once, by whichever thread comes first (initialisation of a function-local
    // static is synchronised by the language)
    static const std::vector<double> grid = [] {
        std::vector<double> g(KLMI_GRID_POINTS);
        make_grid(g.data());
        return g;
    }();
    return grid_index(grid.data(), frequency);
}

extern "C" int klmi_butter2_lowpass(double wn, double b[3], double a[3])
{
    if (!(wn > 0. && wn < 1.))
        return fail(KLMI_E_ARG, "klmi_butter2_lowpass: Wn must lie in (0, 1), got %g", wn);
    butter2(wn, b, a);
    return 0;
}

extern "C" int klmi_low_pass_coefficients(double frequency, double b[1], double a[2])
{
    if (!b || !a || !(frequency >= 0.))
        return fail(KLMI_E_ARG, "klmi_low_pass_coefficients: frequency must be >= 0");
    a[0] = 1.;
    if (frequency == 0.) {
        b[0] = 0.;
        a[1] = -1.;
    } else if (std::isinf(frequency)) {
        b[0] = 1.;
        a[1] = 0.;
    } else {
        a[1] = -std::exp(-kTau * frequency / kSampleRate);
        b[0] = 1. - std::fabs(a[1]);
    }
    return 0;
}

extern "C" int klmi_envelope_coefficients(double attack, double decay, double sustain,
                                          double release, double dt, double overshoot,
                                          double out[6])
{
    if (attack < 0 || decay < 0 || release < 0 || !(sustain >= 0. && sustain <= 1.) || !(dt > 0))
        return fail(KLMI_E_ARG, "klmi_envelope_coefficients: invalid envelope parameters");
    env_coefficients(attack, decay, sustain, release, dt, clip(overshoot, 1e-9, 1e9), out);
    return 0;
}

extern "C" int klmi_panning_amplitudes(double level, int mode, int has_pan_law, double pan_law,
                                       double lr[2])
{
    if (mode < 0 || mode > 2)
        return fail(KLMI_E_ARG, "Invalid panning mode %d!", mode);  // panning.py:49-50
    if (has_pan_law && pan_law > 0)
        return fail(KLMI_E_ARG, "Pan law has to be negative!");      // panning.py:51-52
    panning(level, mode, has_pan_law, pan_law, lr);
    return 0;
}

extern "C" int klmi_reverb_design(double decay, double pre_delay, int n_echos, int32_t *taps,
                                  double *alphas)
{
    if (n_echos <= 0)
        return fail(KLMI_E_ARG, "klmi_reverb_design: nEchos must be > 0");  // effects.py:584
    int v = (int)(pre_delay * kSampleRate);  // effects.py:589
    if (v < 1)
        v = 1;                               // primes.py:133 default start
    for (int k = 0; k < n_echos; ++k) {
        while (!is_prime(v))
            ++v;
        taps[k] = v;
        // effects.py:592-606 compute_alpha(k / SAMPLING_RATE, decay)
        alphas[k] = decay <= 0. ? 0. : std::exp(-kPi * ((double)v / kSampleRate) / decay);
        ++v;
    }
    return 0;
}

// ---------------------------------------------------------------------------------
// voice bank
// ---------------------------------------------------------------------------------
struct klmi_bank {
    klmi_ctx *ctx = nullptr;
    int n = 0;
    klmi_chain_desc desc{};
    double mix_divisor = 1.;

    // host shadows of the parameters the device arrays are derived from
    std::vector<double> attack, decay, sustain, release, overshoot, gain, pan;
    std::vector<uint8_t> loop, retrigger;
    std::vector<uint8_t> is_external;   // wave id == KLMI_WAVE_EXTERNAL, per voice
    int n_external = 0;                 // how many of them: only then is the input span checked

    // device parameters
    DevBuf<double> duty, env_coef, gain_l, gain_r;
    DevBuf<int32_t> wave;
    DevBuf<uint32_t> env_flags;
    int f64_slots = 0;                   // rows of the float64 state slab
    // effect stages (desc.n_fx > 0): parameters as the blocks take them [8][KLMI_FX_PARAMS][n] on the
    // host, what the kernel needs [8][kFxSlots][n] on the device, the delay stage's rings
    std::vector<double> fx_raw;
    DevBuf<double> fx_par, fx_ring, fx_ring_backup;
    int fx_ring_cap = 0;                 // longest line the allocation holds; 0 = not allocated yet
    bool fx_ring_dirty = true;           // a delay time was set since the lines were last sized
    int fx_ring_stride = 0;              // block-major layout (a line shorter than a buffer):
                                         // doubles per voice row, [length slots][256 tail]
    int fx_ring_tiles = 0;               // tiled layout (all lines >= one buffer): time tiles of 8
                                         // samples held, [tile][voice][8] by ABSOLUTE sample time
    size_t fx_ring_len = 0;              // doubles allocated to the lines, either layout
    int fx_delay_stage = -1;
    // some line is shorter than one buffer: lines are [length][256 tail], the Delay runs
    // block-major through a staging area of [waves][2][256][64] doubles (mod_kernels.h)
    bool fx_block_major = false;
    DevBuf<double> fx_stage;
    DevBuf<double> lfo_par, mod_par;    // [3][n] frequency, scale, offset (allocated when used)
    DevBuf<int32_t> lfo_wave, mod_wave;
    // device state: one slab so that a preview can snapshot/restore it with two copies
    //   f64: phase | env_value | z0 | z1 | frequency | amplitude | lfo phase | mod phase  (8n)
    //   u32: env_state | coef_index | ev_cursor                       (3n)
    DevBuf<double> state_f64, backup_f64;
    DevBuf<uint32_t> state_u32, backup_u32;

    // events: kept with absolute block numbers (block_clock at submission + relative
    // block), uploaded sorted by (voice, block) when the queue changed; the device
    // cursor per voice persists across render calls
    std::vector<klmi_event> pending;
    bool events_dirty = false;
    uint32_t block_clock = 0;  // blocks rendered so far
    DevBuf<uint32_t> ev_offset, ev_block, ev_kind;
    DevBuf<int32_t> ev_index;
    DevBuf<double> ev_velocity, ev_frequency;

    // notes of polyphonic banks (desc.poly_voices > 0): same life cycle as the events,
    // CSR by instrument; every voice keeps its own cursor into its instrument's notes
    std::vector<klmi_note> pending_notes;
    bool notes_dirty = false;
    DevBuf<uint32_t> nt_offset, nt_block;
    DevBuf<int32_t> nt_pitch;
    DevBuf<double> nt_velocity, nt_frequency;

    // scratch
    DevBuf<float> partial, dev_mix, dev_voices_f32;
    DevBuf<double> stage;  // bus reduction, slice sums
    DevBuf<double> dev_voices_f64, ext_input, bus_f64;
    int last_mix_blocks = 0;  // blocks of the last klmi_bank_render, still in dev_mix
    int ext_blocks = 0;   // blocks supplied by klmi_bank_set_input
    int ext_cursor = 0;   // blocks of it consumed by renders so far

    // offline bounce (klmi_bank_bounce)
    DevBuf<float> bounce_frames, bounce_peak;
    DevBuf<double> bounce_time;
    DevBuf<int16_t> bounce_pcm;

    // reverb (has_reverb)
    ReverbState reverb;
    std::vector<int32_t> taps;
    std::vector<double> alphas;
    DevBuf<double> rings, d_alphas;      // float64 delay lines (desc.reverb_rings_f64 != 0)
    DevBuf<float> rings32;               // float32 storage of the same lines
    DevBuf<int32_t> d_taps, d_ring_offset;
    std::vector<int32_t> ring_pos;  // host-tracked, identical for every voice
    DevBuf<float> rv_partial;
    hipStream_t aux_stream = nullptr;            // dry-voice kernels of reverb banks
    DevBuf<BankView> d_view;                     // the view, for kernels that read it in memory
    hipEvent_t rv_start = nullptr, rv_ready[2] = {nullptr, nullptr}, rv_done[2] = {nullptr, nullptr};

    // timing
    std::vector<hipEvent_t> ev_start, ev_stop;
    int timed_launches = 0;

    double *f64(int slot) { return state_f64.p + (size_t)slot * n; }
    uint32_t *u32(int slot) { return state_u32.p + (size_t)slot * n; }
};

enum { SL_PHASE = 0, SL_ENV_VALUE, SL_Z0, SL_Z1, SL_FREQ, SL_AMP, SL_LFO_PHASE, SL_MOD_PHASE, SL_F64_COUNT,
       SL_FX_STATE = SL_F64_COUNT };  // banks with effect stages: + [8][2] slots of stage state
enum { SL_ENV_STATE = 0, SL_COEF_INDEX, SL_EV_CURSOR, SL_NT_CURSOR, SL_ROUND_ROBIN, SL_CUR_PITCH,
       SL_U32_COUNT };

static BankView make_view(klmi_bank *b)
{
    BankView v{};
    v.n = b->n;
    v.duty = b->duty.p;
    v.wave = b->wave.p;
    v.env_coef = b->env_coef.p;
    v.env_flags = b->env_flags.p;
    v.coef_table = b->ctx->d_table.p;
    v.gain_l = b->gain_l.p;
    v.gain_r = b->gain_r.p;
    v.phase = b->f64(SL_PHASE);
    v.env_value = b->f64(SL_ENV_VALUE);
    v.z0 = b->f64(SL_Z0);
    v.z1 = b->f64(SL_Z1);
    v.frequency_rw = b->f64(SL_FREQ);
    v.amplitude_rw = b->f64(SL_AMP);
    v.env_state = b->u32(SL_ENV_STATE);
    v.coef_index_rw = reinterpret_cast<int32_t *>(b->u32(SL_COEF_INDEX));
    v.ev_cursor = b->u32(SL_EV_CURSOR);
    v.ev_offset = b->ev_offset.p;
    v.ev_block = b->ev_block.p;
    v.ev_kind = b->ev_kind.p;
    v.ev_index = b->ev_index.p;
    v.ev_velocity = b->ev_velocity.p;
    v.ev_frequency = b->ev_frequency.p;
    v.ext_input = b->ext_input.p ? b->ext_input.p + (size_t)b->ext_cursor * kBlock * b->n : nullptr;
    v.lfo_par = b->lfo_par.p;
    v.lfo_wave = b->lfo_wave.p;
    v.lfo_phase = b->f64(SL_LFO_PHASE);
    v.grid = b->ctx->d_grid.p;
    v.mod_target = b->desc.mod_target;
    v.mod_par = b->mod_par.p;
    v.mod_wave = b->mod_wave.p;
    v.mod_phase = b->f64(SL_MOD_PHASE);
    v.n_fx = b->desc.n_fx;
    for (int i = 0; i < KLMI_MAX_FX; ++i)
        v.fx_kind[i] = i < b->desc.n_fx ? b->desc.fx[i] : 0;
    v.fx_par = b->fx_par.p;
    v.fx_state = b->desc.n_fx ? b->f64(SL_FX_STATE) : nullptr;
    v.fx_ring = b->fx_ring.p;
    v.fx_ring_stride = b->fx_ring_stride;
    v.fx_ring_tiles = b->fx_ring_tiles;
    v.fx_block_major = b->fx_block_major ? 1 : 0;
    v.fx_stage = b->fx_stage.p;
    v.gated = b->desc.gated || b->desc.poly_voices > 0;
    v.poly = b->desc.poly_voices;
    v.nt_offset = b->nt_offset.p;
    v.nt_block = b->nt_block.p;
    v.nt_pitch = b->nt_pitch.p;
    v.nt_velocity = b->nt_velocity.p;
    v.nt_frequency = b->nt_frequency.p;
    v.nt_cursor = b->u32(SL_NT_CURSOR);
    v.round_robin = b->u32(SL_ROUND_ROBIN);
    v.current_pitch = reinterpret_cast<int32_t *>(b->u32(SL_CUR_PITCH));
    return v;
}

static int upload_env(klmi_bank *b, int first, int count)
{
    std::vector<double> coef((size_t)7 * count);
    std::vector<uint32_t> flags(count);
    for (int i = 0; i < count; ++i) {
        const int v = first + i;
        double c[6];
        env_coefficients(b->attack[v], b->decay[v], b->sustain[v], b->release[v], kDt,
                         b->overshoot[v], c);
        for (int k = 0; k < 6; ++k)
            coef[(size_t)k * count + i] = c[k];
        coef[(size_t)6 * count + i] = b->sustain[v];
        flags[i] = (b->loop[v] ? 1u : 0u) | (b->retrigger[v] ? 2u : 0u);
    }
    for (int k = 0; k < 7; ++k)
        HIP_TRY(hipMemcpyAsync(b->env_coef.p + (size_t)k * b->n + first, &coef[(size_t)k * count],
                               sizeof(double) * count, hipMemcpyHostToDevice, b->ctx->stream));
    HIP_TRY(hipMemcpyAsync(b->env_flags.p + first, flags.data(), sizeof(uint32_t) * count,
                           hipMemcpyHostToDevice, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return 0;
}

static int upload_gains(klmi_bank *b, int first, int count)
{
    std::vector<double> gl(count), gr(count);
    for (int i = 0; i < count; ++i) {
        const int v = first + i;
        if (b->desc.stereo) {
            double lr[2];
            panning(b->pan[v], b->desc.pan_mode, b->desc.has_pan_law, b->desc.pan_law, lr);
            gl[i] = b->gain[v] * lr[0];  // mixer.py:99  (gain * pan) * channel
            gr[i] = b->gain[v] * lr[1];
        } else {
            gl[i] = b->gain[v];
            gr[i] = 0.;
        }
    }
    HIP_TRY(hipMemcpyAsync(b->gain_l.p + first, gl.data(), sizeof(double) * count,
                           hipMemcpyHostToDevice, b->ctx->stream));
    HIP_TRY(hipMemcpyAsync(b->gain_r.p + first, gr.data(), sizeof(double) * count,
                           hipMemcpyHostToDevice, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return 0;
}

template <typename T>
static int fill_device(klmi_bank *b, T *dst, size_t count, T value)
{
    std::vector<T> h(count, value);
    HIP_TRY(hipMemcpy(dst, h.data(), sizeof(T) * count, hipMemcpyHostToDevice));
    return 0;
}

// Device parameters of one effect stage for voices [first, first+count), derived from the
// host shadow (what the blocks take) exactly like the reference derives them.
static int upload_fx(klmi_bank *b, int stage, int first, int count)
{
    const size_t n = (size_t)b->n;
    const int kind = b->desc.fx[stage];
    std::vector<double> dev((size_t)kFxSlots * count, 0.);
    auto raw = [&](int k, int i) {
        return b->fx_raw[((size_t)stage * KLMI_FX_PARAMS + k) * n + first + i];
    };
    for (int i = 0; i < count; ++i) {
        double d[kFxSlots] = {raw(0, i), raw(1, i), raw(2, i), raw(3, i), 0., 0.};
        if (kind == KLMI_FX_TREMOLO) {
            // effects.py:124-125: decay = max(1e-6, smoothness * min(duty, 1 - duty));
            // coeffs = low_pass_coefficients(1. / decay)
            const double duty = raw(3, i);
            double decay = raw(2, i) * std::min(duty, 1. - duty);
            if (!(decay > 1e-6))
                decay = 1e-6;
            double lb[1], la[2];
            (void)klmi_low_pass_coefficients(1. / decay, lb, la);
            d[2] = duty, d[3] = lb[0], d[4] = la[1];  // rate, depth, duty, b0, a1
        } else if (kind == KLMI_FX_DELAY) {
            d[0] = (double)(int)(raw(0, i) * KLMI_SAMPLING_RATE);  // effects.py:155 length
        } else if (kind == KLMI_FX_RINGMOD) {
            d[4] = raw(4, i);  // frequency, rate, amount, dryWet, lfo wave
        }
        for (int k = 0; k < kFxSlots; ++k)
            dev[(size_t)k * count + i] = d[k];
    }
    for (int k = 0; k < kFxSlots; ++k)
        HIP_TRY(hipMemcpyAsync(b->fx_par.p + ((size_t)stage * kFxSlots + k) * n + first,
                               &dev[(size_t)k * count], sizeof(double) * count,
                               hipMemcpyHostToDevice, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return 0;
}

// Line lengths (samples) the Delay stage would have with `times` for voices [first, first+count)
// and the bank's current times elsewhere: -> longest line, and whether any is shorter than a buffer.
static void delay_line_shape(const klmi_bank *b, int first, int count, const double *times,
                             int *longest, bool *any_short)
{
    const size_t n = (size_t)b->n;
    const double *cur = &b->fx_raw[((size_t)b->fx_delay_stage * KLMI_FX_PARAMS) * n];
    *longest = 0;
    *any_short = false;
    for (size_t v = 0; v < n; ++v) {
        const bool edited = times && (int)v >= first && (int)v < first + count;
        const int len = (int)((edited ? times[v - first] : cur[v]) * KLMI_SAMPLING_RATE);
        *longest = std::max(*longest, len);
        *any_short = *any_short || len < KLMI_BUFFER_SIZE;
    }
}

// The delay stage's lines, zero-initialised (ring_buffer.py:42), in one of two layouts fixed by
// the bank's first render:
//  * tiled by ABSOLUTE time (every line >= one buffer): tiles[(t / 8) % T][voice][t % 8], t =
//    samples since the bank was created.  What a voice reads at time t is what it wrote at
//    t - length, so no per-voice position exists in memory: every write of a wavefront is one
//    contiguous 4 KiB row piece, and so is every read of voices with equal delay times
//    (T = longest line / 8 + 3 tiles);
//  * the RingBuffer's own data array per voice, [n][length slots + 256 tail] (some line is
//    shorter than one buffer: mod_kernels.h delay_block).
// Sized when a delay time was set since the last render; an allocation that is too short
// grows and keeps what the lines hold.
static int ensure_fx_ring(klmi_bank *b, bool may_alloc = true)
{
    if (b->fx_delay_stage < 0 || !b->fx_ring_dirty)
        return 0;
    const size_t n = (size_t)b->n;
    int need = 0;
    bool any_short = false;
    delay_line_shape(b, 0, 0, nullptr, &need, &any_short);
    // (klmi_bank_set_params refuses edits that would change the layout of lines in use)
    if (b->fx_ring_cap > 0 && any_short != b->fx_block_major)
        return fail(KLMI_E_STATE, "Delay times shorter than one buffer (%d samples) have to be set "
                                  "before the bank's first render, for all voices at once, and "
                                  "cannot be left again later", KLMI_BUFFER_SIZE);
    if (need <= b->fx_ring_cap) {
        b->fx_ring_dirty = false;
        return 0;
    }
    if (!may_alloc)
        return fail(KLMI_E_STATE, "the Delay stage's lines are not allocated yet (or a longer "
                                  "time was set): one allocating render (or preview) must come first");
    hipStream_t st = b->ctx->stream;
    if (any_short && !b->fx_stage.p) {
        const size_t waves = ((size_t)b->n + kGroup - 1) / kGroup * kWavesPerGroup;
        hipError_t se = b->fx_stage.alloc(waves * 2 * kBlock * 64);
        if (se != hipSuccess)
            return fail(KLMI_E_MEMORY, "the block-major Delay's staging area (%zu MiB) does not fit: %s",
                        (waves * 2 * kBlock * 64 * sizeof(double)) >> 20, hipGetErrorString(se));
    }
    b->fx_block_major = any_short;
    const int stride = any_short ? (need + KLMI_BUFFER_SIZE + 15) & ~15 : 0;
    const int tiles = any_short ? 0 : need / kDelayTile + 3;
    const size_t len = any_short ? (size_t)stride * n : delay_tiles_len(n, tiles);
    DevBuf<double> grown;
    hipError_t e = grown.alloc(len);
    if (e != hipSuccess)
        return fail(KLMI_E_MEMORY, "delay lines (%d samples x %d voices) do not fit: %s", need, b->n,
                    hipGetErrorString(e));
    // ordered on the context's stream like the kernels that read the lines (a caller's
    // non-blocking stream is not ordered after the null stream)
    HIP_TRY(hipMemsetAsync(grown.p, 0, sizeof(double) * len, st));
    if (b->fx_ring_cap > 0 && any_short) {
        HIP_TRY(hipMemcpy2DAsync(grown.p, sizeof(double) * stride, b->fx_ring.p,
                                 sizeof(double) * b->fx_ring_stride,
                                 sizeof(double) * b->fx_ring_stride, n, hipMemcpyDeviceToDevice, st));
    } else if (b->fx_ring_cap > 0) {
        // the tiles of the last T_old - 1 time groups move to their places modulo the new T
        const unsigned long long now = (unsigned long long)b->block_clock * (kBlock / kDelayTile);
        const int old_tiles = b->fx_ring_tiles;
        for (int k = 1; k < old_tiles && (unsigned long long)k <= now; ++k) {
            const unsigned long long g = now - (unsigned long long)k;
            const int to = (int)(g % (unsigned)tiles), from = (int)(g % (unsigned)old_tiles);
            if (KLMI_DELAY_WAVE_MAJOR) {
                // one 64-voice piece of the tile per wavefront region
                const size_t piece = sizeof(double) * 64 * kDelayTile;
                HIP_TRY(hipMemcpy2DAsync(grown.p + delay_tile_at(n, tiles, to, 0), piece * tiles,
                                         b->fx_ring.p + delay_tile_at(n, old_tiles, from, 0),
                                         piece * old_tiles, piece, (n + 63) / 64,
                                         hipMemcpyDeviceToDevice, st));
            } else {
                HIP_TRY(hipMemcpyAsync(grown.p + delay_tile_at(n, tiles, to, 0),
                                       b->fx_ring.p + delay_tile_at(n, old_tiles, from, 0),
                                       sizeof(double) * n * kDelayTile, hipMemcpyDeviceToDevice, st));
            }
        }
    } else {
        HIP_TRY(hipMemsetAsync(b->f64(SL_FX_STATE + 2 * b->fx_delay_stage), 0, sizeof(double) * n, st));
    }
    HIP_TRY(hipStreamSynchronize(st));   // the old lines are freed below
    std::swap(b->fx_ring.p, grown.p);
    std::swap(b->fx_ring.n, grown.n);
    b->fx_ring_backup.release();
    b->fx_ring_stride = stride;
    b->fx_ring_tiles = tiles;
    b->fx_ring_len = len;
    b->fx_ring_cap = need;
    b->fx_ring_dirty = false;
    return 0;
}

// New delay times for voices [first, first+count): a voice whose line length (in samples)
// changed starts again with an empty line at position 0; the others keep theirs.
static int reset_changed_delay_lines(klmi_bank *b, int first, int count, const double *new_time)
{
    const size_t n = (size_t)b->n;
    const double *old_time = &b->fx_raw[((size_t)b->fx_delay_stage * KLMI_FX_PARAMS) * n + first];
    hipStream_t st = b->ctx->stream;
    int run = -1;   // start of the current run of changed voices
    for (int i = 0; i <= count; ++i) {
        const bool changed = i < count && (int)(old_time[i] * KLMI_SAMPLING_RATE) !=
                                              (int)(new_time[i] * KLMI_SAMPLING_RATE);
        if (changed && run < 0)
            run = i;
        if (!changed && run >= 0) {
            if (b->fx_ring_cap > 0) {
                if (b->fx_block_major)
                    HIP_TRY(hipMemsetAsync(b->fx_ring.p + (size_t)(first + run) * b->fx_ring_stride, 0,
                                           sizeof(double) * (size_t)(i - run) * b->fx_ring_stride, st));
                else if (KLMI_DELAY_WAVE_MAJOR) {   // the voices' columns of every tile, wavefront by wavefront
                    for (int v0 = first + run; v0 < first + i;) {
                        if (v0 % 64 == 0 && first + i - v0 >= 64) {   // whole wavefront regions: contiguous
                            const int v1 = v0 + (first + i - v0) / 64 * 64;
                            HIP_TRY(hipMemsetAsync(b->fx_ring.p + delay_tile_at(n, b->fx_ring_tiles, 0, v0), 0,
                                                   sizeof(double) * (size_t)(v1 - v0) * b->fx_ring_tiles * kDelayTile, st));
                            v0 = v1;
                            continue;
                        }
                        const int v1 = std::min(first + i, (v0 / 64 + 1) * 64);
                        HIP_TRY(hipMemset2DAsync(b->fx_ring.p + delay_tile_at(n, b->fx_ring_tiles, 0, v0),
                                                 sizeof(double) * 64 * kDelayTile, 0,
                                                 sizeof(double) * (size_t)(v1 - v0) * kDelayTile,
                                                 (size_t)b->fx_ring_tiles, st));
                        v0 = v1;
                    }
                } else
                    HIP_TRY(hipMemset2DAsync(b->fx_ring.p + (size_t)(first + run) * kDelayTile,
                                             sizeof(double) * n * kDelayTile, 0,
                                             sizeof(double) * (size_t)(i - run) * kDelayTile,
                                             (size_t)b->fx_ring_tiles, st));
                HIP_TRY(hipMemsetAsync(b->f64(SL_FX_STATE + 2 * b->fx_delay_stage) + first + run, 0,
                                       sizeof(double) * (size_t)(i - run), st));
            }
            b->fx_ring_dirty = true;
            run = -1;
        }
    }
    return 0;
}

static int bank_init_reverb(klmi_bank *b);

extern "C" int klmi_bank_create(klmi_ctx *ctx, int n_voices, const klmi_chain_desc *desc,
                                klmi_bank **out)
{
    TRY(use(ctx));
    if (!out || !desc)
        return fail(KLMI_E_ARG, "klmi_bank_create: null argument");
    if (n_voices <= 0)
        return fail(KLMI_E_ARG, "klmi_bank_create: n_voices must be positive, got %d", n_voices);
    if (desc->struct_size != 0 && desc->struct_size != (int32_t)sizeof(klmi_chain_desc))
        return fail(KLMI_E_ARG, "klmi_bank_create: the caller's klmi_chain_desc is %d bytes, this "
                                "library's (%s) is %zu: rebuild against include/klang_mi.h",
                    desc->struct_size, klmi_version(), sizeof(klmi_chain_desc));
    if (desc->pan_mode < 0 || desc->pan_mode > 2)
        return fail(KLMI_E_ARG, "Invalid panning mode %d!", desc->pan_mode);
    if (desc->has_pan_law && desc->pan_law > 0)
        return fail(KLMI_E_ARG, "Pan law has to be negative!");
    if (desc->cutoff_lfo && !desc->has_filter)
        return fail(KLMI_E_ARG, "klmi_bank_create: cutoff_lfo needs has_filter");
    if (desc->mod_target < KLMI_MOD_NONE || desc->mod_target > KLMI_MOD_DUTY)
        return fail(KLMI_E_ARG, "klmi_bank_create: unknown mod_target %d", desc->mod_target);
    if (desc->n_fx < 0 || desc->n_fx > KLMI_MAX_FX)
        return fail(KLMI_E_ARG, "klmi_bank_create: n_fx %d outside 0..%d", desc->n_fx, KLMI_MAX_FX);
    {
        int seen[KLMI_FX_RINGMOD + 1] = {};
        for (int i = 0; i < desc->n_fx; ++i) {
            if (desc->fx[i] < KLMI_FX_FILTER || desc->fx[i] > KLMI_FX_RINGMOD)
                return fail(KLMI_E_ARG, "klmi_bank_create: unknown effect kind %d at stage %d",
                            desc->fx[i], i);
            seen[desc->fx[i]]++;
        }
        if (seen[KLMI_FX_FILTER] > 1 || seen[KLMI_FX_TREMOLO] > 1 || seen[KLMI_FX_DELAY] > 1 ||
            seen[KLMI_FX_SUBSAMPLE] > 1 || seen[KLMI_FX_RINGMOD] > 1)
            return fail(KLMI_E_ARG, "klmi_bank_create: at most one Filter, Tremolo, Delay, "
                                    "Subsampler and RingModulator stage per chain");
        if (desc->n_fx && (seen[KLMI_FX_FILTER] != 0) != (desc->has_filter != 0))
            return fail(KLMI_E_ARG, "klmi_bank_create: has_filter and a KLMI_FX_FILTER stage go "
                                    "together");
    }
    if (desc->poly_voices < 0 || (desc->poly_voices > 0 && n_voices % desc->poly_voices != 0))
        return fail(KLMI_E_ARG, "klmi_bank_create: %d voices are not a whole number of "
                                "instruments of poly_voices = %d", n_voices, desc->poly_voices);
    if ((desc->gated || desc->poly_voices > 0) &&
        (!desc->has_env || desc->has_filter || desc->n_fx || desc->has_reverb))
        return fail(KLMI_E_ARG, "klmi_bank_create: gated / polyphonic banks are Voice(oscillator, "
                                "envelope) chains: has_env, no filter, effect stages or reverb");
    klmi_bank *b = new klmi_bank;
    b->ctx = ctx;
    b->n = n_voices;
    b->desc = *desc;
    b->mix_divisor = n_voices > 1 ? (double)n_voices : 1.;  // mixer.py:46-47
    if (desc->poly_voices > 0)
        b->mix_divisor = (double)desc->poly_voices;         // synthesizer.py:219
    const size_t n = (size_t)n_voices;
    // defaults: ADSR(.1,.2,.8,1.) envelopes.py:66, overshoot 1e-3, Filter 440 Hz effects.py:353
    b->attack.assign(n, .1);
    b->decay.assign(n, .2);
    b->sustain.assign(n, .8);
    b->release.assign(n, 1.);
    b->overshoot.assign(n, 1e-3);
    b->gain.assign(n, 1.);  // mixer.py:11
    b->pan.assign(n, 0.);   // panning.py:20
    b->loop.assign(n, 0);
    b->retrigger.assign(n, 0);
    b->is_external.assign(n, 0);

    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t r) {
        if (e == hipSuccess)
            e = r;
    };
    chk(b->duty.alloc(n));
    chk(b->env_coef.alloc(7 * n));
    chk(b->gain_l.alloc(n));
    chk(b->gain_r.alloc(n));
    chk(b->wave.alloc(n));
    chk(b->env_flags.alloc(n));
    b->f64_slots = SL_F64_COUNT + (desc->n_fx ? 2 * KLMI_MAX_FX : 0);
    chk(b->state_f64.alloc(b->f64_slots * n));
    chk(b->state_u32.alloc(SL_U32_COUNT * n));
    chk(b->backup_f64.alloc(b->f64_slots * n));
    chk(b->backup_u32.alloc(SL_U32_COUNT * n));
    if (desc->cutoff_lfo) {
        chk(b->lfo_par.alloc(3 * n));
        chk(b->lfo_wave.alloc(n));
    }
    if (desc->mod_target != KLMI_MOD_NONE) {
        chk(b->mod_par.alloc(3 * n));
        chk(b->mod_wave.alloc(n));
    }
    if (desc->n_fx)
        chk(b->fx_par.alloc((size_t)KLMI_MAX_FX * kFxSlots * n));
    chk(b->ev_offset.alloc(n + 1));
    chk(b->ev_block.alloc(1));
    chk(b->ev_kind.alloc(1));
    chk(b->ev_index.alloc(1));
    chk(b->ev_velocity.alloc(1));
    chk(b->ev_frequency.alloc(1));
    if (desc->poly_voices > 0) {
        chk(b->nt_offset.alloc(n / desc->poly_voices + 1));
        chk(b->nt_block.alloc(1));
        chk(b->nt_pitch.alloc(1));
        chk(b->nt_velocity.alloc(1));
        chk(b->nt_frequency.alloc(1));
    }
    if (e != hipSuccess) {
        delete b;
        return fail(KLMI_E_MEMORY, "klmi_bank_create: device allocation failed: %s",
                    hipGetErrorString(e));
    }
    int rc = 0;
    auto step = [&](int r) {
        if (rc == 0)
            rc = r;
    };
    // cleared ON THE CONTEXT'S STREAM, like every launch: a caller's non-blocking stream is not
    // ordered with the null stream, so a null-stream hipMemset could still be running (or
    // run late) when the first render reads the state
    hipStream_t cst = ctx->stream;
    if (hipMemsetAsync(b->state_f64.p, 0, sizeof(double) * b->f64_slots * n, cst) != hipSuccess ||
        hipMemsetAsync(b->state_u32.p, 0, sizeof(uint32_t) * SL_U32_COUNT * n, cst) != hipSuccess ||
        hipMemsetAsync(b->ev_offset.p, 0, sizeof(uint32_t) * (n + 1), cst) != hipSuccess ||
        hipMemsetAsync(b->wave.p, 0, sizeof(int32_t) * n, cst) != hipSuccess ||
        (desc->poly_voices > 0 &&
         hipMemsetAsync(b->nt_offset.p, 0, sizeof(uint32_t) * (n / desc->poly_voices + 1), cst) != hipSuccess))
        step(fail(KLMI_E_DEVICE, "klmi_bank_create: memset failed"));
    step(fill_device<double>(b, b->duty.p, n, .5));           // oscillators.py:245
    step(fill_device<double>(b, b->f64(SL_FREQ), n, 440.));   // oscillators.py:121
    step(fill_device<uint32_t>(b, b->u32(SL_COEF_INDEX), n,
                               (uint32_t)grid_index(ctx->grid.data(), 440.)));
    if (desc->cutoff_lfo) {
        // Lfo defaults (oscillators.py:156-157): 1 Hz, sine, outputRange (0, 1)
        step(fill_device<double>(b, b->lfo_par.p, n, 1.));
        step(fill_device<double>(b, b->lfo_par.p + n, n, .5));
        step(fill_device<double>(b, b->lfo_par.p + 2 * n, n, .5));
        step(fill_device<int32_t>(b, b->lfo_wave.p, n, KLMI_WAVE_SINE));
    }
    if (desc->mod_target != KLMI_MOD_NONE) {
        // FmOscillator defaults (oscillators.py:202-203): modFrequency 10 Hz, intensity 1, sine
        step(fill_device<double>(b, b->mod_par.p, n, 10.));
        step(fill_device<double>(b, b->mod_par.p + n, n, 1.));
        step(fill_device<double>(b, b->mod_par.p + 2 * n, n, 0.));
        step(fill_device<int32_t>(b, b->mod_wave.p, n, KLMI_WAVE_SINE));
    }
    if (desc->n_fx) {
        // the blocks' own defaults (effects.py:62, 89-90, 143, 473; Transformer: identity)
        // (effects.py:62, 89-90, 143, 473, 628-629; Subsampler / Bitcrusher have no defaults
        // worth the name: factor 2, nBits 0)
        static const double defaults[KLMI_FX_RINGMOD + 1][KLMI_FX_PARAMS] = {
            {0}, {0}, {1.}, {1., 0.}, {1.}, {0}, {3., .8, 1., .5}, {1., .1, .5},
            {2.}, {0.}, {500., 2., 20., 1., KLMI_WAVE_SQUARE}};
        b->fx_raw.assign((size_t)KLMI_MAX_FX * KLMI_FX_PARAMS * n, 0.);
        for (int st = 0; st < desc->n_fx; ++st) {
            for (int k = 0; k < KLMI_FX_PARAMS; ++k)
                std::fill_n(b->fx_raw.begin() + ((size_t)st * KLMI_FX_PARAMS + k) * n, n,
                            defaults[desc->fx[st]][k]);
            if (desc->fx[st] == KLMI_FX_DELAY)
                b->fx_delay_stage = st;
            if (rc == 0)
                step(upload_fx(b, st, 0, n_voices));
        }
    }
    if (rc == 0)
        step(upload_env(b, 0, n_voices));
    if (rc == 0)
        step(upload_gains(b, 0, n_voices));
    if (rc == 0 && desc->has_reverb)
        step(bank_init_reverb(b));
    // the fills above went through the null stream: finished before a caller's
    // non-blocking stream sees the bank
    if (rc == 0 && hipDeviceSynchronize() != hipSuccess)
        step(fail(KLMI_E_DEVICE, "klmi_bank_create: device synchronisation failed"));
    if (rc != 0) {
        delete b;
        return rc;
    }
    ctx->live_objects++;
    *out = b;
    return 0;
}

static void destroy_events(klmi_bank *b)
{
    for (auto e : b->ev_start)
        (void)hipEventDestroy(e);
    for (auto e : b->ev_stop)
        (void)hipEventDestroy(e);
    b->ev_start.clear();
    b->ev_stop.clear();
    if (b->aux_stream) {
        (void)hipStreamSynchronize(b->aux_stream);
        (void)hipStreamDestroy(b->aux_stream);
        (void)hipEventDestroy(b->rv_start);
        for (int i = 0; i < 2; ++i) {
            (void)hipEventDestroy(b->rv_ready[i]);
            (void)hipEventDestroy(b->rv_done[i]);
        }
        b->aux_stream = nullptr;
    }
}

extern "C" int klmi_bank_destroy(klmi_bank *bank)
{
    if (!bank)
        return 0;
    (void)hipSetDevice(bank->ctx->device);
    (void)hipStreamSynchronize(bank->ctx->stream);
    destroy_events(bank);
    bank->ctx->live_objects--;
    delete bank;
    return 0;
}

static int check_range(klmi_bank *b, int first, int count, const void *host, const char *what)
{
    if (!b)
        return fail(KLMI_E_ARG, "%s: null bank", what);
    if (!host)
        return fail(KLMI_E_ARG, "%s: null host pointer", what);
    if (first < 0 || count < 0 || (long)first + count > b->n)
        return fail(KLMI_E_ARG, "%s: range [%d, %d) outside bank of %d voices", what, first,
                    first + count, b->n);
    return 0;
}

static int state_slot(klmi_bank *b, int field, bool *is_u32)
{
    *is_u32 = false;
    if (field >= KLMI_F_FX_STATE(0, 0) && field < KLMI_F_FX_STATE(KLMI_MAX_FX, 0))
        return (field - KLMI_F_FX_STATE(0, 0)) / 2 < b->desc.n_fx
                   ? SL_FX_STATE + (field - KLMI_F_FX_STATE(0, 0))
                   : -1;
    switch (field) {
    case KLMI_F_PHASE: return SL_PHASE;
    case KLMI_F_ENV_VALUE: return SL_ENV_VALUE;
    case KLMI_F_Z0: return SL_Z0;
    case KLMI_F_Z1: return SL_Z1;
    case KLMI_F_FREQUENCY: return SL_FREQ;
    case KLMI_F_AMPLITUDE: return SL_AMP;
    case KLMI_F_LFO_PHASE: return SL_LFO_PHASE;
    case KLMI_F_MOD_PHASE: return SL_MOD_PHASE;
    case KLMI_F_ENV_STAGE:
    case KLMI_F_ENV_ENABLED: *is_u32 = true; return SL_ENV_STATE;
    case KLMI_F_CURRENT_PITCH: *is_u32 = true; return b->desc.poly_voices > 0 ? SL_CUR_PITCH : -1;
    case KLMI_F_ROUND_ROBIN: *is_u32 = true; return b->desc.poly_voices > 0 ? SL_ROUND_ROBIN : -1;
    default: return -1;
    }
}

extern "C" int klmi_bank_set_params(klmi_bank *b, int field, int first, int count,
                                    const double *host)
{
    TRY(check_range(b, first, count, host, "klmi_bank_set_params"));
    TRY(use(b->ctx));
    if (count == 0)
        return 0;
    hipStream_t st = b->ctx->stream;
    if (field >= KLMI_F_FX(0, 0) && field < KLMI_F_FX(KLMI_MAX_FX, 0)) {
        const int stage = (field - KLMI_F_FX(0, 0)) / KLMI_FX_PARAMS;
        const int k = (field - KLMI_F_FX(0, 0)) % KLMI_FX_PARAMS;
        if (stage >= b->desc.n_fx)
            return fail(KLMI_E_ARG, "klmi_bank_set_params: the chain has %d effect stage(s), "
                                    "stage %d addressed", b->desc.n_fx, stage);
        if (b->desc.fx[stage] == KLMI_FX_DELAY && k == 0) {
            for (int i = 0; i < count; ++i) {
                if (host[i] > 2.)  // effects.py:140, 160-164
                    return fail(KLMI_E_ARG, "Delay time %.3f sec to long (max %.3f)!", host[i], 2.);
                if (!((int)(host[i] * KLMI_SAMPLING_RATE) >= 1))
                    return fail(KLMI_E_ARG, "klmi_bank_set_params: delay time %g s is shorter than "
                                            "one sample", host[i]);
            }
        }
        if (b->desc.fx[stage] == KLMI_FX_SUBSAMPLE && k == 0) {
            for (int i = 0; i < count; ++i) {  // effects.py:392-400 VALID_FACTORS
                const int f = (int)host[i];
                if ((double)f != host[i] || f < 2 || f > KLMI_BUFFER_SIZE / 2 || (f & (f - 1)))
                    return fail(KLMI_E_ARG, "klmi_bank_set_params: Subsampler factor %g is not a "
                                            "power of 2 in 2..%d", host[i], KLMI_BUFFER_SIZE / 2);
            }
        }
        if (b->desc.fx[stage] == KLMI_FX_BITCRUSH && k == 0) {
            for (int i = 0; i < count; ++i)    // effects.py:431
                if (!(host[i] >= 0. && host[i] < 16.) || host[i] != std::floor(host[i]))
                    return fail(KLMI_E_ARG, "klmi_bank_set_params: Bitcrusher nBits %g outside 0..15",
                                host[i]);
        }
        if (b->desc.fx[stage] == KLMI_FX_RINGMOD && k == 4) {
            for (int i = 0; i < count; ++i)
                if (!(host[i] >= 0. && host[i] <= (double)KLMI_WAVE_TRIANGLE) ||
                    host[i] != std::floor(host[i]))
                    return fail(KLMI_E_ARG, "klmi_bank_set_params: RingModulator Lfo wave id %g "
                                            "(sine, square, sawtooth, triangle are 0..3)", host[i]);
        }
        if (b->desc.fx[stage] == KLMI_FX_DELAY && k == 0) {
            if (b->fx_ring_cap > 0) {
                // the two line layouts do not convert into each other once the lines are in
                // use: refused HERE, before anything is touched, so the bank stays renderable
                int longest = 0;
                bool any_short = false;
                delay_line_shape(b, first, count, host, &longest, &any_short);
                if (any_short != b->fx_block_major)
                    return fail(KLMI_E_STATE, "Delay times shorter than one buffer (%d samples) have "
                                              "to be set before the bank's first render, for all "
                                              "voices at once, and cannot be left again later "
                                              "(nothing was changed)", KLMI_BUFFER_SIZE);
            }
            TRY(reset_changed_delay_lines(b, first, count, host));
        }
        std::copy(host, host + count,
                  b->fx_raw.begin() + ((size_t)stage * KLMI_FX_PARAMS + k) * b->n + first);
        return upload_fx(b, stage, first, count);
    }
    switch (field) {
    case KLMI_F_FREQUENCY:
    case KLMI_F_AMPLITUDE: {
        const int slot = field == KLMI_F_FREQUENCY ? SL_FREQ : SL_AMP;
        HIP_TRY(hipMemcpyAsync(b->f64(slot) + first, host, sizeof(double) * count,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        return 0;
    }
    case KLMI_F_DUTY:
        HIP_TRY(hipMemcpyAsync(b->duty.p + first, host, sizeof(double) * count,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        return 0;
    case KLMI_F_WAVE: {
        std::vector<int32_t> w(count);
        for (int i = 0; i < count; ++i) {
            if (!(host[i] >= 0. && host[i] <= (double)KLMI_WAVE_EXTERNAL))
                return fail(KLMI_E_ARG, "klmi_bank_set_params: unknown wave id %g", host[i]);
            w[i] = (int32_t)host[i];
        }
        for (int i = 0; i < count; ++i) {
            const uint8_t ext = w[i] == KLMI_WAVE_EXTERNAL;
            b->n_external += (int)ext - (int)b->is_external[first + i];
            b->is_external[first + i] = ext;
        }
        HIP_TRY(hipMemcpyAsync(b->wave.p + first, w.data(), sizeof(int32_t) * count,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        return 0;
    }
    case KLMI_F_ATTACK:
    case KLMI_F_DECAY:
    case KLMI_F_RELEASE:
    case KLMI_F_SUSTAIN:
    case KLMI_F_OVERSHOOT: {
        static const char *names[] = {"attack", "decay", "sustain", "release", "overshoot"};
        const char *name = names[field - KLMI_F_ATTACK];
        for (int i = 0; i < count; ++i) {
            if (field == KLMI_F_SUSTAIN) {
                if (!(host[i] >= 0. && host[i] <= 1.))
                    return fail(KLMI_E_ARG, "sustain not within bounds!");  // _envelope.c:299-302
            } else if (!(host[i] >= 0.)) {
                return fail(KLMI_E_ARG, "%s must be positive!", name);      // _envelope.c:229-232
            }
        }
        std::vector<double> &dst = field == KLMI_F_ATTACK    ? b->attack
                                   : field == KLMI_F_DECAY   ? b->decay
                                   : field == KLMI_F_SUSTAIN ? b->sustain
                                   : field == KLMI_F_RELEASE ? b->release
                                                             : b->overshoot;
        for (int i = 0; i < count; ++i)
            dst[first + i] = field == KLMI_F_OVERSHOOT ? clip(host[i], 1e-9, 1e9) : host[i];
        return upload_env(b, first, count);
    }
    case KLMI_F_LOOP:
    case KLMI_F_RETRIGGER: {
        std::vector<uint8_t> &dst = field == KLMI_F_LOOP ? b->loop : b->retrigger;
        for (int i = 0; i < count; ++i)
            dst[first + i] = host[i] != 0.;
        return upload_env(b, first, count);
    }
    case KLMI_F_CUTOFF: {
        std::vector<int32_t> idx(count);
        for (int i = 0; i < count; ++i)
            idx[i] = grid_index(b->ctx->grid.data(), host[i]);
        HIP_TRY(hipMemcpyAsync(b->u32(SL_COEF_INDEX) + first, idx.data(), sizeof(int32_t) * count,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        return 0;
    }
    case KLMI_F_LFO_FREQUENCY:
    case KLMI_F_LFO_SCALE:
    case KLMI_F_LFO_OFFSET:
    case KLMI_F_MOD_FREQUENCY:
    case KLMI_F_MOD_SCALE:
    case KLMI_F_MOD_OFFSET: {
        const bool lfo = field <= KLMI_F_LFO_OFFSET;
        DevBuf<double> &par = lfo ? b->lfo_par : b->mod_par;
        if (!par.p)
            return fail(KLMI_E_STATE, "klmi_bank_set_params: the bank was created without %s",
                        lfo ? "cutoff_lfo" : "a mod_target");
        const int base = lfo ? KLMI_F_LFO_FREQUENCY : KLMI_F_MOD_FREQUENCY;
        const int row = field == base ? 0 : field - base - 1;  // FREQUENCY, (WAVE), SCALE, OFFSET
        HIP_TRY(hipMemcpyAsync(par.p + (size_t)row * b->n + first, host, sizeof(double) * count,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        return 0;
    }
    case KLMI_F_LFO_WAVE:
    case KLMI_F_MOD_WAVE: {
        DevBuf<int32_t> &dst = field == KLMI_F_LFO_WAVE ? b->lfo_wave : b->mod_wave;
        if (!dst.p)
            return fail(KLMI_E_STATE, "klmi_bank_set_params: the bank was created without %s",
                        field == KLMI_F_LFO_WAVE ? "cutoff_lfo" : "a mod_target");
        std::vector<int32_t> w(count);
        for (int i = 0; i < count; ++i) {
            if (!(host[i] >= 0. && host[i] <= (double)KLMI_WAVE_TRIANGLE))
                return fail(KLMI_E_ARG, "klmi_bank_set_params: modulator wave id %g (sine, square, "
                                        "sawtooth, triangle are 0..3)", host[i]);
            w[i] = (int32_t)host[i];
        }
        HIP_TRY(hipMemcpyAsync(dst.p + first, w.data(), sizeof(int32_t) * count,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        return 0;
    }
    case KLMI_F_GAIN:
    case KLMI_F_PAN: {
        std::vector<double> &dst = field == KLMI_F_GAIN ? b->gain : b->pan;
        std::copy(host, host + count, dst.begin() + first);
        return upload_gains(b, first, count);
    }
    default:
        return fail(KLMI_E_ARG, "klmi_bank_set_params: field %d is not a parameter", field);
    }
}

extern "C" int klmi_bank_state_get(klmi_bank *b, int field, int first, int count, double *host)
{
    TRY(check_range(b, first, count, host, "klmi_bank_state_get"));
    TRY(use(b->ctx));
    bool is_u32;
    const int slot = state_slot(b, field, &is_u32);
    if (slot < 0)
        return fail(KLMI_E_ARG, "klmi_bank_state_get: field %d is not a state field", field);
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    if (!is_u32) {
        HIP_TRY(hipMemcpy(host, b->f64(slot) + first, sizeof(double) * count,
                          hipMemcpyDeviceToHost));
        return 0;
    }
    std::vector<uint32_t> raw(count);
    HIP_TRY(hipMemcpy(raw.data(), b->u32(slot) + first, sizeof(uint32_t) * count,
                      hipMemcpyDeviceToHost));
    for (int i = 0; i < count; ++i) {
        if (field == KLMI_F_CURRENT_PITCH || field == KLMI_F_ROUND_ROBIN)
            host[i] = (double)raw[i];
        else
            host[i] = field == KLMI_F_ENV_STAGE ? (double)(raw[i] & 7u) : (double)((raw[i] >> 8) & 1u);
    }
    return 0;
}

extern "C" int klmi_bank_state_set(klmi_bank *b, int field, int first, int count,
                                   const double *host)
{
    TRY(check_range(b, first, count, host, "klmi_bank_state_set"));
    TRY(use(b->ctx));
    bool is_u32;
    const int slot = state_slot(b, field, &is_u32);
    if (slot < 0)
        return fail(KLMI_E_ARG, "klmi_bank_state_set: field %d is not a state field", field);
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    if (!is_u32) {
        HIP_TRY(hipMemcpy(b->f64(slot) + first, host, sizeof(double) * count,
                          hipMemcpyHostToDevice));
        return 0;
    }
    std::vector<uint32_t> raw(count);
    HIP_TRY(hipMemcpy(raw.data(), b->u32(slot) + first, sizeof(uint32_t) * count,
                      hipMemcpyDeviceToHost));
    if (field == KLMI_F_ROUND_ROBIN) {
        // one counter per instrument, replicated in its voices: set the whole instrument(s)
        const int P = b->desc.poly_voices;
        for (int i = 0; i < count; ++i)
            if (!(host[i] >= 0. && host[i] < (double)P))
                return fail(KLMI_E_ARG, "klmi_bank_state_set: round robin position %g outside "
                                        "0..%d", host[i], P - 1);
        for (int i = 0; i < count; ++i) {
            const int inst = (first + i) / P;
            std::vector<uint32_t> row(P, (uint32_t)host[i]);
            HIP_TRY(hipMemcpy(b->u32(slot) + (size_t)inst * P, row.data(), sizeof(uint32_t) * P,
                              hipMemcpyHostToDevice));
        }
        return 0;
    }
    for (int i = 0; i < count; ++i) {
        if (field == KLMI_F_CURRENT_PITCH) {
            if (!(host[i] >= 0. && host[i] < 128.))
                return fail(KLMI_E_ARG, "Invalid pitch %g!", host[i]);  // messages.py:26-27
            raw[i] = (uint32_t)host[i];
        } else if (field == KLMI_F_ENV_STAGE) {
            if (!(host[i] >= 0. && host[i] <= 4.))
                return fail(KLMI_E_ARG, "klmi_bank_state_set: bad envelope stage %g", host[i]);
            raw[i] = (raw[i] & ~7u) | (uint32_t)host[i];
        } else {
            raw[i] = (raw[i] & ~0x100u) | (host[i] != 0. ? 0x100u : 0u);
        }
    }
    HIP_TRY(hipMemcpy(b->u32(slot) + first, raw.data(), sizeof(uint32_t) * count,
                      hipMemcpyHostToDevice));
    return 0;
}

extern "C" int klmi_bank_events(klmi_bank *b, const klmi_event *events, size_t n)
{
    if (!b)
        return fail(KLMI_E_ARG, "klmi_bank_events: null bank");
    if (n && !events)
        return fail(KLMI_E_ARG, "klmi_bank_events: null events");
    for (size_t i = 0; i < n; ++i) {
        if (events[i].voice >= (uint32_t)b->n)
            return fail(KLMI_E_ARG, "klmi_bank_events: event %zu addresses voice %u of %d", i,
                        events[i].voice, b->n);
        if (events[i].kind > KLMI_EV_SET_FREQUENCY)
            return fail(KLMI_E_ARG, "klmi_bank_events: event %zu has unknown kind %u", i,
                        events[i].kind);
    }
    for (size_t i = 0; i < n; ++i) {
        if (events[i].block > 0xFFFFFFFEu - b->block_clock)
            return fail(KLMI_E_ARG, "klmi_bank_events: event %zu lies too far in the future", i);
    }
    for (size_t i = 0; i < n; ++i) {
        klmi_event e = events[i];
        e.block += b->block_clock;
        b->pending.push_back(e);
    }
    if (n)
        b->events_dirty = true;
    return 0;
}

extern "C" int klmi_bank_notes(klmi_bank *b, const klmi_note *notes, size_t n)
{
    if (!b)
        return fail(KLMI_E_ARG, "klmi_bank_notes: null bank");
    if (n && !notes)
        return fail(KLMI_E_ARG, "klmi_bank_notes: null notes");
    if (b->desc.poly_voices <= 0)
        return fail(KLMI_E_STATE, "klmi_bank_notes: the bank is not polyphonic "
                                  "(desc.poly_voices == 0); use klmi_bank_events");
    const uint32_t n_inst = (uint32_t)(b->n / b->desc.poly_voices);
    for (size_t i = 0; i < n; ++i) {
        if (notes[i].instrument >= n_inst)
            return fail(KLMI_E_ARG, "klmi_bank_notes: note %zu addresses instrument %u of %u", i,
                        notes[i].instrument, n_inst);
        if (notes[i].pitch < 0 || notes[i].pitch >= 128)   // messages.py:26-31
            return fail(KLMI_E_ARG, "Invalid pitch %d!", notes[i].pitch);
        if (!(notes[i].velocity >= 0. && notes[i].velocity <= 1.))
            return fail(KLMI_E_ARG, "Invalid velocity %g!", notes[i].velocity);
        if (!(notes[i].frequency >= 0.))
            return fail(KLMI_E_ARG, "Invalid frequency %g!", notes[i].frequency);
        if (notes[i].block > 0xFFFFFFFEu - b->block_clock)
            return fail(KLMI_E_ARG, "klmi_bank_notes: note %zu lies too far in the future", i);
    }
    for (size_t i = 0; i < n; ++i) {
        klmi_note nt = notes[i];
        nt.block += b->block_clock;
        b->pending_notes.push_back(nt);
    }
    if (n)
        b->notes_dirty = true;
    return 0;
}

extern "C" int klmi_bank_set_mix_divisor(klmi_bank *b, double divisor)
{
    if (!b || !(divisor > 0.))
        return fail(KLMI_E_ARG, "klmi_bank_set_mix_divisor: divisor must be > 0");
    b->mix_divisor = divisor;
    return 0;
}

extern "C" int klmi_bank_set_input(klmi_bank *b, const double *host_x, int n_blocks)
{
    if (!b || !host_x || n_blocks <= 0)
        return fail(KLMI_E_ARG, "klmi_bank_set_input: bad argument");
    TRY(use(b->ctx));
    const size_t span = (size_t)n_blocks * kBlock, n = (size_t)b->n;
    std::vector<double> t(span * n);  // [voice][span] -> [span][voice]
    for (size_t v = 0; v < n; ++v)
        for (size_t s = 0; s < span; ++s)
            t[s * n + v] = host_x[v * span + s];
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    HIP_TRY(b->ext_input.reserve(span * n));
    HIP_TRY(hipMemcpy(b->ext_input.p, t.data(), sizeof(double) * span * n, hipMemcpyHostToDevice));
    b->ext_blocks = n_blocks;
    b->ext_cursor = 0;
    return 0;
}

// Upload the event queue if it changed since the last upload: drop what has
// been consumed (block < clock), stable-sort by (voice, block), build the
// per-voice CSR, reset the per-voice cursors.  Renders with an unchanged queue
// do no host-side event work at all.
static int sync_events(klmi_bank *b, bool may_alloc = true)
{
    if (!b->events_dirty)
        return 0;
    std::vector<klmi_event> &q = b->pending;
    q.erase(std::remove_if(q.begin(), q.end(),
                           [&](const klmi_event &e) { return e.block < b->block_clock; }),
            q.end());
    std::stable_sort(q.begin(), q.end(), [](const klmi_event &a, const klmi_event &c) {
        return a.voice != c.voice ? a.voice < c.voice : a.block < c.block;
    });
    const size_t m = q.size();
    std::vector<uint32_t> offset((size_t)b->n + 1, 0), block(m), kind(m);
    std::vector<int32_t> index(m);
    std::vector<double> vel(m), freq(m);
    for (size_t i = 0; i < m; ++i) {
        offset[q[i].voice + 1]++;
        block[i] = q[i].block;
        kind[i] = q[i].kind;
        vel[i] = q[i].velocity;
        freq[i] = q[i].frequency;
        index[i] = q[i].kind == KLMI_EV_SET_CUTOFF ? grid_index(b->ctx->grid.data(), q[i].frequency)
                                                   : 0;
    }
    for (int v = 0; v < b->n; ++v)
        offset[v + 1] += offset[v];
    hipStream_t st = b->ctx->stream;
    if (!may_alloc && m > b->ev_block.n)
        return fail(KLMI_E_STATE, "%zu queued events do not fit the %zu uploaded so far: the "
                                  "device queue would have to grow; queue them before an "
                                  "allocating render", m, b->ev_block.n);
    HIP_TRY(hipStreamSynchronize(st));  // nothing in flight may still read the old arrays
    HIP_TRY(b->ev_block.reserve(m));
    HIP_TRY(b->ev_kind.reserve(m));
    HIP_TRY(b->ev_index.reserve(m));
    HIP_TRY(b->ev_velocity.reserve(m));
    HIP_TRY(b->ev_frequency.reserve(m));
    HIP_TRY(hipMemcpyAsync(b->ev_offset.p, offset.data(), sizeof(uint32_t) * offset.size(),
                           hipMemcpyHostToDevice, st));
    if (m) {
        HIP_TRY(hipMemcpyAsync(b->ev_block.p, block.data(), sizeof(uint32_t) * m,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->ev_kind.p, kind.data(), sizeof(uint32_t) * m,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->ev_index.p, index.data(), sizeof(int32_t) * m,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->ev_velocity.p, vel.data(), sizeof(double) * m,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->ev_frequency.p, freq.data(), sizeof(double) * m,
                               hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipMemsetAsync(b->u32(SL_EV_CURSOR), 0, sizeof(uint32_t) * b->n, st));
    HIP_TRY(hipStreamSynchronize(st));  // host vectors die here
    b->events_dirty = false;
    return 0;
}

// The same for the notes of a polyphonic bank, CSR by instrument.
static int sync_notes(klmi_bank *b, bool may_alloc = true)
{
    if (!b->notes_dirty)
        return 0;
    std::vector<klmi_note> &q = b->pending_notes;
    q.erase(std::remove_if(q.begin(), q.end(),
                           [&](const klmi_note &e) { return e.block < b->block_clock; }),
            q.end());
    std::stable_sort(q.begin(), q.end(), [](const klmi_note &a, const klmi_note &c) {
        return a.instrument != c.instrument ? a.instrument < c.instrument : a.block < c.block;
    });
    const size_t m = q.size();
    const size_t n_inst = (size_t)(b->n / b->desc.poly_voices);
    std::vector<uint32_t> offset(n_inst + 1, 0), block(m);
    std::vector<int32_t> pitch(m);
    std::vector<double> vel(m), freq(m);
    for (size_t i = 0; i < m; ++i) {
        offset[q[i].instrument + 1]++;
        block[i] = q[i].block;
        pitch[i] = q[i].pitch;
        vel[i] = q[i].velocity;
        freq[i] = q[i].frequency;
    }
    for (size_t k = 0; k < n_inst; ++k)
        offset[k + 1] += offset[k];
    hipStream_t st = b->ctx->stream;
    if (!may_alloc && m > b->nt_block.n)
        return fail(KLMI_E_STATE, "%zu queued notes do not fit the %zu uploaded so far: the "
                                  "device queue would have to grow; queue them before an "
                                  "allocating render", m, b->nt_block.n);
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(b->nt_block.reserve(m));
    HIP_TRY(b->nt_pitch.reserve(m));
    HIP_TRY(b->nt_velocity.reserve(m));
    HIP_TRY(b->nt_frequency.reserve(m));
    HIP_TRY(hipMemcpyAsync(b->nt_offset.p, offset.data(), sizeof(uint32_t) * offset.size(),
                           hipMemcpyHostToDevice, st));
    if (m) {
        HIP_TRY(hipMemcpyAsync(b->nt_block.p, block.data(), sizeof(uint32_t) * m,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->nt_pitch.p, pitch.data(), sizeof(int32_t) * m,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->nt_velocity.p, vel.data(), sizeof(double) * m,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->nt_frequency.p, freq.data(), sizeof(double) * m,
                               hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipMemsetAsync(b->u32(SL_NT_CURSOR), 0, sizeof(uint32_t) * b->n, st));
    HIP_TRY(hipStreamSynchronize(st));
    b->notes_dirty = false;
    return 0;
}

// The fast kernel for plain [Filter] chains, the general one for everything else
// (voice_tu.hip / mod_tu.hip).
template <int MODE>
static hipError_t dispatch_voice(bool env, bool filt, hipStream_t st, const BankView &view,
                                 int first, int count, int block_base, uint32_t clock_base,
                                 int n_blocks, int span_total, float *partial, void *out)
{
    const bool plain_chain = view.n_fx == 0 || (view.n_fx == 1 && view.fx_kind[0] == FX_FILTER);
    // synthesizer voices (oscillator x envelope under a mixer) have a gated form of the fast kernel
    const bool fast_gated = env && !filt && view.n_fx == 0 &&
                            (MODE == OUT_MIX_MONO || MODE == OUT_MIX_STEREO);
    // banks that leave most SIMDs idle run oscillator / chain / output as pipeline stages
    int n_cus = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        static int cached[64] = {};
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            if (!cached[dev] && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                cached[dev] = prop.multiProcessorCount;
            if (cached[dev] > 0)
                n_cus = cached[dev];
        }
    }
    const int pipe = pipe_voices_log2(view, filt, count, MODE, n_cus);
    if (pipe >= 0)
        return launch_pipe_kernel(env, filt, MODE, st, view, first, count, block_base, clock_base,
                                  n_blocks, span_total, pipe, n_cus, partial, out);
    // audio-rate modulation inputs in a plain [Filter] chain: the fast kernel's forms
    const char *fm_knob = getenv("KLMI_FM_KERNEL");   // tuning knob: 0 = the general kernel
    if (mod_fast_form(view, plain_chain, env, filt, MODE) && !(fm_knob && fm_knob[0] == '0'))
        return launch_fm_kernel(env, filt, MODE, st, view, first, count, block_base, clock_base,
                                n_blocks, span_total, partial, out);
    // a short Tremolo / Delay / Gain program behind the [Filter]: compiled in
    const char *fx_knob = getenv("KLMI_FX_KERNEL");   // tuning knob: 0 = the general kernel
    const uint32_t program = fx_knob && fx_knob[0] == '0' ? 0 : fx_program_of(view, filt, MODE);
    if (program != 0)
        return launch_fx_kernel(program, env, filt, MODE, st, view, first, count, block_base,
                                clock_base, n_blocks, span_total, partial, out);
    if (view.mod_target != MT_NONE || !plain_chain || (view.gated && !fast_gated))
        return launch_mod_kernel(env, filt, MODE, st, view, first, count, block_base, clock_base,
                                 n_blocks, span_total, partial, out);
    // one workgroup per compute unit at most: the build without the two-waves register cap
    const char *waves_knob = getenv("KLMI_VOICE_WAVES");   // tuning knob: 2 = always the capped build
    if (!view.gated && (count + kGroup - 1) / kGroup <= n_cus && !(waves_knob && waves_knob[0] == '2'))
        return launch_voice_kernel_one_wave(env, filt, MODE, st, view, first, count, block_base,
                                            clock_base, n_blocks, span_total, partial, out);
    return launch_voice_kernel(env, filt, view.gated != 0, MODE, st, view, first, count, block_base,
                               clock_base, n_blocks, span_total, partial, out);
}

static int check_external(klmi_bank *b, int n_blocks)
{
    // W_EXTERNAL voices read ext_input: it has to cover the span.  The bank counts such
    // voices on the host (klmi_bank_set_params, KLMI_F_WAVE), so ordinary banks pay nothing.
    if (b->n_external == 0)
        return 0;
    if (b->ext_input.p == nullptr || b->ext_blocks - b->ext_cursor < n_blocks) {
        int v = 0;
        while (v < b->n && !b->is_external[v])
            ++v;
        return fail(KLMI_E_STATE,
                    "voice %d uses KLMI_WAVE_EXTERNAL but only %d blocks of external input are "
                    "left, %d needed", v, b->ext_input.p ? b->ext_blocks - b->ext_cursor : 0,
                    n_blocks);
    }
    return 0;
}

static int timing_begin(klmi_bank *b, int idx)
{
    while ((int)b->ev_start.size() <= idx) {
        hipEvent_t a, c;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&c));
        b->ev_start.push_back(a);
        b->ev_stop.push_back(c);
    }
    HIP_TRY(hipEventRecord(b->ev_start[idx], b->ctx->stream));
    return 0;
}

static int render_reverb(klmi_bank *b, int n_blocks, float *dev_mix);

// `fixed`: a caller-supplied workspace of `fixed_bytes` (the static scratch array) instead of
// the bank's own growing buffers; nothing is allocated then, and the launches are not timed.
static int render_mix(klmi_bank *b, int n_blocks, float *dev_mix, unsigned char *fixed = nullptr,
                      size_t fixed_bytes = 0)
{
    const bool may_alloc = fixed == nullptr;
    if (n_blocks <= 0)
        return fail(KLMI_E_ARG, "klmi_bank_render: n_blocks must be positive, got %d", n_blocks);
    TRY(check_external(b, n_blocks));
    TRY(sync_events(b, may_alloc));
    TRY(sync_notes(b, may_alloc));
    TRY(ensure_fx_ring(b, may_alloc));
    if (b->desc.has_reverb)
        return render_reverb(b, n_blocks, dev_mix);
    hipStream_t st = b->ctx->stream;
    const int planes = b->desc.stereo ? 2 : 1;
    BankView view = make_view(b);
    // one partial bus per wave; blocks per launch keep them within 256 MiB
    const int rows = voice_partial_rows(view, b->desc.has_filter != 0, b->n,
                                        b->desc.stereo ? OUT_MIX_STEREO : OUT_MIX_MONO,
                                        b->ctx->n_cus);
    const int slices = bus_slices(rows);
    const size_t per_block = (size_t)rows * planes * kBlock * sizeof(float);
    const size_t stage_per_block = (size_t)slices * planes * kBlock * sizeof(double);
    int bpl;
    float *partial;
    double *stage;
    if (may_alloc) {
        size_t cap_mib = 256;
        if (const char *e = getenv("KLMI_PARTIAL_MIB")) {   // tuning knob: partial-bus bytes per launch
            const long v = atol(e);
            if (v >= 1 && v <= 8192)
                cap_mib = (size_t)v;
        }
        bpl = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_blocks, (cap_mib << 20) / per_block));
        HIP_TRY(b->partial.reserve((size_t)rows * planes * kBlock * bpl));
        HIP_TRY(b->stage.reserve((size_t)slices * planes * kBlock * bpl));
        partial = b->partial.p;
        stage = b->stage.p;
    } else {
        // partial buses first, slice sums behind them: both are whole KiB per block, so the
        // second part stays 256-byte aligned
        bpl = (int)std::min<size_t>((size_t)n_blocks, fixed_bytes / (per_block + stage_per_block));
        if (bpl < 1)
            return fail(KLMI_E_ARG, "klmi_bank_render_device_noalloc: one buffer of this bank "
                                    "(%d voices, %d channel(s)) needs %zu bytes of scratch, the "
                                    "static array holds %zu", b->n, planes,
                        per_block + stage_per_block, fixed_bytes);
        partial = reinterpret_cast<float *>(fixed);
        stage = reinterpret_cast<double *>(fixed + per_block * (size_t)bpl);
    }
    b->timed_launches = 0;
    for (int base = 0; base < n_blocks; base += bpl) {
        const int nb = std::min(bpl, n_blocks - base);
        if (may_alloc)
            TRY(timing_begin(b, b->timed_launches));
        hipError_t e;
        if (b->desc.stereo)
            e = dispatch_voice<OUT_MIX_STEREO>(b->desc.has_env, b->desc.has_filter, st, view, 0,
                                               b->n, base, b->block_clock + (uint32_t)base, nb, 0,
                                               partial, nullptr);
        else
            e = dispatch_voice<OUT_MIX_MONO>(b->desc.has_env, b->desc.has_filter, st, view, 0,
                                             b->n, base, b->block_clock + (uint32_t)base, nb, 0,
                                             partial, nullptr);
        if (e != hipSuccess)
            return fail(KLMI_E_DEVICE, "voice kernel launch failed: %s", hipGetErrorString(e));
        if (may_alloc) {
            HIP_TRY(hipEventRecord(b->ev_stop[b->timed_launches], st));
            b->timed_launches++;
        }
        const size_t bus_len = (size_t)nb * planes * kBlock;
        hipLaunchKernelGGL(bus_slice_kernel, dim3((unsigned)((bus_len + 1023) / 1024), slices),
                           dim3(256), 0, st, partial, rows, bus_len, stage);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(bus_finish_kernel, dim3((unsigned)((bus_len + 255) / 256)), dim3(256), 0,
                           st, stage, slices, bus_len, b->mix_divisor,
                           dev_mix + (size_t)base * planes * kBlock);
        HIP_TRY(hipGetLastError());
    }
    b->block_clock += (uint32_t)n_blocks;
    if (b->ext_input.p)
        b->ext_cursor = std::min(b->ext_blocks, b->ext_cursor + n_blocks);
    return 0;
}

extern "C" int klmi_bank_render_device(klmi_bank *b, int n_blocks, float *dev_mix)
{
    if (!b || !dev_mix)
        return fail(KLMI_E_ARG, "klmi_bank_render_device: null argument");
    TRY(use(b->ctx));
    return render_mix(b, n_blocks, dev_mix);
}

// The same render with the static scratch array as its only workspace: no hipMalloc, hipFree
// or other allocating call on this path.  The kernels overwrite every partial row and slice
// sum they later read, so nothing of the previous call's contents has to be cleared.
extern "C" int klmi_bank_render_device_noalloc(klmi_bank *b, int n_blocks, float *dev_mix)
{
    if (!b || !dev_mix)
        return fail(KLMI_E_ARG, "klmi_bank_render_device_noalloc: null argument");
    TRY(use(b->ctx));
    if (b->desc.has_reverb)
        return fail(KLMI_E_STATE, "klmi_bank_render_device_noalloc: reverb banks stage their dry "
                                  "bus in buffers that grow with the call; use "
                                  "klmi_bank_render_device");
    return render_mix(b, n_blocks, dev_mix, b->ctx->scratch_ptr, kScratchBytes);
}

extern "C" int klmi_bank_render(klmi_bank *b, int n_blocks, float *host_mix)
{
    if (!b || !host_mix)
        return fail(KLMI_E_ARG, "klmi_bank_render: null argument");
    TRY(use(b->ctx));
    if (n_blocks <= 0)
        return fail(KLMI_E_ARG, "klmi_bank_render: n_blocks must be positive, got %d", n_blocks);
    const size_t len = (size_t)n_blocks * (b->desc.stereo ? 2 : 1) * kBlock;
    HIP_TRY(b->dev_mix.reserve(len));
    b->last_mix_blocks = 0;
    TRY(render_mix(b, n_blocks, b->dev_mix.p));
    b->last_mix_blocks = n_blocks;
    HIP_TRY(hipMemcpyAsync(host_mix, b->dev_mix.p, sizeof(float) * len, hipMemcpyDeviceToHost,
                           b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return 0;
}

// Bank -> bank on the device: what `source` mixed in its last klmi_bank_render is this
// bank's external input, without the way over the host.
extern "C" int klmi_bank_set_input_from_bank(klmi_bank *b, klmi_bank *source, int n_blocks)
{
    if (!b || !source || b == source || n_blocks <= 0)
        return fail(KLMI_E_ARG, "klmi_bank_set_input_from_bank: bad argument");
    if (b->ctx != source->ctx)
        return fail(KLMI_E_ARG, "klmi_bank_set_input_from_bank: the banks live on different contexts");
    if (source->last_mix_blocks < n_blocks)
        return fail(KLMI_E_STATE, "klmi_bank_set_input_from_bank: the source's last klmi_bank_render "
                                  "left %d buffer(s), %d asked for", source->last_mix_blocks, n_blocks);
    TRY(use(b->ctx));
    const size_t span = (size_t)n_blocks * kBlock, n = (size_t)b->n;
    hipStream_t st = b->ctx->stream;
    if (b->ext_input.n < span * n) {
        HIP_TRY(hipStreamSynchronize(st));   // nothing in flight may still read the old input
        HIP_TRY(b->ext_input.reserve(span * n));
    }
    hipLaunchKernelGGL(bus_to_input_kernel, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, st,
                       source->dev_mix.p, source->desc.stereo ? 2 : 1, b->n, span, b->ext_input.p);
    HIP_TRY(hipGetLastError());
    b->ext_blocks = n_blocks;
    b->ext_cursor = 0;
    return 0;
}

// Preview: snapshot state, render, restore; the event queue is not consumed.
template <int MODE>
static int render_preview(klmi_bank *b, int first, int count, int n_blocks, void *dev_out)
{
    if (n_blocks <= 0 || count <= 0)
        return fail(KLMI_E_ARG, "klmi_bank_render_voices: n_blocks and count must be positive");
    if (first < 0 || (long)first + count > b->n)
        return fail(KLMI_E_ARG, "klmi_bank_render_voices: range [%d, %d) outside bank of %d voices",
                    first, first + count, b->n);
    if (b->desc.has_reverb)
        return fail(KLMI_E_STATE, "klmi_bank_render_voices: previews of reverb banks are not "
                                  "available; use klmi_bank_render");
    TRY(check_external(b, n_blocks));
    TRY(sync_events(b));
    TRY(sync_notes(b));
    TRY(ensure_fx_ring(b));
    hipStream_t st = b->ctx->stream;
    const size_t n = (size_t)b->n;
    const size_t ring_len = b->fx_ring_len;
    if (ring_len) {  // the delay lines are state too
        if (ring_len * sizeof(double) > ((size_t)1 << 30))
            return fail(KLMI_E_STATE, "klmi_bank_render_voices: previews of banks with more than "
                                      "1 GiB of delay lines are not available");
        HIP_TRY(b->fx_ring_backup.reserve(ring_len));
        HIP_TRY(hipMemcpyAsync(b->fx_ring_backup.p, b->fx_ring.p, sizeof(double) * ring_len,
                               hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipMemcpyAsync(b->backup_f64.p, b->state_f64.p, sizeof(double) * b->f64_slots * n,
                           hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->backup_u32.p, b->state_u32.p, sizeof(uint32_t) * SL_U32_COUNT * n,
                           hipMemcpyDeviceToDevice, st));
    BankView view = make_view(b);
    b->timed_launches = 0;
    TRY(timing_begin(b, 0));
    hipError_t e = dispatch_voice<MODE>(b->desc.has_env, b->desc.has_filter, st, view, first, count,
                                        0, b->block_clock, n_blocks, n_blocks * kBlock, nullptr,
                                        dev_out);
    if (e != hipSuccess)
        return fail(KLMI_E_DEVICE, "voice kernel launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipEventRecord(b->ev_stop[0], st));
    b->timed_launches = 1;
    HIP_TRY(hipMemcpyAsync(b->state_f64.p, b->backup_f64.p, sizeof(double) * b->f64_slots * n,
                           hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->state_u32.p, b->backup_u32.p, sizeof(uint32_t) * SL_U32_COUNT * n,
                           hipMemcpyDeviceToDevice, st));
    if (ring_len)
        HIP_TRY(hipMemcpyAsync(b->fx_ring.p, b->fx_ring_backup.p, sizeof(double) * ring_len,
                               hipMemcpyDeviceToDevice, st));
    return 0;
}

extern "C" int klmi_bank_render_voices_device(klmi_bank *b, int first, int count, int n_blocks,
                                              float *dev_out)
{
    if (!b || !dev_out)
        return fail(KLMI_E_ARG, "klmi_bank_render_voices_device: null argument");
    TRY(use(b->ctx));
    return render_preview<OUT_VOICES_F32>(b, first, count, n_blocks, dev_out);
}

extern "C" int klmi_bank_render_voices(klmi_bank *b, int first, int count, int n_blocks,
                                       float *host_out)
{
    if (!b || !host_out)
        return fail(KLMI_E_ARG, "klmi_bank_render_voices: null argument");
    TRY(use(b->ctx));
    if (n_blocks <= 0 || count <= 0)
        return fail(KLMI_E_ARG, "klmi_bank_render_voices: n_blocks and count must be positive");
    const size_t len = (size_t)count * n_blocks * kBlock;
    HIP_TRY(b->dev_voices_f32.reserve(len));
    TRY(render_preview<OUT_VOICES_F32>(b, first, count, n_blocks, b->dev_voices_f32.p));
    HIP_TRY(hipMemcpyAsync(host_out, b->dev_voices_f32.p, sizeof(float) * len,
                           hipMemcpyDeviceToHost, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return 0;
}

extern "C" int klmi_bank_render_voices_f64(klmi_bank *b, int first, int count, int n_blocks,
                                           double *host_out)
{
    if (!b || !host_out)
        return fail(KLMI_E_ARG, "klmi_bank_render_voices_f64: null argument");
    TRY(use(b->ctx));
    if (n_blocks <= 0 || count <= 0)
        return fail(KLMI_E_ARG, "klmi_bank_render_voices_f64: n_blocks and count must be positive");
    const size_t len = (size_t)count * n_blocks * kBlock;
    HIP_TRY(b->dev_voices_f64.reserve(len));
    TRY(render_preview<OUT_VOICES_F64>(b, first, count, n_blocks, b->dev_voices_f64.p));
    HIP_TRY(hipMemcpyAsync(host_out, b->dev_voices_f64.p, sizeof(double) * len,
                           hipMemcpyDeviceToHost, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return 0;
}

// ---------------------------------------------------------------------------------
// offline bounce: klanggeber.py:255-263 (nCycles buffers from play time 0), :81-97 pack,
// :205-209, 243-249 fade-out, wavfile.py:54-63 peak-normalised int16
// ---------------------------------------------------------------------------------
extern "C" int klmi_bounce_cycles(double duration)
{
    // int(duration * SAMPLING_RATE // BUFFER_SIZE); the division by 256 is exact
    const double q = std::floor(duration * (double)KLMI_SAMPLING_RATE / (double)KLMI_BUFFER_SIZE);
    if (!(q >= 0.) || !std::isfinite(q))
        return 0;   // nothing to render; klmi_bank_bounce rejects such durations
    return q > 2147483647. ? 2147483647 : (int)q;
}

// The output side alone: `bus` [n_blocks][ch][256] float32 (mixed, on the device) -> frames /
// pcm / peak.  dev_frames may be the bank's own staging buffer.
static int bounce_finish(klmi_bank *b, const float *bus, int n_blocks, double duration,
                         double fadeout, float *dev_frames, int16_t *dev_pcm, float *dev_peak)
{
    hipStream_t st = b->ctx->stream;
    const int ch = b->desc.stereo ? 2 : 1;
    const size_t len = (size_t)n_blocks * ch * kBlock;
    // playTime += INTERVAL, a running float64 sum (klanggeber.py:199, 248)
    std::vector<double> play_time(n_blocks);
    double t = 0.;
    for (int k = 0; k < n_blocks; ++k) {
        play_time[k] = t;
        t += kInterval;
    }
    HIP_TRY(b->bounce_time.reserve(n_blocks));
    HIP_TRY(b->bounce_peak.reserve((size_t)n_blocks));
    HIP_TRY(hipMemcpyAsync(b->bounce_time.p, play_time.data(), sizeof(double) * n_blocks,
                           hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // play_time dies here
    const double fade_start = duration - fadeout;
    if (ch == 2)
        hipLaunchKernelGGL(bounce_pack_kernel<2>, dim3(n_blocks), dim3(kBlock), 0, st, bus,
                           b->bounce_time.p, fade_start, duration, dev_frames, b->bounce_peak.p);
    else
        hipLaunchKernelGGL(bounce_pack_kernel<1>, dim3(n_blocks), dim3(kBlock), 0, st, bus,
                           b->bounce_time.p, fade_start, duration, dev_frames, b->bounce_peak.p);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(bounce_peak_kernel, dim3(1), dim3(256), 0, st, b->bounce_peak.p, n_blocks,
                       dev_peak);
    HIP_TRY(hipGetLastError());
    if (dev_pcm) {
        const size_t total4 = len / 4;
        hipLaunchKernelGGL(bounce_pcm_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st,
                           dev_frames, total4, dev_peak, dev_pcm);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// render + output side; dev_frames may be the bank's own staging buffer
static int bounce_device(klmi_bank *b, int n_blocks, double duration, double fadeout,
                         float *dev_frames, int16_t *dev_pcm, float *dev_peak)
{
    const int ch = b->desc.stereo ? 2 : 1;
    const size_t len = (size_t)n_blocks * ch * kBlock;
    HIP_TRY(b->dev_mix.reserve(len));
    b->last_mix_blocks = 0;
    TRY(render_mix(b, n_blocks, b->dev_mix.p));
    b->last_mix_blocks = n_blocks;
    return bounce_finish(b, b->dev_mix.p, n_blocks, duration, fadeout, dev_frames, dev_pcm, dev_peak);
}

static int bounce_args(klmi_bank *b, double duration, double fadeout, int *n_blocks)
{
    if (!b)
        return fail(KLMI_E_ARG, "klmi_bank_bounce: null bank");
    if (!(duration > 0.) || !std::isfinite(duration))
        return fail(KLMI_E_ARG, "klmi_bank_bounce: Offline mode and duration set to INF (or not "
                                "positive). You will never get anything.");
    if (!(fadeout >= 0.))
        return fail(KLMI_E_ARG, "klmi_bank_bounce: fadeout must be >= 0, got %g", fadeout);
    *n_blocks = klmi_bounce_cycles(duration);
    if (*n_blocks <= 0)
        return fail(KLMI_E_ARG, "klmi_bank_bounce: duration %g s is shorter than one buffer",
                    duration);
    if (b->block_clock > 0xFFFFFFFEu - (uint32_t)*n_blocks)
        return fail(KLMI_E_ARG, "klmi_bank_bounce: block clock would overflow");
    return 0;
}

extern "C" int klmi_bank_bounce_device(klmi_bank *b, double duration, double fadeout,
                                       float *dev_frames, int16_t *dev_pcm, float *dev_peak)
{
    int n_blocks = 0;
    TRY(bounce_args(b, duration, fadeout, &n_blocks));
    if (!dev_frames || !dev_peak)
        return fail(KLMI_E_ARG, "klmi_bank_bounce_device: null frames / peak pointer");
    TRY(use(b->ctx));
    return bounce_device(b, n_blocks, duration, fadeout, dev_frames, dev_pcm, dev_peak);
}

// frames / pcm / peak to the host after bounce_device / bounce_finish wrote them
static int bounce_download(klmi_bank *b, size_t len, const float *dev_peak, float *host_frames,
                           int16_t *host_pcm, float *host_peak)
{
    hipStream_t st = b->ctx->stream;
    float peak = 0.f;
    HIP_TRY(hipMemcpyAsync(&peak, dev_peak, sizeof(float), hipMemcpyDeviceToHost, st));
    if (host_frames)
        HIP_TRY(hipMemcpyAsync(host_frames, b->bounce_frames.p, sizeof(float) * len,
                               hipMemcpyDeviceToHost, st));
    if (host_pcm)
        HIP_TRY(hipMemcpyAsync(host_pcm, b->bounce_pcm.p, sizeof(int16_t) * len,
                               hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (host_peak)
        *host_peak = peak;
    if (host_pcm && peak == 0.f)
        return fail(KLMI_E_ARG, "Zero amplitude! Can not normalize!");  // math.py:17-19
    return 0;
}

extern "C" int klmi_bank_bounce(klmi_bank *b, double duration, double fadeout, float *host_frames,
                                int16_t *host_pcm, float *host_peak)
{
    int n_blocks = 0;
    TRY(bounce_args(b, duration, fadeout, &n_blocks));
    TRY(use(b->ctx));
    const size_t len = (size_t)n_blocks * (b->desc.stereo ? 2 : 1) * kBlock;
    HIP_TRY(b->bounce_frames.reserve(len + 4));  // + the peak, behind the frames
    float *dev_peak = b->bounce_frames.p + len;
    if (host_pcm)
        HIP_TRY(b->bounce_pcm.reserve(len));
    TRY(bounce_device(b, n_blocks, duration, fadeout, b->bounce_frames.p,
                      host_pcm ? b->bounce_pcm.p : nullptr, dev_peak));
    return bounce_download(b, len, dev_peak, host_frames, host_pcm, host_peak);
}

// The output side of a bounce for a bus that was rendered elsewhere: klmi_bank_render_device
// on every GPU of a sharded patch, the partial buses summed over the ranks (one RCCL reduce)
// and divided by the global input count -- the root then packs / fades / normalises HERE.
// Nothing is rendered and the bank's state does not move.
extern "C" int klmi_bank_bounce_bus(klmi_bank *b, const float *dev_bus, double duration,
                                    double fadeout, float *host_frames, int16_t *host_pcm,
                                    float *host_peak)
{
    int n_blocks = 0;
    TRY(bounce_args(b, duration, fadeout, &n_blocks));
    if (!dev_bus)
        return fail(KLMI_E_ARG, "klmi_bank_bounce_bus: null bus");
    TRY(use(b->ctx));
    const size_t len = (size_t)n_blocks * (b->desc.stereo ? 2 : 1) * kBlock;
    HIP_TRY(b->bounce_frames.reserve(len + 4));
    float *dev_peak = b->bounce_frames.p + len;
    if (host_pcm)
        HIP_TRY(b->bounce_pcm.reserve(len));
    TRY(bounce_finish(b, dev_bus, n_blocks, duration, fadeout, b->bounce_frames.p,
                      host_pcm ? b->bounce_pcm.p : nullptr, dev_peak));
    return bounce_download(b, len, dev_peak, host_frames, host_pcm, host_peak);
}

extern "C" int klmi_bank_last_kernel_ms(klmi_bank *b, float *voice_ms, int *launches)
{
    if (!b)
        return fail(KLMI_E_ARG, "klmi_bank_last_kernel_ms: null bank");
    TRY(use(b->ctx));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    float total = 0.f;
    for (int i = 0; i < b->timed_launches; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, b->ev_start[i], b->ev_stop[i]));
        total += ms;
    }
    if (voice_ms)
        *voice_ms = total;
    if (launches)
        *launches = b->timed_launches;
    return 0;
}

// ---------------------------------------------------------------------------------
// reverb banks
// ---------------------------------------------------------------------------------
static int bank_init_reverb(klmi_bank *b)
{
    const klmi_chain_desc &d = b->desc;
    if (d.reverb_n_echos <= 0 || d.reverb_n_echos > kMaxEchos)
        return fail(KLMI_E_ARG, "Reverb: nEchos must lie in 1..%d, got %d", kMaxEchos,
                    d.reverb_n_echos);
    b->taps.resize(d.reverb_n_echos);
    b->alphas.resize(d.reverb_n_echos);
    TRY(klmi_reverb_design(d.reverb_decay, d.reverb_pre_delay, d.reverb_n_echos, b->taps.data(),
                           b->alphas.data()));
    ReverbState &r = b->reverb;
    r.n_echos = d.reverb_n_echos;
    r.n = b->n;
    r.dry_wet = clip(d.reverb_dry_wet, 0., 1.);  // effects.py:586
    size_t total = 0;
    int min_tap = b->taps[0];
    for (int k = 0; k < r.n_echos; ++k) {
        r.length[k] = b->taps[k];
        r.alpha[k] = b->alphas[k];
        r.offset[k] = (int)total + kReverbMirror;            // spare slots, then the ring,
        total += (size_t)b->taps[k] + 2 * kReverbMirror;     // then its mirror slots
        min_tap = std::min(min_tap, b->taps[k]);
    }
    r.min_length = min_tap;
    r.stride = (long long)total;   // voice-major: one run of all its combs per voice
    b->ring_pos.assign(r.n_echos, 0);
    const bool f32 = d.reverb_rings_f64 == 0;
    // + spare slots behind the last voice's last mirror (second stores may reach them)
    const size_t slots = total * (size_t)b->n + 2 * kReverbMirror;
    hipError_t e = f32 ? b->rings32.alloc(slots) : b->rings.alloc(slots);
    if (e != hipSuccess)
        return fail(KLMI_E_MEMORY, "Reverb: could not allocate %zu ring samples: %s",
                    total * (size_t)b->n, hipGetErrorString(e));
    r.rings = f32 ? (void *)b->rings32.p : (void *)b->rings.p;
    // _filters.c:57 zero-initialised; ordered on the context's stream like the kernels
    HIP_TRY(hipMemsetAsync(r.rings, 0, (f32 ? sizeof(float) : sizeof(double)) * slots,
                           b->ctx->stream));
    return 0;
}

static int reverb_streams(klmi_bank *b)
{
    if (b->aux_stream)
        return 0;
    HIP_TRY(hipStreamCreateWithFlags(&b->aux_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&b->rv_start, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipEventCreateWithFlags(&b->rv_ready[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&b->rv_done[i], hipEventDisableTiming));
    }
    return 0;
}

template <typename RingT, bool STEREO>
static void launch_reverb_echos(int groups, hipStream_t st, const ReverbState &r, const double *x,
                                size_t x_stride, int total, int pass, int vpw, const double *gl,
                                const double *gr, float *partial, size_t row_stride)
{
    // The fast kernel is built for the Reverb's default of 10 combs (effects.py:575; its
    // registers are sized for them) and passes of 256 samples; a comb shorter than that or
    // another comb count takes the sample-per-lane kernel.
    if (pass == kReverbPass && r.n_echos == 10)
        hipLaunchKernelGGL((reverb_kernel<RingT, STEREO, 10>), dim3(groups), dim3(kReverbThreads), 0,
                           st, r, x, x_stride, total, vpw, gl, gr, partial, row_stride);
    else
        hipLaunchKernelGGL((reverb_short_kernel<RingT, STEREO>), dim3(groups), dim3(kReverbThreads),
                           0, st, r, x, x_stride, total, std::min(pass, 64), vpw, gl, gr, partial,
                           row_stride);
}

static void launch_reverb(bool f32, bool stereo, int groups, hipStream_t st, const ReverbState &r,
                          const double *x, size_t x_stride, int total, int pass, int vpw,
                          const double *gl, const double *gr, float *partial, size_t row_stride)
{
    if (f32) {
        if (stereo)
            launch_reverb_echos<float, true>(groups, st, r, x, x_stride, total, pass, vpw, gl, gr, partial, row_stride);
        else
            launch_reverb_echos<float, false>(groups, st, r, x, x_stride, total, pass, vpw, gl, gr, partial, row_stride);
    } else {
        if (stereo)
            launch_reverb_echos<double, true>(groups, st, r, x, x_stride, total, pass, vpw, gl, gr, partial, row_stride);
        else
            launch_reverb_echos<double, false>(groups, st, r, x, x_stride, total, pass, vpw, gl, gr, partial, row_stride);
    }
}

// Dry voices and reverb are pipelined over chunks of a few blocks: the voice kernel of
// chunk j+1 (few waves, float64-VALU bound) runs on a side stream while the reverb kernel
// of chunk j (HBM bound) runs on the context's stream; two dry buses alternate.  Order is
// carried by events only: ready[i] (dry bus i written), done[i] (reverb finished reading
// dry bus i).  One reverb launch covers the whole chunk: its waves own their voices and
// walk the chunk's time passes themselves (reverb_kernels.h).
// Voice -> Reverb -> Mixer in one kernel (reverb_voice_kernels.h): chains without filter,
// effect stages or modulation, the default ten combs, none shorter than a buffer.
static bool reverb_fused_form(const klmi_bank *b)
{
    const char *knob = getenv("KLMI_REVERB_FUSED");   // 0: the two-kernel path (tests compare)
    const bool enabled = !(knob && atoi(knob) == 0);
    const klmi_chain_desc &d = b->desc;
    return enabled && !d.has_filter && d.n_fx == 0 && d.mod_target == 0 && !d.cutoff_lfo &&
           !d.gated && d.poly_voices == 0 && b->reverb.n_echos == 10 &&
           b->reverb.min_length >= kReverbPass;
}

static int render_reverb_fused(klmi_bank *b, int n_blocks, float *dev_mix)
{
    hipStream_t st = b->ctx->stream;
    ReverbState &r = b->reverb;
    const bool stereo = b->desc.stereo != 0, f32 = b->desc.reverb_rings_f64 == 0;
    const int planes = stereo ? 2 : 1;
    // voices per wave: the kernel runs two workgroups (8 waves) per compute unit; one voice per
    // wave while that leaves waves free, two beyond (measured at 4,096 / 16,384 channels: 1
    // voice 10.0 / 9.9 ms, 2 voices 7.4 / 7.7 ms, 4 voices 7.6 / 10.0 ms per 512 / 128 buffers)
    const char *vpw_knob = getenv("KLMI_REVERB_VPW");  // tuning knob (the tests run all three)
    const int vpw_forced = vpw_knob ? atoi(vpw_knob) : 0;
    // float64 rings (twice the bytes per voice in flight): one voice per wave throughout
    // (4,096 / 16,384 channels: 15.4 / 16.4 ms against 19.3 / 19.2 with two)
    int vpw = (!f32 || b->n <= 8 * b->ctx->n_cus) ? 1 : 2;
    if (vpw_forced == 1 || vpw_forced == 2 || vpw_forced == 4)
        vpw = vpw_forced;
    const int rows = reverb_voice_groups(b->n, vpw);   // one partial row per workgroup
    const int slices = bus_slices(rows);
    // Buffers per launch.  Rings that fit the 256 MiB Infinity Cache (4,096 channels of float32
    // rings: 0.23 GB) are served from it, and the partial rows of a long launch stream through
    // the same cache: with one launch for a 512-buffer call (256 MiB of rows) the kernel runs
    // at 5.2-5.9 TB/s depending on where the two allocations lie relative to each other (a 16
    // MiB period, measured by shifting the rings inside a padded allocation); with 32 buffers
    // per launch (16 MiB of rows, reused by every launch, stored and read non-temporally) it
    // runs at 5.75 in every placement tried.  Rings beyond the cache gain nothing from short
    // launches and lose the launch boundaries (float64 rings at 4,096 channels: 5.2 against
    // 5.7 TB/s): as many buffers as keep the rows within 256 MiB.
    static const int chunk_forced = [] {
        const char *e = getenv("KLMI_REVERB_CHUNK");   // tuning knob
        const int v = e ? atoi(e) : 0;
        return v >= 1 && v <= 4096 ? v : 0;
    }();
    const size_t per_block = (size_t)rows * planes * kBlock * sizeof(float);
    const int fit = (int)std::max<size_t>(1, ((size_t)256 << 20) / per_block);
    const size_t ring_bytes = (size_t)r.stride * (size_t)b->n * (f32 ? sizeof(float) : sizeof(double));
    const bool rings_cached = ring_bytes <= ((size_t)256 << 20);
    const int chunk_blocks = std::min(n_blocks, chunk_forced ? chunk_forced
                                                : rings_cached ? std::min(fit, 32) : fit);
    const size_t row_stride = (size_t)chunk_blocks * kBlock * planes;
    HIP_TRY(b->rv_partial.reserve((size_t)rows * row_stride));
    HIP_TRY(b->stage.reserve((size_t)slices * row_stride));
    HIP_TRY(b->d_view.reserve(1));
    const BankView view = make_view(b);
    // (pageable source: the copy is staged before the call returns, `view` may die)
    HIP_TRY(hipMemcpyAsync(b->d_view.p, &view, sizeof view, hipMemcpyHostToDevice, st));
    b->timed_launches = 0;
    for (int base = 0; base < n_blocks; base += chunk_blocks) {
        const int nb = std::min(chunk_blocks, n_blocks - base);
        const size_t bus_len = (size_t)nb * kBlock * planes;   // [block][plane][256]
        for (int k = 0; k < r.n_echos; ++k)
            r.position[k] = b->ring_pos[k];
        TRY(timing_begin(b, b->timed_launches));
        hipError_t e = launch_reverb_voice(f32, stereo, b->desc.has_env != 0, vpw, st, r, b->d_view.p,
                                           base, b->block_clock + (uint32_t)base, nb,
                                           b->rv_partial.p, bus_len);
        if (e != hipSuccess)
            return fail(KLMI_E_DEVICE, "reverb kernel launch failed: %s", hipGetErrorString(e));
        HIP_TRY(hipEventRecord(b->ev_stop[b->timed_launches], st));
        b->timed_launches++;
        for (int k = 0; k < r.n_echos; ++k)
            b->ring_pos[k] = (int)(((long long)b->ring_pos[k] + (long long)nb * kBlock) % r.length[k]);
        hipLaunchKernelGGL(bus_slice_kernel, dim3((unsigned)((bus_len + 1023) / 1024), slices),
                           dim3(256), 0, st, b->rv_partial.p, rows, bus_len, b->stage.p);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(bus_finish_kernel, dim3((unsigned)((bus_len + 255) / 256)), dim3(256), 0,
                           st, b->stage.p, slices, bus_len, b->mix_divisor,
                           dev_mix + (size_t)base * planes * kBlock);
        HIP_TRY(hipGetLastError());
    }
    b->block_clock += (uint32_t)n_blocks;
    if (b->ext_input.p)
        b->ext_cursor = std::min(b->ext_blocks, b->ext_cursor + n_blocks);
    return 0;
}

static int render_reverb(klmi_bank *b, int n_blocks, float *dev_mix)
{
    if (reverb_fused_form(b))
        return render_reverb_fused(b, n_blocks, dev_mix);
    TRY(reverb_streams(b));
    hipStream_t st = b->ctx->stream, aux = b->aux_stream;
    ReverbState &r = b->reverb;
    const size_t n = (size_t)b->n;
    const bool stereo = b->desc.stereo != 0, f32 = b->desc.reverb_rings_f64 == 0;
    const int planes = stereo ? 2 : 1;
    // time pass: as many samples as are independent of each other (<= shortest ring)
    const int pass = r.min_length >= kReverbPass ? kReverbPass : std::min(r.min_length, 64);
    // blocks per chunk: 32 amortise the launch boundaries between the stages (measured: 8
    // blocks cost 20 % of the whole call at 4,096 channels), fewer when the two dry buses
    // would pass 2 GiB
    static const int chunk_forced = [] {
        const char *e = getenv("KLMI_REVERB_CHUNK");   // tuning knob
        const int v = e ? atoi(e) : 0;
        return v >= 1 && v <= 64 ? v : 0;
    }();
    const long fit = ((long)1 << 31) / ((long)4096 * b->n);
    const int chunk_pref = chunk_forced ? chunk_forced : (int)std::max(8L, std::min(32L, fit));
    const int chunk_blocks = std::min(n_blocks, chunk_pref);
    const size_t chunk_samples = (size_t)chunk_blocks * kBlock;
    // dry bus [voice][bus_stride]: rows are 256 bytes longer than the chunk, so that the
    // rows of consecutive voices -- read at the same sample offset by all waves at about
    // the same time -- do not start a power-of-two apart
    const size_t bus_stride = chunk_samples + 32;
    const size_t bus_elems = bus_stride * n;
    HIP_TRY(b->bus_f64.reserve(2 * bus_elems));
    // voices per wave: enough waves for 16 per compute unit, at most 16 voices each
    static const int vpw_forced = [] {
        const char *e = getenv("KLMI_REVERB_VPW");     // tuning knob
        return e ? atoi(e) : 0;
    }();
    int vpw = (b->n + 16 * b->ctx->n_cus - 1) / (16 * b->ctx->n_cus);
    vpw = std::max(1, std::min(vpw, kReverbVoicesPerWaveMax));
    if (vpw_forced >= 1 && vpw_forced <= 64)
        vpw = vpw_forced;
    const int waves = (b->n + vpw - 1) / vpw;
    const int groups = (waves + kReverbWaves - 1) / kReverbWaves;
    const int rows = groups * kReverbWaves;          // partial rows, one per wave
    const int slices = bus_slices(rows);
    const size_t row_stride = chunk_samples * planes;
    const size_t partial_elems = (size_t)rows * row_stride;
    HIP_TRY(b->rv_partial.reserve(2 * partial_elems));
    HIP_TRY(b->stage.reserve((size_t)slices * row_stride));
    BankView view = make_view(b);
    b->timed_launches = 0;
    // the side stream starts after everything already queued on the main stream
    HIP_TRY(hipEventRecord(b->rv_start, st));
    HIP_TRY(hipStreamWaitEvent(aux, b->rv_start, 0));
    int chunk = 0;
    for (int base = 0; base < n_blocks; base += chunk_blocks, ++chunk) {
        const int nb = std::min(chunk_blocks, n_blocks - base);
        const int buf = chunk & 1;
        double *bus = b->bus_f64.p + (size_t)buf * bus_elems;
        float *partial = b->rv_partial.p + (size_t)buf * partial_elems;
        // 1) dry voices for nb blocks -> bus[voice][sample], on the side stream
        if (chunk >= 2)
            HIP_TRY(hipStreamWaitEvent(aux, b->rv_done[buf], 0));
        hipError_t e = dispatch_voice<OUT_VOICES_F64>(b->desc.has_env, b->desc.has_filter, aux, view,
                                                      0, b->n, base, b->block_clock + (uint32_t)base,
                                                      nb, (int)bus_stride, nullptr, bus);
        if (e != hipSuccess)
            return fail(KLMI_E_DEVICE, "voice kernel launch failed: %s", hipGetErrorString(e));
        HIP_TRY(hipEventRecord(b->rv_ready[buf], aux));
        HIP_TRY(hipStreamWaitEvent(st, b->rv_ready[buf], 0));
        // 2) combs + dry/wet + per-wave voice sums for the whole chunk
        const int total = nb * kBlock;
        const size_t bus_len = (size_t)total * planes;   // [block][plane][256]
        for (int k = 0; k < r.n_echos; ++k)
            r.position[k] = b->ring_pos[k];
        TRY(timing_begin(b, b->timed_launches));
        launch_reverb(f32, stereo, groups, st, r, bus, bus_stride, total, pass, vpw, b->gain_l.p,
                      b->gain_r.p, partial, bus_len);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(b->ev_stop[b->timed_launches], st));
        b->timed_launches++;
        for (int k = 0; k < r.n_echos; ++k)
            b->ring_pos[k] = (int)(((long long)b->ring_pos[k] + total) % r.length[k]);
        HIP_TRY(hipEventRecord(b->rv_done[buf], st));
        // 3) fixed-order fold of the rows (the mixer's sum over voices), / divisor -> float32
        hipLaunchKernelGGL(bus_slice_kernel, dim3((unsigned)((bus_len + 1023) / 1024), slices),
                           dim3(256), 0, st, partial, rows, bus_len, b->stage.p);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(bus_finish_kernel, dim3((unsigned)((bus_len + 255) / 256)), dim3(256), 0,
                           st, b->stage.p, slices, bus_len, b->mix_divisor,
                           dev_mix + (size_t)base * planes * kBlock);
        HIP_TRY(hipGetLastError());
    }
    b->block_clock += (uint32_t)n_blocks;
    if (b->ext_input.p)
        b->ext_cursor = std::min(b->ext_blocks, b->ext_cursor + n_blocks);
    return 0;
}

extern "C" int klmi_bank_reverb_state_get(klmi_bank *b, int voice, double *rings,
                                          int32_t *positions)
{
    if (!b || !rings || !positions)
        return fail(KLMI_E_ARG, "klmi_bank_reverb_state_get: null argument");
    if (!b->desc.has_reverb)
        return fail(KLMI_E_STATE, "klmi_bank_reverb_state_get: bank has no reverb");
    if (voice < 0 || voice >= b->n)
        return fail(KLMI_E_ARG, "klmi_bank_reverb_state_get: voice %d out of range", voice);
    TRY(use(b->ctx));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    // rings are [voice][comb: ring + mirror slots]: one contiguous run per voice
    const size_t run = (size_t)b->reverb.stride;
    std::vector<double> tmp64;
    std::vector<float> tmp32;
    if (b->desc.reverb_rings_f64 == 0) {
        tmp32.resize(run);
        HIP_TRY(hipMemcpy(tmp32.data(), b->rings32.p + (size_t)voice * run, sizeof(float) * run,
                          hipMemcpyDeviceToHost));
    } else {
        tmp64.resize(run);
        HIP_TRY(hipMemcpy(tmp64.data(), b->rings.p + (size_t)voice * run, sizeof(double) * run,
                          hipMemcpyDeviceToHost));
    }
    size_t out = 0;
    for (size_t k = 0; k < b->taps.size(); ++k)
        for (int i = 0; i < b->taps[k]; ++i, ++out)
            rings[out] = b->desc.reverb_rings_f64 == 0 ? (double)tmp32[(size_t)b->reverb.offset[k] + i]
                                                  : tmp64[(size_t)b->reverb.offset[k] + i];
    for (size_t k = 0; k < b->taps.size(); ++k)
        positions[k] = b->ring_pos[k];
    return 0;
}

// ---------------------------------------------------------------------------------
// envelope batch
// ---------------------------------------------------------------------------------
struct klmi_env {
    klmi_ctx *ctx = nullptr;
    int count = 0;
    double dt = kDt;
    std::vector<double> attack, decay, sustain, release, overshoot;
    std::vector<uint8_t> loop, retrigger;
    DevBuf<double> coef, value, out;
    DevBuf<uint32_t> flags, state;
};

static int env_upload(klmi_env *e, int first, int count)
{
    std::vector<double> coef((size_t)7 * count);
    std::vector<uint32_t> flags(count);
    for (int i = 0; i < count; ++i) {
        const int v = first + i;
        double c[6];
        env_coefficients(e->attack[v], e->decay[v], e->sustain[v], e->release[v], e->dt,
                         e->overshoot[v], c);
        for (int k = 0; k < 6; ++k)
            coef[(size_t)k * count + i] = c[k];
        coef[(size_t)6 * count + i] = e->sustain[v];
        flags[i] = (e->loop[v] ? 1u : 0u) | (e->retrigger[v] ? 2u : 0u);
    }
    HIP_TRY(hipStreamSynchronize(e->ctx->stream));
    for (int k = 0; k < 7; ++k)
        HIP_TRY(hipMemcpy(e->coef.p + (size_t)k * e->count + first, &coef[(size_t)k * count],
                          sizeof(double) * count, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->flags.p + first, flags.data(), sizeof(uint32_t) * count,
                      hipMemcpyHostToDevice));
    return 0;
}

extern "C" int klmi_env_create(klmi_ctx *ctx, int count, double attack, double decay,
                               double sustain, double release, double dt, double overshoot,
                               int retrigger, int loop, klmi_env **out)
{
    TRY(use(ctx));
    if (!out || count <= 0)
        return fail(KLMI_E_ARG, "klmi_env_create: bad argument");
    // The reference's constructor (_envelope.c:518-535) stores its arguments unchecked;
    // only the setters validate.  dt must be usable as a divisor here.
    if (!(dt > 0.))
        return fail(KLMI_E_ARG, "klmi_env_create: dt must be > 0");
    klmi_env *e = new klmi_env;
    e->ctx = ctx;
    e->count = count;
    e->dt = dt;
    e->attack.assign(count, attack);
    e->decay.assign(count, decay);
    e->sustain.assign(count, sustain);
    e->release.assign(count, release);
    e->overshoot.assign(count, clip(overshoot, 1e-9, 1e9));  // _envelope.c:532
    e->loop.assign(count, loop != 0);
    e->retrigger.assign(count, retrigger != 0);
    hipError_t err = e->coef.alloc((size_t)7 * count);
    if (err == hipSuccess) err = e->value.alloc(count);
    if (err == hipSuccess) err = e->flags.alloc(count);
    if (err == hipSuccess) err = e->state.alloc(count);
    if (err == hipSuccess) err = hipMemsetAsync(e->value.p, 0, sizeof(double) * count, ctx->stream);
    if (err == hipSuccess) err = hipMemsetAsync(e->state.p, 0, sizeof(uint32_t) * count, ctx->stream);
    if (err != hipSuccess) {
        delete e;
        return fail(KLMI_E_MEMORY, "klmi_env_create: %s", hipGetErrorString(err));
    }
    int rc = env_upload(e, 0, count);
    if (rc != 0) {
        delete e;
        return rc;
    }
    ctx->live_objects++;
    *out = e;
    return 0;
}

extern "C" int klmi_env_destroy(klmi_env *env)
{
    if (!env)
        return 0;
    (void)hipSetDevice(env->ctx->device);
    (void)hipStreamSynchronize(env->ctx->stream);
    env->ctx->live_objects--;
    delete env;
    return 0;
}

static int env_range(klmi_env *e, int first, int count, const void *host, const char *what)
{
    if (!e || !host)
        return fail(KLMI_E_ARG, "%s: null argument", what);
    if (first < 0 || count < 0 || (long)first + count > e->count)
        return fail(KLMI_E_ARG, "%s: range [%d, %d) outside batch of %d", what, first,
                    first + count, e->count);
    return 0;
}

extern "C" int klmi_env_set(klmi_env *e, int field, int first, int count, const double *host)
{
    TRY(env_range(e, first, count, host, "klmi_env_set"));
    TRY(use(e->ctx));
    switch (field) {
    case KLMI_F_ATTACK:
    case KLMI_F_DECAY:
    case KLMI_F_RELEASE:
    case KLMI_F_OVERSHOOT:
        for (int i = 0; i < count; ++i)
            if (!(host[i] >= 0.))
                return fail(KLMI_E_ARG, "%s must be positive!",
                            field == KLMI_F_ATTACK ? "attack" : field == KLMI_F_DECAY ? "decay"
                            : field == KLMI_F_RELEASE ? "release" : "overshoot");
        break;
    case KLMI_F_SUSTAIN:
        for (int i = 0; i < count; ++i)
            if (!(host[i] >= 0. && host[i] <= 1.))
                return fail(KLMI_E_ARG, "sustain not within bounds!");
        break;
    case KLMI_F_LOOP:
    case KLMI_F_RETRIGGER:
        break;
    case KLMI_F_ENV_VALUE:
        HIP_TRY(hipStreamSynchronize(e->ctx->stream));
        HIP_TRY(hipMemcpy(e->value.p + first, host, sizeof(double) * count, hipMemcpyHostToDevice));
        return 0;
    default:
        return fail(KLMI_E_ARG, "klmi_env_set: field %d not settable", field);
    }
    for (int i = 0; i < count; ++i) {
        const int v = first + i;
        switch (field) {
        case KLMI_F_ATTACK: e->attack[v] = host[i]; break;
        case KLMI_F_DECAY: e->decay[v] = host[i]; break;
        case KLMI_F_SUSTAIN: e->sustain[v] = host[i]; break;
        case KLMI_F_RELEASE: e->release[v] = host[i]; break;
        case KLMI_F_OVERSHOOT: e->overshoot[v] = clip(host[i], 1e-9, 1e9); break;  // :374
        case KLMI_F_LOOP: e->loop[v] = host[i] != 0.; break;
        case KLMI_F_RETRIGGER: e->retrigger[v] = host[i] != 0.; break;
        }
    }
    return env_upload(e, first, count);
}

extern "C" int klmi_env_get(klmi_env *e, int field, int first, int count, double *host)
{
    TRY(env_range(e, first, count, host, "klmi_env_get"));
    TRY(use(e->ctx));
    std::vector<uint32_t> raw;
    switch (field) {
    case KLMI_F_ATTACK: std::copy_n(e->attack.begin() + first, count, host); return 0;
    case KLMI_F_DECAY: std::copy_n(e->decay.begin() + first, count, host); return 0;
    case KLMI_F_SUSTAIN: std::copy_n(e->sustain.begin() + first, count, host); return 0;
    case KLMI_F_RELEASE: std::copy_n(e->release.begin() + first, count, host); return 0;
    case KLMI_F_OVERSHOOT: std::copy_n(e->overshoot.begin() + first, count, host); return 0;
    case KLMI_F_LOOP:
        for (int i = 0; i < count; ++i) host[i] = e->loop[first + i];
        return 0;
    case KLMI_F_RETRIGGER:
        for (int i = 0; i < count; ++i) host[i] = e->retrigger[first + i];
        return 0;
    case KLMI_F_ENV_VALUE:
        HIP_TRY(hipStreamSynchronize(e->ctx->stream));
        HIP_TRY(hipMemcpy(host, e->value.p + first, sizeof(double) * count, hipMemcpyDeviceToHost));
        return 0;
    case KLMI_F_ENV_STAGE:
    case KLMI_F_ENV_ENABLED:
        raw.resize(count);
        HIP_TRY(hipStreamSynchronize(e->ctx->stream));
        HIP_TRY(hipMemcpy(raw.data(), e->state.p + first, sizeof(uint32_t) * count,
                          hipMemcpyDeviceToHost));
        for (int i = 0; i < count; ++i)
            host[i] = field == KLMI_F_ENV_STAGE ? (double)(raw[i] & 7u)
                                                : (double)((raw[i] >> 8) & 1u);
        return 0;
    default:
        return fail(KLMI_E_ARG, "klmi_env_get: unknown field %d", field);
    }
}

extern "C" int klmi_env_gate(klmi_env *e, int first, int count, int on)
{
    if (!e)
        return fail(KLMI_E_ARG, "klmi_env_gate: null envelope");
    if (first < 0 || count < 0 || (long)first + count > e->count)
        return fail(KLMI_E_ARG, "klmi_env_gate: range outside batch");
    TRY(use(e->ctx));
    if (count == 0)
        return 0;
    hipLaunchKernelGGL(env_gate_kernel, dim3((count + 255) / 256), dim3(256), 0, e->ctx->stream,
                       e->count, first, count, on, e->flags.p, e->state.p);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int klmi_env_sample(klmi_env *e, long n, double *host_out)
{
    if (!e)
        return fail(KLMI_E_ARG, "klmi_env_sample: null envelope");
    if (n < 0)
        return fail(KLMI_E_ARG, "Could not cast *bufferSize to size_t!");  // _envelope.c:640-644
    if (n == 0)
        return 0;
    if (!host_out)
        return fail(KLMI_E_ARG, "klmi_env_sample: null output");
    TRY(use(e->ctx));
    const size_t len = (size_t)n * e->count;
    HIP_TRY(e->out.reserve(len));
    hipLaunchKernelGGL(env_sample_kernel, dim3((e->count + 63) / 64), dim3(64), 0, e->ctx->stream,
                       e->count, n, e->coef.p, e->flags.p, e->value.p, e->state.p, e->out.p);
    HIP_TRY(hipGetLastError());
    std::vector<double> t(len);
    HIP_TRY(hipMemcpyAsync(t.data(), e->out.p, sizeof(double) * len, hipMemcpyDeviceToHost,
                           e->ctx->stream));
    HIP_TRY(hipStreamSynchronize(e->ctx->stream));
    for (int v = 0; v < e->count; ++v)  // [n][count] -> [count][n]
        for (long i = 0; i < n; ++i)
            host_out[(size_t)v * n + i] = t[(size_t)i * e->count + v];
    return 0;
}

extern "C" int klmi_env_active(klmi_env *e, int32_t *host_out)
{
    if (!e || !host_out)
        return fail(KLMI_E_ARG, "klmi_env_active: null argument");
    TRY(use(e->ctx));
    std::vector<uint32_t> raw(e->count);
    HIP_TRY(hipStreamSynchronize(e->ctx->stream));
    HIP_TRY(hipMemcpy(raw.data(), e->state.p, sizeof(uint32_t) * e->count, hipMemcpyDeviceToHost));
    for (int i = 0; i < e->count; ++i)
        host_out[i] = (raw[i] & 7u) != 0;  // _envelope.c:385-393
    return 0;
}

// ---------------------------------------------------------------------------------
// ring-buffer filter batch
// ---------------------------------------------------------------------------------
struct klmi_comb {
    klmi_ctx *ctx = nullptr;
    int kind = 0, count = 0, length = 0, position = 0;
    DevBuf<double> alpha, ring, x, y;
};

extern "C" int klmi_comb_create(klmi_ctx *ctx, int kind, int count, int length, double alpha,
                                klmi_comb **out)
{
    TRY(use(ctx));
    if (!out || count <= 0 || kind < 0 || kind > 2)
        return fail(KLMI_E_ARG, "klmi_comb_create: bad argument");
    if (length <= 0 || (long)length > 60L * KLMI_SAMPLING_RATE)  // _filters.c:43-46
        return fail(KLMI_E_MEMORY, "Could not initialize RingBuffer!");
    klmi_comb *c = new klmi_comb;
    c->ctx = ctx;
    c->kind = kind;
    c->count = count;
    c->length = length;
    hipError_t e = c->alpha.alloc(count);
    if (e == hipSuccess) e = c->ring.alloc((size_t)length * count);
    if (e == hipSuccess)
        e = hipMemsetAsync(c->ring.p, 0, sizeof(double) * (size_t)length * count, ctx->stream);
    if (e == hipSuccess) {
        std::vector<double> a(count, alpha);
        e = hipMemcpy(c->alpha.p, a.data(), sizeof(double) * count, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        delete c;
        return fail(KLMI_E_MEMORY, "klmi_comb_create: %s", hipGetErrorString(e));
    }
    ctx->live_objects++;
    *out = c;
    return 0;
}

extern "C" int klmi_comb_destroy(klmi_comb *c)
{
    if (!c)
        return 0;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    c->ctx->live_objects--;
    delete c;
    return 0;
}

extern "C" int klmi_comb_set_alpha(klmi_comb *c, int first, int count, const double *host)
{
    if (!c || !host || first < 0 || count < 0 || (long)first + count > c->count)
        return fail(KLMI_E_ARG, "klmi_comb_set_alpha: bad argument");
    TRY(use(c->ctx));
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    HIP_TRY(hipMemcpy(c->alpha.p + first, host, sizeof(double) * count, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int klmi_comb_filter(klmi_comb *c, const double *host_x, long n, double *host_y)
{
    if (!c)
        return fail(KLMI_E_ARG, "klmi_comb_filter: null filter");
    if (n < 0)
        return fail(KLMI_E_ARG, "klmi_comb_filter: negative length");
    if (n == 0)
        return 0;
    if (!host_x || !host_y)
        return fail(KLMI_E_ARG, "klmi_comb_filter: null buffer");
    TRY(use(c->ctx));
    const size_t len = (size_t)n * c->count;
    std::vector<double> t(len);
    for (int f = 0; f < c->count; ++f)  // [count][n] -> [n][count]
        for (long i = 0; i < n; ++i)
            t[(size_t)i * c->count + f] = host_x[(size_t)f * n + i];
    HIP_TRY(c->x.reserve(len));
    HIP_TRY(c->y.reserve(len));
    hipStream_t st = c->ctx->stream;
    HIP_TRY(hipMemcpyAsync(c->x.p, t.data(), sizeof(double) * len, hipMemcpyHostToDevice, st));
    const dim3 grid((c->count + 63) / 64), block(64);
    if (c->kind == 0)
        hipLaunchKernelGGL(ring_filter_kernel<0>, grid, block, 0, st, c->count, c->length,
                           c->position, n, c->alpha.p, c->ring.p, c->x.p, c->y.p);
    else if (c->kind == 1)
        hipLaunchKernelGGL(ring_filter_kernel<1>, grid, block, 0, st, c->count, c->length,
                           c->position, n, c->alpha.p, c->ring.p, c->x.p, c->y.p);
    else
        hipLaunchKernelGGL(ring_filter_kernel<2>, grid, block, 0, st, c->count, c->length,
                           c->position, n, c->alpha.p, c->ring.p, c->x.p, c->y.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(t.data(), c->y.p, sizeof(double) * len, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    c->position = (int)((c->position + n) % c->length);
    for (int f = 0; f < c->count; ++f)
        for (long i = 0; i < n; ++i)
            host_y[(size_t)f * n + i] = t[(size_t)i * c->count + f];
    return 0;
}

// ---------------------------------------------------------------------------------
// kick drums (klang.audio.synthesizer.Kick x count)
// ---------------------------------------------------------------------------------
struct klmi_kick {
    klmi_ctx *ctx = nullptr;
    int count = 0;
    DevBuf<double> par, time, phase, out, t_table;
    DevBuf<unsigned char> trig;
    std::vector<std::pair<uint32_t, uint32_t>> pending;  // (kick, block relative to next render)
};

extern "C" int klmi_kick_create(klmi_ctx *ctx, int count, klmi_kick **out)
{
    TRY(use(ctx));
    if (!out || count <= 0)
        return fail(KLMI_E_ARG, "klmi_kick_create: bad argument");
    klmi_kick *k = new klmi_kick;
    k->ctx = ctx;
    k->count = count;
    std::vector<double> par((size_t)4 * count), t(kBlock);
    const double defaults[4] = {40., .8, 2., .3};  // synthesizer.py:257
    for (int f = 0; f < 4; ++f)
        std::fill_n(par.begin() + (size_t)f * count, count, defaults[f]);
    for (int i = 0; i < kBlock; ++i)
        t[i] = kDt * (double)i;
    hipError_t e = k->par.alloc(par.size());
    if (e == hipSuccess) e = k->time.alloc(count);
    if (e == hipSuccess) e = k->phase.alloc((size_t)count * kBlock);
    if (e == hipSuccess) e = k->t_table.alloc(kBlock);
    if (e == hipSuccess) e = hipMemsetAsync(k->time.p, 0, sizeof(double) * count, ctx->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(k->phase.p, 0, sizeof(double) * count * kBlock, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpy(k->par.p, par.data(), sizeof(double) * par.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(k->t_table.p, t.data(), sizeof(double) * kBlock, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        delete k;
        return fail(KLMI_E_MEMORY, "klmi_kick_create: %s", hipGetErrorString(e));
    }
    ctx->live_objects++;
    *out = k;
    return 0;
}

extern "C" int klmi_kick_destroy(klmi_kick *kick)
{
    if (!kick)
        return 0;
    (void)hipSetDevice(kick->ctx->device);
    (void)hipStreamSynchronize(kick->ctx->stream);
    kick->ctx->live_objects--;
    delete kick;
    return 0;
}

extern "C" int klmi_kick_set(klmi_kick *k, int field, int first, int count, const double *host)
{
    if (!k || !host)
        return fail(KLMI_E_ARG, "klmi_kick_set: null argument");
    if (field < KLMI_KICK_FREQUENCY || field > KLMI_KICK_PITCH_DECAY)
        return fail(KLMI_E_ARG, "klmi_kick_set: unknown field %d", field);
    if (first < 0 || count < 0 || (long)first + count > k->count)
        return fail(KLMI_E_ARG, "klmi_kick_set: range [%d, %d) outside %d kicks", first,
                    first + count, k->count);
    if (field == KLMI_KICK_DECAY || field == KLMI_KICK_PITCH_DECAY)
        for (int i = 0; i < count; ++i)
            if (!(host[i] > 0.))  // the reference divides by it (synthesizer.py:34)
                return fail(KLMI_E_ARG, "klmi_kick_set: decay times must be > 0, got %g", host[i]);
    TRY(use(k->ctx));
    HIP_TRY(hipStreamSynchronize(k->ctx->stream));
    HIP_TRY(hipMemcpy(k->par.p + (size_t)field * k->count + first, host, sizeof(double) * count,
                      hipMemcpyHostToDevice));
    return 0;
}

extern "C" int klmi_kick_trigger(klmi_kick *k, const uint32_t *which, const uint32_t *block,
                                 size_t n)
{
    if (!k || (n && (!which || !block)))
        return fail(KLMI_E_ARG, "klmi_kick_trigger: null argument");
    for (size_t i = 0; i < n; ++i)
        if (which[i] >= (uint32_t)k->count)
            return fail(KLMI_E_ARG, "klmi_kick_trigger: trigger %zu addresses kick %u of %d", i,
                        which[i], k->count);
    for (size_t i = 0; i < n; ++i)
        k->pending.emplace_back(which[i], block[i]);
    return 0;
}

extern "C" int klmi_kick_render(klmi_kick *k, int n_blocks, double *host_out)
{
    if (!k || !host_out)
        return fail(KLMI_E_ARG, "klmi_kick_render: null argument");
    if (n_blocks <= 0)
        return fail(KLMI_E_ARG, "klmi_kick_render: n_blocks must be positive, got %d", n_blocks);
    TRY(use(k->ctx));
    hipStream_t st = k->ctx->stream;
    const size_t cells = (size_t)k->count * n_blocks;
    std::vector<unsigned char> trig(cells, 0);
    std::vector<std::pair<uint32_t, uint32_t>> later;
    for (const auto &p : k->pending) {
        if (p.second < (uint32_t)n_blocks)
            trig[(size_t)p.first * n_blocks + p.second] = 1;
        else
            later.emplace_back(p.first, p.second - (uint32_t)n_blocks);
    }
    k->pending.swap(later);
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(k->trig.reserve(cells));
    HIP_TRY(k->out.reserve(cells * kBlock));
    HIP_TRY(hipMemcpyAsync(k->trig.p, trig.data(), cells, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(kick_kernel, dim3(k->count), dim3(kBlock), 0, st, k->count, n_blocks,
                       k->par.p, k->time.p, k->phase.p, k->trig.p, k->t_table.p, k->out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host_out, k->out.p, sizeof(double) * cells * kBlock,
                           hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));  // `trig` dies here
    return 0;
}

extern "C" int klmi_kick_state_get(klmi_kick *k, double *host_time, double *host_phase)
{
    if (!k)
        return fail(KLMI_E_ARG, "klmi_kick_state_get: null kick");
    TRY(use(k->ctx));
    HIP_TRY(hipStreamSynchronize(k->ctx->stream));
    if (host_time)
        HIP_TRY(hipMemcpy(host_time, k->time.p, sizeof(double) * k->count, hipMemcpyDeviceToHost));
    if (host_phase)
        HIP_TRY(hipMemcpy(host_phase, k->phase.p, sizeof(double) * k->count * kBlock,
                          hipMemcpyDeviceToHost));
    return 0;
}

// ---------------------------------------------------------------------------------
// chirp oscillators (klang.audio.oscillators.Chirper x count)
// ---------------------------------------------------------------------------------
struct klmi_chirp {
    klmi_ctx *ctx = nullptr;
    int count = 0;
    DevBuf<double> par, t, out;
    DevBuf<int32_t> kind;
};

static int chirp_reset_range(klmi_chirp *c, int first, int count)
{
    std::vector<double> t((size_t)count * kBlock);
    for (int k = 0; k < count; ++k)
        for (int i = 0; i < kBlock; ++i)
            t[(size_t)k * kBlock + i] = kDt * (double)i;  // helpers.py:30 get_time
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    HIP_TRY(hipMemcpy(c->t.p + (size_t)first * kBlock, t.data(), sizeof(double) * t.size(),
                      hipMemcpyHostToDevice));
    return 0;
}

extern "C" int klmi_chirp_create(klmi_ctx *ctx, int count, const double *start_frequency,
                                 const double *duration, const double *end_frequency,
                                 const int32_t *method, const int32_t *wave, klmi_chirp **out)
{
    TRY(use(ctx));
    if (!out || count <= 0 || !start_frequency || !duration || !end_frequency || !method || !wave)
        return fail(KLMI_E_ARG, "klmi_chirp_create: bad argument");
    for (int k = 0; k < count; ++k) {
        const double f0 = start_frequency[k], f1 = end_frequency[k];
        if (method[k] < KLMI_CHIRP_LINEAR || method[k] > KLMI_CHIRP_HYPERBOLIC)
            return fail(KLMI_E_ARG, "method must be 'linear', 'quadratic', 'logarithmic', or "
                                    "'hyperbolic', but a value of %d was given.", method[k]);
        if (wave[k] < KLMI_WAVE_SINE || wave[k] > KLMI_WAVE_TRIANGLE)
            return fail(KLMI_E_ARG, "klmi_chirp_create: wave id %d (sine, square, sawtooth, "
                                    "triangle are 0..3)", wave[k]);
        // the sweep formula is only reached by a chirper whose first buffer starts before
        // tEnd and whose ends differ (oscillators.py:23-25); the others never raise
        if (!(duration[k] > 0.) || f0 == f1)
            continue;
        if (method[k] == KLMI_CHIRP_LOGARITHMIC && f0 * f1 <= 0.0)
            return fail(KLMI_E_ARG, "For a logarithmic chirp, f0 and f1 must be nonzero and have "
                                    "the same sign.");
        if (method[k] == KLMI_CHIRP_HYPERBOLIC && (f0 == 0 || f1 == 0))
            return fail(KLMI_E_ARG, "For a hyperbolic chirp, f0 and f1 must be nonzero.");
    }
    klmi_chirp *c = new klmi_chirp;
    c->ctx = ctx;
    c->count = count;
    std::vector<double> par((size_t)3 * count);
    std::vector<int32_t> kind((size_t)2 * count);
    for (int k = 0; k < count; ++k) {
        par[k] = start_frequency[k];
        par[(size_t)count + k] = duration[k];
        par[2 * (size_t)count + k] = end_frequency[k];
        kind[k] = method[k];
        kind[(size_t)count + k] = wave[k];
    }
    hipError_t e = c->par.alloc(par.size());
    if (e == hipSuccess) e = c->kind.alloc(kind.size());
    if (e == hipSuccess) e = c->t.alloc((size_t)count * kBlock);
    if (e == hipSuccess)
        e = hipMemcpy(c->par.p, par.data(), sizeof(double) * par.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(c->kind.p, kind.data(), sizeof(int32_t) * kind.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        delete c;
        return fail(KLMI_E_MEMORY, "klmi_chirp_create: %s", hipGetErrorString(e));
    }
    const int rc = chirp_reset_range(c, 0, count);
    if (rc != 0) {
        delete c;
        return rc;
    }
    ctx->live_objects++;
    *out = c;
    return 0;
}

extern "C" int klmi_chirp_destroy(klmi_chirp *chirp)
{
    if (!chirp)
        return 0;
    (void)hipSetDevice(chirp->ctx->device);
    (void)hipStreamSynchronize(chirp->ctx->stream);
    chirp->ctx->live_objects--;
    delete chirp;
    return 0;
}

extern "C" int klmi_chirp_reset(klmi_chirp *c, int first, int count)
{
    if (!c)
        return fail(KLMI_E_ARG, "klmi_chirp_reset: null chirp");
    if (first < 0 || count < 0 || (long)first + count > c->count)
        return fail(KLMI_E_ARG, "klmi_chirp_reset: range [%d, %d) outside %d chirpers", first,
                    first + count, c->count);
    TRY(use(c->ctx));
    return count ? chirp_reset_range(c, first, count) : 0;
}

extern "C" int klmi_chirp_render(klmi_chirp *c, int n_blocks, double *host_out)
{
    if (!c || !host_out)
        return fail(KLMI_E_ARG, "klmi_chirp_render: null argument");
    if (n_blocks <= 0)
        return fail(KLMI_E_ARG, "klmi_chirp_render: n_blocks must be positive, got %d", n_blocks);
    TRY(use(c->ctx));
    hipStream_t st = c->ctx->stream;
    const size_t len = (size_t)c->count * n_blocks * kBlock;
    HIP_TRY(c->out.reserve(len));
    hipLaunchKernelGGL(chirp_kernel, dim3(c->count), dim3(kBlock), 0, st, c->count, n_blocks,
                       c->par.p, c->kind.p, c->t.p, c->out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host_out, c->out.p, sizeof(double) * len, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

extern "C" int klmi_chirp_state_get(klmi_chirp *c, double *host_t)
{
    if (!c || !host_t)
        return fail(KLMI_E_ARG, "klmi_chirp_state_get: null argument");
    TRY(use(c->ctx));
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    HIP_TRY(hipMemcpy(host_t, c->t.p, sizeof(double) * c->count * kBlock, hipMemcpyDeviceToHost));
    return 0;
}
