// bank_kernels.h -- the fused voice kernel (oscillator -> envelope -> biquad ->
// mix / materialise) and its deterministic bus reduction, for gfx950.
//
// Mapping: one lane = one voice; phase, envelope stage/value and the IIR
// z-state live in VGPRs for the whole launch and touch HBM once per launch.
// SoA device arrays, voice index fastest => every load/store is a coalesced
// 64-lane access.  The mix-down goes through a wave-private LDS tile
// [CHUNK samples][64 voices] that is read back transposed (lane = sample), so
// the cross-voice sum costs one ds_write_b32 per sample and 1/4 ds_read_b128
// + one add per sample instead of a 6-step cross-lane reduction per sample.
#pragma once
#include "dsp_device.h"
#include <type_traits>
#include <utility>

namespace klmi {

// t[n] = DT * n, n = 0..256 (helpers.py:27-32), padded to whole groups of 8
constexpr int kTimeSlots = kBlock + 8;
// one copy per translation unit (each uploads its own: launch.h upload_time_table_*)
static __constant__ __attribute__((aligned(64))) double c_time[kTimeSlots];
struct __attribute__((aligned(64))) Time8 {
    double t[8];
};

struct BankView {
    int n;  // voices in the bank
    // parameters
    const double *duty;
    const int32_t *wave;
    const double *env_coef;    // [7][n]: aCoef aBase dCoef dBase rCoef rBase sustain
    const uint32_t *env_flags; // bit0 loop, bit1 retrigger
    const double *coef_table;  // [rows][5]: b0 b1 b2 a1 a2
    const double *gain_l, *gain_r;  // gain (mono) or gain*pan_l / gain*pan_r (stereo)
    // state
    double *phase, *env_value, *z0, *z1;
    uint32_t *env_state;       // bits 0-2 stage, bit 8 enabled
    // frequency / amplitude are state too once NOTE_ON events have been applied
    double *frequency_rw, *amplitude_rw;
    int32_t *coef_index_rw;
    // events of this render call, sorted by (voice, block); CSR by voice
    const uint32_t *ev_offset;  // [n+1]
    const uint32_t *ev_block, *ev_kind;
    const int32_t *ev_index;    // SET_CUTOFF: table row
    const double *ev_velocity, *ev_frequency;
    uint32_t *ev_cursor;        // [n] progress through the voice's events
    // external input [span][n] (time-major) for W_EXTERNAL voices
    const double *ext_input;
    // modulation inputs; null / 0 when the bank has none
    const double *lfo_par;     // [3][n]: frequency, scale, offset of the cutoff Lfo
    const int32_t *lfo_wave;
    double *lfo_phase;         // state
    const double *grid;        // [1000] filter grid frequencies
    int mod_target;            // ModTarget
    const double *mod_par;     // [3][n]: frequency, scale (FM: intensity), offset
    const int32_t *mod_wave;
    double *mod_phase;         // state
    // effect stages, in patch order; n_fx == 0: the plain [Filter] chain
    int n_fx;
    unsigned char fx_kind[8];  // FxKind
    const double *fx_par;      // [8][kFxSlots][n] per-stage device parameters
    double *fx_state;          // [8][2][n]: tremolo lfo phase, zi; delay position; ring modulator
                               // lfo phase, modulator phase
    double *fx_ring;           // the delay stage's lines: tiles[(t / 8) % fx_ring_tiles][n][8] by
                               // absolute sample time t (delay_tiles below), or, block-major,
                               // [n][fx_ring_stride] RingBuffer arrays
    int fx_ring_stride;
    int fx_ring_tiles;
    // some voice's delay line is shorter than one buffer: the lines are RingBuffer data arrays
    // ([length slots][256 tail]) and the Delay runs block-major (mod_kernels.h delay_block)
    int fx_block_major;
    double *fx_stage;          // its staging area, [waves][2][256][64]
    // synthesizers: activity gating and device-side note routing
    int gated;                  // voices are only updated while Voice.active
    int poly;                   // voices per PolyphonicSynthesizer, 0: none
    const uint32_t *nt_offset;  // notes of this render call, CSR by instrument [n/poly + 1]
    const uint32_t *nt_block;
    const int32_t *nt_pitch;
    const double *nt_velocity, *nt_frequency;
    uint32_t *nt_cursor;        // [n] state: notes of its instrument a voice has seen
    uint32_t *round_robin;      // [n] state: the instrument's cycle position (replicated)
    int32_t *current_pitch;     // [n] state: Voice.currentPitch
};

enum FxKind : int { FX_NONE = 0, FX_FILTER, FX_GAIN, FX_TRANSFORM, FX_TANH, FX_OCTAVE, FX_TREMOLO, FX_DELAY,
                    FX_SUBSAMPLE, FX_BITCRUSH, FX_RINGMOD };
// device parameter rows per stage: GAIN gain | TRANSFORM scale, offset | TANH drive |
// TREMOLO rate, depth, duty, b0, a1 (one-pole design done on the host, effects.py:124-125) |
// DELAY length (samples), feedback, drywet | SUBSAMPLE factor | BITCRUSH nBits |
// RINGMOD frequency, rate, amount, dryWet, lfo wave
constexpr int kFxSlots = 6;

enum ModTarget : int { MT_NONE = 0, MT_FM = 1, MT_FREQUENCY = 2, MT_DUTY = 3 };
constexpr int kGridPoints = 1000;
constexpr double kInterval = kDt * kBlock;  // klang/audio/helpers.py:12

enum OutMode : int {
    OUT_MIX_MONO = 0,    // partial bus per workgroup
    OUT_MIX_STEREO = 1,
    OUT_VOICES_F32 = 2,  // out[voice][span] float32
    OUT_VOICES_F64 = 3,  // out[voice][span] float64
    OUT_BUS_F64 = 4      // out[sample][voice] float64 (input of the reverb kernel)
};

template <int MODE> struct TileTraits;
template <> struct TileTraits<OUT_MIX_MONO> { using T = float; static constexpr int chunk = 64, stride = 68, planes = 1; };
template <> struct TileTraits<OUT_MIX_STEREO> { using T = float; static constexpr int chunk = 32, stride = 68, planes = 2; };
template <> struct TileTraits<OUT_VOICES_F32> { using T = float; static constexpr int chunk = 64, stride = 65, planes = 1; };
template <> struct TileTraits<OUT_VOICES_F64> { using T = double; static constexpr int chunk = 32, stride = 65, planes = 1; };
template <> struct TileTraits<OUT_BUS_F64> { using T = double; static constexpr int chunk = 64, stride = 1, planes = 0; };

constexpr int kWavesPerGroup = 4;
constexpr int kGroup = 64 * kWavesPerGroup;

// LDS bytes of a workgroup's mix-down / transpose tiles
template <int MODE> __host__ __device__ constexpr size_t tile_bytes()
{
    using TT = TileTraits<MODE>;
    return sizeof(typename TT::T) * (size_t)kWavesPerGroup * (TT::planes ? TT::planes : 1) * TT::chunk *
           TT::stride;
}
// general kernel with effect stages: samples per wave strip (float64 [kFxStrip][64] per wave)
constexpr int kFxStrip = 32;
template <int MODE> __host__ __device__ constexpr size_t fx_strip_bytes()
{
    using TT = TileTraits<MODE>;
    return sizeof(double) * (size_t)kWavesPerGroup * (kFxStrip < TT::chunk ? kFxStrip : TT::chunk) * 64;
}

// high 32 bits of TAU = 0x401921FB54442D18
constexpr uint32_t kTauHighWord = 0x401921FBu;

// Sum of one tile row (64 voices) read as 16 x ds_read_b128, pairwise order.
__device__ __forceinline__ float row_sum64(const float *row)
{
    const float4 *r4 = reinterpret_cast<const float4 *>(row);
    float s[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float4 v = r4[j];
        s[j] = (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
        for (int j = 0; j < w; ++j)
            s[j] = s[j] + s[j + w];
    return s[0];
}

// One sample of one voice into the wave's tile (or straight to the time-major bus).
template <int MODE, bool UNIT = false>
__device__ __forceinline__ void emit_sample(typename TileTraits<MODE>::T *tile, int i, int lane,
                                            double x, double gl, double gr, bool live, void *out,
                                            int s0, int count, int local)
{
    using TT = TileTraits<MODE>;
    if constexpr (MODE == OUT_MIX_MONO) {
        tile[i * TT::stride + lane] = UNIT ? (float)x : (float)(gl * x);
    } else if constexpr (MODE == OUT_MIX_STEREO) {
        tile[i * TT::stride + lane] = (float)(gl * x);
        tile[(TT::chunk + i) * TT::stride + lane] = (float)(gr * x);
    } else if constexpr (MODE == OUT_VOICES_F32 || MODE == OUT_VOICES_F64) {
        tile[i * TT::stride + lane] = (typename TT::T)x;
    } else {
        if (live)
            reinterpret_cast<double *>(out)[(size_t)(s0 + i) * count + local] = x;
    }
}

// End of a chunk of TT::chunk samples: the tile leaves the wave.
//   mix modes: lane = sample, sum the 64 voices of this wave; one partial row per WAVE,
//              so the waves of a workgroup never wait for each other
//   voices:    lane = sample, 64 voice rows of TT::chunk contiguous samples each
template <int MODE>
__device__ __forceinline__ void flush_chunk(typename TileTraits<MODE>::T *tile, int lane, int wave,
                                            int b, int c, int n_blocks, int count, int span_total,
                                            int s0, float *__restrict__ partial,
                                            void *__restrict__ out)
{
    using TT = TileTraits<MODE>;
    using T = typename TT::T;
    constexpr int CH = TT::chunk;
    constexpr int PLANES = TT::planes;
    // the tile is wave-private: LDS operations of one wave execute in order;
    // the fence only stops the compiler from moving them across this point
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (MODE == OUT_MIX_MONO || MODE == OUT_MIX_STEREO) {
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
            for (int r = lane; r < CH; r += 64) {
                const size_t o = ((size_t)b * PLANES + pl) * kBlock + c * CH + r;
                partial[(size_t)(blockIdx.x * kWavesPerGroup + wave) * n_blocks * PLANES * kBlock + o] =
                    row_sum64(tile + (size_t)(pl * CH + r) * TT::stride);
            }
        }
    } else if constexpr (MODE == OUT_VOICES_F32 || MODE == OUT_VOICES_F64) {
        T *o = reinterpret_cast<T *>(out);
        const int base_local = blockIdx.x * kGroup + wave * 64;
        // A voice's CH samples are 256 contiguous bytes of its row.  16 lanes x 16 bytes cover
        // one row, so a wavefront stores 4 rows per instruction (16 instead of 64 stores per
        // chunk): all LDS reads first (one wait), then the stores back to back.  Rows that
        // are not 16-byte aligned take the element-wise loop.
        constexpr int E = 16 / (int)sizeof(T);   // elements per 16-byte store
        static_assert(CH / E == 16, "16 lanes per voice row");
        struct alignas(16) Pack {
            T e[E];
        };
        const bool wide = ((reinterpret_cast<uintptr_t>(o) | ((size_t)span_total * sizeof(T))) & 15) == 0;
        if (wide) {
            const int q = lane >> 4, seg = lane & 15;
            Pack pk[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
#pragma unroll
                for (int k = 0; k < E; ++k)
                    pk[it].e[k] = tile[(seg * E + k) * TT::stride + it * 4 + q];
            }
            T *row = o + (size_t)(base_local + q) * span_total + s0 + seg * E;
            if (base_local + 64 <= count) {   // wave-uniform: a full tile
#pragma unroll
                for (int it = 0; it < 16; ++it)
                    *reinterpret_cast<Pack *>(row + (size_t)(it * 4) * span_total) = pk[it];
            } else {
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    if (base_local + it * 4 + q < count)
                        *reinterpret_cast<Pack *>(row + (size_t)(it * 4) * span_total) = pk[it];
                }
            }
        } else {
            for (int j = 0; j < 64; ++j) {
                if (base_local + j < count) {
                    for (int r = lane; r < CH; r += 64)
                        o[(size_t)(base_local + j) * span_total + s0 + r] = tile[r * TT::stride + j];
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- Delay lines tiled by absolute time (effects.py:136-174, ring_buffer.py:62-109) -----------
// tiles[(t / TS) % T][voice][t % TS] (TS = kDelayTile samples): what a voice wrote at time t.  The delayed
// sample of time t is the one written at t - length (zero before the line's start: the tiles
// start as zeros and a voice's column is cleared when its length changes), so a voice needs
// no position in memory.  A wavefront's eight new samples are ONE contiguous 4 KiB piece of a
// tile row; its eight delayed samples lie in two neighbouring tile rows, contiguous across the
// lanes whenever the voices' delay times agree.  T >= longest / TS + 3: the tile being written
// is never one being read.
#ifndef KLMI_DELAY_TILE
#define KLMI_DELAY_TILE 8
#endif
constexpr int kDelayTile = KLMI_DELAY_TILE;   // samples of one voice per time tile (8 or 16)
// Which tiles are neighbours in memory.  Wave-major (1; measured slower, kept as a build option): the T tiles of one wavefront's
// 64 voices are one contiguous region, tiles[voice / 64][(t / TS) % T][voice % 64][TS] -- a
// wavefront whose voices have 64 different delay times still reads inside its own T * 4 KiB
// (one or two pages of the address translation, a few DRAM rows) instead of 64 tile rows that lie
// n * 64 bytes apart.  0 (the default): tiles[(t / TS) % T][voice][TS], a tile row over all voices.
#ifndef KLMI_DELAY_WAVE_MAJOR
#define KLMI_DELAY_WAVE_MAJOR 0
#endif
// first element of tile t (already reduced modulo T) of voice v, and the allocation's length
__host__ __device__ __forceinline__ size_t delay_tile_at(size_t n, int T, int t, int v)
{
    if (KLMI_DELAY_WAVE_MAJOR)
        return ((((size_t)v >> 6) * (size_t)T + (size_t)t) * 64 + ((size_t)v & 63)) * kDelayTile;
    return ((size_t)t * n + (size_t)v) * kDelayTile;
}
__host__ __device__ __forceinline__ size_t delay_tiles_len(size_t n, int T)
{
    return (KLMI_DELAY_WAVE_MAJOR ? (n + 63) / 64 * 64 : n) * (size_t)T * kDelayTile;
}
static_assert(kDelayTile == 8 || kDelayTile == 16, "a group of 8 samples is a tile or half of one");

// The eight samples written `length` samples before the group that starts at sample `s8` * 8
// of tile `g` (tile index already reduced modulo T) of voice v; length / kDelayTile < T - 2.
__device__ __forceinline__ void delay_read8(const double *tiles, size_t n, int T, int g, int s8,
                                            int length, int v, double (&old)[8])
{
    constexpr int TS = kDelayTile;
    // first delayed sample: TS * (g - q) + 8 * s8 - r, with length = TS * q + r
    const int q = length / TS, r = length % TS;
    int off = 8 * s8 - r;
    int t0 = g - q;
    if (off < 0)
        off += TS, t0 -= 1;
    t0 = t0 < 0 ? t0 + T : t0;
    int t1 = t0 + 1;
    t1 = t1 >= T ? t1 - T : t1;
    const double *p0 = tiles + delay_tile_at(n, T, t0, v) + off;         // elements j < TS - off
    const double *p1 = tiles + delay_tile_at(n, T, t1, v) + off - TS;    // the others
    const int split = TS - off;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        old[j] = (j < split ? p0 : p1)[j];
}

__device__ __forceinline__ void delay_write8(double *tiles, size_t n, int T, int g, int s8, int v,
                                             const double (&w)[8])
{
    double2 *dst = reinterpret_cast<double2 *>(tiles + delay_tile_at(n, T, g, v) + 8 * s8);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        dst[j] = make_double2(w[2 * j], w[2 * j + 1]);
}

// a whole tile of 16 samples (kDelayTile == 16): the group before this one (`lo`) and this one
// (`hi`) as ONE 128-byte piece per voice -- a full cache line, never half of one
__device__ __forceinline__ void delay_write16(double *tiles, size_t n, int T, int g, int v,
                                              const double (&lo)[8], const double (&hi)[8])
{
    double2 *dst = reinterpret_cast<double2 *>(tiles + delay_tile_at(n, T, g, v));
#pragma unroll
    for (int j = 0; j < 4; ++j)
        dst[j] = make_double2(lo[2 * j], lo[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        dst[4 + j] = make_double2(hi[2 * j], hi[2 * j + 1]);
}

// The same eight samples from a window the lane carries in registers: `prev` and `cur` are the
// voice's time groups a and a + 1 (eight samples each), a = this group - length / 8 - 1; the delayed samples are
// elements 8 - length % 8 .. of the two, i.e. window[1 + k + j] with k = 7 - length % 8.  A lane
// then reads every tile of its line exactly once, as one aligned 64-byte piece, whatever its
// delay time; read in two unaligned parts (delay_read8), each tile is fetched in two consecutive
// groups, and with 65,536 different times the second fetch misses the cache (measured: twice
// the algorithmic read traffic).  The shift by k is three rounds of selects.
// (written with constant indices throughout -- index packs, no loops: a loop's variable index
// lets the optimiser turn "c ? w[i + 4] : w[i]" into one load at a selected ADDRESS, which
// puts the window into scratch memory)
template <int BY, int... I>
__device__ __forceinline__ void delay_shift_round(const double *in, double *out, bool c,
                                                  std::integer_sequence<int, I...>)
{
    ((out[I] = c ? in[I + BY] : in[I]), ...);
}
template <int... I>
__device__ __forceinline__ void delay_window_fill(const double (&prev)[8], const double (&cur)[8],
                                                  double *w0, std::integer_sequence<int, I...>)
{
    ((w0[I] = I < 7 ? prev[(I + 1) & 7] : cur[(I - 7) & 7]), ...);
}
__device__ __forceinline__ void delay_window8(const double (&prev)[8], const double (&cur)[8], int k,
                                              double (&old)[8])
{
    double w0[15], w1[11], w2[9];
    delay_window_fill(prev, cur, w0, std::make_integer_sequence<int, 15>{});
    delay_shift_round<4>(w0, w1, (k & 4) != 0, std::make_integer_sequence<int, 11>{});
    delay_shift_round<2>(w1, w2, (k & 2) != 0, std::make_integer_sequence<int, 9>{});
    delay_shift_round<1>(w2, old, (k & 1) != 0, std::make_integer_sequence<int, 8>{});
}

// the eight samples of time group `grp` (absolute sample time / 8, reduced modulo T * TS / 8) of
// voice v: four 16-byte loads of an aligned 64-byte piece -- a tile, or half of a 16-sample one
__device__ __forceinline__ void delay_load_group8(const double *tiles, size_t n, int T, int grp, int v,
                                                  double (&d)[8])
{
    constexpr int GPT = kDelayTile / 8;
    const double2 *src = reinterpret_cast<const double2 *>(tiles + delay_tile_at(n, T, grp / GPT, v) +
                                                           8 * (grp % GPT));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double2 p = src[j];
        d[2 * j] = p.x, d[2 * j + 1] = p.y;
    }
}

// tile index of the first sample of absolute block `clock`: (clock * tiles per block) % T
__device__ __forceinline__ int delay_tile_of_block(uint32_t clock, int T)
{
    return (int)(((unsigned long long)clock * (kBlock / kDelayTile)) % (unsigned)T);
}

// Lfo(shape=1) | Filter.frequency, once per block (oscillators.py:175-181,
// effects.py:271-274, 365-373): value = scale * wave(phase) + offset; the phasor steps
// by TAU*f*INTERVAL; row = searchsorted(grid, value) clipped to the last row.
__device__ __forceinline__ void lfo_cutoff_step(const BankView &bk, int v, double &lfo_ph,
                                                int &coef_row, Biquad &bq)
{
    const size_t n = (size_t)bk.n;
    const double f = bk.lfo_par[v], scale = bk.lfo_par[n + v], offset = bk.lfo_par[2 * n + v];
    const double value = scale * wave_eval_any(bk.lfo_wave[v], lfo_ph, kPi) + offset;
    lfo_ph = np_mod_tau(kTau * f * kInterval + lfo_ph);
    int lo = 0, hi = kGridPoints;  // first row whose frequency is >= value
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (bk.grid[mid] < value)
            lo = mid + 1;
        else
            hi = mid;
    }
    const int row = lo < kGridPoints - 1 ? lo : kGridPoints - 1;
    if (row != coef_row) {
        coef_row = row;
        const double *c = bk.coef_table + 5 * (size_t)row;
        bq.b0 = c[0], bq.b1 = c[1], bq.b2 = c[2], bq.a1 = c[3], bq.a2 = c[4];
    }
}

// grid.x = ceil(count / 256) workgroups of 4 waves.
// partial: [gridDim.x * 4 waves][n_blocks*planes*256] float32 (mix modes)
// out:     voices / bus buffer (other modes)
// GATED (synthesizer banks, synthesizer.py:151-219): a voice is only updated in a buffer in
// which it is active or receives a note; polyphonic banks route their instrument's notes here
// (round robin), exactly as voice_mod_kernel does for the chains this kernel does not cover.
// FM: the oscillator is an FmOscillator (oscillators.py:193-234): its own Oscillator(modFrequency)
// sampled at the same time points, and the wave function evaluated at phase + intensity *
// modulator -- a phase of any sign.  The eight-sample region speculates one more remainder (of
// that sum) under the same single check; wavefronts whose carrier and modulator do not share
// one of sine / square / sawtooth / triangle (the reference builds both from one wave_func)
// take the exact per-sample form.
// MOD = MT_FREQUENCY / MT_DUTY: an Lfo(shape=256) patched into Oscillator.frequency (the
// np.cumsum branch of sample_phase, oscillators.py:61-64: phase[n] = (TAU*DT) * sum(f[0..n-1])
// + start) or PwmOscillator.dutyCycle (oscillators.py:256-260).  The Lfo's eight values are a
// region of their own in front of the straight-line region (independent samples, the wave
// function chosen by one scalar jump, the remainder repaired in place); the running frequency
// sum is one more serial recurrence -- a single add per sample -- beside envelope and biquad.
// OCC: waves per SIMD the register budget allows.  2 (256 registers per lane) lets banks above
// 65,536 voices run two workgroups per compute unit side by side (131,072 voices: 1.01e12 ->
// 1.24e12 voice-samples/s); with ONE workgroup per compute unit the unconstrained allocation
// (256 + 18 registers) is 2 % faster, so plain chains are also built with OCC = 1
// (voice1_tu.hip) for grids that do not exceed the compute units.
template <bool ENV, bool FILT, int MODE, bool GATED = false, int MOD = MT_NONE, int OCC = 2>
__global__ __launch_bounds__(kGroup, OCC) void voice_kernel(BankView bk, int first, int count,
                                                       int block_base,      // first block, relative to this call
                                                       uint32_t clock_base,  // absolute number of block_base (events)
                                                       int n_blocks,
                                                       int span_total,  // samples per voice row in `out`
                                                       float *__restrict__ partial,
                                                       void *__restrict__ out)
{
    using TT = TileTraits<MODE>;
    using T = typename TT::T;
    constexpr int CH = TT::chunk;
    constexpr int NCHUNK = kBlock / CH;
    constexpr int PLANES = TT::planes;
    constexpr bool MIX = (MODE == OUT_MIX_MONO || MODE == OUT_MIX_STEREO);
    constexpr bool FM = MOD == MT_FM, VIB = MOD == MT_FREQUENCY, PWMOD = MOD == MT_DUTY;
    constexpr bool LFO = VIB || PWMOD;   // the modulator is an audio-rate Lfo with scale / offset

    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // wave-private tile(s): [PLANES][CH][stride]
    T *tile = reinterpret_cast<T *>(smem_raw) + (size_t)wave * (PLANES ? PLANES : 1) * CH * TT::stride;


    const int local = blockIdx.x * kGroup + threadIdx.x;
    const bool live = local < count;
    const int v = first + (live ? local : 0);

    TauConsts tc = {kInvTau, kTau, 2. * kInvTau, kPi};   // opaque to the optimiser: dsp_device.h
    asm volatile("" : "+s"(tc.inv_tau), "+s"(tc.tau), "+s"(tc.inv_pi), "+s"(tc.pi));

    // ---- load parameters and state ------------------------------------------------
    double freq = bk.frequency_rw[v];
    double amp = bk.amplitude_rw[v];
    const int wave_id = bk.wave[v];
    const double duty_edge = kTau * bk.duty[v];  // oscillators.py:259
    // wave function shared by the whole wavefront, or -1
    const int first_wave = __builtin_amdgcn_readfirstlane(wave_id);
    const int uniform_wave = __all(wave_id == first_wave) ? first_wave : -1;
    double phase = bk.phase[v];
    static_assert(!(GATED && (MOD == MT_FREQUENCY || MOD == MT_DUTY)),
                  "gated chains with an Lfo on frequency / duty cycle run in voice_mod_kernel");
    double mod_freq = 0., mod_scale = 0., mod_offset = 0., mod_phase = 0.;
    int mod_wave = 0;
    bool fm_paired = false;  // wave-uniform: carrier and modulator share a plain wave function
    int uniform_mod_wave = -1;  // the Lfo's wave function when the whole wavefront agrees on it
    if (MOD != MT_NONE) {
        mod_freq = bk.mod_par[v], mod_scale = bk.mod_par[(size_t)bk.n + v];
        mod_offset = bk.mod_par[2 * (size_t)bk.n + v];
        mod_wave = bk.mod_wave[v];
        mod_phase = bk.mod_phase[v];
        fm_paired = uniform_wave >= W_SINE && uniform_wave <= W_TRIANGLE && __all(mod_wave == wave_id);
        const int first_mod_wave = __builtin_amdgcn_readfirstlane(mod_wave);
        uniform_mod_wave = __all(mod_wave == first_mod_wave) ? first_mod_wave : -1;
    }

    EnvParams ep;
    EnvState es;
    if (ENV) {
        ep.aCoef = bk.env_coef[0 * (size_t)bk.n + v];
        ep.aBase = bk.env_coef[1 * (size_t)bk.n + v];
        ep.dCoef = bk.env_coef[2 * (size_t)bk.n + v];
        ep.dBase = bk.env_coef[3 * (size_t)bk.n + v];
        ep.rCoef = bk.env_coef[4 * (size_t)bk.n + v];
        ep.rBase = bk.env_coef[5 * (size_t)bk.n + v];
        ep.sustain = bk.env_coef[6 * (size_t)bk.n + v];
        const uint32_t fl = bk.env_flags[v];
        ep.loop = fl & 1u;
        ep.retrigger = fl & 2u;
        const uint32_t st = bk.env_state[v];
        es.stage = (int)(st & 7u);
        es.enabled = (st >> 8) & 1u;
        es.value = bk.env_value[v];
        es.select(ep);
    }

    Biquad bq;
    int coef_row = 0;
    if (FILT) {
        coef_row = bk.coef_index_rw[v];
        const double *c = bk.coef_table + 5 * (size_t)coef_row;
        bq.b0 = c[0], bq.b1 = c[1], bq.b2 = c[2], bq.a1 = c[3], bq.a2 = c[4];
        bq.z0 = bk.z0[v];
        bq.z1 = bk.z1[v];
    }

    const bool cutoff_lfo = FILT && bk.lfo_par != nullptr;  // uniform over the launch
    double lfo_ph = cutoff_lfo ? bk.lfo_phase[v] : 0.;

    double gl = 0., gr = 0.;
    bool unit_gain = false;  // wave-uniform: every live lane's gain is exactly 1 (plain Mixer)
    if (MIX) {
        gl = live ? bk.gain_l[v] : 0.;
        if (MODE == OUT_MIX_STEREO)
            gr = live ? bk.gain_r[v] : 0.;
        else
            unit_gain = __all(gl == 1.);
    }

    // ---- event cursor -------------------------------------------------------------
    uint32_t ev_i = bk.ev_offset[v] + bk.ev_cursor[v];
    const uint32_t ev_end = bk.ev_offset[v + 1];
    uint32_t next_ev = (live && ev_i < ev_end) ? bk.ev_block[ev_i] : 0xFFFFFFFFu;

    // polyphonic banks: the voice's place in its instrument and the instrument's notes
    const int poly = GATED ? bk.poly : 0;
    int slot = 0, rr = 0, cur_pitch = 0;
    uint32_t nt_i = 0, nt_end = 0, next_nt = 0xFFFFFFFFu;
    if (GATED && poly > 0) {
        const int inst = v / poly;
        slot = v - inst * poly;
        rr = (int)bk.round_robin[v];
        cur_pitch = bk.current_pitch[v];
        nt_i = bk.nt_offset[inst] + bk.nt_cursor[v];
        nt_end = bk.nt_offset[inst + 1];
        next_nt = (live && nt_i < nt_end) ? bk.nt_block[nt_i] : 0xFFFFFFFFu;
    }

    for (int b = 0; b < n_blocks; ++b) {
        const uint32_t abs_block = clock_base + (uint32_t)b;
        // Voice.active (voices.py:45-53) is asked before the notes are processed
        const bool was_active = !ENV || es.stage != 0;
        bool had_note = false;
        // voices.py:55-64: notes are consumed at the block boundary
        while (next_ev <= abs_block) {
            const uint32_t kind = bk.ev_kind[ev_i];
            if (kind == 1u) {
                if (ENV)
                    es.gate(ep, true);
                amp = bk.ev_velocity[ev_i];
                freq = bk.ev_frequency[ev_i];
                had_note = true;
            } else if (kind == 0u) {
                if (ENV)
                    es.gate(ep, false);
                had_note = true;
            } else if (kind == 2u) {
                if (FILT) {
                    coef_row = bk.ev_index[ev_i];
                    const double *c = bk.coef_table + 5 * (size_t)coef_row;
                    bq.b0 = c[0], bq.b1 = c[1], bq.b2 = c[2], bq.a1 = c[3], bq.a2 = c[4];
                }
            } else {
                freq = bk.ev_frequency[ev_i];
            }
            ++ev_i;
            next_ev = (ev_i < ev_end) ? bk.ev_block[ev_i] : 0xFFFFFFFFu;
        }
        if (GATED && poly > 0) {
            // PolyphonicSynthesizer.process_note (synthesizer.py:200-209): every voice reads
            // its instrument's notes of this buffer and picks out its own; note-offs match the
            // pitch the voice's LAST update left (see voice_mod_kernel)
            const int start_pitch = cur_pitch;
            while (next_nt <= abs_block) {
                const double velocity = bk.nt_velocity[nt_i];
                const int pitch = bk.nt_pitch[nt_i];
                if (velocity > 0.) {
                    if (rr == slot) {
                        if (ENV)
                            es.gate(ep, true);
                        amp = velocity;
                        freq = bk.nt_frequency[nt_i];
                        cur_pitch = pitch;
                        had_note = true;
                    }
                    rr = rr + 1 == poly ? 0 : rr + 1;
                } else if (start_pitch == pitch) {
                    if (ENV)
                        es.gate(ep, false);
                    cur_pitch = 0;
                    had_note = true;
                }
                ++nt_i;
                next_nt = (nt_i < nt_end) ? bk.nt_block[nt_i] : 0xFFFFFFFFu;
            }
        }
        // synthesizer.py:177-179, 214-217: an idle voice is not updated: its envelope is OFF
        // (samples exactly 0, state unchanged) and its oscillator stays where it is
        const bool run = !GATED || had_note || was_active;
        if (GATED && !__any(run)) {
            // nobody in this wavefront sounds: a buffer of zeros, no arithmetic
            for (int c = 0; c < NCHUNK; ++c) {
                const int s0 = b * kBlock + c * CH;
                for (int i = 0; i < CH; ++i)
                    emit_sample<MODE, false>(tile, i, lane, 0., gl, gr, live, out, s0, count, local);
                flush_chunk<MODE>(tile, lane, wave, b, c, n_blocks, count, span_total, s0, partial, out);
            }
            continue;
        }

        if (cutoff_lfo)
            lfo_cutoff_step(bk, v, lfo_ph, coef_row, bq);

        const double omega = kTau * freq;  // oscillators.py:60  (TAU * frequency) * t
        const double start = phase;
        // wave-uniform: may this block use the fast remainder?
        bool fast_block =
            __all(omega >= 0. && start >= 0. && omega * c_time[kBlock] + start < 1e9);
        const double mod_omega = kTau * mod_freq;
        const double mod_start = mod_phase;
        if (FM)  // ... and the modulator's, and |phase + intensity * modulator| < 1e9
            fast_block = fast_block && fm_paired &&
                         __all(mod_omega >= 0. && mod_start >= 0. &&
                               mod_omega * c_time[kBlock] + mod_start < 1e9 && fabs(mod_scale) < 1e8);
        if (VIB)  // the phase is a running sum of Lfo values: bounded by the Lfo's range
            fast_block = __all(start >= 0. && (kTau * kDt) * kBlock * (fabs(mod_scale) + fabs(mod_offset)) +
                                                      start < 1e9);
        // the Lfo's own phase: non-negative and bounded, one plain wave function per wavefront
        const bool fast_mod = LFO && uniform_mod_wave >= W_SINE && uniform_mod_wave <= W_TRIANGLE &&
                              __all(mod_omega >= 0. && mod_start >= 0. &&
                                    mod_omega * c_time[kBlock] + mod_start < 1e9);
        double running = 0.;   // np.cumsum(frequency) within this block (VIB)
        // Lfo.sample for eight time points (oscillators.py:175-190): scale * wave(phase) + offset
        auto lfo_group = [&](const Time8 &tt, double (&m)[8]) {
            double marg[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                marg[j] = mod_omega * tt.t[j] + mod_start;
            if (fast_mod) {
                double mp[8];
                uint32_t hi_max = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double k = floor(marg[j] * kInvTau);
                    mp[j] = __builtin_fma(-k, kTau, marg[j]);
                    hi_max = max(hi_max, (uint32_t)__double2hiint(mp[j]));
                }
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(hi_max >= kTauHighWord) != 0, 0)) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        mp[j] = fmod_tau_fast(marg[j]);
                }
                switch (uniform_mod_wave) {
                case W_SINE:
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        m[j] = wave_eval<true>(W_SINE, mp[j], kPi);
                    break;
                case W_SQUARE:
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        m[j] = wave_eval<true>(W_SQUARE, mp[j], kPi);
                    break;
                case W_SAWTOOTH:
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        m[j] = wave_eval<true>(W_SAWTOOTH, mp[j], kPi);
                    break;
                default:
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        m[j] = wave_eval<true>(W_TRIANGLE, mp[j], kPi);
                    break;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    m[j] = wave_eval<false>(mod_wave, np_mod_tau(marg[j]), kPi);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                m[j] = mod_scale * m[j] + mod_offset;
        };
        // FmOscillator.sample, exactly, for one time point (oscillators.py:215-220)
        auto fm_sample = [&](int w, double t) {
            const double mp = np_mod_tau(mod_omega * t + mod_start);
            const double m = wave_eval<false>(mod_wave, mp, kPi);
            const double p = np_mod_tau(omega * t + start);
            return wave_eval_any<false>(w, p + mod_scale * m, duty_edge);
        };

        for (int c = 0; c < NCHUNK; ++c) {
            const int s0 = b * kBlock + c * CH;  // sample offset inside this launch
            // One chunk of CH samples.  WK is the wave function when the whole
            // wavefront agrees on it (compile-time dispatch), or -1 (per-lane switch).
            // FAST: every lane's phase argument of this block is >= 0 and < 1e9, so
            // the remainder needs no sign handling and no library fallback.
            auto run_chunk = [&](auto wk, auto fast, auto unit) {
                constexpr int WK = decltype(wk)::value;
                constexpr bool FAST = decltype(fast)::value;
                constexpr bool UNIT = decltype(unit)::value;  // gains are 1: 1. * x == x
                // 8 time points per scalar load (wave-uniform index).  Fetching them a group
                // ahead was measured slower (more scalar registers live across the group).
                const Time8 *times = reinterpret_cast<const Time8 *>(c_time) + ((c * CH) >> 3);
                auto group8 = [&](int i8, const Time8 &tt) {
                    constexpr int G = 8;
                    const int wsel = (WK >= 0) ? WK : wave_id;
                    double m[G];   // the Lfo's values (LFO)
                    if constexpr (LFO)
                        lfo_group(tt, m);
                    const bool first_group = c == 0 && i8 == 0;
                    // Eight samples as ONE straight-line region: oscillator (8 independent
                    // samples), envelope and biquad recurrences (serial chains) are all
                    // computed speculatively so that the scheduler can interleave the chains
                    // with the independent work.  Speculation: "no lane needs the remainder
                    // repair and no lane's envelope switches stage in these 8 samples".  The
                    // lane masks of the checks are OR-ed on the scalar unit; on a hit the whole
                    // wave redoes the 8 samples with the exact per-sample code from the state
                    // at the start of the group (nothing was committed).
                    unsigned long long redo = 0;
                    double arg[G], osc[G], y[G];
                    const bool has_phase = (wsel != W_EXTERNAL && wsel != W_ONE);

                    // -- oscillator --------------------------------------------------------
                    if (wsel == W_EXTERNAL) {
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            osc[j] = bk.ext_input[(size_t)(block_base * kBlock + s0 + i8 + j) * bk.n + v];
                        // Waited for on the spot (vmcnt 0).  Left pending, these eight loads reach
                        // the chunk loop's head as "maybe in flight" for the registers that every
                        // other wave-function variant computes its samples into, and the compiler
                        // then guards each of the eight samples of EVERY variant's straight-line
                        // region with an s_waitcnt vmcnt(7..0): eight no-op waits per group, each
                        // an issue slot of the lone wave of a 65,536-voice bank.
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                    } else if (wsel == W_ONE) {
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            osc[j] = 1.;
                    } else {
                        // 0 <= ph < TAU, conservatively, on the high words only: non-negative
                        // doubles order like their bit patterns and a set sign bit compares
                        // high, so one unsigned maximum over the 8 high words and ONE compare
                        // replace 8 compares (a false alarm just takes the exact path)
                        uint32_t hi_max = 0;
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            arg[j] = omega * tt.t[j] + start;  // oscillators.py:60
                            if constexpr (VIB) {
                                // oscillators.py:61-64: phase[0] = start, phase[n] = (TAU*DT) *
                                // cumsum(f)[n-1] + start; np.cumsum is a sequential float64 sum
                                const bool first = first_group && j == 0;
                                arg[j] = first ? start : (kTau * kDt) * running + start;
                                running = first ? m[j] : running + m[j];
                            }
                            if constexpr (PWMOD && FAST) {          // oscillators.py:259-260
                                const double k = floor(arg[j] * kInvTau);
                                const double ph = __builtin_fma(-k, kTau, arg[j]);
                                hi_max = max(hi_max, (uint32_t)__double2hiint(ph));
                                osc[j] = (ph < kTau * m[j]) ? 1. : -1.;
                            } else if constexpr (PWMOD) {
                                osc[j] = (np_mod_tau(arg[j]) < kTau * m[j]) ? 1. : -1.;
                            } else if constexpr (FM && FAST) {
                                // modulator: Oscillator.sample of the same wave function
                                const double marg = mod_omega * tt.t[j] + mod_start;
                                const double mk = floor(marg * kInvTau);
                                const double mp = __builtin_fma(-mk, kTau, marg);
                                const double m = wave_eval<true>(wsel, mp, kPi);
                                const double k = floor(arg[j] * kInvTau);
                                const double ph = __builtin_fma(-k, kTau, arg[j]);
                                // oscillators.py:220: wave(phase + intensity * modulator); the
                                // remainder of that sum (sine, sawtooth) or of the sum + PI
                                // (square, triangle: math.py:9-11) is x - floor(x/TAU) TAU for
                                // either sign, as NumPy's fmod-then-sign-fix yields it
                                const double q = ph + mod_scale * m;
                                const double a = (wsel == W_SQUARE || wsel == W_TRIANGLE) ? q + kPi : q;
                                const double qk = floor(a * kInvTau);
                                const double r = __builtin_fma(-qk, kTau, a);
                                hi_max = max(hi_max, max(max((uint32_t)__double2hiint(mp),
                                                             (uint32_t)__double2hiint(ph)),
                                                         (uint32_t)__double2hiint(r)));
                                if (wsel == W_SINE) {
                                    osc[j] = sin_wrapped(r);
                                } else if (wsel == W_SQUARE) {
                                    osc[j] = (r - kPi >= 0.) ? 1. : -1.;
                                } else if (wsel == W_SAWTOOTH) {
                                    osc[j] = div_pi(r) - 1.;
                                } else {
                                    double t4 = div_tau(4. * (r - kPi));
                                    t4 = t4 < 0. ? t4 + 4. : t4;
                                    osc[j] = 1. - fabs(t4 - 2.);
                                }
                            } else if constexpr (FM) {
                                osc[j] = fm_sample(wsel, tt.t[j]);
                            } else if constexpr (FAST) {
                                const double k = floor(arg[j] * tc.inv_tau);
                                const double ph = __builtin_fma(-k, tc.tau, arg[j]);
                                hi_max = max(hi_max, (uint32_t)__double2hiint(ph));
                                osc[j] = wave_eval_wrapped(wsel, ph, duty_edge, tc);
                            } else {
                                osc[j] = wave_eval<false>(wsel, np_mod_tau(arg[j]), duty_edge);
                            }
                        }
                        if constexpr (FAST)
                            redo |= __builtin_amdgcn_ballot_w64(hi_max >= kTauHighWord);
                    }

                    // -- envelope: v -> fl(base + fl(v*coef)) is monotone in v, so the 8
                    // values move one way: if any crossed the stage's limit, the last did
                    double vv = 0.;
                    if (ENV) {
                        vv = es.value;
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            vv = es.base + vv * es.coef;           // _envelope.c:579-624
                            y[j] = amp * osc[j] * vv;              // voices.py:73 (amplitude*osc)*env
                        }
                        redo |= __builtin_amdgcn_ballot_w64(es.reached(vv));
                    } else {
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            y[j] = osc[j];
                    }

                    // -- biquad (serial chain) on a copy of the state ------------------------
                    Biquad sp = bq;
                    if (FILT) {
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            y[j] = sp.step(y[j]);
                    }

                    if (__builtin_expect(redo != 0, 0)) {
                        // exact path, from the uncommitted state
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            double x = osc[j];
                            if constexpr (FM && FAST) {
                                x = fm_sample(wsel, tt.t[j]);
                            } else if constexpr (VIB && FAST) {   // a negative phase argument lands here
                                if (has_phase)
                                    x = wave_eval<false>(wsel, np_mod_tau(arg[j]), duty_edge);
                            } else if constexpr (PWMOD && FAST) {
                                if (has_phase)
                                    x = (fmod_tau_fast(arg[j]) < kTau * m[j]) ? 1. : -1.;
                            } else if constexpr (FAST) {
                                if (has_phase)
                                    x = wave_eval<true>(wsel, fmod_tau_fast(arg[j]), duty_edge);
                            }
                            if (ENV)
                                x = amp * x * es.step(ep);
                            if (FILT)
                                x = bq.step(x);
                            y[j] = x;
                        }
                    } else {
                        if (ENV)
                            es.value = vv;
                        if (FILT)
                            bq.z0 = sp.z0, bq.z1 = sp.z1;
                    }

                    // -- output ----------------------------------------------------------------
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        emit_sample<MODE, UNIT>(tile, i8 + j, lane, y[j], gl, gr, live, out, s0, count,
                                                local);
                };
                for (int i8 = 0; i8 < CH; i8 += 8)
                    group8(i8, times[i8 >> 3]);
            };
            using std::integral_constant;
            using Fast = std::true_type;
            auto dispatch = [&](auto unit) {
                if (fast_block) {
                    switch (uniform_wave) {
                    case W_SINE: run_chunk(integral_constant<int, W_SINE>{}, Fast{}, unit); break;
                    case W_SQUARE: run_chunk(integral_constant<int, W_SQUARE>{}, Fast{}, unit); break;
                    case W_SAWTOOTH: run_chunk(integral_constant<int, W_SAWTOOTH>{}, Fast{}, unit); break;
                    case W_TRIANGLE: run_chunk(integral_constant<int, W_TRIANGLE>{}, Fast{}, unit); break;
                    default:   // (FM: fast_block implies one of the four above)
                        if constexpr (!FM) {
                            if (uniform_wave == W_PWM)
                                run_chunk(integral_constant<int, W_PWM>{}, Fast{}, unit);
                            else
                                run_chunk(integral_constant<int, -1>{}, Fast{}, unit);
                        }
                        break;
                    }
                } else {
                    run_chunk(integral_constant<int, -1>{}, std::false_type{}, unit);
                }
            };
            if constexpr (MODE == OUT_MIX_MONO && MOD == MT_NONE) {
                if (unit_gain)
                    dispatch(std::true_type{});
                else
                    dispatch(std::false_type{});
            } else {
                dispatch(std::false_type{});
            }
            flush_chunk<MODE>(tile, lane, wave, b, c, n_blocks, count, span_total, s0, partial, out);
        }
        // oscillators.py:66-67: next start phase is sample 256 of this block
        if (wave_id != W_EXTERNAL && wave_id != W_ONE)
            phase = VIB ? np_mod_tau((kTau * kDt) * running + start)
                        : np_mod_tau(omega * c_time[kBlock] + start);
        if (MOD != MT_NONE)
            mod_phase = np_mod_tau(mod_omega * c_time[kBlock] + mod_start);
        if (GATED && !run) {     // an idle voice's oscillator (and its modulator) stay where they are
            phase = start;
            if (MOD != MT_NONE)
                mod_phase = mod_start;
        }
    }

    // ---- store state ----------------------------------------------------------------
    if (live) {
        bk.phase[v] = phase;
        if (MOD != MT_NONE)
            bk.mod_phase[v] = mod_phase;
        bk.frequency_rw[v] = freq;
        bk.amplitude_rw[v] = amp;
        if (ENV) {
            bk.env_value[v] = es.value;
            bk.env_state[v] = (uint32_t)es.stage | (es.enabled ? 0x100u : 0u);
        }
        if (FILT) {
            bk.z0[v] = bq.z0;
            bk.z1[v] = bq.z1;
            bk.coef_index_rw[v] = coef_row;
            if (cutoff_lfo)
                bk.lfo_phase[v] = lfo_ph;
        }
        bk.ev_cursor[v] = ev_i - bk.ev_offset[v];
        if (GATED && poly > 0) {
            bk.nt_cursor[v] = nt_i - bk.nt_offset[v / poly];
            bk.round_robin[v] = (uint32_t)rr;
            bk.current_pitch[v] = cur_pitch;
        }
    }
}

}  // namespace klmi
