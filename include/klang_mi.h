/*
 * klang_mi.h -- C-ABI of libklang_mi.so, the MI355X (gfx950) evaluator for
 * klang's per-block DSP hot path.
 *
 * The reference (atheler/klang) has no FFI registry; its own native swap point
 * is the pair of CPython extension types klang.audio._envelope.Envelope and
 * klang.audio._filters.{Forward,Backward}CombFilter/EchoFilter, selected by
 * try/except import (klang/audio/envelopes.py:4-9, klang/audio/filters.py:19-31,
 * 100-108), and the Block.update() protocol (klang/block.py:159-238).  This
 * header is the batched restatement of that seam: every entry point cites the
 * reference interface it replaces.  INTEGRATION.md shows the ctypes stubs a
 * klang maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative klmi_status otherwise;
 *     klmi_last_error() returns a thread-local message for the last failure.
 *   - one klmi_ctx == one GPU (one process per GPU; shard voices across
 *     processes and reduce the mix bus with RCCL, see klang_amd/sharding.py).
 *   - the caller owns host buffers; the library owns device memory.  Pointers
 *     named dev_* are device pointers the caller owns (e.g. a torch tensor).
 *   - not re-entrant per ctx (the reference is single-threaded per graph);
 *     callable from any host thread (each call sets the device).
 *   - HIP only.  There is no CPU backend and no fallback: without a GPU
 *     klmi_ctx_create fails with KLMI_E_DEVICE.
 *   - all arithmetic is float64 in the reference's operation order (phase,
 *     envelope and IIR recurrences are bit-identical to the reference's
 *     float64 results); mix/voice outputs are rounded to float32 once.
 */
#ifndef KLANG_MI_H
#define KLANG_MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KLMI_BUFFER_SIZE 256      /* klang/config.py:5  BUFFER_SIZE */
#define KLMI_SAMPLING_RATE 44100  /* klang/config.py:11 SAMPLING_RATE */
#define KLMI_GRID_POINTS 1000     /* klang/audio/effects.py:260-265 */

typedef enum {
    KLMI_OK = 0,
    KLMI_E_ARG = -1,     /* bad argument (ValueError in the reference's ext types) */
    KLMI_E_DEVICE = -2,  /* HIP runtime error / no GPU */
    KLMI_E_MEMORY = -3,  /* allocation failure (RuntimeError in _filters.c:124-136) */
    KLMI_E_STATE = -4    /* call not valid in the object's current state */
} klmi_status;

/* Wave functions: klang/audio/waves.py:22-39, PwmOscillator oscillators.py:256-260. */
typedef enum {
    KLMI_WAVE_SINE = 0,
    KLMI_WAVE_SQUARE = 1,
    KLMI_WAVE_SAWTOOTH = 2,
    KLMI_WAVE_TRIANGLE = 3,
    KLMI_WAVE_PWM = 4,
    KLMI_WAVE_ONE = 5,      /* no oscillator: constant 1.0 (bare envelope blocks) */
    KLMI_WAVE_EXTERNAL = 6  /* samples supplied with klmi_bank_set_input (Filter/Reverb
                               blocks fed by something outside the bank) */
} klmi_wave;

/* Envelope stages: klang/audio/_envelope.c:59-65. */
typedef enum {
    KLMI_STAGE_OFF = 0,
    KLMI_STAGE_ATTACKING = 1,
    KLMI_STAGE_DECAYING = 2,
    KLMI_STAGE_SUSTAINING = 3,
    KLMI_STAGE_RELEASING = 4
} klmi_stage;

/* Per-voice parameter / state fields for klmi_bank_set_params / _state_get / _state_set.
 * All travel as float64 (integers and flags converted). */
typedef enum {
    /* parameters */
    KLMI_F_FREQUENCY = 0,   /* Phasor.frequency in Hz (oscillators.py:81-89) */
    KLMI_F_AMPLITUDE = 1,   /* Voice.amplitude (voices.py:41,60) */
    KLMI_F_WAVE = 2,        /* klmi_wave */
    KLMI_F_DUTY = 3,        /* PwmOscillator.dutyCycle (oscillators.py:245-254) */
    KLMI_F_ATTACK = 4,      /* seconds; _envelope.c:220-237 (negative -> KLMI_E_ARG) */
    KLMI_F_DECAY = 5,       /* _envelope.c:255-272 */
    KLMI_F_SUSTAIN = 6,     /* [0,1]; _envelope.c:290-307 */
    KLMI_F_RELEASE = 7,     /* _envelope.c:325-342 */
    KLMI_F_OVERSHOOT = 8,   /* clipped to [1e-9,1e9]; _envelope.c:360-377 */
    KLMI_F_LOOP = 9,        /* _envelope.c:173-179 */
    KLMI_F_RETRIGGER = 10,  /* _envelope.c:166-172 */
    KLMI_F_CUTOFF = 11,     /* Filter.frequency in Hz, quantised to the 1000-point grid
                               (effects.py:271-274, 365-369) */
    KLMI_F_GAIN = 12,       /* Mixer.gains[i] (mixer.py:33,43-44) */
    KLMI_F_PAN = 13,        /* StereoMixer.pannings[i] (mixer.py:88,97-99) */
    /* the voice's cutoff Lfo (desc.cutoff_lfo): Lfo(frequency, wave_func, shape=1,
     * outputRange) | Filter.frequency (oscillators.py:143-181, effects.py:365-373);
     * scale / offset = linear_mapping((-1, 1), outputRange) (oscillators.py:171) */
    KLMI_F_LFO_FREQUENCY = 14,
    KLMI_F_LFO_WAVE = 15,   /* klmi_wave, sine .. triangle */
    KLMI_F_LFO_SCALE = 16,
    KLMI_F_LFO_OFFSET = 17,
    /* the voice's audio-rate modulator (desc.mod_target): FmOscillator.modulator +
     * intensity (oscillators.py:202-220) or an Lfo(shape=256) (oscillators.py:178-181) */
    KLMI_F_MOD_FREQUENCY = 18,
    KLMI_F_MOD_WAVE = 19,   /* klmi_wave, sine .. triangle */
    KLMI_F_MOD_SCALE = 20,  /* FM: intensity; Lfo: scale */
    KLMI_F_MOD_OFFSET = 21, /* Lfo: offset (unused by FM) */
    /* state */
    KLMI_F_PHASE = 32,        /* Phasor.currentPhase (oscillators.py:89) */
    KLMI_F_ENV_VALUE = 33,    /* _envelope.c:147 */
    KLMI_F_ENV_STAGE = 34,    /* _envelope.c:146 (klmi_stage) */
    KLMI_F_ENV_ENABLED = 35,  /* _envelope.c:148 */
    KLMI_F_Z0 = 36,           /* _Filter.state[0] (effects.py:305-316) */
    KLMI_F_Z1 = 37,           /* _Filter.state[1] */
    KLMI_F_LFO_PHASE = 38,    /* the cutoff Lfo's currentPhase */
    KLMI_F_MOD_PHASE = 39,    /* the modulator's currentPhase */
    /* routing state of polyphonic banks (desc.poly_voices > 0) */
    KLMI_F_CURRENT_PITCH = 40, /* Voice.currentPitch (voices.py:42, 62-64) */
    KLMI_F_ROUND_ROBIN = 41    /* position of PolyphonicSynthesizer.freeVoice, the
                                  itertools.cycle over the voices (synthesizer.py:198, 203):
                                  per instrument, readable/settable through any of its voices */
} klmi_field;

/* Audio-rate modulation input of the voice's oscillator (one modulator per voice). */
typedef enum {
    KLMI_MOD_NONE = 0,
    KLMI_MOD_FM = 1,         /* FmOscillator: wave(phase + intensity * modulator)
                                (oscillators.py:193-234) */
    KLMI_MOD_FREQUENCY = 2,  /* Lfo(shape=256) | Oscillator.frequency: array branch of
                                sample_phase (oscillators.py:61-64) */
    KLMI_MOD_DUTY = 3        /* Lfo(shape=256) | PwmOscillator.dutyCycle
                                (oscillators.py:256-260) */
} klmi_mod_target;

/* Effect stages of a voice chain, in patch order between the voice and the
 * reverb / mixer (klang/audio/effects.py).  Parameters are set per voice with
 * klmi_bank_set_params(bank, KLMI_F_FX(stage, k), ...), k as listed. */
typedef enum {
    KLMI_FX_NONE = 0,
    KLMI_FX_FILTER = 1,     /* the chain's Filter (desc.has_filter) sits here; KLMI_F_CUTOFF */
    KLMI_FX_GAIN = 2,       /* Gain (effects.py:58-70): k0 gain */
    KLMI_FX_TRANSFORM = 3,  /* Transformer (effects.py:522-561): k0 scale, k1 offset */
    KLMI_FX_TANH = 4,       /* TanhDistortion (effects.py:469-487): k0 drive */
    KLMI_FX_OCTAVE = 5,     /* OctaveDistortion (effects.py:452-466) */
    KLMI_FX_TREMOLO = 6,    /* Tremolo (effects.py:73-133): k0 rate (Hz), k1 depth,
                               k2 smoothness, k3 dutyCycle; state k0 lfo phase, k1 zi */
    KLMI_FX_DELAY = 7,      /* Delay (effects.py:136-174): k0 time (s, <= 2, at least one
                               sample), k1 feedback, k2 drywet; state k0 ring position.
                               Setting k0 empties the lines of those voices whose length in
                               samples changed; the others keep their contents.  Lines
                               shorter than one buffer (256 samples) behave as the
                               reference's RingBuffer does (ring_buffer.py:88-109: its tail
                               and head copies land inside the window the Delay reads
                               back); a bank with such a line runs the stage block by block
                               (slower), and whether a bank has one is fixed by its first
                               render. */
    KLMI_FX_SUBSAMPLE = 8,  /* Subsampler (effects.py:388-417): k0 factor, a power of 2 in 2..128 */
    KLMI_FX_BITCRUSH = 9,   /* Bitcrusher (effects.py:420-449): k0 nBits in 0..15; samples go
                               through int16 like convert_samples_to_int (wavfile.py:23-30) */
    KLMI_FX_RINGMOD = 10    /* RingModulator (effects.py:618-662): k0 frequency, k1 rate,
                               k2 amount, k3 dryWet, k4 the Lfo's klmi_wave (sine..triangle);
                               state k0 lfo phase, k1 modulator phase */
} klmi_fx_kind;
#define KLMI_MAX_FX 8
#define KLMI_FX_PARAMS 8
#define KLMI_F_FX(stage, k) (64 + KLMI_FX_PARAMS * (stage) + (k)) /* parameter k = 0..7 of a stage */
#define KLMI_F_FX_STATE(stage, k) (128 + 2 * (stage) + (k)) /* state k = 0..1 of a stage */

/* What one bank evaluates per voice, in the reference's terms:
 *   osc = Oscillator(wave_func)                     oscillators.py:111-140
 *         (mod_target: FmOscillator / modulated frequency / modulated duty cycle)
 *   x   = has_env ? (amplitude*osc)*ADSR/AR/D/R     voices.py:66-74, envelopes.py:18-114
 *   x   = has_filter ? Filter(x)                    effects.py:342-385
 *         (cutoff_lfo: Filter.frequency driven per buffer by the voice's own Lfo)
 *   x   = fx[0](x), fx[1](x), ...                   when n_fx > 0: the stages in patch
 *         order, the Filter among them as KLMI_FX_FILTER (at most one stage of
 *         each stateful kind per chain: Filter, Tremolo, Delay, Subsampler, RingModulator); n_fx == 0 is the plain [Filter] chain
 *   x   = has_reverb ? Reverb(x)                    effects.py:564-615
 *   bus = stereo ? StereoMixer : Mixer over voices  mixer.py:15-104
 */
typedef struct {
    int32_t has_env;
    int32_t has_filter;
    int32_t has_reverb;
    int32_t stereo;
    int32_t pan_mode;     /* 0 linear, 1 constant_power, 2 mixture (panning.py:57-65) */
    int32_t has_pan_law;  /* panning.py:67-76 */
    double pan_law;       /* dB, <= 0 */
    double reverb_decay;      /* effects.py:575 decay (s) */
    double reverb_pre_delay;  /* preDelay (s) */
    double reverb_dry_wet;    /* dryWet, clipped to [0,1] (effects.py:586) */
    int32_t reverb_n_echos;   /* nEchos > 0 */
    int32_t cutoff_lfo;       /* 1: Lfo(shape=1) | Filter.frequency per voice (needs has_filter) */
    int32_t mod_target;       /* klmi_mod_target */
    int32_t n_fx;             /* number of effect stages, 0..KLMI_MAX_FX */
    uint8_t fx[KLMI_MAX_FX];  /* klmi_fx_kind per stage */
    /* synthesizers (klang/audio/synthesizer.py:151-219) */
    int32_t gated;            /* 1: a voice is only updated while Voice.active (voices.py:45-53:
                                 a note of this buffer or envelope stage != OFF), as Mono-/
                                 PolyphonicSynthesizer.update do (synthesizer.py:177-179,
                                 214-217); an idle voice keeps its oscillator phase.  Needs
                                 has_env; no filter, effect stages or reverb */
    int32_t poly_voices;      /* P > 0: the bank is n_voices / P PolyphonicSynthesizers of P
                                 voices each (instrument i owns voices [i*P, (i+1)*P)); notes
                                 are addressed to instruments with klmi_bank_notes and routed
                                 on the device (synthesizer.py:200-209).  Implies gated; the
                                 default mix divisor is P (synthesizer.py:219) */
    int32_t reverb_rings_f64; /* storage of the Reverb's delay lines: 0 float32 (80 B of HBM
                                 traffic per channel-sample with 10 combs; the comb arithmetic
                                 is float64 either way, each ring write is rounded once:
                                 measured 1.4e-7 on a channel's output), 1 float64
                                 as in _filters.c:33 (160 B; delay lines bit-identical to the
                                 reference's) */
    int32_t struct_size;      /* sizeof(klmi_chain_desc) as the CALLER compiled it, or 0 (not
                                 checked).  klmi_bank_create refuses any other value: a client
                                 built against an older, shorter descriptor fails loudly
                                 instead of handing over a short struct (the descriptor grew
                                 from 88 to 96 bytes between 0.1 and 0.2) */
} klmi_chain_desc;

/* Block-boundary events, the batched form of Voice.process_incoming_notes
 * (voices.py:55-64) and of Input.set_value on frequency inputs. */
typedef enum {
    KLMI_EV_NOTE_OFF = 0,    /* envelope.gate(False)                                */
    KLMI_EV_NOTE_ON = 1,     /* gate(True); amplitude = velocity; frequency = note  */
    KLMI_EV_SET_CUTOFF = 2,  /* Filter.frequency.set_value(frequency)               */
    KLMI_EV_SET_FREQUENCY = 3 /* Oscillator.frequency.set_value(frequency)           */
} klmi_event_kind;

typedef struct {
    uint32_t voice;
    uint32_t block;   /* block index relative to the next klmi_bank_render call */
    uint32_t kind;    /* klmi_event_kind */
    uint32_t reserved;
    double velocity;  /* NOTE_ON */
    double frequency; /* NOTE_ON, SET_CUTOFF, SET_FREQUENCY (Hz) */
} klmi_event;

/* A Note message (klang/messages.py:8-61) for one PolyphonicSynthesizer of a polyphonic bank. */
typedef struct {
    uint32_t instrument;
    uint32_t block;    /* relative to the next klmi_bank_render call */
    int32_t pitch;     /* 0..127 */
    uint32_t reserved;
    double velocity;   /* > 0: note-on, 0: note-off (messages.py:49-57) */
    double frequency;  /* Hz */
} klmi_note;

typedef struct klmi_ctx klmi_ctx;
typedef struct klmi_bank klmi_bank;
typedef struct klmi_env klmi_env;
typedef struct klmi_comb klmi_comb;
typedef struct klmi_kick klmi_kick;
typedef struct klmi_chirp klmi_chirp;

const char *klmi_last_error(void);
const char *klmi_version(void);

/* ---- context ------------------------------------------------------------ */
/* One GPU.  `stream` is the hipStream_t every launch and copy is ordered on
 * (e.g. torch.cuda.current_stream().cuda_stream); NULL is the device's default
 * (null) stream, as in every HIP call -- which is also what torch reports (0)
 * while no side stream is current. */
int klmi_ctx_create(int device, void *stream, klmi_ctx **out);
int klmi_ctx_destroy(klmi_ctx *ctx);
int klmi_ctx_synchronize(klmi_ctx *ctx);

/* Measurement aid (no reference counterpart): the rate this GPU's memory system gives a
 * plain hand-written streaming kernel -- one 16-byte access per lane, one workgroup per
 * 4 KiB in address order -- over `bytes` per buffer (at most 64 GiB), `reps` timed launches (HIP events on the context's stream).  kind 0
 * copy (read + write), 1 in-place update (read-modify-write, the Reverb kernel's access
 * mix), 2 fill (write only), 3 sum (read only).  gb_per_s counts bytes read + written.
 * bench.py quotes the HBM-bound kernels against it ("achievable") next to the 8 TB/s spec. */
int klmi_ctx_stream_rate(klmi_ctx *ctx, int kind, size_t bytes, int reps, double *gb_per_s);

/* Host-side design helpers (no GPU work).
 * effects.py:260-269: the 1000-point log2 grid and scipy.signal.butter(N=2,
 * 'lowpass', Wn) per grid point; b/a as scipy returns them (a[0] == 1). */
int klmi_filter_grid(double *freqs /* [KLMI_GRID_POINTS] */);
int klmi_filter_grid_index(double frequency);  /* effects.py:271-274; returns the row */
int klmi_butter2_lowpass(double wn, double b[3], double a[3]);
/* effects.py:33-55 low_pass_coefficients: single-pole low-pass, a1 = -exp(-TAU f / SR),
 * b0 = 1 - |a1|; f == 0 -> ([0], [1, -1]); f == inf -> ([1], [1, 0]). */
int klmi_low_pass_coefficients(double frequency, double b[1], double a[2]);
/* _envelope.c:112-119, 187-198: {attackCoef, attackBase, decayCoef, decayBase,
 * releaseCoef, releaseBase}. */
int klmi_envelope_coefficients(double attack, double decay, double sustain, double release,
                               double dt, double overshoot, double out[6]);
/* panning.py:35-81 */
int klmi_panning_amplitudes(double level, int mode, int has_pan_law, double pan_law,
                            double lr[2]);
/* primes.py:133-141 and effects.py:587-606: tap lengths and comb gains of a Reverb. */
int klmi_reverb_design(double decay, double pre_delay, int n_echos, int32_t *taps,
                       double *alphas);

/* ---- voice bank: the hot path -------------------------------------------- */
/* Replaces N x {Oscillator.update, EnvelopeBase.update, Voice.update,
 * Filter.update, Reverb.update} + Mixer.update per 256-sample buffer
 * (klang/klang.py:76-78 hot loop). */
int klmi_bank_create(klmi_ctx *ctx, int n_voices, const klmi_chain_desc *desc, klmi_bank **out);
int klmi_bank_destroy(klmi_bank *bank);

int klmi_bank_set_params(klmi_bank *bank, int field, int first, int count, const double *host);
int klmi_bank_state_get(klmi_bank *bank, int field, int first, int count, double *host);
int klmi_bank_state_set(klmi_bank *bank, int field, int first, int count, const double *host);

/* Queue events for the next render (any order; applied per voice in block
 * order, same-block events in submission order).  Events whose block lies
 * beyond the next render stay queued, re-based. */
int klmi_bank_events(klmi_bank *bank, const klmi_event *events, size_t n);

/* PolyphonicSynthesizer.process_note (synthesizer.py:200-209), batched and moved onto the
 * device: queue notes for the instruments of a polyphonic bank (any order; applied per
 * instrument in block order, same-block notes in submission order).  Per buffer every voice
 * scans its instrument's notes: a note-on goes to the voice the round robin points at, a
 * note-off to every voice whose currentPitch -- as the previous buffer left it -- equals the
 * note's pitch. */
int klmi_bank_notes(klmi_bank *bank, const klmi_note *notes, size_t n);

/* Mixer divisor: mixer.py:46-47 divides the bus by nInputs when nInputs > 1.
 * Default is that rule on this bank's n_voices; a sharded patch sets 1.0 on
 * every shard and divides by the global voice count after the reduce. */
int klmi_bank_set_mix_divisor(klmi_bank *bank, double divisor);

/* External input for KLMI_WAVE_EXTERNAL voices: x[n_voices][n_blocks*256]
 * float64.  Renders consume it block by block (several renders may walk through
 * one input); previews read from the current position without consuming. */
int klmi_bank_set_input(klmi_bank *bank, const double *host_x, int n_blocks);

/* The same from the device: the first n_blocks buffers `source` mixed in its last
 * klmi_bank_render (they are still in its device bus) become this bank's external input;
 * voice v listens to channel v % channels of that bus.  It replaces the per-buffer hop
 * Output.value -> Input.value between two fused groups of blocks (connections.py:274-302),
 * e.g. `mixer | Reverb | Dac` (examples/reverberation_demo.py:33-34): both banks can then
 * render many buffers per call.  Same context (GPU) for both banks. */
int klmi_bank_set_input_from_bank(klmi_bank *bank, klmi_bank *source, int n_blocks);

/* THE hot call.  Renders n_blocks buffers of every voice, advances all state,
 * consumes due events, writes the mix bus [n_blocks][channels][256] float32
 * (channels = stereo ? 2 : 1).  _device writes to a device pointer and does
 * not synchronise. */
int klmi_bank_render(klmi_bank *bank, int n_blocks, float *host_mix);
int klmi_bank_render_device(klmi_bank *bank, int n_blocks, float *dev_mix);

/* klmi_bank_render_device without any allocation.  It has no counterpart in the
 * reference: it is for callers that cannot allocate -- tight serving loops, harnesses that
 * forbid hipMalloc, code under HIP graph capture (for capture two more preconditions hold,
 * because the call otherwise synchronises the stream or sets a kernel attribute: no events /
 * notes queued since the last render, and the bank's kernels launched once before).  The per-wave partial buses and
 * the slice sums of the bus fold, which klmi_bank_render_device keeps in buffers that grow
 * with the call, live in ONE static device array of KLMI_NOALLOC_SCRATCH_BYTES per GPU
 * (held by every GPU the library runs on until the process ends); the call makes no
 * hipMalloc / hipFree and no allocating library call.  Same kernels, same fold order:
 * the bus is bit-identical to klmi_bank_render_device's, state and queues advance alike.
 * Limits, each reported with a clear error and nothing rendered:
 *   - one buffer of the bank must fit the array (n_voices * channels * 16 bytes + at
 *     most 64 KiB * channels; 4,128,768 voices mono): KLMI_E_ARG; longer calls are split
 *     into as many launches as the array requires;
 *   - reverb banks (their dry bus and rings are sized per call): KLMI_E_STATE;
 *   - events / notes queued since the last render must fit the device queues already
 *     uploaded, and a Delay stage's lines must exist (one allocating render or preview
 *     first): KLMI_E_STATE;
 *   - klmi_bank_last_kernel_ms reports 0 launches after it (timing events are created
 *     on demand by the allocating calls only).
 * Every call on a GPU shares that GPU's array, whatever bank or context it comes from:
 * calls that overlap in time on one GPU (different streams or threads) race; calls
 * ordered on one stream do not.  Nothing in the array has to start at zero -- each launch
 * overwrites every element it later reads. */
#define KLMI_NOALLOC_SCRATCH_BYTES (64u << 20)
int klmi_bank_render_device_noalloc(klmi_bank *bank, int n_blocks, float *dev_mix);

/* Materialise per-voice streams (Output.value of the fused chain's last block
 * for voices [first, first+count)): out[count][n_blocks*256].  A *preview*:
 * state and the event queue are left untouched, so the next klmi_bank_render
 * of the same span mixes exactly these samples.  _f64 returns the unrounded
 * float64 values (parity diagnostics). */
int klmi_bank_render_voices(klmi_bank *bank, int first, int count, int n_blocks, float *host_out);
int klmi_bank_render_voices_f64(klmi_bank *bank, int first, int count, int n_blocks,
                                double *host_out);
int klmi_bank_render_voices_device(klmi_bank *bank, int first, int count, int n_blocks,
                                   float *dev_out);

/* ---- output side of an offline bounce ---------------------------------------- */
/* run_audio_engine(offline=True) for a patch whose Dac is fed by this bank
 * (klang/audio/klanggeber.py:179-300): nCycles = int(duration * SAMPLING_RATE //
 * BUFFER_SIZE) buffers from play time 0 (:258); per buffer Dac.collect_samples /
 * pack_signals to interleaved float32 (:81-97, 164-176) and, for buffers with
 * playTime >= duration - fadeout, the np.interp fade-out (:205-209, 243-249); then
 * write_wave's peak normalisation (klang/audio/wavfile.py:54-63, klang/math.py:14-21):
 * int16(frames / max|frames| * 32767) in float32.  One device pass instead of one host
 * round trip per buffer.
 *   frames [nCycles*256][channels] float32 (optional for _bounce), pcm the same shape
 *   int16 (optional), peak max|frames|.  A silent piece with pcm requested fails with
 *   KLMI_E_ARG ("Zero amplitude! Can not normalize!", the reference's ValueError) after
 *   frames and peak have been delivered.  State, events and notes advance as in
 *   klmi_bank_render. */
int klmi_bounce_cycles(double duration);  /* nCycles for a duration (0 if not positive) */
int klmi_bank_bounce(klmi_bank *bank, double duration, double fadeout, float *host_frames,
                     int16_t *host_pcm, float *host_peak);
int klmi_bank_bounce_device(klmi_bank *bank, double duration, double fadeout, float *dev_frames,
                            int16_t *dev_pcm, float *dev_peak);
/* The output side alone, for a bus rendered elsewhere: one process per GPU renders its
 * voice shard with klmi_bank_render_device (mix divisor 1), ONE reduce sums the partial
 * buses over the ranks (klang/audio/mixer.py:41-49 is the only cross-voice operation; the
 * root divides by the global input count, :46-47), and the root hands the summed bus
 * dev_bus [nCycles][channels][256] float32 to this call for Dac packing, fade-out and the
 * peak-normalised int16 samples exactly as above (klanggeber.py:81-97, 205-249,
 * wavfile.py:54-63).  Renders nothing; the bank's state, events and block clock stay. */
int klmi_bank_bounce_bus(klmi_bank *bank, const float *dev_bus, double duration, double fadeout,
                         float *host_frames, int16_t *host_pcm, float *host_peak);

/* Reverb delay-line state of one voice: rings concatenated in tap order
 * (sum of tap lengths float64) and the ring positions [n_echos]. */
int klmi_bank_reverb_state_get(klmi_bank *bank, int voice, double *rings, int32_t *positions);

/* Kernel timing of the last render call, measured with HIP events on the
 * ctx stream: milliseconds spent in the voice kernel(s) and launches made. */
int klmi_bank_last_kernel_ms(klmi_bank *bank, float *voice_ms, int *launches);

/* ---- envelope batch: klang.audio._envelope.Envelope, x count --------------- */
/* _envelope.c:518-535 (ctor; overshoot clipped), :549-567 gate, :636-663
 * sample(n) -> float64[count][n], :385-393 active.  count == 1 is the ext type. */
int klmi_env_create(klmi_ctx *ctx, int count, double attack, double decay, double sustain,
                    double release, double dt, double overshoot, int retrigger, int loop,
                    klmi_env **out);
int klmi_env_destroy(klmi_env *env);
int klmi_env_set(klmi_env *env, int field, int first, int count, const double *host);
int klmi_env_get(klmi_env *env, int field, int first, int count, double *host);
int klmi_env_gate(klmi_env *env, int first, int count, int on);
int klmi_env_sample(klmi_env *env, long n, double *host_out);
int klmi_env_active(klmi_env *env, int32_t *host_out /* [count] */);

/* ---- ring-buffer filter batch: klang.audio._filters.*, x count ------------- */
/* _filters.c:113-139 ctor (length int, alpha=.9; length > 60*44100 fails),
 * :231-306 filter(x) for kind 0 forward comb, 1 backward comb, 2 echo.
 * x, y: float64[count][n]. */
int klmi_comb_create(klmi_ctx *ctx, int kind, int count, int length, double alpha,
                     klmi_comb **out);
int klmi_comb_destroy(klmi_comb *comb);
int klmi_comb_set_alpha(klmi_comb *comb, int first, int count, const double *host);
int klmi_comb_filter(klmi_comb *comb, const double *host_x, long n, double *host_y);

/* ---- kick drums: klang.audio.synthesizer.Kick, x count ---------------------- */
/* synthesizer.py:253-283 (with sample_exponential_decay / sample_pitch_decay :22-55 and
 * waves.py:56-61 sample_wave): ctor defaults frequency 40, decay .8, intensity 2,
 * pitchDecay .3.  klmi_kick_trigger queues the buffers (relative to the next render) at
 * which a note-on with pitch > 0 reaches a kick (currentTime and currentPhase back to 0);
 * klmi_kick_render = update() x n_blocks -> float64[count][n_blocks*256].  State fields:
 * currentTime [count], currentPhase [count][256] (a vector after the first buffer, as in
 * the reference). */
typedef enum {
    KLMI_KICK_FREQUENCY = 0,
    KLMI_KICK_DECAY = 1,
    KLMI_KICK_INTENSITY = 2,
    KLMI_KICK_PITCH_DECAY = 3
} klmi_kick_field;
int klmi_kick_create(klmi_ctx *ctx, int count, klmi_kick **out);
int klmi_kick_destroy(klmi_kick *kick);
int klmi_kick_set(klmi_kick *kick, int field, int first, int count, const double *host);
int klmi_kick_trigger(klmi_kick *kick, const uint32_t *which, const uint32_t *block, size_t n);
int klmi_kick_render(klmi_kick *kick, int n_blocks, double *host_out);
int klmi_kick_state_get(klmi_kick *kick, double *host_time /* [count] */,
                        double *host_phase /* [count][256] */);

/* ---- chirp oscillators: klang.audio.oscillators.Chirper, x count ------------- */
/* oscillators.py:280-308: wave_func(chirp_phase(t, startFrequency, duration, endFrequency,
 * method)); chirp_phase (oscillators.py:18-38) is scipy.signal's _chirp_phase
 * (vertex_zero=True) until `duration` and the end frequency held from there on; t is the
 * chirper's own time vector (starts as DT * arange(256), gains DT * 256 per element and
 * buffer; _reset puts it back).  wave: klmi_wave sine..triangle.  Combinations that reach
 * the sweep (duration > 0, start != end) and are invalid there fail with scipy's messages
 * (logarithmic: f0 * f1 <= 0; hyperbolic: a zero frequency).
 * klmi_chirp_render = update() x n_blocks -> float64[count][n_blocks*256]. */
typedef enum {
    KLMI_CHIRP_LINEAR = 0,
    KLMI_CHIRP_QUADRATIC = 1,
    KLMI_CHIRP_LOGARITHMIC = 2,
    KLMI_CHIRP_HYPERBOLIC = 3
} klmi_chirp_method;
int klmi_chirp_create(klmi_ctx *ctx, int count, const double *start_frequency,
                      const double *duration, const double *end_frequency,
                      const int32_t *method, const int32_t *wave, klmi_chirp **out);
int klmi_chirp_destroy(klmi_chirp *chirp);
int klmi_chirp_reset(klmi_chirp *chirp, int first, int count);
int klmi_chirp_render(klmi_chirp *chirp, int n_blocks, double *host_out);
int klmi_chirp_state_get(klmi_chirp *chirp, double *host_t /* [count][256] */);

#ifdef __cplusplus
}
#endif
#endif
