/*
 * klang_oracle.h -- CPU restatement of atheler/klang's per-block DSP hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.  Nothing under klang_amd/
 * links, imports or calls it; the product path is the HIP library
 * (klang_amd/csrc) and fails loudly when that is missing.
 *
 * Every function restates, in scalar float64 with the reference's operation
 * order (built -O2 -ffp-contract=off), one piece of the reference and cites
 * the file:line (relative to the reference root) it follows.
 *
 * Pinning: tests/test_oracle_golden.py checks every function here against
 * fixtures in tests/golden/ that tools/gen_golden.py produced by running the
 * real reference (its NumPy/SciPy code and its two C extensions compiled from
 * the reference sources into oracle/_ref/), and against the known-answer tests
 * of the reference's own test-suite (tests/test_audio__envelope.py,
 * tests/test_audio_filters.py, tests/test_audio_oscillators.py).
 */
#ifndef KLANG_ORACLE_H
#define KLANG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KO_BUFFER_SIZE 256      /* klang/config.py:5  */
#define KO_SAMPLING_RATE 44100  /* klang/config.py:11 */
#define KO_GRID_POINTS 1000     /* klang/audio/effects.py:260-265 */

enum ko_wave {
    KO_WAVE_SINE = 0,      /* klang/audio/waves.py:22 */
    KO_WAVE_SQUARE = 1,    /* klang/audio/waves.py:27 */
    KO_WAVE_SAWTOOTH = 2,  /* klang/audio/waves.py:32 */
    KO_WAVE_TRIANGLE = 3,  /* klang/audio/waves.py:37 */
    KO_WAVE_PWM = 4,       /* klang/audio/oscillators.py:256-260 */
    KO_WAVE_ONE = 5        /* no oscillator: constant 1.0 (envelope-only chains) */
};

enum ko_stage { KO_OFF = 0, KO_ATTACKING, KO_DECAYING, KO_SUSTAINING, KO_RELEASING };

/* ---- oscillators.py / waves.py / math.py -------------------------------- */
double ko_np_mod(double a, double b);
void ko_sample_phase_const(double frequency, double start, double *phase, double *next);
void ko_sample_phase_array(const double *frequency, double start, double *phase, double *next);
double ko_phasor_step(double frequency, double current);
double ko_wrap(double phase);
void ko_wave(int wave_id, const double *phase, size_t n, double duty, double *out);

/* ---- _envelope.c -------------------------------------------------------- */
typedef struct {
    double attack, decay, sustain, release, dt, overshoot;
    int retrigger, loop;
    int stage;
    double value;
    int enabled;
    double attackCoef, attackBase, decayCoef, decayBase, releaseCoef, releaseBase;
} ko_envelope;

double ko_env_coefficient(double rate, double overshoot);
void ko_envelope_init(ko_envelope *e, double attack, double decay, double sustain,
                      double release, double dt, double overshoot, int retrigger, int loop);
void ko_envelope_recompute(ko_envelope *e);
void ko_envelope_gate(ko_envelope *e, int on);
double ko_envelope_step(ko_envelope *e);
void ko_envelope_sample(ko_envelope *e, size_t n, double *out);

/* ---- effects.py Filter (scipy butter / lfilter boundary) ---------------- */
void ko_butter2_lowpass(double wn, double b[3], double a[3]);
void ko_filter_grid(double *freqs /* KO_GRID_POINTS */);
int ko_filter_grid_index(const double *freqs, int n, double frequency);
void ko_lfilter2(const double b[3], const double a[3], const double *x, size_t n,
                 double z[2], double *y);
void ko_lfilter1(const double b[1], const double a[2], const double *x, size_t n,
                 double z[1], double *y);
void ko_low_pass_coefficients(double frequency, double b[1], double a[2]);

/* ---- _filters.c --------------------------------------------------------- */
enum ko_ring_kind { KO_COMB_FORWARD = 0, KO_COMB_BACKWARD = 1, KO_ECHO = 2 };
typedef struct {
    size_t length, position;
    double alpha;
    double *data;
} ko_ring_filter;

int ko_ring_filter_init(ko_ring_filter *f, size_t length, double alpha);
void ko_ring_filter_free(ko_ring_filter *f);
void ko_ring_filter_run(ko_ring_filter *f, int kind, const double *x, size_t n, double *y);

/* ---- effects.py Reverb / primes.py -------------------------------------- */
void ko_next_primes(int n, int start, int *out);
double ko_reverb_alpha(double delay_time, double decay);
typedef struct {
    int n_echos;
    double dry_wet;
    ko_ring_filter *filters;
} ko_reverb;
int ko_reverb_init(ko_reverb *r, double decay, double pre_delay, double dry_wet, int n_echos);
void ko_reverb_free(ko_reverb *r);
void ko_reverb_run(ko_reverb *r, const double *x, size_t n, double *out);

/* ---- mixer.py / panning.py ---------------------------------------------- */
void ko_mixer(const double *inputs /* [n][len] */, const double *gains, int n, size_t len,
              double *out);
void ko_panning_amplitudes(double level, int mode /*0 linear,1 constant_power,2 mixture*/,
                           int has_pan_law, double pan_law, double lr[2]);
void ko_stereo_mixer(const double *inputs /* [n][len] */, const double *gains,
                     const double *pan_lr /* [n][2] */, int n, size_t len,
                     double *out /* [2][len] */);

/* ---- voices.py + chain: a bank of independent voices --------------------- */
typedef struct {
    uint32_t voice;
    uint32_t block;
    uint32_t kind;      /* 0 note-off, 1 note-on, 2 set cutoff coefficients index */
    uint32_t index;     /* kind 2: coefficient-table row */
    double velocity;
    double frequency;
} ko_event;

typedef struct {
    int n_voices;
    int has_env, has_filter, stereo;
    /* parameters */
    double *frequency, *amplitude, *duty;
    int32_t *wave;
    ko_envelope *env;          /* [n_voices] */
    int32_t *coef_index;       /* row of coef_table per voice */
    const double *coef_table;  /* [rows][5] = b0 b1 b2 a1 a2 */
    double *gain, *pan_l, *pan_r;
    /* state */
    double *phase, *z0, *z1;
    /* modulation inputs (oscillators.py:143-267, effects.py:365-373); arrays may be
     * NULL when the feature is off */
    int cutoff_lfo;            /* 1: Lfo(shape=1) | Filter.frequency, one Lfo per voice */
    double *lfo_frequency, *lfo_scale, *lfo_offset, *lfo_phase /* state */;
    int32_t *lfo_wave;
    const double *grid;        /* [KO_GRID_POINTS] filter frequencies (effects.py:260-265) */
    int mod_target;            /* KO_MOD_* below: one audio-rate modulator per voice */
    double *mod_frequency, *mod_scale, *mod_offset, *mod_phase /* state */;
    int32_t *mod_wave;
    /* effect stages between the voice and the mixer, in patch order (effects.py:58-174,
     * 452-561).  n_fx == 0: the chain is just [Filter] when has_filter. */
    int n_fx;
    int fx_kind[8];            /* KO_FX_* */
    double *fx_par;            /* [8][KO_FX_PARAMS][n_voices] parameters as the blocks take them */
    double *fx_state;          /* [8][2][n_voices]: tremolo lfo phase, zi; delay position;
                                  ring modulator lfo phase, modulator phase */
    double *fx_ring;           /* the delay stage's rings, [n_voices][fx_ring_capacity] */
    int fx_ring_capacity;
} ko_bank;

enum {
    KO_FX_NONE = 0,
    KO_FX_FILTER = 1,     /* the bank's Filter (has_filter) at this position */
    KO_FX_GAIN = 2,       /* p0 gain */
    KO_FX_TRANSFORM = 3,  /* p0 scale, p1 offset */
    KO_FX_TANH = 4,       /* p0 drive */
    KO_FX_OCTAVE = 5,
    KO_FX_TREMOLO = 6,    /* p0 rate, p1 depth, p2 smoothness, p3 dutyCycle */
    KO_FX_DELAY = 7,      /* p0 time (s), p1 feedback, p2 drywet; length >= one buffer */
    KO_FX_SUBSAMPLE = 8,  /* p0 factor (2..128, power of 2) */
    KO_FX_BITCRUSH = 9,   /* p0 nBits (0..15) */
    KO_FX_RINGMOD = 10    /* p0 frequency, p1 rate, p2 amount, p3 dryWet, p4 lfo wave */
};
#define KO_FX_PARAMS 8

enum {
    KO_MOD_NONE = 0,
    KO_MOD_FM = 1,         /* FmOscillator: wave(phase + scale * Oscillator(mod_frequency, mod_wave)) */
    KO_MOD_FREQUENCY = 2,  /* Lfo(shape=256) | Oscillator.frequency (array branch of sample_phase) */
    KO_MOD_DUTY = 3        /* Lfo(shape=256) | PwmOscillator.dutyCycle */
};

/* Renders n_blocks blocks.  events must be sorted by (voice, block), stable.
 * mix: [n_blocks][channels][256] float64 (sum in voice order, / n if n > 1).
 * voices: optional [n_voices][n_blocks*256] float64 per-voice streams.
 * n_threads > 1 uses OpenMP over voices for the timed CPU baseline only (the
 * per-thread partial buses are then added in thread order). */
void ko_bank_render(ko_bank *bank, int n_blocks, const ko_event *events, size_t n_events,
                    double *mix, double *voices, int n_threads);

/* ---- synthesizer.py: note routing + activity gating over a bank ------------ */
typedef struct {
    uint32_t instrument;
    uint32_t block;
    int32_t pitch;
    uint32_t reserved;
    double velocity;   /* 0: note-off (messages.py:49-57) */
    double frequency;
} ko_note;

/* MonophonicSynthesizer (mono != 0, poly == 1; notes are what the host's NoteScheduler
 * hands to the voice) / PolyphonicSynthesizer (synthesizer.py:151-219).  bank: has_env,
 * n_voices = instruments * poly.  notes sorted by (instrument, block), stable.
 * round_robin [instruments], current_pitch [n_voices]: routing state, zero at start.
 * out [instruments][n_blocks*256]; voices optional [n_voices][n_blocks*256] (zeros where a
 * voice was not updated). */
void ko_synth_render(ko_bank *bank, int poly, int mono, int32_t *round_robin,
                     int32_t *current_pitch, int n_blocks, const ko_note *notes, size_t n_notes,
                     double *out, double *voices);

/* ---- klanggeber.py / wavfile.py: output side of an offline bounce --------- */
/* pack_signals + fade-out + peak-normalised int16 (klanggeber.py:81-97, 205-249,
 * wavfile.py:54-63, math.py:14-21).  bus [n_blocks][channels][256] float32 ->
 * frames [n_blocks*256][channels] float32, pcm (optional) int16; returns the peak. */
float ko_bounce(const float *bus, int n_blocks, int channels, double duration, double fadeout,
                float *frames, int16_t *pcm);

/* ---- synthesizer.py Kick ---------------------------------------------------- */
typedef struct {
    double frequency, decay, intensity, pitch_decay; /* synthesizer.py:257: 40, .8, 2, .3 */
    double current_time;
    double current_phase[KO_BUFFER_SIZE];
} ko_kick;
void ko_kick_block(ko_kick *kick, int trigger, double *out /* 256 */);

/* ---- oscillators.py Chirper ------------------------------------------------ */
enum { KO_CHIRP_LINEAR = 0, KO_CHIRP_QUADRATIC, KO_CHIRP_LOGARITHMIC, KO_CHIRP_HYPERBOLIC };
typedef struct {
    double f0, t1, f1; /* startFrequency, duration, endFrequency */
    int method;        /* KO_CHIRP_* */
    int wave;          /* ko_wave */
    double t[KO_BUFFER_SIZE];
} ko_chirper;
void ko_chirp_reset(ko_chirper *c);
void ko_chirp_block(ko_chirper *c, double *out /* 256 */);

int ko_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
