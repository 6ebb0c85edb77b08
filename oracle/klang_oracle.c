/*
 * klang_oracle.c -- CPU restatement of atheler/klang's per-block DSP hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see klang_oracle.h).  Scalar float64,
 * reference operation order, build with -O2 -ffp-contract=off.
 * Citations are file:line relative to the reference root.
 */
#include "klang_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const double KO_PI = 3.141592653589793;           /* klang/constants.py:5 */
static const double KO_TAU = 2. * 3.141592653589793;      /* klang/constants.py:8 */
static const double KO_DT = 1. / KO_SAMPLING_RATE;        /* klang/audio/helpers.py:9 */
static const double KO_INTERVAL = (1. / KO_SAMPLING_RATE) * KO_BUFFER_SIZE; /* helpers.py:12 */

/* ------------------------------------------------------------------------ */
/* NumPy's float remainder (np.mod / Python %): C fmod, then the result takes
 * the divisor's sign.  Used by oscillators.py:66, waves.py:34,39, math.py:11. */
double ko_np_mod(double a, double b)
{
    double m = fmod(a, b);
    if (b == 0.)
        return m;
    if (m != 0.) {
        if ((b < 0.) != (m < 0.))
            m += b;
    } else {
        m = copysign(0., b);
    }
    return m;
}

/* oscillators.py:57-60,66-67 (scalar frequency branch).
 * t = DT * arange(257) (helpers.py:28-32); phase = (TAU*f)*t + start; mod TAU. */
void ko_sample_phase_const(double frequency, double start, double *phase, double *next)
{
    const double w = KO_TAU * frequency;
    for (int n = 0; n <= KO_BUFFER_SIZE; ++n) {
        const double t = KO_DT * (double)n;
        const double p = ko_np_mod(w * t + start, KO_TAU);
        if (n < KO_BUFFER_SIZE)
            phase[n] = p;
        else
            *next = p;
    }
}

/* oscillators.py:61-67 (array frequency branch): phase[0] = start;
 * phase[1:] = (TAU*DT) * cumsum(f) + start.  np.cumsum on float64 is a plain
 * sequential running sum. */
void ko_sample_phase_array(const double *frequency, double start, double *phase, double *next)
{
    const double k = KO_TAU * KO_DT;
    double running = 0.;
    phase[0] = ko_np_mod(start, KO_TAU);
    for (int n = 1; n <= KO_BUFFER_SIZE; ++n) {
        running = (n == 1) ? frequency[0] : running + frequency[n - 1];
        const double p = ko_np_mod(k * running + start, KO_TAU);
        if (n < KO_BUFFER_SIZE)
            phase[n] = p;
        else
            *next = p;
    }
}

/* oscillators.py:91-96 (Phasor.sample; also Lfo scalar shape :175-177). */
double ko_phasor_step(double frequency, double current)
{
    return ko_np_mod(KO_TAU * frequency * KO_INTERVAL + current, KO_TAU);
}

/* math.py:9-11 */
double ko_wrap(double phase)
{
    return ko_np_mod(phase + KO_PI, KO_TAU) - KO_PI;
}

/* waves.py:22-39; oscillators.py:259-260 for PWM. */
void ko_wave(int wave_id, const double *phase, size_t n, double duty, double *out)
{
    for (size_t i = 0; i < n; ++i) {
        const double p = phase[i];
        double v;
        switch (wave_id) {
        case KO_WAVE_SINE:
            v = sin(p);
            break;
        case KO_WAVE_SQUARE:
            v = 2. * (ko_wrap(p) >= 0. ? 1. : 0.) - 1.;
            break;
        case KO_WAVE_SAWTOOTH:
            v = 2. * ko_np_mod(p, KO_TAU) / KO_TAU - 1.;
            break;
        case KO_WAVE_TRIANGLE:
            v = 1. - fabs(ko_np_mod(4. * ko_wrap(p) / KO_TAU, 4.) - 2.);
            break;
        case KO_WAVE_PWM:
            v = 2. * (p < KO_TAU * duty ? 1. : 0.) - 1.;
            break;
        default:
            v = 1.;
            break;
        }
        out[i] = v;
    }
}

/* ------------------------------------------------------------------------ */
/* _envelope.c:112-119 */
double ko_env_coefficient(double rate, double overshoot)
{
    if (rate <= 0.)
        return 0.;
    return exp(-log((1.0 + overshoot) / overshoot) / rate);
}

/* _envelope.c:187-198 */
void ko_envelope_recompute(ko_envelope *e)
{
    e->attackCoef = ko_env_coefficient(e->attack / e->dt, e->overshoot);
    e->attackBase = (1. + e->overshoot) * (1. - e->attackCoef);
    e->decayCoef = ko_env_coefficient(e->decay / e->dt, e->overshoot);
    e->decayBase = (e->sustain - e->overshoot) * (1. - e->decayCoef);
    e->releaseCoef = ko_env_coefficient(e->release / e->dt, e->overshoot);
    e->releaseBase = (0. - e->overshoot) * (1. - e->releaseCoef);
}

/* _envelope.c:461-490 (defaults), :518-535 (init incl. overshoot clip :532) */
void ko_envelope_init(ko_envelope *e, double attack, double decay, double sustain,
                      double release, double dt, double overshoot, int retrigger, int loop)
{
    e->attack = attack;
    e->decay = decay;
    e->sustain = sustain;
    e->release = release;
    e->dt = dt;
    e->overshoot = fmin(fmax(overshoot, 1e-9), 1e9);
    e->retrigger = retrigger != 0;
    e->loop = loop != 0;
    e->stage = KO_OFF;
    e->value = 0.;
    e->enabled = 0;
    ko_envelope_recompute(e);
}

/* _envelope.c:549-567 */
void ko_envelope_gate(ko_envelope *e, int on)
{
    if (on) {
        e->enabled = 1;
        if (e->retrigger || e->stage == KO_OFF || e->stage == KO_RELEASING)
            e->stage = KO_ATTACKING;
    } else {
        e->enabled = 0;
        if (e->stage == KO_ATTACKING || e->stage == KO_DECAYING || e->stage == KO_SUSTAINING)
            e->stage = KO_RELEASING;
    }
}

/* _envelope.c:579-624 */
double ko_envelope_step(ko_envelope *e)
{
    double v = 0.;
    switch (e->stage) {
    case KO_OFF:
        v = 0.;
        break;
    case KO_ATTACKING:
        v = e->attackBase + e->value * e->attackCoef;
        if (v >= 1.) {
            v = 1.;
            e->stage = KO_DECAYING;
        }
        break;
    case KO_DECAYING:
        v = e->decayBase + e->value * e->decayCoef;
        if (v <= e->sustain) {
            v = e->sustain;
            e->stage = e->loop ? KO_RELEASING : KO_SUSTAINING;
        }
        break;
    case KO_SUSTAINING:
        v = e->sustain;
        break;
    case KO_RELEASING:
        v = e->releaseBase + e->value * e->releaseCoef;
        if (v <= 0.) {
            v = 0.;
            e->stage = (e->loop && e->enabled) ? KO_ATTACKING : KO_OFF;
        }
        break;
    }
    e->value = v;
    return v;
}

/* _envelope.c:636-663 */
void ko_envelope_sample(ko_envelope *e, size_t n, double *out)
{
    for (size_t i = 0; i < n; ++i)
        out[i] = ko_envelope_step(e);
}

/* ------------------------------------------------------------------------ */
/* effects.py:266-269, 354: scipy.signal.butter(N=2, Wn, btype='lowpass').
 * scipy is an un-vendored dependency (requirements.txt:2, no version pin;
 * fixtures were made with scipy 1.15.3).  Published algorithm: 2nd-order
 * Butterworth analog prototype, frequency pre-warp, bilinear transform.  With
 * K = tan(pi*Wn/2):  b = K^2/(1+sqrt2 K+K^2) * [1,2,1],
 * a = [1, 2(K^2-1)/(..), (1-sqrt2 K+K^2)/(..)].  Agrees with scipy's
 * zpk->tf route to <= 2e-15 relative (tests/test_oracle_golden.py). */
void ko_butter2_lowpass(double wn, double b[3], double a[3])
{
    const double K = tan(KO_PI * wn / 2.);
    const double K2 = K * K;
    const double r2 = sqrt(2.);
    const double norm = 1. / (1. + r2 * K + K2);
    b[0] = K2 * norm;
    b[1] = 2. * b[0];
    b[2] = b[0];
    a[0] = 1.;
    a[1] = 2. * (K2 - 1.) * norm;
    a[2] = (1. - r2 * K + K2) * norm;
}

/* effects.py:260-265: np.logspace(log2(20), log2(20000), num=1000, base=2).
 * np.linspace computes arange(num)*step + start with the last point forced to
 * stop; logspace is power(base, y). */
void ko_filter_grid(double *freqs)
{
    const double start = log2(20.);
    const double stop = log2(20000.);
    const double step = (stop - start) / (double)(KO_GRID_POINTS - 1);
    for (int i = 0; i < KO_GRID_POINTS; ++i) {
        double y = (double)i * step + start;
        if (i == KO_GRID_POINTS - 1)
            y = stop;
        freqs[i] = pow(2., y);
    }
}

/* effects.py:271-274: np.searchsorted (side='left') then clip to [0, n-1]. */
int ko_filter_grid_index(const double *freqs, int n, double frequency)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        if (freqs[mid] < frequency)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo > n - 1)
        lo = n - 1;
    return lo;
}

/* effects.py:309-316: scipy.signal.lfilter(b, a, x, zi) for len(b)=len(a)=3,
 * a[0]=1.  scipy's direct-form-II-transposed loop:
 *   y = z0 + b0*x;  z0 = z1 + b1*x - a1*y;  z1 = b2*x - a2*y. */
void ko_lfilter2(const double b[3], const double a[3], const double *x, size_t n,
                 double z[2], double *y)
{
    double z0 = z[0], z1 = z[1];
    for (size_t i = 0; i < n; ++i) {
        const double xi = x[i];
        const double yi = z0 + b[0] * xi;
        z0 = z1 + b[1] * xi - a[1] * yi;
        z1 = b[2] * xi - a[2] * yi;
        y[i] = yi;
    }
    z[0] = z0;
    z[1] = z1;
}

/* effects.py:128 (Tremolo): lfilter with b=[b0], a=[1,a1] -> scipy pads b to
 * len 2: y = z0 + b0*x; z0 = 0*x - a1*y. */
void ko_lfilter1(const double b[1], const double a[2], const double *x, size_t n,
                 double z[1], double *y)
{
    double z0 = z[0];
    for (size_t i = 0; i < n; ++i) {
        const double yi = z0 + b[0] * x[i];
        z0 = 0. * x[i] - a[1] * yi;
        y[i] = yi;
    }
    z[0] = z0;
}

/* effects.py:33-55 */
void ko_low_pass_coefficients(double frequency, double b[1], double a[2])
{
    if (frequency == 0.) {
        b[0] = 0.;
        a[0] = 1.;
        a[1] = -1.;
        return;
    }
    if (isinf(frequency)) {
        b[0] = 1.;
        a[0] = 1.;
        a[1] = 0.;
        return;
    }
    a[0] = 1.;
    a[1] = -exp(-KO_TAU * frequency / KO_SAMPLING_RATE);
    b[0] = 1. - fabs(a[1]);
}

/* ------------------------------------------------------------------------ */
/* _filters.c:41-59 */
int ko_ring_filter_init(ko_ring_filter *f, size_t length, double alpha)
{
    if (length == 0 || length > (size_t)60 * KO_SAMPLING_RATE)
        return -1;
    f->data = (double *)calloc(length, sizeof(double));
    if (!f->data)
        return -1;
    f->length = length;
    f->position = 0;
    f->alpha = alpha;
    return 0;
}

void ko_ring_filter_free(ko_ring_filter *f)
{
    free(f->data);
    f->data = NULL;
}

/* _filters.c:231-254 (forward), :257-280 (backward), :283-306 (echo);
 * peek :77-81, append :87-92. */
void ko_ring_filter_run(ko_ring_filter *f, int kind, const double *x, size_t n, double *y)
{
    for (size_t i = 0; i < n; ++i) {
        const double old = f->data[f->position];
        double store;
        if (kind == KO_COMB_FORWARD) {
            y[i] = x[i] + f->alpha * old;
            store = x[i];
        } else if (kind == KO_COMB_BACKWARD) {
            y[i] = x[i] + f->alpha * old;
            store = y[i];
        } else {
            y[i] = old;
            store = f->alpha * y[i] + x[i];
        }
        f->data[f->position++] = store;
        f->position %= f->length;
    }
}

/* ------------------------------------------------------------------------ */
static int ko_is_prime(int v)
{
    if (v < 2)
        return 0;
    for (int d = 2; (long)d * d <= v; ++d)
        if (v % d == 0)
            return 0;
    return 1;
}

/* primes.py:124-141: next n primes >= start. */
void ko_next_primes(int n, int start, int *out)
{
    int v = start;
    for (int i = 0; i < n; ++i) {
        while (!ko_is_prime(v))
            ++v;
        out[i] = v;
        ++v;
    }
}

/* effects.py:592-606 */
double ko_reverb_alpha(double delay_time, double decay)
{
    if (decay <= 0.)
        return 0.;
    return exp(-KO_PI * delay_time / decay);
}

/* effects.py:575-590 */
int ko_reverb_init(ko_reverb *r, double decay, double pre_delay, double dry_wet, int n_echos)
{
    int *taps = (int *)malloc(sizeof(int) * (size_t)n_echos);
    r->n_echos = n_echos;
    r->dry_wet = fmin(fmax(dry_wet, 0.), 1.);
    r->filters = (ko_ring_filter *)calloc((size_t)n_echos, sizeof(ko_ring_filter));
    ko_next_primes(n_echos, (int)(pre_delay * KO_SAMPLING_RATE), taps);
    for (int k = 0; k < n_echos; ++k) {
        const double alpha = ko_reverb_alpha((double)taps[k] / KO_SAMPLING_RATE, decay);
        if (ko_ring_filter_init(&r->filters[k], (size_t)taps[k], alpha) != 0) {
            free(taps);
            return -1;
        }
    }
    free(taps);
    return 0;
}

void ko_reverb_free(ko_reverb *r)
{
    for (int k = 0; k < r->n_echos; ++k)
        ko_ring_filter_free(&r->filters[k]);
    free(r->filters);
    r->filters = NULL;
}

/* effects.py:608-615: y = sum_k comb_k(x) (in tap order); y /= n;
 * out = (1-dw)*x + dw*y  (math.py:68-75). */
void ko_reverb_run(ko_reverb *r, const double *x, size_t n, double *out)
{
    double *y = (double *)malloc(sizeof(double) * n);
    double *acc = (double *)malloc(sizeof(double) * n);
    for (int k = 0; k < r->n_echos; ++k) {
        ko_ring_filter_run(&r->filters[k], KO_COMB_BACKWARD, x, n, y);
        if (k == 0)
            memcpy(acc, y, sizeof(double) * n);
        else
            for (size_t i = 0; i < n; ++i)
                acc[i] += y[i];
    }
    for (size_t i = 0; i < n; ++i) {
        const double wet = acc[i] / (double)r->n_echos;
        out[i] = (1. - r->dry_wet) * x[i] + r->dry_wet * wet;
    }
    free(y);
    free(acc);
}

/* ------------------------------------------------------------------------ */
/* mixer.py:41-49 */
void ko_mixer(const double *inputs, const double *gains, int n, size_t len, double *out)
{
    for (size_t i = 0; i < len; ++i)
        out[i] = 0.;
    for (int c = 0; c < n; ++c)
        for (size_t i = 0; i < len; ++i)
            out[i] += gains[c] * inputs[(size_t)c * len + i];
    if (n > 1)
        for (size_t i = 0; i < len; ++i)
            out[i] = out[i] / (double)n;
}

/* panning.py:27-31 */
static double ko_pan_law_exponent(double pan_law, double center)
{
    return (pan_law - 20. * log10(2.)) / (2. * 20. * log10(center));
}

/* panning.py:35-81 */
void ko_panning_amplitudes(double level, int mode, int has_pan_law, double pan_law,
                           double lr[2])
{
    double left, right, alpha = 1.;
    level = fmin(fmax(level, -1.), 1.);
    const double angle = KO_TAU / 8 * (1. + level);
    if (mode == 0) {
        left = .5 * (1. - level);
        right = .5 * (1. + level);
    } else if (mode == 1) {
        left = cos(angle);
        right = sin(angle);
    } else {
        left = sqrt(.5 * (1. - level) * cos(angle));
        right = sqrt(.5 * (1. + level) * sin(angle));
    }
    if (has_pan_law) {
        if (mode == 0)
            alpha = ko_pan_law_exponent(pan_law, .5);
        else if (mode == 1)
            alpha = ko_pan_law_exponent(pan_law, cos(KO_TAU / 8));
        else
            alpha = ko_pan_law_exponent(pan_law, sqrt(.5 * cos(KO_TAU / 8)));
        left = pow(left, alpha);
        right = pow(right, alpha);
    }
    lr[0] = left;
    lr[1] = right;
}

/* mixer.py:95-104: sum += (gain * pan) * x per channel, / n if n > 1. */
void ko_stereo_mixer(const double *inputs, const double *gains, const double *pan_lr, int n,
                     size_t len, double *out)
{
    for (size_t i = 0; i < 2 * len; ++i)
        out[i] = 0.;
    for (int c = 0; c < n; ++c) {
        const double gl = gains[c] * pan_lr[2 * c + 0];
        const double gr = gains[c] * pan_lr[2 * c + 1];
        for (size_t i = 0; i < len; ++i) {
            out[i] += gl * inputs[(size_t)c * len + i];
            out[len + i] += gr * inputs[(size_t)c * len + i];
        }
    }
    if (n > 1)
        for (size_t i = 0; i < 2 * len; ++i)
            out[i] = out[i] / (double)n;
}

/* Filter.update for one voice and one block (effects.py:371-385), with the optional
 * Lfo(shape=1) on Filter.frequency (oscillators.py:175-181, effects.py:271-274, 365-373). */
static void ko_filter_stage(ko_bank *bk, int v, double *buf)
{
    if (bk->cutoff_lfo) {
        const double p = bk->lfo_phase[v];
        double w;
        bk->lfo_phase[v] = ko_phasor_step(bk->lfo_frequency[v], p);
        ko_wave(bk->lfo_wave[v], &p, 1, .5, &w);
        const double cutoff = bk->lfo_scale[v] * w + bk->lfo_offset[v];
        bk->coef_index[v] = ko_filter_grid_index(bk->grid, KO_GRID_POINTS, cutoff);
    }
    const double *c = bk->coef_table + 5 * (size_t)bk->coef_index[v];
    const double b[3] = {c[0], c[1], c[2]};
    const double a[3] = {1., c[3], c[4]};
    double z[2] = {bk->z0[v], bk->z1[v]};
    ko_lfilter2(b, a, buf, KO_BUFFER_SIZE, z, buf);
    bk->z0[v] = z[0];
    bk->z1[v] = z[1];
}

/* ------------------------------------------------------------------------ */
/* One voice, one block: voices.py:55-74 (note handling + (amp*osc)*env),
 * oscillators.py:130-133, envelopes.py:40-45, effects.py:371-385. */
static void ko_voice_block(ko_bank *bk, int v, double *buf /* 256 */)
{
    double phase[KO_BUFFER_SIZE], env[KO_BUFFER_SIZE], mod[KO_BUFFER_SIZE];
    if (bk->mod_target != KO_MOD_NONE) {
        /* the modulator: Oscillator.sample / Lfo.sample with shape == BUFFER_SIZE
         * (oscillators.py:130-133, 178-181): scale * wave(phase) + offset */
        ko_sample_phase_const(bk->mod_frequency[v], bk->mod_phase[v], phase, &bk->mod_phase[v]);
        ko_wave(bk->mod_wave[v], phase, KO_BUFFER_SIZE, .5, mod);
        if (bk->mod_target != KO_MOD_FM)
            for (int i = 0; i < KO_BUFFER_SIZE; ++i)
                mod[i] = bk->mod_scale[v] * mod[i] + bk->mod_offset[v];
    }
    if (bk->wave[v] == KO_WAVE_ONE) {
        for (int i = 0; i < KO_BUFFER_SIZE; ++i)
            buf[i] = 1.;
    } else if (bk->mod_target == KO_MOD_FM) {
        /* oscillators.py:215-220: wave_func(phase + intensity * modSamples) */
        ko_sample_phase_const(bk->frequency[v], bk->phase[v], phase, &bk->phase[v]);
        for (int i = 0; i < KO_BUFFER_SIZE; ++i)
            phase[i] = phase[i] + bk->mod_scale[v] * mod[i];
        ko_wave(bk->wave[v], phase, KO_BUFFER_SIZE, bk->duty[v], buf);
    } else if (bk->mod_target == KO_MOD_FREQUENCY) {
        ko_sample_phase_array(mod, bk->phase[v], phase, &bk->phase[v]);
        ko_wave(bk->wave[v], phase, KO_BUFFER_SIZE, bk->duty[v], buf);
    } else if (bk->mod_target == KO_MOD_DUTY) {
        /* oscillators.py:256-260 with an array dutyCycle */
        ko_sample_phase_const(bk->frequency[v], bk->phase[v], phase, &bk->phase[v]);
        for (int i = 0; i < KO_BUFFER_SIZE; ++i)
            buf[i] = 2. * (phase[i] < KO_TAU * mod[i] ? 1. : 0.) - 1.;
    } else {
        ko_sample_phase_const(bk->frequency[v], bk->phase[v], phase, &bk->phase[v]);
        ko_wave(bk->wave[v], phase, KO_BUFFER_SIZE, bk->duty[v], buf);
    }
    if (bk->has_env) {
        ko_envelope_sample(&bk->env[v], KO_BUFFER_SIZE, env);
        for (int i = 0; i < KO_BUFFER_SIZE; ++i)
            buf[i] = bk->amplitude[v] * buf[i] * env[i];
    }
    if (bk->n_fx == 0) {
        if (bk->has_filter)
            ko_filter_stage(bk, v, buf);
        return;
    }
    const size_t n = (size_t)bk->n_voices;
    for (int s = 0; s < bk->n_fx; ++s) {
        const double *par = bk->fx_par + (size_t)s * KO_FX_PARAMS * n + v; /* par[k * n] */
        double *state = bk->fx_state + (size_t)s * 2 * n + v;
        switch (bk->fx_kind[s]) {
        case KO_FX_FILTER:
            ko_filter_stage(bk, v, buf);
            break;
        case KO_FX_GAIN: /* effects.py:67-70 */
            for (int i = 0; i < KO_BUFFER_SIZE; ++i)
                buf[i] = par[0] * buf[i];
            break;
        case KO_FX_TRANSFORM: /* effects.py:558-561 */
            for (int i = 0; i < KO_BUFFER_SIZE; ++i)
                buf[i] = par[0] * buf[i] + par[n];
            break;
        case KO_FX_TANH: /* effects.py:480-483 */
            for (int i = 0; i < KO_BUFFER_SIZE; ++i)
                buf[i] = tanh(par[0] * buf[i]);
            break;
        case KO_FX_OCTAVE: /* effects.py:460-462 */
            for (int i = 0; i < KO_BUFFER_SIZE; ++i)
                buf[i] = 2. * (buf[i] * buf[i]) - 1.;
            break;
        case KO_FX_TREMOLO: { /* effects.py:117-133 */
            const double rate = par[0], depth = par[n], smooth = par[2 * n], duty = par[3 * n];
            const double lim = duty < 1. - duty ? duty : 1. - duty;
            double decay = smooth * lim;
            if (!(decay > 1e-6))
                decay = 1e-6;
            double b[1], a[2], pwm[KO_BUFFER_SIZE];
            ko_low_pass_coefficients(1. / decay, b, a);
            ko_sample_phase_const(rate, state[0], phase, &state[0]);
            ko_wave(KO_WAVE_PWM, phase, KO_BUFFER_SIZE, duty, pwm);
            ko_lfilter1(b, a, pwm, KO_BUFFER_SIZE, &state[n], pwm);
            for (int i = 0; i < KO_BUFFER_SIZE; ++i) {
                const double e = (1. - depth) + depth * (.5 + .5 * pwm[i]);
                buf[i] = e * buf[i];
            }
            break;
        }
        case KO_FX_DELAY: { /* effects.py:166-174 + ring_buffer.py:62-109, for any length >= 1.
            The line is the RingBuffer's data array: `length` slots and a tail of one buffer.
            peek() returns a VIEW of data[pos : pos + 256]; extend() then writes the new
            values there, copies those that fall into the first buffer's worth of slots to the
            tail and those that run over the end to the head -- three slice assignments in
            this order -- and the blend reads the view afterwards.  For length >= 256 that is
            "wet = new + feedback * delayed"; for shorter lines the copies land inside the
            window (fixtures delay_short_* from the real reference). */
            const int length = (int)(par[0] * KO_SAMPLING_RATE);
            const double feedback = par[n], drywet = par[2 * n];
            double *ring = bk->fx_ring + (size_t)v * bk->fx_ring_capacity;
            const int start = (int)state[0], stop = start + KO_BUFFER_SIZE;
            double values[KO_BUFFER_SIZE];
            for (int i = 0; i < KO_BUFFER_SIZE; ++i)
                values[i] = buf[i] + feedback * ring[start + i];
            for (int i = 0; i < KO_BUFFER_SIZE; ++i) /* ring_buffer.py:95-97 */
                ring[start + i] = values[i];
            if (start < KO_BUFFER_SIZE) { /* ring_buffer.py:98-102: copy to the tail */
                const int a = start + length;
                const int b = length + KO_BUFFER_SIZE < stop + length ? length + KO_BUFFER_SIZE : stop + length;
                for (int k = 0; k < b - a; ++k)
                    ring[a + k] = values[k];
            }
            if (stop > length) { /* ring_buffer.py:104-108: copy to the head */
                const int a = start - length > 0 ? start - length : 0;
                const int b = stop - length;
                for (int k = 0; k < b - a; ++k)
                    ring[a + k] = values[KO_BUFFER_SIZE - (b - a) + k];
            }
            state[0] = (double)(stop % length);
            for (int i = 0; i < KO_BUFFER_SIZE; ++i) /* math.py:68-75 blend(new, view) */
                buf[i] = (1. - drywet) * buf[i] + drywet * ring[start + i];
            break;
        }
        case KO_FX_SUBSAMPLE: { /* effects.py:400-415: np.repeat(x[::skip], skip) */
            const int skip = (int)par[0];
            for (int i = 0; i < KO_BUFFER_SIZE; ++i)
                buf[i] = buf[i - i % skip];
            break;
        }
        case KO_FX_BITCRUSH: { /* effects.py:434-447 with wavfile.py:14-30: float64 ->
            int16 (C cast: truncation, low 16 bits), arithmetic shifts, / 32768 */
            const int bits = (int)par[0];
            for (int i = 0; i < KO_BUFFER_SIZE; ++i) {
                const int16_t q = (int16_t)(int32_t)(32767. * buf[i]);
                const int16_t c = (int16_t)((int16_t)(q >> bits) << bits);
                buf[i] = (double)c / 32768.;
            }
            break;
        }
        case KO_FX_RINGMOD: { /* effects.py:618-662: lfo -> modulator frequency array ->
            sine AM signal -> blend(x, am * x, dryWet) */
            const double frequency = par[0], rate = par[n], amount = par[2 * n], drywet = par[3 * n];
            double lfo[KO_BUFFER_SIZE], am[KO_BUFFER_SIZE];
            ko_sample_phase_const(rate, state[0], phase, &state[0]);
            ko_wave((int)par[4 * n], phase, KO_BUFFER_SIZE, .5, lfo);
            for (int i = 0; i < KO_BUFFER_SIZE; ++i)
                lfo[i] = frequency + amount * lfo[i];
            ko_sample_phase_array(lfo, state[n], phase, &state[n]);
            ko_wave(KO_WAVE_SINE, phase, KO_BUFFER_SIZE, .5, am);
            for (int i = 0; i < KO_BUFFER_SIZE; ++i)
                buf[i] = (1. - drywet) * buf[i] + drywet * (am[i] * buf[i]);
            break;
        }
        default:
            break;
        }
    }
}

static void ko_apply_event(ko_bank *bk, const ko_event *ev)
{
    const int v = (int)ev->voice;
    if (ev->kind == 2) {
        bk->coef_index[v] = (int32_t)ev->index;
        return;
    }
    if (bk->has_env)
        ko_envelope_gate(&bk->env[v], ev->kind == 1);
    if (ev->kind == 1) {
        bk->amplitude[v] = ev->velocity;
        bk->frequency[v] = ev->frequency;
    }
}

static void ko_render_range(ko_bank *bk, int v0, int v1, int n_blocks, const ko_event *events,
                            size_t n_events, double *bus, double *voices)
{
    const int ch = bk->stereo ? 2 : 1;
    const size_t span = (size_t)n_blocks * KO_BUFFER_SIZE;
    double buf[KO_BUFFER_SIZE];
    size_t e = 0;
    while (e < n_events && (int)events[e].voice < v0)
        ++e;
    for (int v = v0; v < v1; ++v) {
        for (int blk = 0; blk < n_blocks; ++blk) {
            while (e < n_events && (int)events[e].voice == v && (int)events[e].block == blk)
                ko_apply_event(bk, &events[e++]);
            ko_voice_block(bk, v, buf);
            if (voices)
                memcpy(voices + (size_t)v * span + (size_t)blk * KO_BUFFER_SIZE, buf, sizeof(buf));
            double *out = bus + (size_t)blk * ch * KO_BUFFER_SIZE;
            if (bk->stereo) {
                const double gl = bk->gain[v] * bk->pan_l[v];
                const double gr = bk->gain[v] * bk->pan_r[v];
                for (int i = 0; i < KO_BUFFER_SIZE; ++i) {
                    out[i] += gl * buf[i];
                    out[KO_BUFFER_SIZE + i] += gr * buf[i];
                }
            } else {
                const double g = bk->gain[v];
                for (int i = 0; i < KO_BUFFER_SIZE; ++i)
                    out[i] += g * buf[i];
            }
        }
        while (e < n_events && (int)events[e].voice == v)
            ++e; /* events beyond the rendered span */
    }
}

void ko_bank_render(ko_bank *bk, int n_blocks, const ko_event *events, size_t n_events,
                    double *mix, double *voices, int n_threads)
{
    const int ch = bk->stereo ? 2 : 1;
    const size_t bus_len = (size_t)n_blocks * ch * KO_BUFFER_SIZE;
    memset(mix, 0, sizeof(double) * bus_len);
    if (n_threads <= 1) {
        ko_render_range(bk, 0, bk->n_voices, n_blocks, events, n_events, mix, voices);
    } else {
#ifdef _OPENMP
        double *partial = (double *)calloc(bus_len * (size_t)n_threads, sizeof(double));
#pragma omp parallel for num_threads(n_threads) schedule(static, 1)
        for (int t = 0; t < n_threads; ++t) {
            const int v0 = (int)((long)bk->n_voices * t / n_threads);
            const int v1 = (int)((long)bk->n_voices * (t + 1) / n_threads);
            ko_render_range(bk, v0, v1, n_blocks, events, n_events, partial + bus_len * (size_t)t,
                            voices);
        }
        for (int t = 0; t < n_threads; ++t)
            for (size_t i = 0; i < bus_len; ++i)
                mix[i] += partial[bus_len * (size_t)t + i];
        free(partial);
#else
        ko_render_range(bk, 0, bk->n_voices, n_blocks, events, n_events, mix, voices);
#endif
    }
    if (bk->n_voices > 1)
        for (size_t i = 0; i < bus_len; ++i)
            mix[i] = mix[i] / (double)bk->n_voices;
}

/* ------------------------------------------------------------------------ */
/* synthesizer.py:151-219: MonophonicSynthesizer / PolyphonicSynthesizer over a bank of
 * n_instruments * poly Voice(oscillator, envelope) voices; instrument i owns voices
 * [i*poly, (i+1)*poly).  Per buffer and instrument:
 *   Synthesizer.update (:82-84): every note of this buffer, in order, is routed --
 *     poly (:200-209): note-on -> next voice of the itertools.cycle (round robin),
 *                      note-off -> every voice whose currentPitch == note.pitch
 *                      (currentPitch as the voices' LAST update left it: the voices
 *                      process their queues only afterwards, voices.py:55-64);
 *     mono (:170-172): every note (already chosen by the host's NoteScheduler) -> the voice;
 *   then only ACTIVE voices are updated (:177-179, :214-217): Voice.active
 *   (voices.py:45-53) = pending notes or envelope stage != OFF; an inactive voice keeps
 *   its oscillator phase.  poly: out = (sum of the active voices, in order) / poly (:219,
 *   also for poly == 1); mono: out = the voice's samples, or silence. */
void ko_synth_render(ko_bank *bk, int poly, int mono, int32_t *round_robin,
                     int32_t *current_pitch, int n_blocks, const ko_note *notes, size_t n_notes,
                     double *out, double *voices)
{
    const int n_inst = bk->n_voices / poly;
    const size_t span = (size_t)n_blocks * KO_BUFFER_SIZE;
    double buf[KO_BUFFER_SIZE];
    size_t e = 0;
    memset(out, 0, sizeof(double) * span * (size_t)n_inst);
    if (voices)
        memset(voices, 0, sizeof(double) * span * (size_t)bk->n_voices);
    for (int inst = 0; inst < n_inst; ++inst) {
        const int v0 = inst * poly;
        while (e < n_notes && (int)notes[e].instrument < inst)
            ++e;
        for (int blk = 0; blk < n_blocks; ++blk) {
            size_t e1 = e;
            while (e1 < n_notes && (int)notes[e1].instrument == inst && (int)notes[e1].block == blk)
                ++e1;
            /* routing: target[k - e] = voice bit mask would overflow for poly > 64, so keep
             * one (note, voice) list in delivery order */
            const size_t cap = (e1 - e) * (size_t)poly + 1;
            int *who = (int *)malloc(sizeof(int) * cap);
            size_t *which = (size_t *)malloc(sizeof(size_t) * cap);
            size_t n_del = 0;
            for (size_t k = e; k < e1; ++k) {
                if (mono) {
                    who[n_del] = v0, which[n_del++] = k;
                } else if (notes[k].velocity > 0.) { /* messages.py:49-52 Note.on */
                    who[n_del] = v0 + round_robin[inst], which[n_del++] = k;
                    round_robin[inst] = (round_robin[inst] + 1) % poly;
                } else {
                    for (int j = 0; j < poly; ++j)
                        if (current_pitch[v0 + j] == notes[k].pitch)
                            who[n_del] = v0 + j, which[n_del++] = k;
                }
            }
            double *dst = out + (size_t)inst * span + (size_t)blk * KO_BUFFER_SIZE;
            for (int j = 0; j < poly; ++j) {
                const int v = v0 + j;
                int pending = 0;
                for (size_t d = 0; d < n_del; ++d)
                    pending += who[d] == v;
                if (!pending && bk->env[v].stage == KO_OFF)
                    continue;
                for (size_t d = 0; d < n_del; ++d) { /* voices.py:55-64 */
                    if (who[d] != v)
                        continue;
                    const ko_note *nt = &notes[which[d]];
                    const int on = nt->velocity > 0.;
                    ko_envelope_gate(&bk->env[v], on);
                    if (on) {
                        bk->amplitude[v] = nt->velocity;
                        bk->frequency[v] = nt->frequency;
                        current_pitch[v] = nt->pitch;
                    } else {
                        current_pitch[v] = 0;
                    }
                }
                ko_voice_block(bk, v, buf);
                if (voices)
                    memcpy(voices + (size_t)v * span + (size_t)blk * KO_BUFFER_SIZE, buf, sizeof(buf));
                for (int i = 0; i < KO_BUFFER_SIZE; ++i)
                    dst[i] += buf[i];
            }
            if (!mono)
                for (int i = 0; i < KO_BUFFER_SIZE; ++i)
                    dst[i] = dst[i] / (double)poly;
            free(who);
            free(which);
            e = e1;
        }
        while (e < n_notes && (int)notes[e].instrument == inst)
            ++e;
    }
}

/* ------------------------------------------------------------------------ */
/* The output side of an offline bounce: klanggeber.py:81-97 pack_signals (float64 ->
 * float32, interleaved), :205-209, 243-249 fade-out (np.interp over [fadeOutStart,
 * duration] at playTime + T, only for buffers with playTime >= fadeOutStart; playTime
 * is the running float64 sum of INTERVAL), then wavfile.py:54-63 write_wave with
 * math.py:14-21 normalize_values: float32 audio / its float32 peak * 32767 (float32
 * arithmetic, NumPy keeps the array's dtype against Python scalars), truncated to int16.
 * bus: [n_blocks][channels][256] float32 (what the Dac's inputs hold, rounded once).
 * frames: [n_blocks*256][channels] float32.  pcm may be NULL.  Returns the peak
 * (0: silence -- the reference raises ValueError, pcm is then left untouched). */
float ko_bounce(const float *bus, int n_blocks, int channels, double duration, double fadeout,
                float *frames, int16_t *pcm)
{
    const double dt = 1. / KO_SAMPLING_RATE;
    const double fade_start = duration - fadeout;
    const double slope = (0. - 1.) / (duration - fade_start); /* numpy arr_interp */
    double play_time = 0.;
    for (int b = 0; b < n_blocks; ++b) {
        for (int i = 0; i < KO_BUFFER_SIZE; ++i) {
            double env = 1.;
            const int fading = play_time >= fade_start;
            if (fading) {
                const double x = play_time + dt * (double)i; /* helpers.py:30 T */
                if (x > duration)
                    env = 0.; /* right */
                else if (x < fade_start)
                    env = 1.; /* left */
                else if (x == duration)
                    env = 0.;
                else
                    env = slope * (x - fade_start) + 1.;
            }
            for (int c = 0; c < channels; ++c) {
                float v = bus[((size_t)b * channels + c) * KO_BUFFER_SIZE + i];
                if (fading)
                    v = (float)((double)v * env); /* float32 array *= float64 array */
                frames[((size_t)b * KO_BUFFER_SIZE + i) * channels + c] = v;
            }
        }
        play_time += KO_INTERVAL;
    }
    const size_t total = (size_t)n_blocks * KO_BUFFER_SIZE * channels;
    float peak = 0.f;
    for (size_t k = 0; k < total; ++k) {
        const float a = fabsf(frames[k]);
        if (a > peak)
            peak = a;
    }
    if (pcm && peak != 0.f)
        for (size_t k = 0; k < total; ++k)
            pcm[k] = (int16_t)(frames[k] / peak * 32767.f);
    return peak;
}

/* ------------------------------------------------------------------------ */
/* synthesizer.py:22-55, 253-283 Kick with waves.py:56-61 sample_wave: exponential amplitude
 * and pitch decay over a sine.  sample_wave returns wrap(phase[-1] + TAU * frequency * DT)
 * with the frequency ARRAY, so after the first buffer the kick's currentPhase is a vector of
 * 256 start phases, one per sample position; a trigger (note-on with pitch > 0) resets it
 * to the scalar 0 and currentTime to 0. */
void ko_kick_block(ko_kick *k, int trigger, double *out)
{
    const double dt = 1. / KO_SAMPLING_RATE;
    if (trigger) {
        k->current_time = 0.;
        for (int i = 0; i < KO_BUFFER_SIZE; ++i)
            k->current_phase[i] = 0.;
    }
    const double cd = -KO_PI / k->decay, cp = -KO_PI / k->pitch_decay;
    const double amp_d = exp(cd * k->current_time), amp_p = exp(cp * k->current_time);
    double w[KO_BUFFER_SIZE], last = 0.;
    for (int i = 0; i < KO_BUFFER_SIZE; ++i) {
        const double t = dt * (double)i;
        const double env = amp_d * exp(cd * t);
        const double pitch = k->frequency * (1. + k->intensity * (amp_p * exp(cp * t)));
        w[i] = KO_TAU * pitch;
        const double phase = w[i] * t + k->current_phase[i];
        out[i] = env * sin(phase);
        last = phase;
    }
    for (int i = 0; i < KO_BUFFER_SIZE; ++i)
        k->current_phase[i] = ko_wrap(last + w[i] * dt);
    k->current_time = k->current_time + dt * KO_BUFFER_SIZE;
}

/* ------------------------------------------------------------------------ */
/* oscillators.py:280-308 Chirper over chirp_phase (oscillators.py:18-38), which is
 * scipy.signal's _chirp_phase with a held frequency from tEnd on (scipy 1.15.3
 * scipy/signal/_waveforms.py, vertex_zero=True; un-vendored: restated from its source as
 * installed in the survey container).  The chirper's time vector starts as T and gains
 * DT * BUFFER_SIZE per element and buffer (a float64 running sum per element). */
void ko_chirp_block(ko_chirper *c, double *out)
{
    const double f0 = c->f0, t1 = c->t1, f1 = c->f1;
    double phase[KO_BUFFER_SIZE];
    /* oscillators.py:23-25: past the sweep (or equal ends): constant frequency, no offset */
    const int constant = (t1 <= c->t[0]) || (f0 == f1);
    for (int i = 0; i < KO_BUFFER_SIZE; ++i) {
        if (constant) {
            phase[i] = KO_TAU * f1 * c->t[i];
            continue;
        }
        const double t = c->t[i];
        double p;
        switch (c->method) {
        case KO_CHIRP_QUADRATIC: {
            const double beta = (f1 - f0) / (t1 * t1);       /* t1 ** 2 */
            p = KO_TAU * (f0 * t + beta * pow(t, 3.) / 3.);
            break;
        }
        case KO_CHIRP_LOGARITHMIC:
            if (f0 == f1) {
                p = KO_TAU * f0 * t;
            } else {
                const double beta = t1 / log(f1 / f0);
                p = KO_TAU * beta * f0 * (pow(f1 / f0, t / t1) - 1.0);
            }
            break;
        case KO_CHIRP_HYPERBOLIC:
            if (f0 == f1) {
                p = KO_TAU * f0 * t;
            } else {
                const double sing = -f1 * t1 / (f0 - f1);
                p = KO_TAU * (-sing * f0) * log(fabs(1 - t / sing));
            }
            break;
        default: { /* linear */
            const double beta = (f1 - f0) / t1;
            p = KO_TAU * (f0 * t + 0.5 * beta * t * t);
            break;
        }
        }
        phase[i] = p;
    }
    if (!constant && !(c->t[KO_BUFFER_SIZE - 1] < t1)) {
        /* oscillators.py:33-37: the buffer straddles tEnd; from the first sample at or after
         * it the frequency is held and the sweep's phase there carries on */
        int mid = 0;
        while (!(c->t[mid] >= t1))
            ++mid;
        const double offset = phase[mid] - KO_TAU * f1 * c->t[mid];
        for (int i = mid; i < KO_BUFFER_SIZE; ++i)
            phase[i] = KO_TAU * f1 * c->t[i] + offset;
    }
    ko_wave(c->wave, phase, KO_BUFFER_SIZE, .5, out);
    const double step = (1. / KO_SAMPLING_RATE) * KO_BUFFER_SIZE; /* oscillators.py:308 */
    for (int i = 0; i < KO_BUFFER_SIZE; ++i)
        c->t[i] = c->t[i] + step;
}

void ko_chirp_reset(ko_chirper *c)
{
    for (int i = 0; i < KO_BUFFER_SIZE; ++i)
        c->t[i] = (1. / KO_SAMPLING_RATE) * (double)i; /* helpers.py:30 */
}

int ko_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
