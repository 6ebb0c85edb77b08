// dsp_device.h -- float64 device primitives for klang's DSP blocks on gfx950.
//
// Built with -ffp-contract=off: every a*b+c written here is a separate multiply
// and add exactly like the reference's NumPy / C code; __builtin_fma is used
// only where an exact result is wanted (the remainder below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace klmi {

constexpr double kPi = 3.141592653589793;          // klang/constants.py:5
constexpr double kTau = 2. * 3.141592653589793;     // klang/constants.py:8
constexpr double kInvTau = 1. / kTau;
constexpr double kDt = 1. / 44100;                  // klang/audio/helpers.py:9
constexpr int kBlock = 256;                         // klang/config.py:5

enum Wave : int { W_SINE = 0, W_SQUARE, W_SAWTOOTH, W_TRIANGLE, W_PWM, W_ONE, W_EXTERNAL };
enum Stage : int { S_OFF = 0, S_ATTACK, S_DECAY, S_SUSTAIN, S_RELEASE };

// Exact fmod(x, TAU) for x >= 0 (C fmod is exact; so is this).
// k ~ floor(x/TAU) may be off by one; x - k*TAU is a multiple of ulp(TAU) below
// 8, hence representable, so the fused multiply-add returns it unrounded for
// the right k, and with the right sign / side of TAU for a neighbouring k.
__device__ __forceinline__ double fmod_tau_nonneg(double x)
{
    if (x < kTau)
        return x;
    if (!(x < 1e9))
        return fmod(x, kTau);  // huge / inf / nan: library path
    double k = floor(x * kInvTau);
    double r = __builtin_fma(-k, kTau, x);
    if (r < 0.)
        r = __builtin_fma(-(k - 1.), kTau, x);
    else if (r >= kTau)
        r = __builtin_fma(-(k + 1.), kTau, x);
    return r;
}

// The same for 0 <= x < 1e9 (checked per block by the caller): no early-outs, the
// off-by-one repair is one rarely taken branch.
__device__ __forceinline__ double fmod_tau_fast(double x)
{
    const double k = floor(x * kInvTau);
    double r = __builtin_fma(-k, kTau, x);
    if (__builtin_expect(r < 0. || r >= kTau, 0))
        r = __builtin_fma(-(r < 0. ? k - 1. : k + 1.), kTau, x);
    return r;
}

// np.mod(x, TAU) (oscillators.py:66): C fmod, then the divisor's sign.
__device__ __forceinline__ double np_mod_tau(double x)
{
    double r = fmod_tau_nonneg(fabs(x));
    if (x < 0. && r != 0.)
        r = kTau - r;  // (-r) + TAU
    return r;
}

// sin on [0, TAU] (np.sin of a wrapped phase, waves.py:22-24): fold to
// [-pi/2, pi/2], odd Taylor polynomial to x^19; |error| < 2e-15.
__device__ __forceinline__ double sin_wrapped(double p)
{
    double r = p;
    if (p > 1.5 * kPi)
        r = p - kTau;
    else if (p > .5 * kPi)
        r = kPi - p;
    const double r2 = r * r;
    double s = -8.22063524662433e-18;              // -1/19!
    s = __builtin_fma(s, r2, 2.8114572543455206e-15);   // 1/17!
    s = __builtin_fma(s, r2, -7.647163731819816e-13);   // -1/15!
    s = __builtin_fma(s, r2, 1.6059043836821613e-10);   // 1/13!
    s = __builtin_fma(s, r2, -2.505210838544172e-08);   // -1/11!
    s = __builtin_fma(s, r2, 2.7557319223985893e-06);   // 1/9!
    s = __builtin_fma(s, r2, -0.0001984126984126984);   // -1/7!
    s = __builtin_fma(s, r2, 0.008333333333333333);     // 1/5!
    s = __builtin_fma(s, r2, -0.16666666666666666);     // -1/3!
    return __builtin_fma(s * r2, r, r);
}

// (p + PI) % TAU for p in [0, TAU] (math.py:9-11 before the "- PI").
__device__ __forceinline__ double wrap_shifted(double p)
{
    const double s = p + kPi;
    return s >= kTau ? s - kTau : s;  // exact remainder for s in [TAU, 2 TAU)
}

// a / TAU, correctly rounded (== the IEEE division the reference performs):
// q = a * fl(1/TAU) is within 1 ulp; one fused residual step rounds it
// correctly (Markstein: y the correctly rounded reciprocal, q within 1 ulp).
__device__ __forceinline__ double div_tau(double a)
{
    const double q = a * kInvTau;
    const double r = __builtin_fma(-q, kTau, a);
    return __builtin_fma(r, kInvTau, q);
}

// (2 a) / TAU, correctly rounded, without forming 2a: scaling by two is exact,
// so this is div_tau(2 a) with every operand halved or doubled accordingly.
__device__ __forceinline__ double div_pi(double a)
{
    constexpr double kInvPi = 2. * kInvTau;
    const double q = a * kInvPi;
    const double r = __builtin_fma(-q, kPi, a);
    return __builtin_fma(r, kInvPi, q);
}

// The four constants of the remainder and the two divisions, as run-time values.  As
// literals the compiler notices that TAU / PI and 1/TAU / (2/TAU) share their low words,
// keeps one copy of each low word and re-assembles the four register pairs with scalar moves
// in every trip through the sample loop; handed over as opaque values (see voice_kernel) they
// are four plain register pairs.  Same numbers, same operations, same results.
struct TauConsts {
    double inv_tau, tau, inv_pi, pi;
};
__device__ __forceinline__ double div_tau(double a, const TauConsts &c)
{
    const double q = a * c.inv_tau;
    const double r = __builtin_fma(-q, c.tau, a);
    return __builtin_fma(r, c.inv_tau, q);
}
__device__ __forceinline__ double div_pi(double a, const TauConsts &c)
{
    const double q = a * c.inv_pi;
    const double r = __builtin_fma(-q, c.pi, a);
    return __builtin_fma(r, c.inv_pi, q);
}

// 0 <= p < TAU as ONE integer compare: non-negative doubles order like their
// bit patterns, and any set sign bit (negative values, -0.0) compares high.
__device__ __forceinline__ bool in_zero_tau(double p)
{
    return (unsigned long long)__double_as_longlong(p) <
           (unsigned long long)__double_as_longlong(kTau);
}

// waves.py:22-39 / oscillators.py:259-260.  `p` is the wrapped phase.
// WRAPPED: the caller guarantees 0 <= p < TAU (the np.mod(phase, TAU) of
// waves.py:34 is then the identity).
template <bool WRAPPED = false>
__device__ __forceinline__ double wave_eval(int wave, double p, double duty_edge)
{
    switch (wave) {
    case W_SINE:
        return sin_wrapped(p);
    case W_SQUARE:
        return (wrap_shifted(p) - kPi >= 0.) ? 1. : -1.;
    case W_SAWTOOTH: {
        const double m = (!WRAPPED && p >= kTau) ? p - kTau : p;  // np.mod(phase, TAU)
        return div_pi(m) - 1.;   // 2. * m / TAU - 1.
    }
    case W_TRIANGLE: {
        const double w = wrap_shifted(p) - kPi;
        double q = div_tau(4. * w);
        q = q < 0. ? q + 4. : q;                    // % 4 (NumPy sign rule)
        return 1. - fabs(q - 2.);
    }
    case W_PWM:
        return (p < duty_edge) ? 1. : -1.;
    default:
        return 1.;
    }
}

// wave_eval<true> with the constants handed over (the fast kernel's straight-line region)
__device__ __forceinline__ double wave_eval_wrapped(int wave, double p, double duty_edge,
                                                    const TauConsts &c)
{
    switch (wave) {
    case W_SINE:
        return sin_wrapped(p);
    case W_SQUARE:
        return (wrap_shifted(p) - kPi >= 0.) ? 1. : -1.;
    case W_SAWTOOTH:
        return div_pi(p, c) - 1.;   // 2. * m / TAU - 1.
    case W_TRIANGLE: {
        const double w = wrap_shifted(p) - kPi;
        double q = div_tau(4. * w, c);
        q = q < 0. ? q + 4. : q;                    // % 4 (NumPy sign rule)
        return 1. - fabs(q - 2.);
    }
    case W_PWM:
        return (p < duty_edge) ? 1. : -1.;
    default:
        return 1.;
    }
}

// np.mod(x, TAU) for finite |x| < 1e9 WITHOUT branches (selects only): the general
// kernel calls it several times per sample, and lane-divergent early-outs cost far more
// than the few operations they save.  r = |x| - k*TAU is exact for k = floor(|x|/TAU) or
// a neighbour (see fmod_tau_nonneg); the true remainder is representable too, so the
// repair by one exact +-TAU returns it unrounded.
__device__ __forceinline__ double np_mod_tau_bounded(double x)
{
    const double a = fabs(x);
    const double k = floor(a * kInvTau);
    double r = __builtin_fma(-k, kTau, a);
    r = r < 0. ? r + kTau : r;
    r = r >= kTau ? r - kTau : r;
    return (x < 0. && r != 0.) ? kTau - r : r;
}

template <bool BOUNDED> __device__ __forceinline__ double np_mod_tau_sel(double x)
{
    if constexpr (BOUNDED)
        return np_mod_tau_bounded(x);
    else
        return np_mod_tau(x);
}

// The wave functions for ANY phase argument (FmOscillator evaluates them at
// phase + intensity * modulator, oscillators.py:220; an Lfo's first phase is whatever
// startPhase was given): math.py:9-11 wrap = np.mod(p + PI, TAU) - PI.
// BOUNDED: the caller guarantees finite |p| < 1e9 - 4 (branch-free remainders).
template <bool BOUNDED = false>
__device__ __forceinline__ double wave_eval_any(int wave, double p, double duty_edge)
{
    switch (wave) {
    case W_SINE:
        return (p >= 0. && p <= kTau) ? sin_wrapped(p) : sin(p);
    case W_SQUARE:
        return (np_mod_tau_sel<BOUNDED>(p + kPi) - kPi >= 0.) ? 1. : -1.;
    case W_SAWTOOTH:
        return div_pi(np_mod_tau_sel<BOUNDED>(p)) - 1.;
    case W_TRIANGLE: {
        const double w = np_mod_tau_sel<BOUNDED>(p + kPi) - kPi;
        double q = div_tau(4. * w);
        q = q < 0. ? q + 4. : q;
        return 1. - fabs(q - 2.);
    }
    case W_PWM:
        return (p < duty_edge) ? 1. : -1.;
    default:
        return 1.;
    }
}

// _envelope.c state machine with the per-stage recurrence held in registers.
// OFF and SUSTAINING use coef 0: base + value*0 == base exactly.
struct EnvParams {
    double aCoef, aBase, dCoef, dBase, rCoef, rBase, sustain;
    bool loop, retrigger;
};

struct EnvState {
    double value;
    int stage;
    bool enabled;
    // recurrence of the current stage
    double coef, base, limit;
    bool rising;
    uint32_t flip;  // 0 when rising, the sign bit (of a high word) when falling

    __device__ __forceinline__ void select(const EnvParams &p)
    {
        switch (stage) {
        case S_ATTACK:
            coef = p.aCoef, base = p.aBase, limit = 1., rising = true;
            break;
        case S_DECAY:
            coef = p.dCoef, base = p.dBase, limit = p.sustain, rising = false;
            break;
        case S_RELEASE:
            coef = p.rCoef, base = p.rBase, limit = 0., rising = false;
            break;
        case S_SUSTAIN:
            coef = 0., base = p.sustain, limit = -__builtin_inf(), rising = false;
            break;
        default:
            coef = 0., base = 0., limit = -__builtin_inf(), rising = false;
            break;
        }
        flip = rising ? 0u : 0x80000000u;
    }

    // "value v reached the stage's limit" (v >= limit when rising, v <= limit when
    // falling) as ONE compare: d = fl(v - limit) is >= 0 resp. <= 0 exactly when v is,
    // and flipping d's sign for a falling stage turns "<= 0" into ">= 0" (-0 == +0).
    // limit = -inf (sustain, off): d = +inf, flipped -inf: never a hit.
    __device__ __forceinline__ bool reached(double v) const
    {
        const double d = v - limit;
        const double f = __hiloint2double((int)((uint32_t)__double2hiint(d) ^ flip), __double2loint(d));
        return f >= 0.;
    }

    // _envelope.c:549-567
    __device__ __forceinline__ void gate(const EnvParams &p, bool on)
    {
        if (on) {
            enabled = true;
            if (p.retrigger || stage == S_OFF || stage == S_RELEASE)
                stage = S_ATTACK;
        } else {
            enabled = false;
            if (stage == S_ATTACK || stage == S_DECAY || stage == S_SUSTAIN)
                stage = S_RELEASE;
        }
        select(p);
    }

    // _envelope.c:579-624
    __device__ __forceinline__ double step(const EnvParams &p)
    {
        double nv = base + value * coef;
        const bool hit = rising ? (nv >= limit) : (nv <= limit);
        if (hit) {
            nv = limit;
            if (stage == S_ATTACK)
                stage = S_DECAY;
            else if (stage == S_DECAY)
                stage = p.loop ? S_RELEASE : S_SUSTAIN;
            else
                stage = (p.loop && enabled) ? S_ATTACK : S_OFF;
            select(p);
        }
        value = nv;
        return nv;
    }
};

// scipy.signal.lfilter, order 2, a0 == 1 (effects.py:309-316): direct form II
// transposed, y = z0 + b0 x; z0 = z1 + b1 x - a1 y; z1 = b2 x - a2 y.
// The Butterworth low-pass numerator is b0 * [1, 2, 1] with b1 == 2 b0 and
// b2 == b0 exactly (scipy's zpk2tf and the closed form both produce it; the
// host checks the table).  Doubling is exact, so b1*x == 2*(b0*x) and
// z1 + b1*x == fma(2, b0*x, z1) bit for bit (outside the subnormal range, where
// results may differ by one unit of 2^-1074): 7 operations instead of 9.
struct Biquad {
    double b0, b1, b2, a1, a2;
    double z0, z1;
    __device__ __forceinline__ double step(double x)
    {
        const double t = b0 * x;
        const double y = z0 + t;
        z0 = __builtin_fma(2., t, z1) - a1 * y;
        z1 = t - a2 * y;
        return y;
    }
};

}  // namespace klmi
