"""Shared builders for the parity tests (test infrastructure)."""
import numpy as np

from klang_amd import patches as P

B = 256


def oracle_bank_from_patch(oracle, patch, table, grid):
    """OracleBank with the patch's parameters; `table`/`grid` choose whose
    Butterworth design is used ([rows][5] = b0 b1 b2 a1 a2)."""
    bank = oracle.OracleBank(patch.n_voices, patch.has_env, patch.has_filter, patch.stereo)
    bank.wave[:] = patch.wave
    bank.frequency[:] = patch.frequency
    bank.set_envelopes(patch.attack, patch.decay, patch.sustain, patch.release, patch.overshoot,
                       patch.retrigger, patch.loop)
    bank.coef_table = np.ascontiguousarray(table)
    bank.coef_index[:] = [oracle.filter_grid_index(grid, c) for c in patch.cutoff]
    bank.gain[:] = patch.gain
    pan = np.array([oracle.panning_amplitudes(p) for p in patch.pan])
    bank.pan_l[:], bank.pan_r[:] = pan[:, 0], pan[:, 1]
    return bank


def hip_bank_from_patch(hip, ctx, patch, **kw):
    bank = hip.Bank(ctx, patch.n_voices, has_env=patch.has_env, has_filter=patch.has_filter,
                    stereo=patch.stereo, has_reverb='reverb' in patch,
                    reverb=patch.get('reverb'), **kw)
    bank.load_patch(patch)
    return bank


def reference_table(golden):
    """scipy's own coefficient table from the fixtures, [1000][5]."""
    g = golden('filter')
    return np.ascontiguousarray(g['table'][:, [0, 1, 2, 4, 5]]), g['grid']


def library_table(hip):
    """The product's closed-form table, [1000][5] (host-side helpers, no GPU)."""
    grid = hip.filter_grid()
    rows = []
    for f in grid:
        b, a = hip.butter2_lowpass(f / 22050)
        rows.append((b[0], b[1], b[2], a[1], a[2]))
    return np.array(rows), grid


def synth_notes(oracle, rows, instrument=0):
    """Fixture rows [block, pitch, velocity, frequency] -> oracle NOTE_DTYPE array."""
    rows = np.asarray(rows, dtype=float).reshape(-1, 4)
    notes = np.zeros(len(rows), dtype=oracle.NOTE_DTYPE)
    notes['instrument'] = instrument
    notes['block'], notes['pitch'] = rows[:, 0].astype(int), rows[:, 1].astype(int)
    notes['velocity'], notes['frequency'] = rows[:, 2], rows[:, 3]
    return notes


def oracle_synth_bank(oracle, n_voices, template, wave='sawtooth'):
    """n_voices x Voice(Oscillator(wave), ADSR(*template)), as PolyphonicSynthesizer's
    duplicate_voice builds them (synthesizer.py:58-60, 197)."""
    bank = oracle.OracleBank(n_voices, has_env=True, has_filter=False)
    bank.wave[:] = oracle.WAVE_IDS[wave]
    bank.set_envelopes(*[float(t) for t in template])
    return bank


def bounce_signals(name, n_blocks, channels):
    """The seeded float64 buffers tools/gen_golden.py placed on the reference's Dac for the
    'bounce' fixtures (same recipe; the fixture stores a checksum)."""
    r = np.random.default_rng(len(name))
    return (2. * r.random((n_blocks, channels, B)) - 1.) * np.linspace(.2, 1.3, n_blocks)[:, None, None]


# ---- parity report: max abs error of the float32 output vs the reference's float64, per block
# type (north_star).  The GPU tests call report(); tests/conftest.py writes the table to
# gpurun_out/parity_report.json when the session ends.
REPORT = {}


def report(block_type, err):
    REPORT[block_type] = max(REPORT.get(block_type, 0.), float(err))
    return err


# Chirper fixtures (tests/golden/chirper.npz): name, constructor arguments, wave, tolerance.
# Linear sweeps are plain float64 arithmetic (non-sine waves bit-exact, sine within libm
# ulps); pow / log sweeps differ by libm ulps inside a phase of a few thousand radians.  The
# last four cover chirp_phase's held region (oscillators.py:18-38); all ten run past tEnd.
CHIRPS = [('linear_sine', (100., .4, 2000., 'linear'), 'sine', 1e-12),
          ('linear_saw', (2000., .3, 50., 'linear'), 'sawtooth', 0.),
          ('linear_square', (30., .5, 900., 'linear'), 'square', 0.),
          ('quadratic_sine', (100., .4, 2000., 'quadratic'), 'sine', 1e-9),
          ('logarithmic_sine', (100., .4, 2000., 'logarithmic'), 'sine', 1e-9),
          ('hyperbolic_sine', (100., .4, 2000., 'hyperbolic'), 'sine', 1e-9),
          ('equal_ends', (300., .2, 300., 'linear'), 'sine', 1e-12),
          ('no_sweep', (100., 0., 500., 'linear'), 'sawtooth', 0.),
          ('short_log', (100., .003, 900., 'logarithmic'), 'sine', 1e-9),
          ('short_hyp', (900., .01, 100., 'hyperbolic'), 'triangle', 1e-9)]
