"""CPU-side checks of the C-ABI: the library loads, exports every symbol the
header declares, its host-side design helpers agree with the reference
fixtures, and it fails loudly (no CPU fallback) where there is no GPU."""
import ctypes
import re

import numpy as np
import pytest

from klang_amd import native
from tests.conftest import has_gpu


def declared_symbols():
    text = open(native.HEADER_PATH).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(klmi_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = native.load()
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), 'libklang_mi.so does not export %s' % name
    assert sorted(native.EXPORTS) == names
    assert b'gfx950' in lib.klmi_version()


def test_struct_layouts_match_header():
    assert ctypes.sizeof(native.ChainDesc) == 96
    assert native.EVENT_DTYPE.itemsize == 32 and native.NOTE_DTYPE.itemsize == 32


def test_host_helpers_against_reference_fixtures(golden):
    g = golden('filter')
    grid = native.filter_grid()
    np.testing.assert_allclose(grid, g['grid'], rtol=2.3e-16, atol=0)
    assert grid[0] == g['grid'][0] and grid[-1] == g['grid'][-1]
    for i in (0, 137, 500, 667, 999):
        b, a = native.butter2_lowpass(g['grid'][i] / 22050)
        np.testing.assert_allclose(np.r_[b, a], g['table'][i], rtol=4e-15, atol=1e-15)
    for req, want in zip(g['index_requests'], g['index_results']):
        assert native.filter_grid_index(req) == want
    for f, (b0, a1) in zip(g['one_pole_freqs'], g['one_pole_coeffs']):
        b, a = native.low_pass_coefficients(f)
        assert abs(b[0] - b0) <= 2e-16 and abs(a[1] - a1) <= 2e-16 and a[0] == 1.
    assert native.low_pass_coefficients(0.) == ([0.], [1., -1.])
    assert native.low_pass_coefficients(float('inf')) == ([1.], [1., 0.])
    r = golden('reverb')
    taps, alphas = native.reverb_design(3., .03, 10)
    np.testing.assert_array_equal(taps, r['taps_default'])
    np.testing.assert_allclose(alphas, r['decay3_alphas'], rtol=2e-16)
    m = golden('mixer')
    for mode in ('linear', 'constant_power', 'mixture'):
        for p, want in zip(m['pan_levels'], m['pan_' + mode]):
            np.testing.assert_allclose(native.panning_amplitudes(p, mode), want, atol=2e-16)
    with pytest.raises(ValueError):
        native.panning_amplitudes(0., 'constant_power', pan_law=3.)
    with pytest.raises(ValueError):
        native.panning_amplitudes(0., 'bogus')


def test_envelope_coefficients_match_oracle(oracle):
    for args in ((.1, .2, .8, 1.), (0., 0., 1., .05), (.01, .1, .7, .5)):
        got = native.envelope_coefficients(*args, 1. / 44100, 1e-3)
        env = oracle.OracleEnvelope(*args)
        want = [env.c.attackCoef, env.c.attackBase, env.c.decayCoef, env.c.decayBase,
                env.c.releaseCoef, env.c.releaseBase]
        np.testing.assert_array_equal(got, want)


@pytest.mark.skipif(has_gpu(), reason='checks the no-GPU failure mode')
def test_no_gpu_means_loud_failure_not_fallback():
    with pytest.raises(native.NativeError) as e:
        native.Context(0)
    assert e.value.code == native.E_DEVICE
    assert 'no CPU path' in str(e.value)
