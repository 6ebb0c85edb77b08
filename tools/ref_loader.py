"""Container-only loader for the real reference (atheler/klang @ /root/reference).

TEST INFRASTRUCTURE ONLY.  Used by tools/gen_golden.py to produce the committed
fixtures under tests/golden/.  Nothing under klang_amd/ imports this, and it is
never used on the GPU box (/root/reference does not exist there).

What it does:
  * loads the reference's two CPython extensions, compiled by oracle/Makefile
    from the sources where they lie into oracle/_ref/, under the module names
    the reference expects (klang.audio._envelope / klang.audio._filters);
  * registers import stand-ins for three modules the reference star-imports
    but which are not on the DSP path and are not installed here (pyaudio,
    samplerate) and re-exposes scipy's private ``_chirp_phase`` under the
    pre-1.15 name the reference imports.  None of them touches hot-path
    arithmetic.
"""
import enum
import importlib.machinery
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
REF_BUILD = os.path.join(os.path.dirname(HERE), 'oracle', '_ref')


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'klang'))


def _load_extension(fullname, filename):
    path = os.path.join(REF_BUILD, filename)
    loader = importlib.machinery.ExtensionFileLoader(fullname, path)
    spec = importlib.util.spec_from_file_location(fullname, path, loader=loader)
    module = importlib.util.module_from_spec(spec)
    loader.exec_module(module)
    sys.modules[fullname] = module
    return module


def _install_stand_ins():
    if 'pyaudio' not in sys.modules:
        pa = types.ModuleType('pyaudio')
        pa.paFloat32, pa.paContinue, pa.paComplete, pa.paAbort = 1, 0, 1, 2
        sys.modules['pyaudio'] = pa

    if 'samplerate' not in sys.modules:
        sr = types.ModuleType('samplerate')
        conv = types.ModuleType('samplerate.converters')

        class ConverterType(enum.Enum):
            sinc_best = 0
            sinc_medium = 1
            sinc_fastest = 2
            zero_order_hold = 3
            linear = 4

        class CallbackResampler:
            def __init__(self, *args, **kwargs):
                raise RuntimeError('samplerate is not installed (stand-in)')

        conv.ConverterType = ConverterType
        sr.converters = conv
        sr.CallbackResampler = CallbackResampler
        sys.modules['samplerate'] = sr
        sys.modules['samplerate.converters'] = conv

    import scipy.signal.waveforms as public
    if not hasattr(public, '_chirp_phase'):
        import scipy.signal._waveforms as private
        public._chirp_phase = private._chirp_phase


def load_reference():
    """Import the reference package with its C extensions. Returns the
    top-level ``klang`` module."""
    if not reference_available():
        raise RuntimeError('reference tree not present at %s' % REFERENCE_ROOT)

    _install_stand_ins()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    import logging
    import klang  # noqa: F401  (the reference; configures logging at import)
    logging.getLogger().setLevel(logging.WARNING)

    # The C extensions must be registered before klang.audio is imported so
    # that envelopes.py / filters.py pick the C path (the oracle), not the
    # numerically different Python fallbacks.
    _load_extension('klang.audio._envelope', '_envelope.so')
    _load_extension('klang.audio._filters', '_filters.so')
    import klang.audio
    import klang.audio.envelopes as envs
    import klang.audio.filters as fil
    assert envs.Envelope.__module__ == 'klang.audio._envelope', envs.Envelope
    assert not fil.USE_PYTHON_FALLBACK
    return klang
