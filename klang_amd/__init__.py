"""klang_amd -- MI355X-native evaluator for klang's per-block DSP hot path."""
__version__ = '0.1.0'
