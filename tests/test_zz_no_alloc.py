"""klmi_bank_render_device_noalloc, the entry point that allocates nothing (it has no
counterpart in the reference): through the C-ABI, against klmi_bank_render_device on the same
patch, bit for bit -- the bus is integer-exact the same because both run the same kernels and
the same fixed-order fold; only the workspace differs (the library's static scratch array)."""
import numpy as np
import pytest

from klang_amd import patches as P
from tests.helpers import B, hip_bank_from_patch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


def _render_pair(hip, ctx, patch, spans, **kw):
    """The patch through both entry points, span by span (state carried); -> (noalloc, device)."""
    import torch
    out = []
    for entry in ('render_device_noalloc', 'render_device'):
        bank = hip_bank_from_patch(hip, ctx, patch, **kw)
        if entry == 'render_device_noalloc':
            # the event queue reaches the device with the first allocating call; a preview
            # is one (it leaves state and queue untouched)
            bank.render_voices(1, first=0, count=1)
        parts = []
        for nb in spans:
            bus = torch.full((nb, bank.channels, B), 3., dtype=torch.float32, device='cuda')
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info()[0]
            getattr(bank, entry)(nb, bus.data_ptr())
            ctx.synchronize()
            if entry == 'render_device_noalloc':
                # hipMemGetInfo: the call took no device memory (a stray reserve() would show)
                assert torch.cuda.mem_get_info()[0] == free_before
            parts.append(bus.cpu().numpy())
        out.append((np.concatenate(parts), bank.get_state(hip.F_PHASE), bank.get_state(hip.F_Z0)))
        bank.close()
    return out


def test_noalloc_equals_render_device_two_calls_second_smaller(hip, ctx):
    """Two calls in a row on one GPU, the second with a smaller input (fewer voices, fewer
    blocks, stereo instead of mono): the scratch array keeps the first call's contents and
    the second must not see them."""
    big = P.polyphonic_saw_patch(n_voices=20000, n_blocks=40, off_base=10, off_span=20)
    (got, ph, z0), (want, wph, wz0) = _render_pair(hip, ctx, big, (24, 16))
    assert np.abs(want).max() > 1e-3
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(ph, wph)
    np.testing.assert_array_equal(z0, wz0)
    small = P.polyphonic_saw_patch(n_voices=700, n_blocks=9, off_base=2, off_span=4)
    small['stereo'] = True
    small['pan'] = np.linspace(-1., 1., 700)
    (got, ph, z0), (want, wph, wz0) = _render_pair(hip, ctx, small, (9,))
    assert got.shape == (9, 2, B) and np.abs(want).max() > 1e-3
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(ph, wph)
    np.testing.assert_array_equal(z0, wz0)


def test_noalloc_splits_long_calls_and_rejects_what_does_not_fit(hip, ctx):
    import torch
    # 65,536 voices: 1 MiB + 64 KiB of scratch per buffer, so 100 buffers take two launches
    patch = P.polyphonic_saw_patch(n_voices=65536, n_blocks=100, off_base=30, off_span=40)
    (got, ph, _), (want, wph, _) = _render_pair(hip, ctx, patch, (100,))
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(ph, wph)
    bus = torch.zeros((4, 1, B), dtype=torch.float32, device='cuda')
    # a reverb bank is refused, loudly
    rv = hip.Bank(ctx, 8, has_env=True, has_reverb=True)
    with pytest.raises(hip.NativeError) as e:
        rv.render_device_noalloc(4, bus.data_ptr())
    assert e.value.code == hip.E_STATE and 'reverb' in str(e.value)
    rv.close()
    # events queued after the last upload that need a larger device queue are refused
    bank = hip.Bank(ctx, 64, has_env=True, has_filter=True)
    bank.render_device_noalloc(1, bus.data_ptr())          # no events: nothing to upload
    for v in range(64):
        bank.event(v, 0, hip.EV_NOTE_ON, 1., 220.)
    with pytest.raises(hip.NativeError) as e:
        bank.render_device_noalloc(1, bus.data_ptr())
    assert e.value.code == hip.E_STATE and 'events' in str(e.value)
    bank.render_device(1, bus.data_ptr())                  # the allocating call takes them
    bank.event(3, 0, hip.EV_NOTE_OFF)                      # fits the uploaded capacity
    bank.render_device_noalloc(1, bus.data_ptr())
    ctx.synchronize()
    assert np.isfinite(bus.cpu().numpy()).all()
    with pytest.raises(hip.NativeError):
        bank.render_device_noalloc(0, bus.data_ptr())
    bank.close()
    # a bank whose single buffer is larger than the static array is refused with its size
    huge = hip.Bank(ctx, 2200000, has_env=False, stereo=True)
    with pytest.raises(hip.NativeError) as e:
        huge.render_device_noalloc(1, bus.data_ptr())
    assert e.value.code == hip.E_ARG and 'scratch' in str(e.value)
    huge.close()


def test_noalloc_covers_the_pipelined_kernel_and_compiled_effect_programs(hip, ctx):
    """Small banks take voice_pipe_kernel, Filter -> Tremolo -> Gain chains voice_fx_kernel:
    both through the no-allocation entry point, bit-equal to the allocating one."""
    small = P.polyphonic_saw_patch(n_voices=3000, n_blocks=20, off_base=5, off_span=10)
    (got, ph, z0), (want, wph, wz0) = _render_pair(hip, ctx, small, (12, 8))
    assert np.abs(want).max() > 1e-3
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(ph, wph)
    np.testing.assert_array_equal(z0, wz0)
    fx = P.polyphonic_saw_patch(n_voices=30000, n_blocks=12, off_base=4, off_span=6)
    kw = dict(effects=[hip.FX_FILTER, hip.FX_TREMOLO, hip.FX_GAIN])
    (got, ph, z0), (want, wph, wz0) = _render_pair(hip, ctx, fx, (7, 5), **kw)
    assert np.abs(want).max() > 1e-3
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(ph, wph)
    np.testing.assert_array_equal(z0, wz0)
