"""Host-side mirror of the reference's Block / connection / scheduler API.
Expectations restate the reference's own unit tests (tests/test_block.py,
test_connections.py, test_graph.py:140-243, test_execution.py, test_composite.py,
test_audio_mixer.py, test_audio_voices.py there); nothing here needs a GPU."""
import numpy as np
import pytest

from klang_amd.audio import (ADSR, AR, Dac, Mixer, Oscillator, StereoMixer, Voice, sawtooth)
from klang_amd.block import Block, fetch_input, fetch_output
from klang_amd.composite import Composite, temporarily_unpatch
from klang_amd.messages import Note
from klang_amd.schedule import determine_execution_order, find_back_edges, topological_order
from klang_amd.wiring import (IncompatibleConnection, Input, InputAlreadyConnected,
                              MessageInput, MessageOutput, MessageRelay, Output, Relay,
                              are_connected, is_connected)


def connected(a, b):
    out, inp = fetch_output(a), fetch_input(b)
    return out is inp.incomingConnection and inp in out.outgoingConnections


# ---- connections -------------------------------------------------------------------
def test_connect_disconnect_and_one_to_many():
    src, a, b = Output(), Input(), Input()
    src.connect(a)
    b.connect(src)
    assert is_connected(src, a) and is_connected(src, b) and src.connected
    with pytest.raises(InputAlreadyConnected):
        Output().connect(a)
    src.disconnect(a)
    assert not a.connected and not is_connected(src, a)
    with pytest.raises(IncompatibleConnection):
        MessageOutput().connect(Input())
    with pytest.raises(IncompatibleConnection):
        Output().connect(MessageInput())
    assert not is_connected(Output(), MessageInput())


def test_value_flow_through_relays():
    src, r1, r2, dst = Output(), Relay(), Relay(), Input(value=7.)
    assert dst.value == 7.
    src.connect(r1)
    r1.connect(r2)
    r2.connect(dst)
    assert are_connected(src, r1, r2, dst)
    src.set_value('payload')
    assert dst.value == 'payload' and dst.get_value() == 'payload'
    r2.disconnect(dst)
    assert dst.value == 7.


def test_message_flow_and_queue_semantics():
    src, relay, a, b = MessageOutput(), MessageRelay(), MessageInput(), MessageInput()
    src.connect(relay)
    relay.connect(a)
    src.connect(b)
    for i in range(3):
        src.send(i)
    assert list(a.receive()) == [0, 1, 2] and not a.queue
    assert b.receive_latest() == 2 and not b.queue and b.receive_latest() is None
    for i in range(MessageInput.MAX_MESSAGES + 5):
        a.push(i)
    assert len(a.queue) == MessageInput.MAX_MESSAGES


# ---- operators (tests/test_block.py) ----------------------------------------------------
def test_pipe_operator():
    with pytest.raises(AttributeError):
        Block() | Block()
    with pytest.raises(AttributeError):
        Block(0, 1) | Block()
    with pytest.raises(AttributeError):
        Block() | Block(1, 0)
    a, b = Block(0, 1), Block(1, 0)
    assert (a | b) is b and connected(a, b)
    a, b = Block(1, 1), Block(1, 1)
    assert (a | b.input) is b and connected(a, b)
    with pytest.raises(TypeError):
        Block(1, 1) | Block(1, 1).output
    with pytest.raises(TypeError):
        Block(1, 1).input | Block(1, 1)
    a, b = Block(1, 1), Block(1, 1)
    assert (a.output | b) is b and connected(a, b)


def test_mix_operator():
    with pytest.raises(AttributeError):
        Block() + Block()
    with pytest.raises(AttributeError):
        Block(0, 1) + Block()
    a, b, c = Block(0, 1), Block(0, 1), Block(0, 1)
    mixer = a + b + c
    assert isinstance(mixer, Mixer) and mixer.nInputs == 3 and mixer.gains == [1., 1., 1.]
    assert all(connected(x, mixer.inputs[i]) for i, x in enumerate((a, b, c)))
    with pytest.raises(AttributeError):
        Block(0, 1) + Block(0, 1).input
    with pytest.raises(TypeError):
        Block(1, 1).input + Block(1, 1)
    a, b, c = Block(0, 1), Block(0, 1), Block(0, 1)
    mixer = a.output + b + c.output
    assert all(connected(x, mixer.inputs[i]) for i, x in enumerate((a, b, c)))
    sm = StereoMixer()
    sm += a
    assert sm.pannings == [0.] and sm.nInputs == 1


# ---- scheduler: the reference's golden orders (tests/test_graph.py:140-243) -----------------
GOLDEN_ORDERS = [
    ([(0, 1), (1, 2)], [0, 1, 2]),
    ([(0, 1), (1, 2), (1, 3), (3, 1)], [0, 1, 3, 2]),
    ([(0, 1), (1, 2), (2, 3), (0, 4), (4, 3)], [0, 4, 1, 2, 3]),
    ([(0, 1), (1, 2), (2, 3), (0, 4), (4, 3), (2, 4)], [0, 1, 2, 4, 3]),
    ([(0, 1), (1, 2), (2, 3), (0, 4), (4, 3), (4, 2)], [0, 4, 1, 2, 3]),
    ([(0, 1), (1, 0)], [0, 1]),
    ([(0, 1), (1, 0), (2, 3), (3, 2)], [0, 1, 2, 3]),
    ([(0, 1), (1, 2), (2, 0)], [0, 1, 2]),
    ([(0, 1), (0, 2), (1, 2), (1, 3), (2, 1), (2, 3)], [0, 1, 2, 3]),
    ([(0, 1), (1, 2), (2, 6), (3, 4), (4, 5), (5, 6), (2, 4), (5, 1)], [0, 1, 2, 3, 4, 5, 6]),
    ([(0, 1), (1, 2), (2, 3), (1, 2)], [0, 1, 2, 3]),
    ([(1, 0), (2, 1), (3, 2), (4, 3), (5, 4), (4, 5)], [4, 5, 3, 2, 1, 0]),
    ([(0, 1), (0, 2), (0, 3), (1, 2), (3, 4), (4, 5), (5, 3)], [0, 3, 4, 5, 1, 2]),
]


@pytest.mark.parametrize('edges,want', GOLDEN_ORDERS)
def test_topological_order_goldens(edges, want):
    n = max(max(e) for e in edges) + 1
    assert topological_order(n, edges) == want


def test_back_edges():
    assert find_back_edges(3, [(0, 1), (1, 2), (2, 0)]) == [(2, 0)]
    assert find_back_edges(3, [(0, 1), (1, 2)]) == []


def test_execution_order_of_block_networks():
    assert determine_execution_order([]) == []
    blk = Block()
    assert determine_execution_order([blk]) == [blk]
    a, b, c = Block(1, 1), Block(1, 1), Block(1, 1)
    assert determine_execution_order([a, b, c]) == [a, b, c]
    b.output.connect(c.input)
    assert determine_execution_order([a, b]) == [a, b, c]
    x, y, z = Block(0, 1), Block(1, 1), Block(1, 0)
    x | y | z
    assert determine_execution_order([z]) == [x, y, z]


def test_composite_and_unpatch():
    src, block, d0, d1 = Block(0, 1), Block(1, 1), Block(1, 0), Block(1, 0)
    src | block | d0
    block | d1
    with temporarily_unpatch(block):
        assert not block.input.connected and not block.output.connected
    assert block.input.incomingConnection is src.output
    assert block.output.outgoingConnections == {d0.input, d1.input}
    comp = Composite()
    comp.inputs = [Relay(owner=comp), Relay(owner=comp)]
    comp.outputs = [Relay(owner=comp), Relay(owner=comp)]
    a, b = Block(1, 1), Block(1, 1)
    comp.inputs[0] | a | comp.outputs[0]
    comp.inputs[1] | b | comp.outputs[1]
    comp.update_internal_exec_order()
    assert comp.execOrder == [a, b]


# ---- voice / note (tests/test_audio_voices.py) -----------------------------------------------
def test_note_validation():
    # the reference's Temperament arithmetic (tunings.py:56-69) gives 440 Hz minus one ulp
    assert Note(69).frequency == 439.99999999999994 and Note(69).on and Note(69, 0.).off
    for bad in (dict(pitch=-1), dict(pitch=128), dict(pitch=60, velocity=1.5),
                dict(pitch=60, frequency=-1.)):
        with pytest.raises(ValueError):
            Note(**bad)
    assert Note.from_json(Note(60, .5).to_json()) == Note(60, .5)


def test_voice_lifecycle_on_host():
    voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.001, .001, .5, .001))
    assert [type(b).__name__ for b in voice.execOrder] == ['Oscillator', 'ADSR']
    voice.update()
    assert not voice.active and np.all(voice.output.value == 0.)   # silence before a note
    voice.input.push(Note(60, .7))
    assert voice.active                                            # pending message
    voice.update()
    assert voice.active and voice.amplitude == .7 and voice.currentPitch == 60
    assert voice.oscillator.frequency.value == Note(60).frequency
    voice.input.push(Note(60, 0.))
    for _ in range(4):
        voice.update()
    assert not voice.active and voice.amplitude == .7 and voice.currentPitch == 0


def test_envelope_attribute_validation_on_host():
    env = AR(.1, 1.)
    assert (env.decay, env.sustain) == (0., 1.)
    for name in ('attack', 'decay', 'release', 'overshoot'):
        with pytest.raises(ValueError):
            setattr(env, name, -1.)
    with pytest.raises(ValueError):
        env.sustain = 1.5
    env.overshoot = 0.
    assert env.overshoot == 1e-9
    with pytest.raises(ValueError):
        env.sample(-1)
    assert env.sample(0).size == 0


def test_dac_channel_collection():
    dac = Dac(nChannels=2)
    dac.inputs[0].set_value(np.ones((2, 256)))
    out = dac.collect_samples()
    assert out.shape == (256, 2) and out.dtype == np.float32


# ---- graph compiler: modulation sources -------------------------------------------
def test_fuser_recognises_modulation_sources():
    from klang_amd import fusion, native
    from klang_amd.audio import (ADSR, Filter, FmOscillator, Lfo, Mixer, Oscillator, PwmOscillator,
                                 Voice, sawtooth, triangle)
    from klang_amd.schedule import determine_execution_order

    def chains_of(top):
        groups, errors = fusion.find_fusable(determine_execution_order([top]))
        return groups, errors

    # arpeggiator_demo.py:35-49 in small: Lfo | fil.frequency, voice | fil
    lfo = Lfo(frequency=.1, wave_func=triangle, outputRange=(100, 3000))
    fil = Filter(frequency=220.)
    lfo | fil.frequency
    voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.1, .2, .3, .4))
    mixer = Mixer()
    mixer += (voice | fil)
    groups, errors = chains_of(mixer)
    assert not errors and len(groups) == 1
    chain = groups[0][1][0]
    assert chain.cutoff_lfo is lfo and chain.mod_target == native.MOD_NONE
    assert lfo in chain.blocks

    fm = FmOscillator(440., intensity=2., modFrequency=55.)
    groups, errors = chains_of(fm)
    assert not errors and groups[0][1][0].mod_target == native.MOD_FM
    assert groups[0][1][0].modulator is fm.modulator

    vib = Lfo(5., shape=256, outputRange=(430, 450))
    osc = Oscillator(440., wave_func=sawtooth)
    vib | osc.frequency
    groups, errors = chains_of(osc)
    assert not errors and groups[0][1][0].mod_target == native.MOD_FREQUENCY

    duty = Lfo(2., shape=256, outputRange=(.1, .9))
    pwm = PwmOscillator(220.)
    duty | pwm.dutyCycle
    groups, errors = chains_of(pwm)
    assert not errors and groups[0][1][0].mod_target == native.MOD_DUTY

    # refused: control-rate Lfo into an audio-rate input, shared Lfo, modulated Lfo
    slow = Lfo(1.)
    osc2 = Oscillator(440.)
    slow | osc2.frequency
    groups, errors = chains_of(osc2)
    assert not groups and 'shape=256' in errors[0]
    shared = Lfo(1., outputRange=(100, 1000))
    f1, f2 = Filter(), Filter()
    shared | f1.frequency
    shared | f2.frequency
    m2 = Mixer()
    m2 += (Oscillator(220.) | f1)
    m2 += (Oscillator(330.) | f2)
    groups, errors = chains_of(m2)
    assert any('more than one input' in e for e in errors)


def test_fuser_recognises_effect_stages():
    from klang_amd import fusion, native
    from klang_amd.audio import (ADSR, Delay, Filter, Gain, Lfo, Mixer, Oscillator, Reverb,
                                 TanhDistortion, Tremolo, Voice, sawtooth)
    from klang_amd.schedule import determine_execution_order

    def build():   # examples/arpeggiator_demo.py:29-49 in small
        voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.1, .2, .3, .4))
        return voice | Filter(frequency=220.) | Delay(time=.076, feedback=.9) | Gain(.5)

    mixer = Mixer()
    mixer += build()
    mixer += build()
    groups, errors = fusion.find_fusable(determine_execution_order([mixer]))
    assert not errors and len(groups) == 1
    chain = groups[0][1][0]
    assert chain.program == (native.FX_FILTER, native.FX_DELAY, native.FX_GAIN)
    assert [type(b).__name__ for b in chain.blocks] == ['Voice', 'Filter', 'Delay', 'Gain']
    # the plain [Filter] chain keeps the empty program (fast kernel)
    plain = Oscillator() | Filter()
    groups, _ = fusion.find_fusable(determine_execution_order([plain]))
    assert groups[0][1][0].program == ()
    # effects before a Reverb belong to its chain; after it they form a chain of their own
    # that takes the Reverb's output as external input
    ok = Oscillator() | TanhDistortion(2.) | Reverb()
    groups, errors = fusion.find_fusable(determine_execution_order([ok]))
    assert not errors and groups[0][1][0].program == (native.FX_TANH,)
    rv = Reverb()
    after = Oscillator() | rv | Gain(.5)
    groups, errors = fusion.find_fusable(determine_execution_order([after]))
    assert not errors and len(groups) == 2
    tail = [c for _, (c,), _ in groups if c.external is not None]
    assert len(tail) == 1 and tail[0].external is rv.output and tail[0].program == (native.FX_GAIN,)
    # refused: two delays, a gain modulated at audio rate; a gain driven at block rate by a
    # host Lfo is fused (the Lfo stays a host block), and so is a delay shorter than a buffer
    two = Oscillator() | Delay(.01) | Delay(.02)
    _, errors = fusion.find_fusable(determine_execution_order([two]))
    assert any('one Delay per chain' in e for e in errors)
    gain = Gain(.5)
    Lfo(1.) | gain.gain
    groups, errors = fusion.find_fusable(determine_execution_order([Oscillator() | gain]))
    assert not errors and groups[0][1][0].program == (native.FX_GAIN,)
    gain = Gain(.5)
    Oscillator(3.) | gain.gain
    _, errors = fusion.find_fusable(determine_execution_order([Oscillator() | gain]))
    assert any('audio rate' in e for e in errors)
    groups, errors = fusion.find_fusable(determine_execution_order([Oscillator() | Delay(.001)]))
    assert not errors and groups[0][1][0].program == (native.FX_DELAY,)   # 44 samples: block-major
    trem = Oscillator() | Tremolo(rate=4.)
    groups, errors = fusion.find_fusable(determine_execution_order([trem]))
    assert not errors and groups[0][1][0].program == (native.FX_TREMOLO,)


# ---- RingBuffer (klang/ring_buffer.py; the reference's tests/test_ring_buffer.py restated) ----
def test_ring_buffer_known_answers():
    from klang_amd.ring_buffer import RingBuffer
    eq = np.testing.assert_array_equal
    ring = RingBuffer(7, bufferSize=4)
    eq(ring.data, 11 * [0.])
    assert ring.pos == 0
    for i in range(5):
        ring.append(10 + i)
    assert ring.pos == 5
    eq(ring.data, [10, 11, 12, 13, 14, 0, 0, 10, 11, 12, 13])
    ring = RingBuffer(7, bufferSize=4)
    ring.extend([1, 2, 3, 4])
    assert ring.pos == 4
    eq(ring.data, [1, 2, 3, 4, 0, 0, 0, 1, 2, 3, 4])
    ring.extend([11, 12, 13, 14])
    assert ring.pos == 1
    eq(ring.data, [14, 2, 3, 4, 11, 12, 13, 14, 2, 3, 4])
    ring.extend([111, 112, 113, 114])
    assert ring.pos == 5
    eq(ring.data, [14, 111, 112, 113, 114, 12, 13, 14, 111, 112, 113])
    ring = RingBuffer(7, bufferSize=4)       # ragged writes
    for chunk in ([0.], [1., 2.], [3., 4., 5.], [6., 7.]):
        ring.extend(chunk)
    eq(ring.data, [7, 1, 2, 3, 4, 5, 6, 7, 1, 2, 3])
    with pytest.raises(ValueError):
        ring.extend([1, 2, 3, 4, 5])
    with pytest.raises(ValueError):
        ring.length = 8
    ring = RingBuffer(7, bufferSize=4)       # peeking
    eq(ring.peek(), [0, 0, 0, 0])
    ring.extend([1, 2, 3, 4])
    eq(ring.peek(), [0, 0, 0, 1])
    ring.extend([5, 6, 7, 8])
    eq(ring.peek(), [2, 3, 4, 5])
    assert ring.peek().base is ring.data     # a view, not a copy
    ring = RingBuffer(7, bufferSize=4)
    eq(ring.peek_extend([1, 2, 3, 4]), [0, 0, 0, 0])
    eq(ring.peek_extend([5, 6, 7, 8]), [0, 0, 0, 1])
    eq(ring.peek_extend([9, 10, 11, 12]), [2, 3, 4, 5])
    assert repr(RingBuffer(5, capacity=9)) == 'RingBuffer(length=5, capacity=9)'


def test_ring_buffer_as_delay_line():
    from klang_amd.ring_buffer import RingBuffer
    eq = np.testing.assert_array_equal
    ring = RingBuffer(7, bufferSize=4)       # a delay of 7 samples
    x = np.zeros(20)
    x[0] = 1.
    y = np.concatenate([ring.peek_extend(row) for row in x.reshape(-1, 4)])
    want = np.zeros(20)
    want[7] = 1.
    eq(y, want)
    ring = RingBuffer(7, bufferSize=4, nChannels=2)
    audio = np.array([[1., 0.], [0., 1.], [0., 0.], [0., 0.]])
    silence = np.zeros((4, 2))
    eq(ring.peek_extend(audio), np.zeros((4, 2)))
    eq(ring.peek_extend(silence), [[0, 0], [0, 0], [0, 0], [1, 0]])
    eq(ring.peek_extend(silence), [[0, 1], [0, 0], [0, 0], [0, 0]])
    ring = RingBuffer(7, bufferSize=4)       # shortening a line that holds samples
    impulse, silence = np.array([1., 0., 0., 0.]), np.zeros(4)
    eq(ring.peek_extend(impulse), [0, 0, 0, 0])
    eq(ring.peek_extend(silence), [0, 0, 0, 1])
    eq(ring.peek_extend(silence), [0, 0, 0, 0])
    ring.length = 5
    ring.pos = 2
    eq(ring.peek_extend(silence), [0, 0, 0, 0])
    eq(ring.peek_extend(impulse), [0, 0, 0, 0])
    eq(ring.peek_extend(silence), [0, 1, 0, 0])


def test_audio_helper_known_answers():
    """tests/test_audio.py:9-47 restated: silence and time arrays are read-only."""
    from klang_amd.audio.signals import get_silence, get_time
    mono = get_silence(10)
    assert mono.shape == (10,) and not mono.any()
    with pytest.raises(ValueError):
        mono[0] = 1
    stereo = get_silence((2, 10))
    assert stereo.shape == (2, 10) and not stereo.any()
    with pytest.raises(ValueError):
        stereo[0, 0] = 1
    t = get_time(10, .1)
    np.testing.assert_array_equal(t, .1 * np.arange(10))
    with pytest.raises(ValueError):
        t[0] = 1


def test_linear_mapping_hits_both_ends():
    """tests/test_math.py:6-15."""
    from klang_amd.audio.signals import linear_mapping
    a, b = linear_mapping((-1, 2), (-2, 3))
    assert a * -1 + b == -2 and a * 2 + b == 3


def test_note_durations():
    """tests/test_tempo.py:20-25, with the 4/4 metre the reference's constants name."""
    from fractions import Fraction
    from klang_amd.tempo import compute_duration, compute_rate, note_duration
    four_four = Fraction(4, 4, _normalize=False)
    for value, want in [(Fraction(1, 8), .25), (Fraction(1, 4), .5), (Fraction(1, 2), 1.),
                        (Fraction(1, 1), 2.)]:
        assert note_duration(value, 120, four_four) == want
        assert compute_duration(value, 120, four_four) == want
        assert compute_rate(value, 120, four_four) == 1. / want
    assert compute_duration(.3) == .3 and compute_rate(5.) == 5.


def test_bar_periods_and_note_values():
    """tests/test_tempo.py:13-18 (bar_period with simple, waltz and compound metres) and the
    note-value table (note_values.py:40-90)."""
    from fractions import Fraction
    from klang_amd import tempo
    assert tempo.bar_period(120) == 2.0 and tempo.bar_period(100) == 2.4
    assert tempo.bar_period(120, tempo.create_metre(3, 4)) == 1.5
    assert tempo.bar_period(80, tempo.create_metre(6, 8)) == 1.5
    beat = tempo.default_beat_value(tempo.create_metre(7, 8))       # un-reduced 2/8
    assert (beat.numerator, beat.denominator) == (2, 8)
    assert tempo.create_metre(6, 8).denominator == 8            # not reduced
    assert tempo.note_duration(tempo.EIGHT_NOTE, 120, tempo.FOUR_FOUR_METRE) == .25
    assert (tempo.LARGE_NOTE, tempo.DOTTED_LARGE_NOTE, tempo.DOUBLE_DOTTED_QUARTER_NOTE,
            tempo.TRIPLE_DOTTED_TWO_HUNDRED_FIFTY_SIXTH_NOTE) == (8, 12, Fraction(7, 16),
                                                                 Fraction(15, 2048))
    assert len(tempo.NOTE_VALUES) == 48
    assert tempo.dot_note(tempo.QUARTER_NOTE) == tempo.DOTTED_QUARTER_NOTE
    assert tempo.dot_note(tempo.HALF_NOTE, 3) == tempo.TRIPLE_DOTTED_HALF_NOTE
    assert tempo.tempo_2_period(120) == .5 and tempo.tempo_2_frequency(90) == 1.5


def test_find_next_primes():
    """tests/test_primes.py:8-54: the primes under 100, from any start."""
    from klang_amd.audio.filtering import find_next_primes
    under_100 = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73,
                 79, 83, 89, 97]
    assert find_next_primes(25, start=-10) == under_100
    for start in range(-10, 97):
        want = next(p for p in under_100 if p >= start)
        assert find_next_primes(1, start) == [want]
    assert find_next_primes(3, 90) == [97, 101, 103]


def test_note_dict_conversion():
    """tests/test_messages.py:14-22."""
    dct = Note(pitch=60).to_dict()
    assert (dct['pitch'], dct['velocity'], dct['type']) == (60, 1., 'Note')
    assert Note.from_dict(dct) == Note(pitch=60)
    assert Note.from_json(Note(pitch=69).to_json()) == Note(pitch=69)


def test_chirp_phase_properties():
    """tests/test_audio_oscillators.py:11-58: frequency and sweep rate of the chirp phase."""
    from klang_amd.audio.signals import chirp_phase
    from klang_amd.constants import TAU
    for lo, hi in ((0., 1.), (1., 2.)):   # inside the sweep and in the held region after it
        t, dt = np.linspace(lo, hi, 10, endpoint=False, retstep=True)
        omega = np.diff(chirp_phase(t, 1.234, 1., 1.234)) / dt
        np.testing.assert_almost_equal(omega / TAU, 1.234)
    t, dt = np.linspace(0, 1., 128, endpoint=False, retstep=True)
    omega = np.diff(chirp_phase(t, 2., 1., 3., method='linear')) / dt
    np.testing.assert_allclose(np.diff(omega) / dt, TAU)
    t, dt = np.linspace(0, 1., 128, retstep=True)
    np.testing.assert_allclose(np.diff(chirp_phase(t, 1., 1., 2.), n=2) / dt**2, TAU)
    t, dt = np.linspace(1., 10., 128, retstep=True)
    np.testing.assert_allclose(np.diff(chirp_phase(t, 1., 1., 2.), n=2) / dt**2, 0., atol=1e-10)
    t = np.linspace(0, 2., 128)
    assert (np.diff(chirp_phase(t, 1., 1., 2.)) < .2).all()


def test_sample_phase_known_answers():
    """tests/test_audio_oscillators.py:60-112."""
    from klang_amd.audio.signals import DT, INTERVAL, sample_phase
    from klang_amd.config import BUFFER_SIZE
    from klang_amd.constants import TAU
    n = np.arange(BUFFER_SIZE)
    phase, _ = sample_phase(440.)
    np.testing.assert_almost_equal(phase, TAU * 440. * DT * n % TAU)
    freqs = np.linspace(440., 880., BUFFER_SIZE)
    phase, _ = sample_phase(freqs)
    want = np.zeros(BUFFER_SIZE)
    want[1:] = TAU * DT * np.cumsum(freqs[:-1])
    np.testing.assert_almost_equal(phase, want % TAU)
    assert sample_phase(440., 1.234)[0][0] == 1.234
    assert sample_phase(freqs, 1.234)[0][0] == 1.234
    phase, _ = sample_phase(2 / INTERVAL)                      # two cycles per buffer
    np.testing.assert_almost_equal(phase, np.linspace(0, 2 * TAU, BUFFER_SIZE, endpoint=False) % TAU)
    assert sample_phase(1 / INTERVAL)[1] == 0.                 # one cycle: back to zero


def test_phasor_and_oscillator_known_answers():
    """tests/test_audio_oscillators.py:114-157."""
    from klang_amd.audio import Phasor
    from klang_amd.audio.signals import DT, INTERVAL
    from klang_amd.config import BUFFER_SIZE, SAMPLING_RATE
    from klang_amd.constants import TAU
    phasor = Phasor(frequency=.5)
    assert phasor.currentPhase == 0.
    phasor.update()
    assert phasor.currentPhase == TAU * .5 * INTERVAL
    phasor = Phasor(frequency=10.)
    cycles = int((SAMPLING_RATE / 10.) // BUFFER_SIZE) + 1
    for _ in range(cycles):
        phasor.update()
    assert abs(phasor.currentPhase - cycles * TAU * 10. * INTERVAL % TAU) < 1e-7
    phasor = Phasor(frequency=.25 / INTERVAL)
    assert [phasor.sample() for _ in range(5)] == [0., .25 * TAU, .5 * TAU, .75 * TAU, 0.]
    osc = Oscillator(frequency=440.)
    osc.update()
    np.testing.assert_almost_equal(osc.output.value, np.sin(TAU * 440. * DT * np.arange(BUFFER_SIZE)))


def test_voice_known_answers():
    """tests/test_audio_voices.py:19-77 with ADSR(0, 0, 1, 0)."""
    from klang_amd.audio import PwmOscillator
    voice = Voice(Oscillator(), ADSR(0., 0., 1., 0.))
    assert not voice.active and voice.amplitude == 0
    voice.update()
    assert not voice.output.value.any()
    voice.input.push(Note(60, velocity=.5))
    assert voice.active
    voice.update()
    assert voice.active and voice.amplitude == .5 and (voice.output.value != 0.).all()
    voice.input.push(Note(60, velocity=0.))
    voice.update()
    assert not voice.active and voice.amplitude == .5
    osc, lfo, env = PwmOscillator(), Oscillator(), ADSR()
    lfo.output.connect(osc.dutyCycle)
    assert Voice(osc, env).execOrder == [env, lfo, osc]


def test_temperament_known_answers(golden):
    """tests/test_music_tunings.py restated, and the default Note frequencies bit for bit as
    the reference's notes carry them (fixture rows written from real Note objects)."""
    from klang_amd.tunings import CENT_PER_OCTAVE, Temperament, cent_2_ratio, ratio_2_cent
    assert ratio_2_cent(2.) == CENT_PER_OCTAVE and cent_2_ratio(CENT_PER_OCTAVE) == 2.
    for kammerton in (440., 435.23):
        equal = Temperament(np.arange(12) * 100, kammerton=kammerton)
        for i in range(-6, 6):
            assert abs(equal.pitch_2_frequency(69 + 12 * i) - kammerton * 2. ** i) < 1e-7
    seen = {}
    for rows in (golden('sequencing')[k] for k in ('split', 'merged', 'quarter', 'arp_up')):
        seen.update({int(r[2]): r[4] for r in rows})
    assert len(seen) >= 12
    for pitch, frequency in seen.items():
        assert Note(pitch).frequency == frequency
    assert Note(60, temperament=Temperament(np.arange(12) * 100, kammerton=432.)).frequency < 260.


def test_metre_known_answers():
    """tests/test_music_metre.py restated."""
    from klang_amd import tempo as T
    m = T.create_metre
    assert (m(6, 4).numerator, m(6, 4).denominator) == (6, 4)
    assert not T.is_irrational(m(3, 4)) and T.is_irrational(m(3, 5))
    assert not any(T.is_complex(x) for x in (m(4, 4), m(3, 4), m(6, 8)))
    assert T.is_complex(m(5, 4)) and T.is_complex(m(7, 8))
    assert all(T.is_compound(x) for x in (m(6, 8), m(9, 8), m(12, 8)))
    assert not T.is_compound(m(3, 4)) and not T.is_compound(m(4, 4))
    assert all(T.is_simple(x) for x in (m(4, 4), m(3, 4), m(2, 4), m(3, 8), m(2, 2)))
    assert not any(T.is_simple(x) for x in (m(6, 8), m(12, 8), m(9, 4)))
    assert T.default_beat_value(T.FOUR_FOUR_METRE) == T.QUARTER_NOTE
    assert T.default_beat_value(T.THREE_FOUR_METRE) == T.QUARTER_NOTE
    assert T.default_beat_value(T.SIX_EIGHT_METRE) == T.DOTTED_QUARTER_NOTE
    assert T.default_beat_value(T.SEVEN_EIGHT_METRE) == m(2, 8)
    assert [T.number_of_beats(x) for x in (T.FOUR_FOUR_METRE, T.THREE_FOUR_METRE, T.SIX_EIGHT_METRE,
                                           T.SEVEN_EIGHT_METRE)] == [4, 3, 2, 3.5]
    assert T.number_of_beats(T.FOUR_FOUR_METRE, beatValue=T.EIGHT_NOTE) == 8
    assert T.number_of_beats(T.SIX_EIGHT_METRE, beatValue=T.EIGHT_NOTE) == 6
    assert T.number_of_beats(T.SEVEN_EIGHT_METRE, beatValue=T.EIGHT_NOTE) == 7


def test_sample_wave_and_pan_signal():
    """waves.py:56-61 and panning.py:84-99: the un-reduced phase of sample_wave (the Kick's
    oscillator) and the amplitudes pan_signal applies."""
    from klang_amd.audio.mixers import pan_signal, panning_amplitudes
    from klang_amd.audio.signals import DT, T, sample_wave, sawtooth, wrap
    from klang_amd.constants import TAU
    x, nxt = sample_wave(440., startPhase=.5)
    np.testing.assert_array_equal(x, np.sin(TAU * 440. * T + .5))
    assert nxt == wrap((TAU * 440. * T + .5)[-1] + TAU * 440. * DT) and -np.pi <= nxt < np.pi
    y, _ = sample_wave(100., wave_func=sawtooth, shape=16)
    assert y.shape == (16,)
    freqs = np.linspace(40., 80., 256)
    _, nxt = sample_wave(freqs)
    assert np.shape(nxt) == (256,)          # an array of frequencies gives an array of phases
    sig = np.arange(4.)
    np.testing.assert_array_equal(pan_signal(sig, .3), panning_amplitudes(.3) * sig)
    assert pan_signal(sig, -1.).shape == (2, 4)


def test_mixer_add_operators():
    """tests/test_audio_mixer.py:16-53: Mixer + thing and thing + Mixer for inputs (refused),
    outputs, blocks without an output (refused) and blocks."""
    with pytest.raises(IncompatibleConnection):
        Mixer(nInputs=0) + Block(1, 1).input
    mixer, block = Mixer(nInputs=0), Block(1, 1)
    assert (mixer + block.output) is mixer and connected(block, mixer.inputs[0])
    with pytest.raises(AttributeError):
        Mixer(nInputs=0) + Block()
    mixer, block = Mixer(nInputs=0), Block(1, 1)
    assert (mixer + block) is mixer and connected(block, mixer.inputs[0])
    with pytest.raises(IncompatibleConnection):
        Block(1, 1).input + Mixer(nInputs=0)
    mixer, block = Mixer(nInputs=0), Block(1, 1)
    assert (block.output + mixer) is mixer and connected(block, mixer.inputs[0])
    with pytest.raises(AttributeError):
        Block() + Mixer(nInputs=0)
    mixer, block = Mixer(nInputs=0), Block(1, 1)
    assert (block + mixer) is mixer and connected(block, mixer.inputs[0])


def test_execution_order_edge_cases_and_queue_reads():
    """tests/test_execution.py:8-36 (no blocks, one block, unconnected blocks in the given order,
    a second network reached through a connection) and tests/test_connections.py:167-190
    (receive drains in order, receive_latest returns the newest and clears the queue)."""
    assert determine_execution_order([]) == []
    block = Block()
    assert determine_execution_order([block]) == [block]
    a, b, c = Block(1, 1), Block(1, 1), Block(1, 1)
    assert determine_execution_order([a, b, c]) == [a, b, c]
    a, b, c = Block(1, 1), Block(1, 1), Block(1, 1)
    b.output.connect(c.input)
    assert determine_execution_order([a, b]) == [a, b, c]
    inp = MessageInput()
    for msg in range(10):
        inp.push(msg)
    assert len(inp.queue) == 10 and list(inp.receive()) == list(range(10)) and len(inp.queue) == 0
    assert inp.receive_latest() is None
    for msg in range(10):
        inp.push(msg)
    assert inp.receive_latest() == 9 and len(inp.queue) == 0
    from klang_amd.wiring import InputBase, OutputBase, RelayBase
    relay = RelayBase()
    OutputBase().connect(relay)
    with pytest.raises(InputAlreadyConnected):
        OutputBase().connect(relay)


def test_temporarily_unpatch_restores_every_connection():
    """tests/test_composite.py:9-23."""
    src, block, dst0, dst1 = Block(0, 1), Block(1, 1), Block(1, 0), Block(1, 0)
    src | block | dst0
    block | dst1
    with temporarily_unpatch(block):
        assert not block.input.connected and not block.output.connected
    assert block.input.incomingConnection is src.output
    assert set(block.output.outgoingConnections) == {dst0.input, dst1.input}


def test_is_valid_connection_table():
    """tests/test_connections.py:192-207."""
    from klang_amd.wiring import InputBase, OutputBase, RelayBase, is_valid_connection as valid
    assert valid(OutputBase(), InputBase()) and valid(OutputBase(), RelayBase())
    assert valid(RelayBase(), RelayBase()) and valid(RelayBase(), InputBase())
    assert not valid(InputBase(), InputBase()) and not valid(OutputBase(), OutputBase())
    assert not valid(MessageOutput(), InputBase()) and not valid(MessageOutput(), RelayBase())
    assert not valid(OutputBase(), MessageInput()) and not valid(RelayBase(), MessageInput())
