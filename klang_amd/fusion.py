"""Graph compiler: hands oscillator -> envelope -> filter -> reverb -> mixer
chains of a patch to the HIP library.

The execution order stays the one derived from the DAG (klang_amd/schedule.py,
klang/execution.py:37-58 in the reference).  In that order every ``Mixer`` /
``StereoMixer`` whose inputs are all *voice chains* of one shape

    Voice(Oscillator|PwmOscillator|FmOscillator, ADSR|AR|D|R)  or a bare oscillator
        [ | Filter | Gain | Transformer | TanhDistortion | OctaveDistortion | Tremolo | Delay
          | Subsampler | Bitcrusher | RingModulator ]*
        [ | Reverb ]   ->  mixer input
    (effect stages in any order, at most one of each stateful kind per chain: Filter, Tremolo,
    Delay, Subsampler, RingModulator)

with, optionally, one modulation source per voice:
``Lfo(shape=1) | filter.frequency``, ``Lfo(shape=256) | oscillator.frequency``,
``Lfo(shape=256) | pwm.dutyCycle`` or an ``FmOscillator`` (its inner modulator)

is replaced, together with the chain blocks, by one ``GpuVoiceBank`` block that
owns a device-resident ``klang_amd.native.Bank`` (one voice per mixer input).
A chain of effect stages / a Reverb whose source is NOT a device oscillator -- the
output of a mixer, a synthesizer or any host block, as in the reference's
examples/reverberation_demo.py:33-34 (``mixer | Reverb | Dac``) -- is fused too: its
samples arrive per buffer as the bank's external input (KLMI_WAVE_EXTERNAL).
``PolyphonicSynthesizer`` / ``MonophonicSynthesizer`` blocks, alone
or as the inputs of one mixer, become *gated* banks: one voice per synthesizer voice,
updated only while active; the polyphonic note routing runs on the device.
``Kick``, ``Chirper`` and ``HiHat`` blocks become device blocks of their own (the hi-hat's
RNG noise is drawn on the host and enters as external input).  Composites that merely run
their inner blocks (``StereoDelay``) are opened first, so that their stages are seen.  A
stereo buffer fed into a chain of stages runs one bank per channel; a ``StereoMixer`` over
external voices takes stereo inputs as two hard-panned voices.
Chains that end anywhere else (``Oscillator(440.) | Dac()``) become one-voice
banks.  Per buffer the bank block drains the voices' note queues into
block-boundary events (voices.py:55-64), renders, and publishes the mix on the
mixer's own ``Output`` object, so downstream connections are untouched.

There is no fallback: under ``backend='hip'`` a hot-path block that cannot be
fused is an error, and a missing library or GPU raises from klang_amd.native.
"""
import copy
import types

import numpy as np

from klang_amd import native
from klang_amd.audio.effects import (Bitcrusher, Delay, Gain, OctaveDistortion, RingModulator,
                                     Subsampler, TanhDistortion, Transformer, Tremolo)
from klang_amd.audio.envelopes import EnvelopeBase
from klang_amd.audio.filtering import BackwardCombFilter, Filter, Reverb
from klang_amd.audio.mixers import Mixer, StereoMixer
from klang_amd.audio.signals import (BUFFER_SIZE, SCALAR, Chirper, FmOscillator, Lfo, Oscillator, Phasor,
                                      PwmOscillator, WAVE_IDS)
from klang_amd.audio.synthesizer import (HiHat, Kick, MonophonicSynthesizer,
                                         PolyphonicSynthesizer, Synthesizer)
from klang_amd.audio.voices import Voice
from klang_amd.block import Block
from klang_amd.composite import Composite
from klang_amd.errors import KlangError
from klang_amd.tempo import compute_rate
from klang_amd.wiring import Output

EFFECT_KINDS = {Filter: native.FX_FILTER, Gain: native.FX_GAIN, Transformer: native.FX_TRANSFORM,
                TanhDistortion: native.FX_TANH, OctaveDistortion: native.FX_OCTAVE,
                Tremolo: native.FX_TREMOLO, Delay: native.FX_DELAY,
                Subsampler: native.FX_SUBSAMPLE, Bitcrusher: native.FX_BITCRUSH,
                RingModulator: native.FX_RINGMOD}
ONE_PER_CHAIN = (Filter, Tremolo, Delay, Subsampler, RingModulator)   # stateful stages
HOT_PATH_TYPES = (Phasor, EnvelopeBase, Voice, Filter, Reverb, Mixer, Synthesizer, Kick, HiHat,
                  Chirper) + tuple(
    t for t in EFFECT_KINDS if t is not Filter)


class NotFusable(KlangError):
    """Why a chain cannot be handed to the device (message says which rule)."""


class Chain:
    """One voice chain, source first: oscillator [+ envelope via Voice]
    [-> Filter] [-> Reverb]; `end` is the block whose output leaves the chain."""

    def __init__(self, end=None, external=None):
        """Chain(end): the chain that ends in block `end`.  Chain(external=output): no block
        at all, just somebody's output as a voice (a mixer input that is not a device chain)."""
        self.end = end
        self.blocks = []
        self.voice = self.oscillator = self.envelope = self.filter = self.reverb = None
        self.stages = []                # effect stages in patch order (the Filter among them)
        self.cutoff_lfo = None          # Lfo(shape=1) | filter.frequency
        self.mod_target = native.MOD_NONE
        self.modulator = None           # Lfo(shape=256) or FmOscillator.modulator
        self.external = external        # the Output whose samples feed the first stage, when the
                                        # source is not a device oscillator
        self.patched_params = False     # some stage parameter input is connected (block rate)
        if end is not None:
            self._walk(end)

    def _check_stage(self, stage):
        kind = type(stage)
        if sum(type(st) is kind for st in self.stages) and kind in ONE_PER_CHAIN:
            raise NotFusable('%s: one %s per chain runs on the device' % (stage, kind.__name__))
        if kind is Filter:
            if stage.frequency.connected:
                self.cutoff_lfo = self._private_lfo(stage.frequency, SCALAR, '%s.frequency' % stage)
            self.filter = stage
        else:
            # A patched parameter input (an Lfo into Tremolo.rate, Gain.gain ...) is followed
            # at block rate: the source stays a host block, runs before the bank, and the
            # value it holds at each buffer is uploaded like a host-side edit
            # (GpuVoiceBank._upload_stage).  Only sources the host evaluates, with one value
            # per buffer, qualify.
            for inp in stage.inputs:
                if inp is stage.input or not inp.connected:
                    continue
                source = inp.incomingConnection.owner
                if type(source) in EFFECT_KINDS or isinstance(source, (Voice, Reverb, Oscillator)):
                    raise NotFusable('%s: a parameter input is fed at audio rate by %s'
                                     % (stage, source))
                if np.ndim(inp.value) != 0:
                    raise NotFusable('%s: a parameter input carries an array per buffer '
                                     '(audio-rate modulation of effect parameters is not fused)'
                                     % stage)
                self.patched_params = True
        if kind is RingModulator:
            if stage.lfo.wave_func not in WAVE_IDS or WAVE_IDS[stage.lfo.wave_func] > native.WAVE_TRIANGLE:
                raise NotFusable('%s: Lfo wave function %r has no device twin'
                                 % (stage, stage.lfo.wave_func))
        if kind is Delay:
            if stage.ring.data.any():
                raise NotFusable('%s: its delay line already holds samples' % stage)

    @staticmethod
    def _private_lfo(inp, shape, what):
        """The Lfo behind `inp`, if it is one the device can evaluate for this
        voice alone: right shape, device wave, constant frequency, one consumer."""
        src = inp.incomingConnection
        lfo = src.owner if src is not None else None
        if type(lfo) is not Lfo or lfo.shape != shape:
            raise NotFusable('%s: only an Lfo(shape=%d) is fused as modulation source'
                             % (what, shape))
        if lfo.wave_func not in WAVE_IDS:
            raise NotFusable('%s: Lfo wave function %r has no device twin' % (what, lfo.wave_func))
        if lfo.frequency.connected or np.ndim(lfo.frequency.value) != 0:
            raise NotFusable('%s: the Lfo\'s own frequency is modulated' % what)
        if len(src.outgoingConnections) != 1:
            raise NotFusable('%s: the Lfo feeds more than one input' % what)
        return lfo

    def _walk(self, block):
        stage, first = block, None      # first: the earliest block of the chain seen so far
        if isinstance(stage, Reverb):
            if stage.echoType is not BackwardCombFilter:
                raise NotFusable('%s: only BackwardCombFilter echoes run on the device' % stage)
            self.reverb, first, stage = stage, stage, self._upstream(stage)
        while type(stage) in EFFECT_KINDS:
            self._check_stage(stage)
            self.stages.insert(0, stage)
            first, stage = stage, self._upstream(stage)
        if first is not None and not isinstance(stage, (Voice, Oscillator, PwmOscillator)):
            # the chain starts at someone else's output: external input, per buffer
            if len(self.stages) > native.MAX_FX:
                raise NotFusable('%s: more than %d effect stages in one chain'
                                 % (block, native.MAX_FX))
            self.external = first.input.incomingConnection
            self.blocks = [b for b in [self.reverb, self.cutoff_lfo] + self.stages if b]
            return
        if len(self.stages) > native.MAX_FX:
            raise NotFusable('%s: more than %d effect stages in one chain' % (block, native.MAX_FX))
        if isinstance(stage, Voice):
            self.voice = stage
            self.oscillator, self.envelope = stage.oscillator, stage.envelope
            if not isinstance(self.envelope, EnvelopeBase):
                raise NotFusable('%s: envelope type %s' % (stage, type(self.envelope).__name__))
            for inner in (self.oscillator, self.envelope):
                if any(o.connected for o in inner.outputs):
                    raise NotFusable('%s: inner block %s is patched elsewhere' % (stage, inner))
        else:
            self.oscillator = stage
        osc = self.oscillator
        if type(osc) not in (Oscillator, PwmOscillator, FmOscillator):
            raise NotFusable('%s is not an oscillator the device evaluates' % osc)
        if type(osc) is not PwmOscillator and osc.wave_func not in WAVE_IDS:
            raise NotFusable('%s: wave function %r has no device twin' % (osc, osc.wave_func))
        if type(osc) is FmOscillator:
            mod = osc.modulator
            if (type(mod) is not Oscillator or mod.wave_func not in WAVE_IDS
                    or mod.frequency.connected or np.ndim(mod.frequency.value) != 0
                    or any(o.connected for o in mod.outputs)):
                raise NotFusable('%s: its modulator is not a plain inner Oscillator' % osc)
            self.mod_target, self.modulator = native.MOD_FM, mod
        if osc.frequency.connected:
            if self.modulator is not None:
                raise NotFusable('%s: one audio-rate modulation source per voice' % osc)
            self.modulator = self._private_lfo(osc.frequency, BUFFER_SIZE, '%s.frequency' % osc)
            self.mod_target = native.MOD_FREQUENCY
        elif np.ndim(osc.frequency.value) != 0:
            raise NotFusable('%s: a constant frequency array is not fused' % osc)
        if type(osc) is PwmOscillator and osc.dutyCycle.connected:
            if self.modulator is not None:
                raise NotFusable('%s: one audio-rate modulation source per voice' % osc)
            self.modulator = self._private_lfo(osc.dutyCycle, BUFFER_SIZE, '%s.dutyCycle' % osc)
            self.mod_target = native.MOD_DUTY
        elif type(osc) is PwmOscillator and np.ndim(osc.dutyCycle.value) != 0:
            raise NotFusable('%s: a constant duty-cycle array is not fused' % osc)
        self.blocks = [b for b in [self.voice or self.oscillator, self.reverb, self.cutoff_lfo,
                                   self.modulator if self.mod_target != native.MOD_FM else None]
                       + self.stages if b]

    @staticmethod
    def _upstream(block):
        src = block.input.incomingConnection
        if src is None or src.owner is None:
            raise NotFusable('%s has no upstream block' % block)
        inside = isinstance(src.owner, (Voice, Phasor)) or type(src.owner) in EFFECT_KINDS
        if inside and len(src.outgoingConnections) != 1:
            raise NotFusable('%s feeds more than one input' % src.owner)
        return src.owner

    @property
    def shape(self):
        rv = self.reverb
        return (self.voice is not None, self.filter is not None,
                (rv.decay, rv.preDelay, rv.dryWet, rv.nEchos) if rv else None,
                self.cutoff_lfo is not None, self.mod_target, self.program,
                self.external is not None)

    @property
    def program(self):
        """Effect kinds in patch order; () for the plain [Filter] chain."""
        kinds = tuple(EFFECT_KINDS[type(st)] for st in self.stages)
        return () if kinds in ((), (native.FX_FILTER,)) else kinds

    def wave_id(self):
        if self.external is not None:
            return native.WAVE_EXTERNAL
        if isinstance(self.oscillator, PwmOscillator):
            return native.WAVE_PWM
        return WAVE_IDS[self.oscillator.wave_func]


class _ChannelTap:
    """One row of somebody's multi-channel output, as the external source of a chain."""

    def __init__(self, output, channel):
        self.output, self.channel, self.owner = output, channel, output.owner

    @property
    def value(self):
        return np.asarray(self.output.value)[self.channel]


class GpuVoiceBank(Block):
    """A fused set of chains evaluated by one device bank.  `sink` is the block
    whose output carries the result (the mixer, or the chain end itself)."""

    def __init__(self, chains, sink, ctx=None, synths=(), shard=None, dropped=()):
        """`shard` (klang_amd.sharding.Shard): `chains` are this rank's contiguous part of the
        mixer's inputs, `dropped` the chains other ranks evaluate.  The bank then renders the
        UN-normalised partial bus and every render ends in one reduce over the ranks
        (mixer.py:41-49 is the only cross-voice operation); the root divides by the global
        input count and is the only rank whose output is the mix (the others deliver silence)."""
        super().__init__()
        self.chains, self.sink = chains, sink
        self.shard, self.dropped = shard, list(dropped)
        # synthesizers whose voices the chains are, in voice order (synthesizer.py:151-219)
        self.synths = list(synths)
        self.poly = (self.synths[0].MAX_VOICES
                     if self.synths and isinstance(self.synths[0], PolyphonicSynthesizer) else 0)
        self.ctx = ctx or native.default_context()
        has_env, has_filter, reverb, cutoff_lfo, mod_target, program, external = chains[0].shape
        self.external = external
        self.cutoff_lfo, self.mod_target, self.program = cutoff_lfo, mod_target, program
        self.stereo = isinstance(sink, StereoMixer)
        self.is_mixer = isinstance(sink, Mixer)
        kw = {}
        if self.stereo:
            kw = dict(pan_mode=sink.mode, pan_law=sink.panLaw)
        self.bank = native.Bank(
            self.ctx, len(chains), has_env=has_env, has_filter=has_filter,
            has_reverb=reverb is not None, stereo=self.stereo,
            reverb=dict(zip(('decay', 'pre_delay', 'dry_wet', 'n_echos'), reverb)) if reverb else None,
            cutoff_lfo=cutoff_lfo, mod_target=mod_target, effects=program,
            gated=bool(self.synths), poly_voices=self.poly, **kw)
        if self.poly and self.is_mixer and len(self.synths) > 1:
            # mean over the synthesizers of (sum of voices / MAX_VOICES): mixer.py:46-47 after
            # synthesizer.py:219
            self.bank.set_mix_divisor(float(self.poly * len(self.synths)))
        # A StereoMixer that takes its inputs as external voices gives every input TWO voices:
        # a mono input uses the first (gain, pan as set), a stereo input (pan amplitudes per
        # channel, mixer.py:95-99) uses both, hard left and hard right.
        self.pairs = bool(self.stereo and all(c.end is None for c in chains))
        if self.pairs:
            self.bank.set_mix_divisor(float(max(1, len(chains) // 2)) if len(chains) > 2 else 1.)
            self._pair_state = None
        if self.shard is not None:
            if self.synths or self.pairs or self.external or not self.is_mixer:
                raise NotFusable('%s: only a mixer of uniform device chains is sharded' % sink)
            self.bank.set_mix_divisor(1.)
            self._shard_bus = None
        self.upstream = None    # the mono bank whose bus feeds this bank's single chain, when
                                # there is one (compile_order): its buffers then arrive device
                                # to device and both banks render ahead in lock-step
        self._fresh = 0         # buffers of the last render() a downstream bank may still take
        self._twins = []        # banks for channels 1.. of a chain fed with a multi-channel signal
        # A connected stage parameter (host Lfo -> Gain.gain ...) changes every buffer in the
        # reference, and its value is only read when a render is prepared: such a bank renders
        # ONE buffer per call whatever lookahead the caller asks for (like a host-fed bank).
        self.patched_params = any(c.patched_params for c in chains)
        self.lookahead = 1      # blocks rendered per device call (set by run_klang)
        self.remaining = None   # buffers still to be delivered, when the caller knows (offline
                                # bounce): rendering ahead never runs past the end of the piece
        self._ready = []
        self._upload()

    # -- parameters / state ------------------------------------------------------------
    def _upload(self):
        F, bank, ch = native, self.bank, self.chains
        oscs = [c.oscillator for c in ch]
        bank.set(F.F_WAVE, [c.wave_id() for c in ch])
        if not self.external:
            if self.mod_target != F.MOD_FREQUENCY:
                bank.set(F.F_FREQUENCY, [float(compute_rate(o.frequency.value)) for o in oscs])
            bank.set(F.F_DUTY, [float(o.dutyCycle.value)
                                if isinstance(o, PwmOscillator) and self.mod_target != F.MOD_DUTY
                                else .5 for o in oscs])
            bank.set_state(F.F_PHASE, [float(o.currentPhase) for o in oscs])
        if self.mod_target != F.MOD_NONE:
            mods = [c.modulator for c in ch]
            fm = self.mod_target == F.MOD_FM
            bank.set(F.F_MOD_FREQUENCY, [float(compute_rate(m.frequency.value)) for m in mods])
            bank.set(F.F_MOD_WAVE, [WAVE_IDS[m.wave_func] for m in mods])
            bank.set(F.F_MOD_SCALE, [float(c.oscillator.intensity if fm else c.modulator.scale)
                                     for c in ch])
            bank.set(F.F_MOD_OFFSET, [0. if fm else float(m.offset) for m in mods])
            bank.set_state(F.F_MOD_PHASE, [float(m.currentPhase) for m in mods])
        if self.cutoff_lfo:
            lfos = [c.cutoff_lfo for c in ch]
            bank.set(F.F_LFO_FREQUENCY, [float(compute_rate(l.frequency.value)) for l in lfos])
            bank.set(F.F_LFO_WAVE, [WAVE_IDS[l.wave_func] for l in lfos])
            bank.set(F.F_LFO_SCALE, [float(l.scale) for l in lfos])
            bank.set(F.F_LFO_OFFSET, [float(l.offset) for l in lfos])
            bank.set_state(F.F_LFO_PHASE, [float(l.currentPhase) for l in lfos])
        if ch[0].voice is not None:
            envs = [c.envelope for c in ch]
            for field, name in ((F.F_ATTACK, 'attack'), (F.F_DECAY, 'decay'),
                                (F.F_SUSTAIN, 'sustain'), (F.F_RELEASE, 'release'),
                                (F.F_OVERSHOOT, 'overshoot'), (F.F_LOOP, 'loop'),
                                (F.F_RETRIGGER, 'retrigger')):
                bank.set(field, [float(getattr(e, name)) for e in envs])
            bank.set(F.F_AMPLITUDE, [float(c.voice.amplitude) for c in ch])
            bank.set_state(F.F_ENV_VALUE, [e.value for e in envs])
            bank.set_state(F.F_ENV_STAGE, [e.stage for e in envs])
            bank.set_state(F.F_ENV_ENABLED, [float(e.enabled) for e in envs])
        if ch[0].filter is not None and not self.cutoff_lfo:
            bank.set(F.F_CUTOFF, [float(c.filter.frequency.value) for c in ch])
            self._cutoffs = [c.filter.frequency.value for c in ch]
        self._fx_params = {}
        for s in range(len(self.program)):
            self._upload_stage(s, force=True)
        if self.is_mixer and not self.pairs:
            bank.set(F.F_GAIN, self._per_voice(self.sink.gains))
            if self.stereo:
                bank.set(F.F_PAN, self._per_voice(self.sink.pannings))
            self._gains = list(self.sink.gains)
            self._pans = list(getattr(self.sink, 'pannings', []))
        if self.poly:
            P = self.poly
            bank.set_state(F.F_CURRENT_PITCH, [float(c.voice.currentPitch) for c in ch])
            for i, syn in enumerate(self.synths):
                bank.set_state(F.F_ROUND_ROBIN, [float(syn.allocated % P)], first=i * P)

    def _per_voice(self, per_input):
        """A mixer's per-input values (gains, pannings), one per bank voice."""
        if self.shard is not None:
            per_input = list(per_input)[self.shard.first:self.shard.first + self.shard.count]
        rep = self.poly or 1
        return [float(x) for x in per_input for _ in range(rep)]

    @staticmethod
    def _stage_params(stage):
        """(parameters k0.., state k0..) of one effect block, as the C-ABI takes them."""
        if isinstance(stage, Gain):
            return (float(stage.gain.value),), ()
        if isinstance(stage, Transformer):
            return (float(stage.scale), float(stage.offset)), ()
        if isinstance(stage, TanhDistortion):
            return (float(stage.drive),), ()
        if isinstance(stage, Tremolo):
            return ((float(compute_rate(stage.rate.value)), float(stage.depth.value),
                     float(stage.smoothness.value), float(stage.dutyCycle.value)),
                    (float(stage.lfo.currentPhase), float(stage.zi[0])))
        if isinstance(stage, Delay):
            return (float(stage.time), float(stage.feedback), float(stage.drywet)), \
                   (float(stage.ring.pos),)
        if isinstance(stage, Subsampler):
            return (float(stage.factor),), ()
        if isinstance(stage, Bitcrusher):
            return (float(stage.nBits),), ()
        if isinstance(stage, RingModulator):
            return ((float(stage.frequency.value), float(compute_rate(stage.rate.value)),
                     float(stage.amount.value), float(stage.dryWet.value),
                     float(WAVE_IDS[stage.lfo.wave_func])),
                    (float(stage.lfo.currentPhase), float(stage.modulator.currentPhase)))
        return (), ()

    def _upload_stage(self, s, force=False):
        """Parameters of effect stage s for every voice; after the first upload only what
        was edited on the host blocks since (Gain.gain, Tremolo.depth, ...)."""
        F, bank = native, self.bank
        per_voice = [self._stage_params(c.stages[s]) for c in self.chains]
        params = [p for p, _ in per_voice]
        if not force and params == self._fx_params.get(s):
            return
        for k in range(len(params[0])):
            column = [p[k] for p in params]
            if force or column != [p[k] for p in self._fx_params[s]]:
                bank.set(F.F_FX(s, k), column)
        self._fx_params[s] = params
        if force:
            for k in range(len(per_voice[0][1])):
                bank.set_state(F.F_FX_STATE(s, k), [st[k] for _, st in per_voice])

    def _collect_events(self):
        """Voice.process_incoming_notes (voices.py:55-64) for every voice, plus
        parameter edits made on the host blocks since the last buffer."""
        F, bank = native, self.bank
        for i, syn in enumerate(self.synths):
            for note in syn.input.receive():            # Synthesizer.update, synthesizer.py:82-84
                if self.poly:
                    # routed on the device; the host block only keeps its cycle in step
                    bank.note(i, 0, note.pitch, note.velocity, note.frequency)
                    if note.on:
                        next(syn.freeVoice)
                else:
                    syn.process_note(note)              # NoteScheduler -> voice.input
        for i, c in enumerate(self.chains):
            if c.voice is not None:
                for note in c.voice.input.receive():
                    if note.on:
                        bank.event(i, 0, F.EV_NOTE_ON, note.velocity, note.frequency)
                        c.voice.amplitude = note.velocity
                        c.voice.currentPitch = note.pitch
                        c.oscillator.frequency.set_value(note.frequency)
                    else:
                        bank.event(i, 0, F.EV_NOTE_OFF)
                        c.voice.currentPitch = 0
                for note in c.envelope.input.receive():        # envelopes.py:40-42
                    bank.event(i, 0, F.EV_NOTE_ON if note.on else F.EV_NOTE_OFF,
                               c.voice.amplitude, float(compute_rate(c.oscillator.frequency.value)))
            if (c.filter is not None and not self.cutoff_lfo
                    and c.filter.frequency.value != self._cutoffs[i]):
                self._cutoffs[i] = c.filter.frequency.value
                bank.event(i, 0, F.EV_SET_CUTOFF, 0., float(self._cutoffs[i]))
        for c in self.dropped:      # another rank plays these voices: their notes go nowhere here
            if c.voice is not None:
                for _ in c.voice.input.receive():
                    pass
                for _ in c.envelope.input.receive():
                    pass
        for s in range(len(self.program)):
            self._upload_stage(s)
        if self.is_mixer and not self.pairs:
            if list(self.sink.gains) != self._gains:
                self._gains = list(self.sink.gains)
                bank.set(F.F_GAIN, self._per_voice(self._gains))
            if self.stereo and list(self.sink.pannings) != self._pans:
                self._pans = list(self.sink.pannings)
                bank.set(F.F_PAN, self._per_voice(self._pans))

    @property
    def pending_messages(self):
        return (any(syn.input.queue for syn in self.synths)
                or any(c.voice is not None and (c.voice.input.queue or c.envelope.input.queue)
                       for c in self.chains))

    def update(self):
        if self._ready and self.pending_messages:
            raise KlangError('GpuVoiceBank: notes arrived while buffers were rendered ahead; '
                             'use lookahead=1 for patches with live message sources')
        if not self._ready:
            self._collect_events()
            chained = self.upstream is not None and self.upstream._fresh > 0
            # float32 [n][channels][256] -> per buffer (256,) mono or (2, 256) stereo
            ahead = (1 if (self.external and not chained) or self.patched_params
                     else max(1, int(self.lookahead)))
            if self.remaining is not None:
                ahead = max(1, min(ahead, int(self.remaining)))
            if chained:
                # the upstream bank has just rendered its next buffers (it runs first and keeps
                # the same pace): they come over on the device
                ahead = min(ahead, self.upstream._fresh)
                self.bank.set_input_from(self.upstream.bank, ahead)
                self.upstream._fresh = 0
            elif self.external:
                self._feed_external()
            mix = self._render(ahead)
            self._fresh = ahead
            self._ready = [blk if self.stereo else blk[0] for blk in mix][::-1]
        if self.remaining is not None:
            self.remaining -= 1
        self.sink.output.set_value(self._ready.pop().astype(np.float64))
        if self._twins:
            rows = [self.sink.output.value]
            for twin in self._twins:
                twin.update()
                rows.append(twin.sink.output.value)
            self.sink.output.set_value(np.stack(rows))

    # -- voice shards (klang_amd.sharding) ----------------------------------------------
    def _reduced(self, n_blocks):
        """n_blocks buffers of this rank's shard, summed over the ranks on the device:
        -> torch float32 [n_blocks][channels][256] (the mix on the root, divided by the global
        input count; meaningless elsewhere).  One collective per call, never per buffer."""
        import contextlib

        import torch
        from klang_amd import sharding
        sh = self.shard
        stream = self.ctx.stream
        scope = (torch.cuda.stream(torch.cuda.ExternalStream(stream)) if stream
                 else contextlib.nullcontext())
        with scope:     # the collective is ordered behind the library's launches by the stream
            bus = self._shard_bus
            if bus is None or bus.shape[0] != n_blocks:
                bus = self._shard_bus = torch.empty((n_blocks, self.bank.channels, BUFFER_SIZE),
                                                    dtype=torch.float32, device='cuda')
            self.bank.render_device(n_blocks, bus.data_ptr())
            sharding.reduce_bus(bus, sh.n_total, sh.world, 0, sh.group, always=True)
        return bus

    def _render(self, n_blocks):
        if self.shard is None:
            return self.bank.render(n_blocks)
        bus = self._reduced(n_blocks)
        if self.shard.rank == 0:
            return bus.cpu().numpy()
        return np.zeros((n_blocks, self.bank.channels, BUFFER_SIZE), dtype=np.float32)

    # -- notes known ahead of time (klang_amd.klang.preschedule) ---------------------------
    def message_targets(self):
        """{MessageInput: handle} for the inputs whose notes this bank consumes (or, for the
        voices another rank plays, swallows)."""
        if self.synths:
            return {syn.input: ('synth', i) for i, syn in enumerate(self.synths)}
        targets = {c.voice.input: ('voice', i) for i, c in enumerate(self.chains)
                   if c.voice is not None}
        targets.update({c.voice.input: ('drop', i) for i, c in enumerate(self.dropped)
                        if c.voice is not None})
        return targets

    def schedule(self, handle, note, block):
        """What update() would do with `note` arriving in buffer `block` (counted from the
        next render), queued now."""
        F, (kind, i) = native, handle
        if kind == 'drop':
            return
        if kind == 'synth':
            syn = self.synths[i]
            if self.poly:
                self.bank.note(i, block, note.pitch, note.velocity, note.frequency)
                if note.on:
                    next(syn.freeVoice)
                return
            note = syn.noteScheduler.get_next_note(note)    # MonophonicSynthesizer.process_note
        voice = self.chains[i].voice
        if note.on:
            self.bank.event(i, block, F.EV_NOTE_ON, note.velocity, note.frequency)
            voice.currentPitch = note.pitch
        else:
            self.bank.event(i, block, F.EV_NOTE_OFF)
            voice.currentPitch = 0

    def _split_channels(self, n_channels):
        """The chain's source turned out to carry n_channels: every channel gets a bank of its
        own over the same stage blocks (same parameters, same initial state), as the host
        blocks treat the rows of a multi-channel buffer alike (effects.py:365-384 for Filter)."""
        chain = self.chains[0]
        source = chain.external
        for ch in range(1, n_channels):
            twin_chain = copy.copy(chain)
            twin_chain.external = _ChannelTap(source, ch)
            twin = GpuVoiceBank([twin_chain], types.SimpleNamespace(output=Output()), self.ctx)
            self._twins.append(twin)
        chain.external = _ChannelTap(source, 0)

    def _feed_pairs(self):
        """StereoMixer over external voices, two voices per input (see __init__)."""
        F, sink = native, self.sink
        rows, gains, pans = [], [], []
        hard_l = float(native.panning_amplitudes(-1., sink.mode, sink.panLaw)[0])
        hard_r = float(native.panning_amplitudes(1., sink.mode, sink.panLaw)[1])
        silence = np.zeros(BUFFER_SIZE)
        for i in range(0, len(self.chains), 2):
            value = np.asarray(self.chains[i].external.value, dtype=np.float64)
            gain, pan = float(sink.gains[i // 2]), float(sink.pannings[i // 2])
            if value.shape == (BUFFER_SIZE,):
                rows += [value, silence]
                gains += [gain, 0.]
                pans += [pan, 0.]
            elif value.shape == (2, BUFFER_SIZE) and hard_l > 0. and hard_r > 0.:
                amp = native.panning_amplitudes(pan, sink.mode, sink.panLaw)
                rows += [value[0], value[1]]
                gains += [gain * float(amp[0]) / hard_l, gain * float(amp[1]) / hard_r]
                pans += [-1., 1.]
            else:
                raise KlangError('%s: input %d carries a buffer of shape %s from %s'
                                 % (self, i // 2, value.shape, self.chains[i].external.owner))
        if (gains, pans) != self._pair_state:
            self._pair_state = (gains, pans)
            self.bank.set(F.F_GAIN, gains)
            self.bank.set(F.F_PAN, pans)
        self.bank.set_input(np.stack(rows))

    def _feed_external(self):
        """This buffer's samples of the chains' sources -> klmi_bank_set_input."""
        if self.pairs:
            return self._feed_pairs()
        if not self.is_mixer and len(self.chains) == 1 and not self._twins:
            c = self.chains[0]
            value = np.asarray(c.external.value)
            if (value.ndim == 2 and c.reverb is None
                    and not (c.stages and type(c.stages[0]) is Delay)):
                self._split_channels(value.shape[0])
        rows = []
        for c in self.chains:
            value = np.asarray(c.external.value, dtype=np.float64)
            if value.ndim == 2 and c.stages and type(c.stages[0]) is Delay:
                value = value.sum(axis=0) / value.shape[0]   # Delay's down-mix (effects.py:166-168)
            if value.shape != (BUFFER_SIZE,):
                raise KlangError('%s: the device chain takes mono buffers of %d samples, got shape '
                                 '%s from %s' % (self, BUFFER_SIZE, value.shape, c.external.owner))
            rows.append(value)
        self.bank.set_input(np.stack(rows))

    def render_for(self, downstream, n_blocks):
        """The whole piece for a bank that listens to this one's bus, device to device."""
        if self._ready or downstream._ready:
            raise KlangError('GpuVoiceBank.render_for: buffers were already rendered ahead')
        self._collect_events()
        if self.external and self.upstream is None:
            raise KlangError('GpuVoiceBank.render_for: %s is fed from the host' % self)
        mix = self.bank.render(n_blocks)
        downstream.bank.set_input_from(self.bank, n_blocks)
        if self.remaining is not None:
            self.remaining = 0
        last = mix[-1].astype(np.float64)
        self.sink.output.set_value(last if self.stereo else last[0])

    def bounce(self, duration, fadeout=0., want_pcm=True):
        """The whole piece in one device pass (klmi_bank_bounce): notes queued so far are
        applied at buffer 0 as usual; -> (frames float32 [cycles*256][channels], int16 pcm)."""
        if self._ready:
            raise KlangError('GpuVoiceBank.bounce: buffers were already rendered ahead')
        self._collect_events()
        if self.shard is not None:
            n_cycles = int(duration * 44100 // BUFFER_SIZE)
            bus = self._reduced(n_cycles)
            if self.shard.rank == 0:
                frames, pcm, _ = self.bank.bounce_bus(bus.data_ptr(), duration, fadeout, want_pcm)
            else:           # the root writes the piece; nothing to deliver here
                frames = np.zeros((n_cycles * BUFFER_SIZE, self.bank.channels), dtype=np.float32)
                pcm = None
        else:
            frames, pcm, _ = self.bank.bounce(duration, fadeout, want_pcm)
        if self.remaining is not None:
            self.remaining = 0
        last = frames[-BUFFER_SIZE:].T.astype(np.float64)
        self.sink.output.set_value(last if self.stereo else last[0])
        return frames, pcm

    def sync_host_blocks(self):
        """Copy device state back into the host blocks (currentPhase, envelope
        stage/value) so that the patch can be inspected or re-run on the host."""
        F, bank = native, self.bank
        if not self.external:
            phase = bank.get_state(F.F_PHASE)
            for i, c in enumerate(self.chains):
                c.oscillator.currentPhase = float(phase[i])
        if self.mod_target != F.MOD_NONE:
            for c, p in zip(self.chains, bank.get_state(F.F_MOD_PHASE)):
                c.modulator.currentPhase = float(p)
        if self.cutoff_lfo:
            for c, p in zip(self.chains, bank.get_state(F.F_LFO_PHASE)):
                c.cutoff_lfo.currentPhase = float(p)
        for s, kind in enumerate(self.program):   # delay-line contents stay on the device
            if kind == F.FX_TREMOLO:
                phase, zi = bank.get_state(F.F_FX_STATE(s, 0)), bank.get_state(F.F_FX_STATE(s, 1))
                for i, c in enumerate(self.chains):
                    c.stages[s].lfo.currentPhase = float(phase[i])
                    c.stages[s].zi = np.array([zi[i]])
            elif kind == F.FX_DELAY:
                for c, p in zip(self.chains, bank.get_state(F.F_FX_STATE(s, 0))):
                    c.stages[s].ring.pos = int(p)
            elif kind == F.FX_RINGMOD:
                lfo, mod = bank.get_state(F.F_FX_STATE(s, 0)), bank.get_state(F.F_FX_STATE(s, 1))
                for i, c in enumerate(self.chains):
                    c.stages[s].lfo.currentPhase = float(lfo[i])
                    c.stages[s].modulator.currentPhase = float(mod[i])
        if self.chains[0].voice is not None:
            value, stage = bank.get_state(F.F_ENV_VALUE), bank.get_state(F.F_ENV_STAGE)
            enabled = bank.get_state(F.F_ENV_ENABLED)
            amp = bank.get_state(F.F_AMPLITUDE)
            for i, c in enumerate(self.chains):
                c.envelope.value, c.envelope.stage = float(value[i]), int(stage[i])
                c.envelope.enabled = bool(enabled[i])
                c.voice.amplitude = float(amp[i])
            if self.mod_target != F.MOD_FREQUENCY:
                for c, f in zip(self.chains, bank.get_state(F.F_FREQUENCY)):
                    if not c.oscillator.frequency.connected:
                        c.oscillator.frequency.set_value(float(f))
        if self.poly:   # what the device-side routing decided
            pitch, amp = bank.get_state(F.F_CURRENT_PITCH), bank.get_state(F.F_AMPLITUDE)
            freq = bank.get_state(F.F_FREQUENCY)
            for i, c in enumerate(self.chains):
                c.voice.currentPitch, c.voice.amplitude = int(pitch[i]), float(amp[i])
                c.oscillator.frequency.set_value(float(freq[i]))

    def close(self):
        for twin in self._twins:
            twin.close()
        self.bank.close()

    def __str__(self):
        return 'GpuVoiceBank(%d voices -> %s)' % (len(self.chains), type(self.sink).__name__)


def synth_chains(synths):
    """One Chain per voice of the synthesizers (all of one kind and one shape)."""
    kinds = {type(s) for s in synths}
    if len(kinds) != 1 or not issubclass(next(iter(kinds)),
                                         (MonophonicSynthesizer, PolyphonicSynthesizer)):
        raise NotFusable('synthesizers of different kinds (%s) in one bank'
                         % ', '.join(sorted(k.__name__ for k in kinds)))
    if len({getattr(s, 'MAX_VOICES', 1) for s in synths}) != 1:
        raise NotFusable('polyphonic synthesizers with different MAX_VOICES in one bank')
    chains = []
    for syn in synths:
        for voice in (syn.voices if isinstance(syn, PolyphonicSynthesizer) else [syn.voice]):
            if not isinstance(voice, Voice):
                raise NotFusable('%s: its voice is a %s, not a Voice' % (syn, type(voice).__name__))
            if any(o.connected for o in voice.outputs):
                raise NotFusable('%s: a voice output is patched elsewhere' % syn)
            chains.append(Chain(voice))
    if len({c.shape for c in chains}) != 1:
        raise NotFusable('synthesizer voices of different shapes')
    if chains[0].filter or chains[0].stages or chains[0].reverb:
        raise NotFusable('synthesizer voices are plain Voice(oscillator, envelope) chains')
    return chains


class GpuKick(Block):
    """One Kick block evaluated on the device (klmi_kick_*, count 1): the kick's own Output
    carries the result.  One block per Kick keeps every kick at its place in the execution
    order; callers that can batch use native.KickBatch directly."""

    def __init__(self, kick, ctx=None):
        super().__init__()
        self.kick = kick
        if kick.currentTime != 0. or np.any(kick.currentPhase):
            raise NotFusable('%s has already played on the host' % kick)
        self.batch = native.KickBatch(ctx or native.default_context(), 1)
        for field, value in ((native.KICK_FREQUENCY, kick.frequency), (native.KICK_DECAY, kick.decay),
                             (native.KICK_INTENSITY, kick.intensity),
                             (native.KICK_PITCH_DECAY, kick.pitchDecay)):
            self.batch.set(field, [float(value)])
        self.lookahead, self.remaining = 1, None    # set by run_klang
        self._ready = []

    def message_targets(self):
        return {self.kick.input: ('kick', 0)}

    def schedule(self, handle, note, block):
        if note.pitch > 0 and note.on:              # synthesizer.py:275-278
            self.batch.trigger(0, block)

    def update(self):
        notes = list(self.kick.input.receive())
        if self._ready and notes:
            raise KlangError('GpuKick: notes arrived while buffers were rendered ahead; use '
                             'lookahead=1 for patches with live message sources')
        if not self._ready:
            for note in notes:
                self.schedule(None, note, 0)
            ahead = max(1, int(self.lookahead))
            if self.remaining is not None:
                ahead = max(1, min(ahead, int(self.remaining)))
            self._ready = list(self.batch.render(ahead)[0].reshape(ahead, BUFFER_SIZE))[::-1]
        if self.remaining is not None:
            self.remaining -= 1
        self.kick.output.set_value(self._ready.pop())

    def sync_host_blocks(self):
        time, phase = self.batch.state()
        self.kick.currentTime, self.kick.currentPhase = float(time[0]), phase[0].copy()

    def close(self):
        self.batch.close()

    def __str__(self):
        return 'GpuKick(%s)' % self.kick


class GpuHiHat(Block):
    """One HiHat block (synthesizer.py:222-250) as a one-voice device bank: the decay envelope
    runs on the device, the white noise is drawn on the host from NumPy's global RNG -- once
    per buffer, at the hi-hat's place in the execution order, exactly where the reference draws
    it -- and goes in as the voice's external input (amplitude 1: 1 * noise * env).  Always one
    buffer per call: drawing ahead would reorder the RNG stream between several hi-hats."""

    def __init__(self, hat, ctx=None):
        super().__init__()
        self.hat = hat
        env = hat.envelope
        if env.stage != 0 or env.value != 0.:
            raise NotFusable('%s has already played on the host' % hat)
        F = native
        self.bank = F.Bank(ctx or F.default_context(), 1, has_env=True)
        self.bank.set(F.F_WAVE, [F.WAVE_EXTERNAL])
        for field, name in ((F.F_ATTACK, 'attack'), (F.F_DECAY, 'decay'), (F.F_SUSTAIN, 'sustain'),
                            (F.F_RELEASE, 'release'), (F.F_OVERSHOOT, 'overshoot'),
                            (F.F_LOOP, 'loop'), (F.F_RETRIGGER, 'retrigger')):
            self.bank.set(field, [float(getattr(env, name))])
        self.bank.set(F.F_AMPLITUDE, [1.])
        self.lookahead, self.remaining = 1, None    # run_klang sets them; one buffer at a time here

    def message_targets(self):
        return {self.hat.input: ('hat', 0)}

    def schedule(self, handle, note, block):
        # every note goes to the envelope's trigger input (synthesizer.py:243-244, envelopes.py:40-42)
        self.bank.event(0, block, native.EV_NOTE_ON if note.on else native.EV_NOTE_OFF, 1., 0.)

    def update(self):
        for note in self.hat.input.receive():
            self.schedule(None, note, 0)
        noise = np.asarray(self.hat.noise_generator(), dtype=np.float64)
        self.bank.set_input(noise[None, :])
        self.hat.output.set_value(self.bank.render(1)[0, 0].astype(np.float64))
        if self.remaining is not None:
            self.remaining -= 1

    def sync_host_blocks(self):
        F, env = native, self.hat.envelope
        env.value = float(self.bank.get_state(F.F_ENV_VALUE)[0])
        env.stage = int(self.bank.get_state(F.F_ENV_STAGE)[0])
        env.enabled = bool(self.bank.get_state(F.F_ENV_ENABLED)[0])

    def close(self):
        self.bank.close()

    def __str__(self):
        return 'GpuHiHat(%s)' % self.hat


class GpuChirper(Block):
    """One Chirper block evaluated on the device (klmi_chirp_*, count 1); the chirper's own
    Output carries the result.  A chirper has no inputs, so it may render ahead."""

    def __init__(self, chirper, ctx=None):
        super().__init__()
        self.chirper = chirper
        if chirper.wave_func not in WAVE_IDS or WAVE_IDS[chirper.wave_func] > native.WAVE_TRIANGLE:
            raise NotFusable('%s: wave function %r has no device twin' % (chirper, chirper.wave_func))
        if np.any(chirper.t != Chirper(1., 1., 1.).t):
            raise NotFusable('%s has already played on the host' % chirper)
        try:
            self.batch = native.ChirpBatch(ctx or native.default_context(), [chirper.startFrequency],
                                           [chirper.duration], [chirper.endFrequency],
                                           chirper.method, WAVE_IDS[chirper.wave_func])
        except native.NativeError as err:
            raise ValueError(str(err)) from None      # scipy's ValueErrors for bad sweeps
        self.lookahead, self.remaining = 1, None
        self._ready = []

    def message_targets(self):
        return {}

    def update(self):
        if not self._ready:
            ahead = max(1, int(self.lookahead))
            if self.remaining is not None:
                ahead = max(1, min(ahead, int(self.remaining)))
            self._ready = list(self.batch.render(ahead)[0].reshape(ahead, BUFFER_SIZE))[::-1]
        if self.remaining is not None:
            self.remaining -= 1
        self.chirper.output.set_value(self._ready.pop())

    def sync_host_blocks(self):
        # buffers rendered ahead but never played do not count
        self.chirper.t = (self.batch.time_vectors()[0]
                          - len(self._ready) * (BUFFER_SIZE / 44100.) if self._ready
                          else self.batch.time_vectors()[0])

    def close(self):
        self.batch.close()

    def __str__(self):
        return 'GpuChirper(%s)' % self.chirper


def _flat(order):
    for blk in order:
        yield blk
        if isinstance(blk, Composite):
            yield from _flat(blk.execOrder)


def open_composites(order):
    """Composites that merely run their inner blocks (Composite.update as it is: StereoDelay)
    and hold hot-path blocks are replaced by those inner blocks, in their inner order: relays
    pass values through, so the flat order computes the same thing and the inner blocks
    become visible to the chain search.  Composites with an update of their own (Voice,
    Tremolo, RingModulator) are device stages as a whole and stay closed."""
    flat = []
    for blk in order:
        if (isinstance(blk, Composite) and type(blk).update is Composite.update
                and any(isinstance(b, HOT_PATH_TYPES) and type(b) not in (Phasor, Lfo)
                        for b in _flat(blk.execOrder))):
            flat.extend(open_composites(blk.execOrder))
        else:
            flat.append(blk)
    return flat


def find_fusable(exec_order):
    """-> (groups, errors): groups = [(sink, [Chain, ...], synthesizers)] for every mixer whose
    inputs are uniform chains and for every other chain end; errors = reasons
    for hot-path blocks that stay outside."""
    groups, errors, claimed = [], [], set()
    for blk in exec_order:
        if isinstance(blk, Mixer) and blk.nInputs > 0:
            try:
                chains, sources = [], []
                for inp in blk.inputs:
                    src = inp.incomingConnection
                    if src is None or src.owner is None:
                        raise NotFusable('%s has an unconnected input' % blk)
                    if len(src.outgoingConnections) != 1:
                        raise NotFusable('%s feeds more than one input' % src.owner)
                    sources.append(src.owner)
                if sources and all(isinstance(s, Synthesizer) for s in sources):
                    chains = synth_chains(sources)       # a mixer of synthesizers: one bank
                    groups.append((blk, chains, sources))
                    claimed.update(sources)
                    claimed.add(blk)
                    continue
                chains = [Chain(s) for s in sources]
                if len({c.shape for c in chains}) != 1:
                    raise NotFusable('%s mixes chains of different shapes' % blk)
                if isinstance(blk, StereoMixer) and chains[0].external is not None:
                    # chains fed from outside may carry stereo buffers: each becomes a bank of
                    # its own (one per channel if need be), the mixer takes their outputs
                    raise NotFusable('%s: externally fed chains under a StereoMixer' % blk)
                groups.append((blk, chains, ()))
                claimed.update(b for c in chains for b in c.blocks)
                claimed.add(blk)
            except NotFusable as err:
                # Not uniform device chains.  The mixer still sums on the device: its inputs
                # arrive per buffer as external voices, and whatever produces them is fused
                # (or not) on its own further down.
                outputs = [inp.incomingConnection for inp in blk.inputs]
                if all(o is not None and o.owner is not None for o in outputs):
                    per_input = 2 if isinstance(blk, StereoMixer) else 1   # see GpuVoiceBank.pairs
                    groups.append((blk, [Chain(external=o) for o in outputs for _ in range(per_input)], ()))
                    claimed.add(blk)
                else:
                    errors.append(str(err))
    # synthesizers that feed something other than a fused mixer: a bank each
    for blk in exec_order:
        if isinstance(blk, Synthesizer) and blk not in claimed:
            try:
                groups.append((blk, synth_chains([blk]), [blk]))
                claimed.add(blk)
            except NotFusable as err:
                errors.append(str(err))
    # chain ends that feed something other than a fused mixer
    for blk in exec_order:
        if blk in claimed or not isinstance(blk, (Phasor, Voice, Reverb) + tuple(EFFECT_KINDS)):
            continue
        if type(blk) in (Phasor, Lfo):
            continue   # control-rate sources stay on the host
        consumers = [d.owner for o in blk.outputs for d in o.outgoingConnections]
        if not isinstance(blk, Reverb) and any(
                isinstance(c, (Reverb,) + tuple(EFFECT_KINDS)) and c not in claimed for c in consumers):
            continue   # not the end of its chain (a Reverb always is: what follows it starts anew)
        try:
            chain = Chain(blk)
            groups.append((blk, [chain], ()))
            claimed.update(chain.blocks)
        except NotFusable as err:
            errors.append(str(err))
    return groups, errors


CHAIN_ON_DEVICE = True   # tests switch it off to get the per-buffer host hop to compare with


def link_banks(order):
    """Bank -> bank on the device.  A bank whose single chain listens to the mono bus of
    another bank (`mixer | Reverb | Dac`, effects after a synthesizer) takes that bus device
    to device (klmi_bank_set_input_from_bank) when the upstream bank runs before it, on the
    same context, and nobody else needs the buffers in between."""
    if not CHAIN_ON_DEVICE:
        return
    banks = [b for b in order if isinstance(b, GpuVoiceBank)]
    by_output = {id(b.sink.output): b for b in banks}
    for k, bank in enumerate(banks):
        if not bank.external or bank.pairs or len(bank.chains) != 1:
            continue
        source = by_output.get(id(bank.chains[0].external))
        if (source is None or source is bank or source.stereo or source.ctx is not bank.ctx
                or source not in banks[:k] or source._twins or source.shard is not None):
            # (a sharded bank's mix exists after the reduce, on the root: what listens to it is
            # a post-reduce stage and takes it from the host, buffer by buffer)
            continue
        bank.upstream = source


def shardable(sink, chains, synths, world):
    """A group the voice sharding applies to: a Mixer / StereoMixer whose inputs are uniform
    device chains (voices are independent until the Mixer sum), at least one per
    rank.  Everything else is evaluated where it is."""
    return (isinstance(sink, Mixer) and not synths and chains[0].external is None
            and len(chains) >= world)


def shard_groups(groups, rank, world, group=None):
    """The plan: [(sink, kept chains, synths, Shard or None, dropped chains)] for `rank`.  A
    shardable group keeps the contiguous part sharding.shard_range gives this rank; notes
    addressed to the other voices are dropped HERE because their owner receives them there
    (every rank builds the same patch, so routing by owner is keeping one's own)."""
    from klang_amd import sharding
    out = []
    for sink, chains, synths in groups:
        if shardable(sink, chains, synths, world):
            sh = sharding.plan(len(chains), rank, world, group)
            kept = chains[sh.first:sh.first + sh.count]
            out.append((sink, kept, synths, sh, chains[:sh.first] + chains[sh.first + sh.count:]))
        else:
            out.append((sink, chains, synths, None, []))
    return out


def compile_order(exec_order, ctx=None, strict=True, shard=None):
    """Replace fusable chains in `exec_order` by GpuVoiceBank blocks.  With
    strict=True every hot-path block must end up fused (KlangError otherwise).
    `shard` = (rank, world[, group]): mixers of uniform device chains keep only this rank's
    voices (shard_groups) and reduce their bus over the ranks."""
    exec_order = open_composites(exec_order)
    groups, errors = find_fusable(exec_order)
    fused_blocks, banks_at = set(), {}
    if shard is not None:
        planned = shard_groups(groups, *shard)
    else:
        planned = [(sink, chains, synths, None, []) for sink, chains, synths in groups]
    for sink, chains, synths, sh, dropped in planned:
        bank = GpuVoiceBank(chains, sink, ctx, synths, shard=sh, dropped=dropped)
        for c in dropped:
            fused_blocks.update(c.blocks)
        for c in chains:
            fused_blocks.update(c.blocks)
        fused_blocks.update(synths)
        fused_blocks.add(sink)
        banks_at[sink] = bank
    for blk in exec_order:
        if isinstance(blk, (Kick, Chirper, HiHat)):
            try:
                make = GpuKick if isinstance(blk, Kick) else GpuHiHat if isinstance(blk, HiHat) else GpuChirper
                banks_at[blk] = make(blk, ctx)
            except NotFusable as err:
                errors.append(str(err))
    new_order = []
    for blk in exec_order:
        if blk in banks_at:
            new_order.append(banks_at[blk])
        elif blk not in fused_blocks:
            new_order.append(blk)
    link_banks(new_order)
    if strict:
        left = [b for b in _flat(new_order) if isinstance(b, HOT_PATH_TYPES)
                and type(b) not in (Phasor, Lfo)]
        if left:
            raise KlangError(
                "backend='hip': %d hot-path block(s) could not be handed to the device (%s). %s "
                "Use backend='host' explicitly to evaluate the patch with the NumPy host blocks."
                % (len(left), ', '.join(str(b) for b in left[:4]), ' / '.join(errors[:4])))
    return new_order, list(banks_at.values())
