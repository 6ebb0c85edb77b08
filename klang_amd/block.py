"""Block base class and the ``|`` / ``+`` patching operators.

The reference's protocol (klang/block.py:24-238), which patches and the scheduler rely on:
a block owns lists of inputs and outputs and an ``update()`` called once per 256-sample
buffer; ``a | b`` connects a's first output to b's first input and returns b's owner;
``a + b`` returns a Mixer fed by both; AttributeError for a missing first input / output,
TypeError for operands that cannot be patched.  Everything about a block's surroundings is
derived from one walk over the connections that cross its border (``border``).
"""
from klang_amd.wiring import InputBase, OutputBase, Input, Output

INWARD, OUTWARD = 'in', 'out'


def border(block):
    """Every connection between `block` and another block, as (direction, source output,
    destination input): inputs first, then outputs, each in port order."""
    crossing = []
    for port in block.inputs:
        source = port.incomingConnection
        if source is not None and source.owner is not block:
            crossing.append((INWARD, source, port))
    for port in block.outputs:
        crossing.extend((OUTWARD, port, sink) for sink in port.outgoingConnections
                        if sink.owner is not block)
    return crossing


def _pairs(block, direction):
    return [(src, dst) for way, src, dst in border(block) if way == direction]


def input_connections(block):
    """(source output, input) pairs feeding `block` from other blocks."""
    return iter(_pairs(block, INWARD))


def output_connections(block):
    """(output, destination input) pairs leaving `block` towards other blocks."""
    return iter(_pairs(block, OUTWARD))


def collect_connections(block):
    return ((src, dst) for _, src, dst in border(block))


def input_neighbors(block):
    """Owners of the outputs that feed `block` (ownerless endpoints have none)."""
    return (src.owner for src, _ in _pairs(block, INWARD) if src.owner)


def output_neighbors(block):
    return (dst.owner for _, dst in _pairs(block, OUTWARD) if dst.owner)


def _first_port(thing, which):
    """A block stands for its first input / output; an endpoint for itself."""
    return getattr(thing, which) if isinstance(thing, Block) else thing


def fetch_input(thing):
    return _first_port(thing, 'input')


def fetch_output(thing):
    return _first_port(thing, 'output')


def pipe_operator(left, right):
    """left.output -> right.input; returns the owner of the input (block.py:113-132)."""
    source, sink = fetch_output(left), fetch_input(right)
    if not (isinstance(source, OutputBase) and isinstance(sink, InputBase)):
        raise TypeError('Can not pipe from %s to %s' % (left, right))
    source.connect(sink)
    return sink.owner


def mix_operator(left, right):
    """New Mixer over two audio outputs (block.py:135-156)."""
    from klang_amd.audio.mixers import Mixer
    if not all(isinstance(fetch_output(side), Output) for side in (left, right)):
        raise TypeError('Can not mix %s with %s' % (left, right))
    mixer = Mixer(nInputs=0)
    for side in (left, right):
        mixer += side
    return mixer


class Block:
    """Unit of a patch: inputs, outputs and an ``update()`` called once per
    256-sample buffer (block.py:159-238)."""

    def __init__(self, nInputs=0, nOutputs=0, name=''):
        self.name = name
        self.inputs = [Input(owner=self) for _ in range(nInputs)]
        self.outputs = [Output(owner=self) for _ in range(nOutputs)]

    nInputs = property(lambda self: len(self.inputs))
    nOutputs = property(lambda self: len(self.outputs))

    def _first(self, kind):
        ports = getattr(self, kind)
        if not ports:
            raise AttributeError('%s has no %s!' % (self, kind))
        return ports[0]

    input = property(lambda self: self._first('inputs'))
    output = property(lambda self: self._first('outputs'))

    def update(self):
        """Called once per buffer by the scheduler."""

    def __str__(self):
        facts = [('name: %r', self.name), ('%d inputs', len(self.inputs)),
                 ('%d outputs', len(self.outputs))]
        return '%s(%s)' % (type(self).__name__, ', '.join(fmt % val for fmt, val in facts if val))

    # patching operators, both operand orders (an endpoint on the left reaches __r*__)
    def __or__(self, other):
        return pipe_operator(self, other)

    def __ror__(self, other):
        return pipe_operator(other, self)

    def __add__(self, other):
        return mix_operator(self, other)

    def __radd__(self, other):
        return mix_operator(other, self)
