"""Block base class and the ``|`` / ``+`` patching operators.

Host-side restatement of klang/block.py:24-238.  ``a | b`` connects a's first
output to b's first input and returns b's owner; ``a + b`` returns a Mixer fed
by both.  AttributeError for a missing first input/output and TypeError for
wrong operand kinds, as in the reference (tests/test_block.py there).
"""
from klang_amd.wiring import InputBase, OutputBase, Input, Output


def input_connections(block):
    """(source output, input) pairs feeding `block` from other blocks."""
    for inp in block.inputs:
        src = inp.incomingConnection
        if src is not None and src.owner is not block:
            yield src, inp


def output_connections(block):
    """(output, destination input) pairs leaving `block` towards other blocks."""
    for out in block.outputs:
        for dst in out.outgoingConnections:
            if dst.owner is not block:
                yield out, dst


def collect_connections(block):
    yield from input_connections(block)
    yield from output_connections(block)


def input_neighbors(block):
    for src, _ in input_connections(block):
        if src.owner:
            yield src.owner


def output_neighbors(block):
    for _, dst in output_connections(block):
        if dst.owner:
            yield dst.owner


def fetch_input(thing):
    return thing.input if isinstance(thing, Block) else thing


def fetch_output(thing):
    return thing.output if isinstance(thing, Block) else thing


def pipe_operator(left, right):
    """left.output -> right.input; returns the owner of the input (block.py:113-132)."""
    out, inp = fetch_output(left), fetch_input(right)
    if not isinstance(out, OutputBase) or not isinstance(inp, InputBase):
        raise TypeError('Can not pipe from %s to %s' % (left, right))
    out.connect(inp)
    return inp.owner


def mix_operator(left, right):
    """New Mixer over two audio outputs (block.py:135-156)."""
    if not isinstance(fetch_output(left), Output) or not isinstance(fetch_output(right), Output):
        raise TypeError('Can not mix %s with %s' % (left, right))
    from klang_amd.audio.mixers import Mixer
    mixer = Mixer(nInputs=0)
    mixer += left
    mixer += right
    return mixer


class Block:
    """Unit of a patch: inputs, outputs and an ``update()`` called once per
    256-sample buffer (block.py:159-238)."""

    def __init__(self, nInputs=0, nOutputs=0, name=''):
        self.inputs = [Input(owner=self) for _ in range(nInputs)]
        self.outputs = [Output(owner=self) for _ in range(nOutputs)]
        self.name = name

    @property
    def nInputs(self):
        return len(self.inputs)

    @property
    def nOutputs(self):
        return len(self.outputs)

    @property
    def input(self):
        if not self.inputs:
            raise AttributeError('%s has no inputs!' % self)
        return self.inputs[0]

    @property
    def output(self):
        if not self.outputs:
            raise AttributeError('%s has no outputs!' % self)
        return self.outputs[0]

    def update(self):
        """Called once per buffer by the scheduler."""

    def __str__(self):
        bits = []
        if self.name:
            bits.append('name: %r' % self.name)
        if self.inputs:
            bits.append('%d inputs' % len(self.inputs))
        if self.outputs:
            bits.append('%d outputs' % len(self.outputs))
        return '%s(%s)' % (type(self).__name__, ', '.join(bits))

    def __or__(self, other):
        return pipe_operator(self, other)

    def __ror__(self, other):
        return pipe_operator(other, self)

    def __add__(self, other):
        return mix_operator(self, other)

    def __radd__(self, other):
        return mix_operator(other, self)
