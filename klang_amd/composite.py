"""Composite blocks: a bundle of blocks behind relays, with an execution order of its own
(the reference's klang/composite.py:9-82).

The inner order is found by running the scheduler on the composite's inner blocks while the
composite is cut out of the patch around it (otherwise the walk would leave through the
relays and order the whole patch).  ``Unpatched`` is that cut: a context manager object that
remembers the connections it removed and puts them back when the block ends, also when the
body raises.
"""
from klang_amd.block import Block, border
from klang_amd.schedule import determine_execution_order, execute
from klang_amd.wiring import RelayBase


class Unpatched:
    """``with Unpatched(block):`` -- `block` without its connections to other blocks."""

    def __init__(self, block):
        self.block = block
        self.removed = []

    def __enter__(self):
        self.removed = [(src, dst) for _, src, dst in border(self.block)]
        for src, dst in self.removed:
            src.disconnect(dst)
        return self.block

    def __exit__(self, exc_type, exc, tb):
        while self.removed:
            src, dst = self.removed.pop(0)
            src.connect(dst)
        return False


def temporarily_unpatch(block):
    """The reference's name for it (composite.py:31-49)."""
    return Unpatched(block)


def introspect(composite):
    """The inner blocks that touch the composite's relays: those an input relay sends to,
    those an output relay listens to (composite.py:9-28).  Plain inputs / outputs of the
    composite have no inner side."""
    inner = []
    for relay in composite.inputs:
        if isinstance(relay, RelayBase):
            inner += [sink.owner for sink in relay.outgoingConnections]
    for relay in composite.outputs:
        if isinstance(relay, RelayBase) and relay.incomingConnection is not None:
            inner.append(relay.incomingConnection.owner)
    return [block for block in inner if block]


class Composite(Block):
    def __init__(self, nInputs=0, nOutputs=0, name=''):
        super().__init__(nInputs, nOutputs, name)
        self.execOrder = []

    def update_internal_exec_order(self, *blocks):
        """Order of the inner network reachable from `blocks` (default: the relays' inner
        neighbours); the composite itself is not part of it."""
        seeds = blocks or introspect(self)
        with Unpatched(self):
            order = determine_execution_order(seeds)
        self.execOrder = [block for block in order if block is not self]

    def update(self):
        execute(self.execOrder)
