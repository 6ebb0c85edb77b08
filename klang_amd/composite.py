"""Composite blocks: a bundle of blocks with its own internal execution order
(host-side restatement of klang/composite.py:9-82)."""
import contextlib

from klang_amd.block import Block, collect_connections
from klang_amd.schedule import determine_execution_order, execute
from klang_amd.wiring import RelayBase


def introspect(composite):
    """Internal blocks adjacent to the composite's relays."""
    for relay in composite.inputs:
        if isinstance(relay, RelayBase):
            for inp in relay.outgoingConnections:
                if inp.owner:
                    yield inp.owner
    for relay in composite.outputs:
        if isinstance(relay, RelayBase):
            src = relay.incomingConnection
            if src and src.owner:
                yield src.owner


@contextlib.contextmanager
def temporarily_unpatch(block):
    """Cut `block` out of its surroundings for the duration of the context."""
    links = list(collect_connections(block))
    for src, dst in links:
        src.disconnect(dst)
    yield block
    for src, dst in links:
        src.connect(dst)


class Composite(Block):
    def __init__(self, nInputs=0, nOutputs=0, name=''):
        super().__init__(nInputs, nOutputs, name)
        self.execOrder = []

    def update_internal_exec_order(self, *blocks):
        if not blocks:
            blocks = introspect(self)
        with temporarily_unpatch(self):
            order = determine_execution_order(blocks)
        self.execOrder = [blk for blk in order if blk is not self]

    def update(self):
        execute(self.execOrder)
