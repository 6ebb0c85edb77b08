"""Connectable outputs / inputs / relays, value and message flavoured.

Host-side restatement of the reference's connection protocol
(klang/connections.py:31-377): one output feeds many inputs, an input accepts
one output, values are passed by reference through ``Output.set_value`` /
``Input.value``, messages are pushed into bounded queues.  Behaviour (names,
argument meaning, exceptions) follows the reference so that its patches and
tests read the same here; the implementation is this package's own.
"""
import collections

from klang_amd.errors import KlangError

__all__ = ['Output', 'Relay', 'Input', 'MessageOutput', 'MessageRelay', 'MessageInput',
           'InputAlreadyConnected', 'IncompatibleConnection', 'is_connected', 'are_connected']

VALUE, MESSAGE = 'value', 'message'


class InputAlreadyConnected(KlangError):
    """An input accepts exactly one incoming connection (connections.py:31-33)."""


class IncompatibleConnection(KlangError):
    """Value and message endpoints do not mix (connections.py:36-38)."""


class _Endpoint:
    """Common part of every connectable: an owner block and a flavour."""
    flavour = None      # VALUE / MESSAGE / None for the abstract bases
    sends = False       # may act as the source of a connection
    accepts = False     # may act as the destination of a connection

    def __init__(self, owner=None):
        self.owner = owner

    def __str__(self):
        bits = []
        if self.owner:
            bits.append('owner: %s' % self.owner)
        bits.append('connected' if self.connected else 'not connected')
        return '%s(%s)' % (type(self).__name__, ', '.join(bits))


def _compatible(src, dst):
    """connections.py:41-65: source must be able to send, destination to accept,
    flavours must agree, and (as in the reference's table) both sides are either
    the concrete value/message classes or the abstract bases."""
    if not (getattr(src, 'sends', False) and getattr(dst, 'accepts', False)):
        return False
    return src.flavour == dst.flavour and type(src) in _KNOWN and type(dst) in _KNOWN


is_valid_connection = _compatible   # the reference's public name (connections.py:41)


def _require_compatible(src, dst):
    if not _compatible(src, dst):
        raise IncompatibleConnection('Can not connect %s with %s!'
                                     % (type(src).__name__, type(dst).__name__))


def _attach(src, dst):
    _require_compatible(src, dst)
    if dst.connected:
        raise InputAlreadyConnected('%s is already connected to another output!' % dst)
    src.outgoingConnections.add(dst)
    dst.incomingConnection = src


def _detach(src, dst):
    _require_compatible(src, dst)
    src.outgoingConnections.remove(dst)
    dst.incomingConnection = None


def is_connected(output, input_):
    """Directional check output -> input_ (connections.py:102-107)."""
    if not _compatible(output, input_):
        return False
    return input_.incomingConnection is output and input_ in output.outgoingConnections


def are_connected(*chain):
    """Every consecutive pair of `chain` is connected (connections.py:110-121)."""
    return all(is_connected(a, b) for a, b in zip(chain, chain[1:]))


class OutputBase(_Endpoint):
    sends = True

    def __init__(self, owner=None):
        super().__init__(owner)
        self.outgoingConnections = set()

    @property
    def connected(self):
        return bool(self.outgoingConnections)

    def connect(self, input_):
        _attach(self, input_)

    def disconnect(self, input_):
        _detach(self, input_)


class InputBase(_Endpoint):
    accepts = True

    def __init__(self, owner=None):
        super().__init__(owner)
        self.incomingConnection = None

    @property
    def connected(self):
        return self.incomingConnection is not None

    def connect(self, output):
        _attach(output, self)

    def disconnect(self, output):
        _detach(output, self)


class RelayBase(InputBase, OutputBase):
    """Input and output at once: the gateway of composite blocks
    (connections.py:202-237)."""

    def __init__(self, owner=None):
        InputBase.__init__(self, owner)
        self.outgoingConnections = set()

    @property
    def connected(self):
        return self.incomingConnection is not None

    def connect(self, other):
        if getattr(other, 'accepts', False):
            _attach(self, other)
        else:
            _attach(other, self)

    def disconnect(self, other):
        if getattr(other, 'accepts', False):
            _detach(self, other)
        else:
            _detach(other, self)


class _Value:
    """Holder of the latest value (connections.py:240-271)."""

    def __init__(self, value=0.):
        self._value = value

    def get_value(self):
        return self._value

    def set_value(self, value):
        self._value = value

    value = property(get_value, set_value)


class Output(OutputBase, _Value):
    flavour = VALUE

    def __init__(self, owner=None, value=0.):
        OutputBase.__init__(self, owner)
        _Value.__init__(self, value)


class Input(InputBase, _Value):
    """Fetches from the connected output, falls back to its own value
    (connections.py:274-302)."""
    flavour = VALUE

    def __init__(self, owner=None, value=0.):
        InputBase.__init__(self, owner)
        _Value.__init__(self, value)

    def get_value(self):
        src = self.incomingConnection
        return src.value if src is not None else self._value

    value = property(get_value, _Value.set_value)


class Relay(RelayBase, Input):
    flavour = VALUE

    def __init__(self, owner=None, value=0.):
        RelayBase.__init__(self, owner)
        _Value.__init__(self, value)


class _Queue:
    """Bounded message queue (connections.py:321-347)."""
    MAX_MESSAGES = 50

    def __init__(self):
        self.queue = collections.deque(maxlen=self.MAX_MESSAGES)

    def push(self, message):
        self.queue.append(message)

    def receive(self):
        while self.queue:
            yield self.queue.popleft()

    def receive_latest(self):
        if not self.queue:
            return None
        latest = self.queue.pop()
        self.queue.clear()
        return latest


class MessageInput(InputBase, _Queue):
    flavour = MESSAGE

    def __init__(self, owner=None):
        InputBase.__init__(self, owner)
        _Queue.__init__(self)


class MessageOutput(OutputBase):
    flavour = MESSAGE

    def send(self, message):
        for dst in self.outgoingConnections:
            dst.push(message)


class MessageRelay(RelayBase, MessageOutput):
    flavour = MESSAGE
    push = MessageOutput.send


_KNOWN = {OutputBase, InputBase, RelayBase, Output, Input, Relay, MessageOutput, MessageInput,
          MessageRelay}
