"""Stand-ins for `gnuradio.gr` and `pmt` (test infrastructure; GNU Radio is not installed in this image).

They exist so that the branch of ofdm_tools.gr_compat that a real flowgraph runs - the blocks deriving from
`gnuradio.gr.sync_block`, ports registered through `pmt.intern`, PDUs as `pmt.cons` pairs - executes in the CPU
suite.  The stand-ins are STRICT where GNU Radio 3.7 is strict, so that a call the real runtime would reject fails
here too:
  * message ports are PMT symbols: registering / publishing / handling on a Python str raises TypeError;
  * `set_msg_handler` and `message_port_pub` raise on a port that was never registered;
  * `message_port_pub` only carries PMTs;
  * `pmt.car` / `pmt.cdr` of something that is not a pair raise (pmt.wrong_type in GNU Radio);
  * `pmt.to_pmt` accepts what pmt_to_python.py accepts (None, bool, str, int, float, complex, dict, list, tuple,
    numpy arrays, PMTs) and raises on anything else - a numpy scalar such as np.float32 included.

`installed()` is a context manager: it swaps fresh `gnuradio`, `gnuradio.gr`, `pmt` modules into sys.modules, drops
every loaded `ofdm_tools*` module so that the package is imported again against them, and restores both on exit.
"""
import contextlib
import sys
import types

import numpy as np


# ---------------------------------------------------------------------------------------------- pmt
class _Pmt(object):
    pass


class _Nil(_Pmt):
    def __str__(self):
        return '()'


class _Sym(_Pmt):
    _table = {}

    def __init__(self, name):
        self.name = name

    def __str__(self):
        return self.name


class _Val(_Pmt):
    def __init__(self, value):
        self.value = value

    def __str__(self):
        return str(self.value)


class _Pair(_Pmt):
    def __init__(self, car, cdr):
        self.car, self.cdr = car, cdr

    def __str__(self):
        return '(%s . %s)' % (self.car, self.cdr)


class _U8Vector(_Pmt):
    def __init__(self, data):
        self.data = bytearray(data)

    def __str__(self):
        return '#[' + ' '.join(str(b) for b in self.data) + ']'


class wrong_type(Exception):
    pass


def make_pmt_module():
    m = types.ModuleType('pmt')
    m.PMT_NIL = _Nil()
    m.wrong_type = wrong_type

    def intern(s):
        if not isinstance(s, str):
            raise TypeError('pmt.intern wants a str, got %r' % (s,))
        return _Sym._table.setdefault(s, _Sym(s))

    def to_pmt(v):
        if isinstance(v, _Pmt):
            return v
        if v is None:
            return m.PMT_NIL
        if isinstance(v, str):
            return intern(v)
        if isinstance(v, (bool, int, float, complex)):
            return _Val(v)
        if isinstance(v, dict):
            return _Val({k: to_pmt(x) for k, x in v.items()})
        if isinstance(v, (list, tuple)):
            return _Val([to_pmt(x) for x in v])
        if isinstance(v, np.ndarray):
            return _Val(v.copy())
        raise ValueError('pmt.to_pmt: cannot convert %r of type %s' % (v, type(v).__name__))

    def to_python(p):
        if not isinstance(p, _Pmt):
            raise TypeError('not a PMT: %r' % (p,))
        if isinstance(p, _Nil):
            return None
        if isinstance(p, _Sym):
            return p.name
        if isinstance(p, _Pair):
            return (to_python(p.car), to_python(p.cdr))
        if isinstance(p, _U8Vector):
            return np.frombuffer(bytes(p.data), np.uint8)
        v = p.value
        if isinstance(v, dict):
            return {k: to_python(x) for k, x in v.items()}
        if isinstance(v, list):
            return [to_python(x) for x in v]
        return v

    def cons(a, b):
        if not (isinstance(a, _Pmt) and isinstance(b, _Pmt)):
            raise TypeError('pmt.cons wants two PMTs, got %r, %r' % (a, b))
        return _Pair(a, b)

    def car(p):
        if not isinstance(p, _Pair):
            raise wrong_type('pmt_car: not a pair: %r' % (p,))
        return p.car

    def cdr(p):
        if not isinstance(p, _Pair):
            raise wrong_type('pmt_cdr: not a pair: %r' % (p,))
        return p.cdr

    def init_u8vector(n, data):
        data = list(data)
        if len(data) != n or any(not isinstance(b, int) or not 0 <= b < 256 for b in data):
            raise ValueError('init_u8vector(%d, ...): %d values' % (n, len(data)))
        return _U8Vector(data)

    m.intern, m.string_to_symbol = intern, intern
    m.to_pmt, m.to_python, m.cons, m.car, m.cdr = to_pmt, to_python, cons, car, cdr
    m.init_u8vector = init_u8vector
    m.u8vector_elements = lambda v: list(v.data)
    m.is_pair = lambda p: isinstance(p, _Pair)
    m.is_symbol = lambda p: isinstance(p, _Sym)
    m.is_u8vector = lambda p: isinstance(p, _U8Vector)
    m.symbol_to_string = lambda p: p.name
    m.length = lambda p: len(p.data)
    return m


# ---------------------------------------------------------------------------------------------- gnuradio.gr
def _want_symbol(port):
    if not isinstance(port, _Sym):
        raise TypeError('message port ids are PMT symbols (pmt.intern), got %r' % (port,))
    return port.name


class sync_block(object):
    """What gr.sync_block's Python gateway offers the blocks of this package (GNU Radio 3.7
    gnuradio/gr/gateway.py): keyword constructor, symbol-keyed message ports, work() driven by the scheduler."""

    def __init__(self, name, in_sig, out_sig):
        self.gr_name, self.gr_in_sig, self.gr_out_sig = name, in_sig, out_sig
        self.gr_in_ports, self.gr_out_ports, self.gr_handlers = [], [], {}
        self.gr_published = []

    def name(self):
        return self.gr_name

    def message_port_register_in(self, port):
        self.gr_in_ports.append(_want_symbol(port))

    def message_port_register_out(self, port):
        self.gr_out_ports.append(_want_symbol(port))

    def set_msg_handler(self, port, fn):
        name = _want_symbol(port)
        if name not in self.gr_in_ports:
            raise RuntimeError('set_msg_handler: port %r is not registered' % name)
        self.gr_handlers[name] = fn

    def message_port_pub(self, port, msg):
        name = _want_symbol(port)
        if name not in self.gr_out_ports:
            raise RuntimeError('message_port_pub: port %r is not registered' % name)
        if not isinstance(msg, _Pmt):
            raise TypeError('message_port_pub carries PMTs, got %r' % (msg,))
        self.gr_published.append((name, msg))

    def scheduler_post(self, port_name, msg):
        """What the scheduler does with a message that arrives on a connected input port."""
        self.gr_handlers[port_name](msg)


def make_gnuradio_modules():
    pkg = types.ModuleType('gnuradio')
    pkg.__path__ = []
    gr = types.ModuleType('gnuradio.gr')
    gr.sync_block = sync_block
    gr.sizeof_gr_complex, gr.sizeof_float = 8, 4
    pkg.gr = gr
    return pkg, gr


@contextlib.contextmanager
def installed():
    """sys.modules with the stand-ins and a freshly imported ofdm_tools; yields (ofdm_tools, pmt, gr)."""
    ours = lambda k: k == 'ofdm_tools' or k.startswith('ofdm_tools.')      # noqa: E731
    saved = {k: v for k, v in sys.modules.items() if ours(k) or k in ('gnuradio', 'gnuradio.gr', 'pmt')}
    for k in saved:
        del sys.modules[k]
    pkg, gr = make_gnuradio_modules()
    pmt = make_pmt_module()
    sys.modules.update({'gnuradio': pkg, 'gnuradio.gr': gr, 'pmt': pmt})
    try:
        import ofdm_tools
        yield ofdm_tools, pmt, gr
    finally:
        for k in [k for k in sys.modules if ours(k) or k in ('gnuradio', 'gnuradio.gr', 'pmt')]:
            del sys.modules[k]
        sys.modules.update(saved)
