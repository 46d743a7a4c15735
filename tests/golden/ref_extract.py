"""Runs the reference's OWN numeric helper functions (build container only).

The reference modules cannot be imported (Python-2 ``print`` statements, ``import gnuradio``), but the
helper functions on the hot path are plain NumPy/SciPy code that parses under Python 3 one function at a
time.  ``load(names)`` reads ``/root/reference/python/<module>.py`` AT GENERATION TIME, cuts out each
named top-level ``def`` by its text extent, checks with ``ast`` that the cut is exactly one function
definition, and ``exec``s it against the real ``numpy`` / ``scipy.signal`` / ``math``.  Nothing of the
reference's text is written anywhere: only the numbers the functions return go into ``tests/golden``
(``make_golden.py``), and nothing here is importable on the GPU box (``/root/reference`` is absent
there; callers skip).

Three helpers need the ENVIRONMENT of the reference's day, not different text.  Their function bodies are exec'd
unmodified as well; only the namespace they run in is the older one (``py2_namespace()``):
  * ``src_power_fft`` calls ``sg.flattop``: SciPy up to 1.12 re-exported the window functions from ``scipy.signal``
    (removed in 1.13).  The ``sg`` handed to it resolves names in ``scipy.signal`` first and in
    ``scipy.signal.windows`` second - the same function object either way.
  * ``xcorr`` / ``fac`` slice with ``h[len(h)/2:]``: under Python 2 ``int / int`` floors.  The ``len`` handed to them
    returns an ``int`` subclass whose ``/`` by an int floors (``Py2Int``, also used for ``ascii_plotter.make_plot``).

Methods that contain a Python-2 ``print`` STATEMENT do not parse under Python 3 at all.  ``load_method(...,
py2_print=True)`` passes the cut text through ``lib2to3``'s ``fix_print`` - the standard library's own Python-2 to
Python-3 translator, restricted to that one fixer - before compiling it: ``print a, b`` becomes ``print(a, b)`` and
nothing else changes (no arithmetic, no control flow).  The thread bodies and packers pinned that way
(``stats_watcher.spectrum_scanner``, ``psd_watcher.run``, ``waterfall_watcher.run``, ``main_thread.run``,
``data_colector.run``, ``spectrum_stitcher.run``, both ``packet_source.send_packet``) run against stand-in queues,
receivers and ports, and against two more pieces of the environment of their day:
  * ``np.maximum(x, None)``: the logger starts its peak arrays at ``None`` and Python 2 ordered ``None`` below every
    number, so the first ``np.maximum`` returned the data (``NumpyOfItsDay.maximum``);
  * ``ord(frame[i])`` over a byte string: indexing a Python-2 ``str`` gives a 1-character ``str``, indexing Python-3
    ``bytes`` gives the ``int`` already (``py2_ord``).
The two consumers of the fragment format (round 5: ``remote_client_qt.handler``, ``data_processor.run`` of the web
server) take a received frame as a Python-2 ``str``: ``Str2`` is that type (bytes whose elements are 1-byte strings);
``np.fromstring(bytes, dtype)`` - removed from NumPy 2 - is ``np.frombuffer`` (same ValueError on a ragged length).
"""
import ast
import math
import os

import numpy as np
import scipy.signal as sg

REF_PY = '/root/reference/python'


class Py2Int(int):
    """An int whose true division by an int floors, as Python 2's ``/`` did."""

    def __truediv__(self, other):
        return Py2Int(int(self) // other) if isinstance(other, int) else int(self) / other


class _SignalOfItsDay(object):
    """scipy.signal as the reference imported it: window functions reachable as ``sg.<name>`` (SciPy <= 1.12)."""

    def __getattr__(self, name):
        if hasattr(sg, name):
            return getattr(sg, name)
        return getattr(sg.windows, name)


class NumpyOfItsDay(object):
    """numpy as the reference's threads used it under Python 2: ``maximum`` with a ``None`` operand returns the other
    operand (``None`` ordered below every number there); everything else is numpy's own."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def maximum(a, b):
        if a is None:
            return b
        if b is None:
            return a
        return np.maximum(a, b)


def py2_ord(c):
    """``ord`` over the elements of a byte string: Python-3 ``bytes`` index to ints already."""
    return c if isinstance(c, int) else ord(c)


def _fix_py2_print(text, fixers=('print',)):
    """The cut text with Python-2 print statements rewritten as calls - lib2to3's fix_print and nothing else (round 5:
    or the named fixers - ``except`` for ``except Exception, e:``, which does not parse under Python 3 either)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        from lib2to3 import refactor
        tool = refactor.RefactoringTool(['lib2to3.fixes.fix_' + f for f in fixers])
        return str(tool.refactor_string(text if text.endswith('\n') else text + '\n', '<reference>'))


def py2_namespace():
    """Globals for the helpers that rely on Python-2 ``/`` on ``len()`` or on the old ``scipy.signal`` re-exports."""
    return {'sg': _SignalOfItsDay(), 'len': lambda v: Py2Int(len(v))}


def available():
    return os.path.isdir(REF_PY)


def _cut(lines, name):
    """Text extent of the top-level ``def name(``: up to the next statement that starts in column 0."""
    start = None
    for i, ln in enumerate(lines):
        if ln.startswith('def %s(' % name):
            start = i
            break
    if start is None:
        raise KeyError(name)
    end = len(lines)
    for j in range(start + 1, len(lines)):
        ln = lines[j]
        if ln.strip() and not ln[0].isspace() and not ln.startswith('#'):
            end = j
            break
    return start, ''.join(lines[start:end])


def load(module, names, namespace=None):
    """-> dict name -> function object compiled from the reference's own lines."""
    path = os.path.join(REF_PY, module)
    with open(path) as fh:
        lines = fh.readlines()
    ns = {'np': np, 'sg': sg, 'math': math, '__builtins__': __builtins__}
    if namespace:
        ns.update(namespace)
    out = {}
    for name in names:
        start, text = _cut(lines, name)
        tree = ast.parse(text)
        assert len(tree.body) == 1 and isinstance(tree.body[0], ast.FunctionDef) and tree.body[0].name == name
        # keep the reference's line numbers in tracebacks
        code = compile('\n' * start + text, path, 'exec')
        exec(code, ns)
        out[name] = ns[name]
    return out


def load_method(module, cls, name, namespace=None, py2_print=False, fixers=('print',)):
    """A method of a reference class as a plain function taking ``self`` (the caller passes a stand-in object
    carrying the attributes the method reads): the ``def`` is cut by its indentation inside ``class cls``.
    py2_print: the method holds Python-2 print statements - see the module docstring."""
    path = os.path.join(REF_PY, module)
    with open(path) as fh:
        lines = fh.readlines()
    c0 = next(i for i, ln in enumerate(lines) if ln.startswith('class %s(' % cls))
    start = None
    for i in range(c0 + 1, len(lines)):
        ln = lines[i]
        if ln.strip() and not ln[0].isspace():
            break
        if ln.lstrip().startswith('def %s(' % name):
            start = i
            break
    if start is None:
        raise KeyError('%s.%s' % (cls, name))
    indent = lines[start][:len(lines[start]) - len(lines[start].lstrip())]
    end = len(lines)
    for j in range(start + 1, len(lines)):
        ln = lines[j]
        if not ln.strip():
            continue
        lead = ln[:len(ln) - len(ln.lstrip())]
        if len(lead) <= len(indent):
            end = j
            break
    text = ''.join(ln[len(indent):] if ln.startswith(indent) else ln.lstrip() for ln in lines[start:end])
    if py2_print:
        text = _fix_py2_print(text, fixers)
    tree = ast.parse(text)
    assert len(tree.body) == 1 and isinstance(tree.body[0], ast.FunctionDef) and tree.body[0].name == name
    ns = {'np': np, 'sg': sg, 'math': math, '__builtins__': __builtins__}
    if namespace:
        ns.update(namespace)
    exec(compile('\n' * start + text, path, 'exec'), ns)
    return ns[name]


class Str2(bytes):
    """A Python-2 ``str``: a byte string whose elements are 1-byte strings (``s[0]`` goes to ``struct.unpack``) and
    that concatenates with the empty text literal the consumers start from (``self.reasembled_frame = ''``)."""

    def __getitem__(self, i):
        return Str2(bytes.__getitem__(self, slice(i, i + 1 if i != -1 else None) if isinstance(i, int) else i))

    def __add__(self, other):
        return Str2(bytes.__add__(self, other))

    def __radd__(self, other):
        assert other == '' or isinstance(other, bytes), 'only the empty text literal starts a byte string'
        return Str2(bytes(other or b'') + bytes(self))


CR_TOOLS = ('frange', 'clc_power_freq', 'movingaverage', 'src_power', 'src_power_welch', 'welch_plot_dB',
            'welch_power_estimate', 'fast_spectrum_scan')
SWEEPER = ('frange', '_src_power')


def cr_tools():
    return load('ofdm_cr_tools.py', CR_TOOLS)


def sweeper():
    return load('spectrum_sweeper.py', SWEEPER)


def load_methods(module, cls, names, namespace=None, py2_print=False):
    """-> a fresh class whose methods ARE the reference's (``load_method`` each): instantiate it without the
    reference's ``__init__`` and set the attributes that constructor would have set."""
    return type('Ref_' + cls, (object,),
                {n: load_method(module, cls, n, namespace, py2_print) for n in names})
