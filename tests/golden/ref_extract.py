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


def load_method(module, cls, name, namespace=None):
    """A method of a reference class as a plain function taking ``self`` (the caller passes a stand-in object
    carrying the attributes the method reads): the ``def`` is cut by its indentation inside ``class cls``."""
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
    tree = ast.parse(text)
    assert len(tree.body) == 1 and isinstance(tree.body[0], ast.FunctionDef) and tree.body[0].name == name
    ns = {'np': np, 'sg': sg, 'math': math, '__builtins__': __builtins__}
    if namespace:
        ns.update(namespace)
    exec(compile('\n' * start + text, path, 'exec'), ns)
    return ns[name]


CR_TOOLS = ('frange', 'clc_power_freq', 'movingaverage', 'src_power', 'src_power_welch', 'welch_plot_dB',
            'welch_power_estimate', 'fast_spectrum_scan')
SWEEPER = ('frange', '_src_power')


def cr_tools():
    return load('ofdm_cr_tools.py', CR_TOOLS)


def sweeper():
    return load('spectrum_sweeper.py', SWEEPER)
