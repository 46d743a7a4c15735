"""ofdm_tools - MI355X-native spectrum-sensing blocks with the gr-ofdm_tools API.

Same import name and class / constructor signatures as the reference package
(python/__init__.py:49-84) for the blocks on the sensing hot path; the PSD
arithmetic runs in hand-written HIP kernels behind libofdmtools_hip.so
(include/ofdm_tools_hip.h).  Importing the package does not touch the GPU; the
first block or helper that computes loads the library and fails loudly if it
(or a GPU) is missing - there is no CPU fallback.
"""
from . import ofdm_cr_tools  # noqa: F401
from . import windows  # noqa: F401
from ._hip import HipError, HipUnavailable  # noqa: F401

_LAZY = {
    'spectrum_sensor_v2': 'spectrum_sensor_v2',
    'psd_logger': 'psd_logger',
    'coherence_detector': 'coherence_detector',
    'coherence_estimator': 'coherence_detector',
    'spectrum_sweeper': 'spectrum_sweeper',
    'multichannel_scanner': 'multichannel_scanner',
    'local_worker': 'local_worker',
    'spectrum_sensor': 'spectrum_sensor',
    'message_pdu': 'message_pdu',
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod = importlib.import_module('.' + _LAZY[name], __name__)
        return getattr(mod, name)
    raise AttributeError('module %r has no attribute %r' % (__name__, name))
