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

# as python/__init__.py:49-84 does, the class replaces the same-named submodule attribute
from .spectrum_sensor import spectrum_sensor  # noqa: F401,E402
from .psd_logger import psd_logger  # noqa: F401,E402
from .spectrum_sensor_v1 import spectrum_sensor_v1  # noqa: F401,E402
from .spectrum_sensor_v2 import spectrum_sensor_v2  # noqa: F401,E402
from .message_pdu import message_pdu  # noqa: F401,E402
from .coherence_detector import coherence_detector, coherence_estimator  # noqa: F401,E402
from .multichannel_scanner import multichannel_scanner  # noqa: F401,E402
from .local_worker import local_worker  # noqa: F401,E402
from .spectrum_sweeper import spectrum_sweeper  # noqa: F401,E402
from .flanck_detector import flanck_detector  # noqa: F401,E402
from .ascii_plot import ascii_plot, ascii_plotter  # noqa: F401,E402
