"""ripcurrents_amd -- MI355X-native dense Farneback flow + rip-current analysis.

Host-side mirror of the reference's call surface for its one hot path
(cv::calcOpticalFlowFarneback and the per-pixel analysis of ripcurrents_module.cpp)
over the C ABI of librcflow.so.  See include/rcflow.h and DESIGN.md.
"""
from ._lib import RC_FARNEBACK_GAUSSIAN, RcflowError, load  # noqa: F401

OPTFLOW_FARNEBACK_GAUSSIAN = RC_FARNEBACK_GAUSSIAN


def __getattr__(name):
    # api pulls in torch; keep `import ripcurrents_amd` light
    if name in ("Context", "FarnebackParams", "Streakline", "api"):
        from . import api
        return api if name == "api" else getattr(api, name)
    raise AttributeError(name)
