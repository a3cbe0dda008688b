"""Drop-in pieces of the reference's ``deep_sort`` package that sit on the hot path:
``Detection`` (output format), ``create_obj_infos`` (detector output -> detections) and the
cosine ``NearestNeighborDistanceMetric`` (HIP-backed).  ``Tracker`` is the native (C++ core + HIP cosine kernel) restatement of the reference's Kalman
filter / matching cascade / assignment / track life cycle."""
from .detection import Detection  # noqa: F401
from .nn_matching import NearestNeighborDistanceMetric  # noqa: F401
from .tracker import Tracker  # noqa: F401
from .utils import create_obj_arrays, create_obj_infos  # noqa: F401
