"""Drop-in pieces of the reference's ``deep_sort`` package that sit on the hot path:
``Detection`` (output format), ``create_obj_infos`` (detector output -> detections) and the
cosine ``NearestNeighborDistanceMetric`` (HIP-backed).  The Kalman filter / Hungarian cascade /
Track state machine stay host code of the reference and are a "next" row."""
from .detection import Detection  # noqa: F401
from .nn_matching import NearestNeighborDistanceMetric  # noqa: F401
from .utils import create_obj_infos  # noqa: F401
