"""TMOT / JDE tracker behind the reference's ``tmot`` surface (reference tmot/multitracker.py)."""
from .multitracker import BaseTrack, JDETracker, STrack, TrackState  # noqa: F401
