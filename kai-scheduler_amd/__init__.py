"""kai-scheduler_amd — host side of the MI355X-native KAI scheduling-cycle core.

Holds only what the hot path needs: `csrc/` (HIP kernels + the C ABI of
include/kai_core.h), `abi` (ctypes mirror + structure-of-arrays snapshot), `core` (the reference's
Session / Action interface re-exposed over the C ABI) and `synth` (synthetic cluster snapshots of
BASELINE.json's configs) and `ingest` (reference-schema snapshot.json / snapshot.zip -> structure-of-arrays snapshot).
"""
from . import abi, core, dist, ingest, synth  # noqa: F401
from .core import KaiCore, KaiError, Session, load_library  # noqa: F401

__all__ = ["abi", "core", "dist", "ingest", "synth", "KaiCore", "KaiError", "Session", "load_library"]
