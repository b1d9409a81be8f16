"""Drop-in ``app`` package (hot path only) that can sit IN FRONT of the reference's ``app`` on sys.path.

The reference's ``api.py:14-18`` imports five modules from ``app.analyzers``; this build provides the ones on the
hot path (``video``, ``fusion``, ``heuristics_v2``) and leaves the rest (``audio``, ``meta``, ``forensic``) to the
reference.  ``extend_path`` appends every other ``app`` directory found on ``sys.path`` to this package's search
path, so ``from app.analyzers import meta`` falls through to the reference checkout while
``from app.analyzers import video`` resolves here (first entry wins).
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
