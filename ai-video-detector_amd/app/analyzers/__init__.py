"""Drop-in for the reference package ``app.analyzers`` (hot path only).

Same module paths and call signatures as reference app/analyzers/{video,fusion,heuristics_v2}.py
so that ``from app.analyzers import video as video_an`` (reference api.py:15) resolves to the
MI355X implementation; ``audio`` is provided as well.  The meta / forensic analyzers are
outside this build's scope: the package search path is extended to any other ``app/analyzers`` directory on
``sys.path`` (see ``app/__init__.py``), so with a reference checkout behind this package
``from app.analyzers import meta`` (api.py:18) imports the reference's own module.
Unlike the reference's ``__init__`` (which eagerly imports all six analyzers, hence cv2 and soundfile),
only the modules of this build are imported here.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from . import audio, fusion, heuristics_v2, video  # noqa: E402,F401
