"""Drop-in for the reference package ``app.analyzers`` (hot path only).

Same module paths and call signatures as reference app/analyzers/{video,fusion,heuristics_v2}.py
so that ``from app.analyzers import video as video_an`` (reference api.py:15) resolves to the
MI355X implementation.  The audio / meta / forensic analyzers are outside this build's scope
(SURVEY.md section 8): the package search path is extended to any other ``app/analyzers`` directory on
``sys.path`` (see ``app/__init__.py``), so with a reference checkout behind this package
``from app.analyzers import audio, meta`` (api.py:14,18) import the reference's own modules.
Unlike the reference's ``__init__`` (which eagerly imports all six analyzers, hence cv2 and soundfile),
only the three hot-path modules are imported here.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from . import fusion, heuristics_v2, video  # noqa: E402,F401
