"""Drop-in for the reference package ``app.analyzers`` (hot path only).

Same module paths and call signatures as reference app/analyzers/{video,fusion,heuristics_v2}.py
so that ``from app.analyzers import video as video_an`` (reference api.py:15) resolves to the
MI355X implementation.  The audio / meta / forensic analyzers are outside this build's scope
 and are not provided here.
"""
from . import fusion, heuristics_v2, video  # noqa: F401
