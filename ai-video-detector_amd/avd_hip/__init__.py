"""MI355X-native per-frame video analysis (hot path of backtato/ai-video-detector).

``avd_hip`` is the host side above the C-ABI of libavd_hip.so:
  _lib      ctypes binding (include/avd.h), fails loudly without the HIP extension
  timeline  float64 scalar tail of reference video.py:54-83 over per-frame records
  analyzer  FrameAnalyzer: frames -> records (HIP) -> result dict, chunked streaming;
            ClipsInFlight: several clips in flight on one GPU (service throughput mode)
  sources   runtime-probed frame sources (decoders are optional host plumbing)
  dist      frame-parallel sharding across ranks + record all-gather (RCCL / gloo)
"""
from ._lib import AvdError, Context, RECORD_DTYPE, build, load  # noqa: F401
from .analyzer import ClipsInFlight, ContextPool, FrameAnalyzer, analyze_frames, default_pool  # noqa: F401
from .timeline import records_to_result, sample_step  # noqa: F401
