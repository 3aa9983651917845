"""Decoder output records: same field names as training/caiman_asr_train/rnnt/response.py:7-80 so
downstream consumers (WER, CTM, websocket serialisation) read them unchanged."""
from dataclasses import dataclass
from typing import List, Optional


@dataclass
class HypothesisResponse:
    y_seq: List[int]
    timesteps: List[int]
    token_seq: List[str]
    confidence: List[float]


@dataclass
class DecodingResponse:
    start_frame_idx: int
    duration_frames: int
    is_provisional: bool
    alternatives: List[HypothesisResponse]


@dataclass
class FrameResponses:
    partials: Optional[DecodingResponse]
    final: Optional[DecodingResponse]
