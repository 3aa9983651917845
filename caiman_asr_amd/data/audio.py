"""Audio files -> mono f32 samples through the library's host decoders (include/caiman_data.h).
Replaces DALI's file reader + audio decoder (training/caiman_asr_train/data/dali/pipeline.py:253-259,400-414)."""
import ctypes
from typing import List, Sequence, Tuple

import numpy as np

from caiman_asr_amd import _lib


def audio_info(data: bytes) -> Tuple[int, int, int]:
    """-> (sample_rate, channels, frames)."""
    sr, ch, fr = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
    buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
    _lib.check(_lib.lib().caiman_audio_info(buf, len(data), ctypes.byref(sr), ctypes.byref(ch), ctypes.byref(fr)))
    return sr.value, ch.value, fr.value


def decode_audio(data: bytes) -> Tuple[np.ndarray, int]:
    """One in-memory FLAC / WAV file -> (samples f32 [n], sample_rate)."""
    sr, ch, frames = audio_info(data)
    cap = frames if frames > 0 else len(data) * 16
    out = np.empty(cap, dtype=np.float32)
    n, rate = ctypes.c_int64(), ctypes.c_int32()
    buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
    _lib.check(_lib.lib().caiman_audio_decode(buf, len(data), out.ctypes.data, cap, ctypes.byref(n), ctypes.byref(rate)))
    return out[: n.value], rate.value


def decode_files(paths: Sequence[str], out: np.ndarray, n_threads: int = 8) -> Tuple[np.ndarray, np.ndarray]:
    """Read + decode `paths` into out [len(paths), max_frames] (f32, C-contiguous, e.g. the numpy view of a pinned
    tensor), zero padded -> (lengths i32 [n], sample_rates i32 [n])."""
    n = len(paths)
    assert out.dtype == np.float32 and out.flags.c_contiguous and out.shape[0] >= n
    arr = (ctypes.c_char_p * max(n, 1))(*[p.encode() for p in paths])
    lens, rates = np.zeros(n, np.int32), np.zeros(n, np.int32)
    _lib.check(_lib.lib().caiman_audio_decode_files(arr, n, out.ctypes.data, out.shape[1], lens.ctypes.data,
                                                    rates.ctypes.data, n_threads))
    return lens, rates
