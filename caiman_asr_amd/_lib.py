"""Build + load the gfx950 shared library and expose typed ctypes entry points.

The library is plain C-ABI (include/caiman_rnnt.h): device pointers, extents, a dtype
tag and a hipStream_t.  torch is used here only to obtain `data_ptr()` and the current
stream — plumbing, not compute.  If the library cannot be loaded every op raises
(`MissingNativeLibrary`); there is deliberately no fallback path.
"""
import ctypes
import glob
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcaiman_rnnt.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "caiman_rnnt.h")
HEADERS = sorted(glob.glob(os.path.join(os.path.dirname(HEADER), "*.h")))

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment"]
# per-file flags.  proj_gemm.hip: keep the MFMA accumulators in VGPRs -- left to itself the register allocator splits
# the 128 accumulator registers of the 256 x 128 tile between VGPRs and AGPRs and moves ~150 of them per K step
FILE_FLAGS = {"proj_gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "joint_gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
              "joint_wgrad.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


class MissingNativeLibrary(RuntimeError):
    pass


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every .hip translation unit for gfx950 and link libcaiman_rnnt.so in-tree."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    objs, procs = [], []
    hdr_t = max(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h")) + HEADERS)
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and os.path.getmtime(obj) > hdr_t):
            continue
        cmd = [HIPCC, *HIP_FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


class FwdSlot(ctypes.Structure):
    _fields_ = [("weights_tiled", ctypes.c_void_p), ("gates", ctypes.c_void_p), ("c", ctypes.c_void_p),
                ("y", ctypes.c_void_p), ("ring", ctypes.c_void_p), ("parity", ctypes.c_int32),
                ("nsteps", ctypes.c_int32), ("y_masked", ctypes.c_void_p), ("drop_counter", ctypes.c_uint64),
                ("drop_p", ctypes.c_float), ("hidden", ctypes.c_int32)]


class BwdSlot(ctypes.Structure):
    _fields_ = [("weights_tiled", ctypes.c_void_p), ("gates", ctypes.c_void_p), ("c", ctypes.c_void_p),
                ("delta", ctypes.c_void_p), ("delta_stride_t", ctypes.c_int64), ("delta_stride_b", ctypes.c_int64),
                ("dG", ctypes.c_void_p), ("ring", ctypes.c_void_p), ("dC", ctypes.c_void_p),
                ("parity", ctypes.c_int32), ("nsteps", ctypes.c_int32), ("has_next", ctypes.c_int32),
                ("drop_p", ctypes.c_float), ("drop_counter", ctypes.c_uint64), ("hidden", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("dbias", ctypes.c_void_p)]

class ProjProblem(ctypes.Structure):
    """caiman_proj_problem_t (include/caiman_rnnt.h): one problem of a grouped input-projection GEMM."""
    _fields_ = [("a", ctypes.c_void_p), ("w", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("c", ctypes.c_void_p),
                ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
                ("a_inner", ctypes.c_int32), ("a_kseg", ctypes.c_int32), ("c_inner", ctypes.c_int32),
                ("c_nseg", ctypes.c_int32),
                ("a_stride_outer", ctypes.c_int64), ("a_stride_inner", ctypes.c_int64), ("a_stride_seg", ctypes.c_int64),
                ("c_stride_outer", ctypes.c_int64), ("c_stride_inner", ctypes.c_int64), ("c_stride_seg", ctypes.c_int64)]


class LstmImages(ctypes.Structure):
    """caiman_lstm_images_t (include/caiman_rnnt.h): one layer of caiman_lstm_weight_images."""
    _fields_ = [("W_ih", ctypes.c_void_p), ("W_hh", ctypes.c_void_p), ("b_ih", ctypes.c_void_p), ("b_hh", ctypes.c_void_p),
                ("Wt", ctypes.c_void_p), ("Wn", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("Rf", ctypes.c_void_p),
                ("Rb", ctypes.c_void_p), ("H", ctypes.c_int32), ("K", ctypes.c_int32)]


class GradItem(ctypes.Structure):
    """caiman_lstm_grad_item_t (include/caiman_rnnt.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("H", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("src_fp32", ctypes.c_int32), ("slabs", ctypes.c_int32)]


class BeamConfig(ctypes.Structure):
    """caiman_beam_config_t (include/caiman_beam.h)."""
    _fields_ = [("blank_idx", ctypes.c_int32), ("beam_width", ctypes.c_int32),
                ("max_symbols_per_step", ctypes.c_int32), ("max_symbol_per_sample", ctypes.c_int32),
                ("beam_prune_score_thresh", ctypes.c_double), ("beam_prune_topk_thresh", ctypes.c_double),
                ("eos_vad_threshold", ctypes.c_double), ("final_emission_thresh", ctypes.c_double),
                ("frame_width", ctypes.c_double), ("eos_terminal_idx", ctypes.c_int32),
                ("return_partials", ctypes.c_int32), ("max_expansions_per_frame", ctypes.c_int32)]


P = ctypes.c_void_p
I64 = ctypes.c_int64
I32 = ctypes.c_int
U32 = ctypes.c_uint32
F64 = ctypes.c_double
F32 = ctypes.c_float

_SIGS = {
    "caiman_abi_version": ([], ctypes.c_int),
    "caiman_last_error": ([], ctypes.c_char_p),
    "caiman_built_for_gfx950": ([], ctypes.c_int),
    "caiman_logsumexp": ([P, I64, I64, I64, I32, P, I32, U32, P], ctypes.c_int),
    "caiman_transducer_loss_forward": (
        [P, P, P, P, P, P, I64, I64, I64, I64, F64, I64, F64, I64, F64, I64, I32, I32, P, P, P, P],
        ctypes.c_int),
    "caiman_transducer_loss_backward": (
        [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, F64, I64, F64, I64, F64, I64, I32, I32,
         P, P], ctypes.c_int),
    "caiman_transducer_loss_backward_colsum": (
        [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, F64, I64, F64, I64, F64, I64, I32, I32,
         P, P, I64, P], ctypes.c_int),
    "caiman_lstm_weight_images": ([P, I32, I32, P], ctypes.c_int),
    "caiman_lstm_grad_deliver": ([P, I32, I32, P], ctypes.c_int),
    "caiman_proj_gemm_supported": ([P, I32], ctypes.c_int),
    "caiman_joint_fc_workspace_elems": ([I64, I64], ctypes.c_int64),
    "caiman_joint_fc_supported": ([I64, I64, I64, I32], ctypes.c_int),
    "caiman_joint_fc_forward": ([P, P, P, P, P, P, I64, I64, I64, I32, P], ctypes.c_int),
    "caiman_joint_fc_wgrad_plan": ([I64, I64, I64, I32, P], ctypes.c_int),
    "caiman_joint_fc_wgrad": ([P, P, P, I64, I64, I64, I32, I64, I32, P], ctypes.c_int),
    "caiman_wgrad_tn_plan": ([I64, I64, I64, I32, I32, P], ctypes.c_int),
    "caiman_wgrad_tn_estimate_us": ([I64, I64, I64, I32, I32], ctypes.c_double),
    "caiman_slab_accumulate": ([P, I32, I64, P, P], ctypes.c_int),
    "caiman_wgrad_tn_covers_remainder": ([I64, I64, I64, I32, I64], ctypes.c_int),
    "caiman_wgrad_tn": ([P, I64, P, I64, P, I32, I64, I64, I64, I32, I64, I32, P], ctypes.c_int),
    "caiman_wgrad_tn2": ([P, I64, P, I64, I32, P, I64, P, I64, I32, P, I64, I64, I64, I32, I64, I32, P], ctypes.c_int),
    "caiman_proj_gemm": ([P, I32, I32, I32, P], ctypes.c_int),
    "caiman_lstm_fused_fwd": ([P, P, P, P, P, I64, I64, I64, I32, I32, P], ctypes.c_int),
    "caiman_lstm_workspace_elems": ([I64, I64, I32], ctypes.c_int64),
    "caiman_lstm_prepare": ([P, P, P, P, P, I64, I64, I32, I32, I32, P], ctypes.c_int),
    "caiman_lstm_wave_fwd": ([P, I32, I32, I64, I64, I32, I32, I32, ctypes.c_uint64, P], ctypes.c_int),
    "caiman_lstm_wave_bwd": ([P, I32, I32, I64, I64, I32, I32, I32, ctypes.c_uint64, P], ctypes.c_int),
    "caiman_lstm_dropout_mask": ([P, I64, ctypes.c_uint64, ctypes.c_uint64, F32, I32, P], ctypes.c_int),
    "caiman_lstm_resident_mode": ([I32], ctypes.c_int),
    "caiman_lstm_resident_failures": ([], ctypes.c_int),
    "caiman_lstm_resident_xcd_roles": ([I32], ctypes.c_int),
    "caiman_lstm_resident_bwd_split": ([I32], ctypes.c_int),
    "caiman_lstm_resident_bt_dma": ([I32], ctypes.c_int),
    "caiman_lstm_resident_profile_bwd2": ([P], ctypes.c_int),
    "caiman_lstm_resident_set_failures": ([I32], ctypes.c_int),
    "caiman_lstm_resident_poison": ([P, P, P], ctypes.c_int),
    "caiman_debug_occupy_cus": ([I32, I32, P], ctypes.c_int),
    "caiman_lstm_resident_launches": ([], ctypes.c_int64),
    "caiman_lstm_resident_would_run": ([I64, I64, I32], ctypes.c_int),
    "caiman_lstm_resident_profile": ([P], ctypes.c_int),
    "caiman_logmel_forward": ([P, P, I64, I64, I32, I32, I32, I32, I32, F32, F32, ctypes.c_uint64, F32, P, P, P, P, P, P,
                               P, P, I64, P], ctypes.c_int),
    "caiman_mel_normalize": ([P, P, I64, I32, I64, P, P, F32, P], ctypes.c_int),
    "caiman_embedding_grad": ([P, I64, P, I32, I64, I64, P, P], ctypes.c_int),
    "caiman_lstm_last_states": ([P, P, I64, I64, I64, I64, I64, I64, I64, I64, P, I32, I32, P, P, P], ctypes.c_int),
    "caiman_specaug_geometry": ([P, P, I32, I64, I64, I64, I32, F32, F32, F32, I32, F32, F32, P, P], ctypes.c_int),
    "caiman_specaug_splice": ([P, I64, I64, I64, P, P, I32, P, P, I32, I32, I32, I64, P, P], ctypes.c_int),
    "caiman_joint_forward": ([P, P, P, P, P, I64, I64, I64, I64, I64, I32, I32, F64, ctypes.c_uint64, I32, P, P],
                             ctypes.c_int),
    "caiman_joint_backward": ([P, P, P, P, P, I64, I64, I64, I64, I32, I32, F64, I32, P, P, P], ctypes.c_int),
    "caiman_lamb_step": ([P, P, P, P, P, P, P, P, I64, P, P, I64, P, P, I32, F32, F32, F32, F32, F32, F32, I32, I32,
                          I32, P, P, P], ctypes.c_int),
    "caiman_lstm_fused_bwd": ([P, P, P, P, I64, I64, P, P, P, I64, I64, I64, I32, I32, P], ctypes.c_int),
    # include/caiman_data.h
    "caiman_audio_info": ([P, I64, P, P, P], ctypes.c_int),
    "caiman_audio_decode": ([P, I64, P, I64, P, P], ctypes.c_int),
    "caiman_audio_decode_files": ([ctypes.POINTER(ctypes.c_char_p), I32, P, I64, P, P, I32], ctypes.c_int),
    "caiman_levenshtein": ([P, I64, P, I64], ctypes.c_int64),
    # include/caiman_beam.h
    "caiman_beam_topk": ([P, I64, I64, I64, I32, F32, I32, I32, I32, F32, F32, I32, P, P, P, P], ctypes.c_int),
    "caiman_beam_gather_inputs": ([P, I64, P, I64, P, P, I64, P, I64, I32, P], ctypes.c_int),
    "caiman_beam_lstm_cell": ([P, I64, P, P, P, P, P, I64, P, I64, I32, P], ctypes.c_int),
    "caiman_lstm_step_gemm": ([P, I64, P, P, I64, I64, I64, P, P, P, P, P, P, I64, I32, P], ctypes.c_int),
    "caiman_beam_joint_act": ([P, P, P, I64, I64, P, I32, P], ctypes.c_int),
    "caiman_beam_create": ([ctypes.POINTER(BeamConfig), I32, ctypes.POINTER(ctypes.c_char_p), I32,
                            ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_double), I32], ctypes.c_void_p),
    "caiman_beam_destroy": ([P], None),
    "caiman_beam_reset_stream": ([P, I32], ctypes.c_int),
    "caiman_beam_push_frame": ([P, P, I32], ctypes.c_int),
    "caiman_beam_requests": ([P, P, P, P, P, P, I64], ctypes.c_int64),
    "caiman_beam_feed": ([P, I64, I32, P, P, P], ctypes.c_int),
    "caiman_beam_close_stream": ([P, I32], ctypes.c_int),
    "caiman_beam_stream_done": ([P, I32], ctypes.c_int),
    "caiman_beam_state_slots": ([P], ctypes.c_int64),
    "caiman_beam_backlog": ([P, I32], ctypes.c_int64),
    "caiman_beam_capped_frames": ([P], ctypes.c_int64),
    "caiman_beam_responses": ([P, ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)), ctypes.POINTER(ctypes.c_int64),
                               ctypes.POINTER(ctypes.POINTER(ctypes.c_float)), ctypes.POINTER(ctypes.c_int64)],
                              ctypes.c_int),
    "caiman_beam_clear_responses": ([P], None),
}


def exported_symbols():
    """Names declared in include/*.h (used by the CPU-side ABI test)."""
    import re

    text = "".join(open(h).read() for h in HEADERS)
    return sorted(set(re.findall(r"\b(caiman_[a-z0-9_]+)\s*\(", text)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MissingNativeLibrary(
                f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'`. "
                "There is no CPU/PyTorch fallback for the RNN-T kernels.")
        try:
            # CAIMAN_LIB_OVERRIDE: a measurement build of the same sources (A/B of compile-time variants on one box)
            L = ctypes.CDLL(os.environ.get("CAIMAN_LIB_OVERRIDE") or LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise MissingNativeLibrary(f"cannot load {LIB_PATH}: {e}") from e
        for name, (argtypes, restype) in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = L
        if os.environ.get("CAIMAN_LSTM_RESIDENT", "") in ("0", "1", "2"):   # weight-resident LSTM chunk kernels (csrc/lstm.hip)
            L.caiman_lstm_resident_mode(int(os.environ["CAIMAN_LSTM_RESIDENT"]))
        if os.environ.get("CAIMAN_LSTM_XCD_ROLES", "") in ("0", "1"):       # layer -> XCD placement of the resident launches (A/B runs)
            L.caiman_lstm_resident_xcd_roles(int(os.environ["CAIMAN_LSTM_XCD_ROLES"]))
        if os.environ.get("CAIMAN_LSTM_BWD_SPLIT", "") in ("0", "1"):       # 2-D split backward resident kernel (A/B runs)
            L.caiman_lstm_resident_bwd_split(int(os.environ["CAIMAN_LSTM_BWD_SPLIT"]))
    return _lib


def check(rc: int):
    if rc != 0:
        raise RuntimeError(lib().caiman_last_error().decode())


_DTYPE_TAG = {torch.float64: 0, torch.float32: 1, torch.float16: 2, torch.bfloat16: 3}


def dtype_tag(dtype) -> int:
    try:
        return _DTYPE_TAG[dtype]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {dtype}; expected one of {list(_DTYPE_TAG)}")


def acc_dtype(dtype):
    """at::acc_type<T, true>: double for double, float otherwise."""
    return torch.float64 if dtype == torch.float64 else torch.float32


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check_input(t, name):
    """MYRTLE_CHECK_INPUT (training/lib/csrc/myrtle/utility.hpp:52-72)."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(
            f"{name} must be contiguous but got shape{list(t.shape)} and strides {list(t.stride())} "
            f"such that numel is {t.numel()}")


# ---- optional per-op HIP event timing (used by bench.py for the roofline line) -----------------
class _Timing:
    """`sample_every[name] = k`: only about one in k brackets of that name is timed (pseudo-randomly chosen) and the
    summary scales the totals back up.  An event pair between two dependent kernels costs more than the few
    microseconds it takes to record: the fence it carries writes back / invalidates the L2 that the next LSTM step
    wants warm.  Timing every one of the ~200 LSTM brackets of a step slowed the step by 13 %."""

    def __init__(self):
        self.enabled = False
        self.records = {}
        self.sample_every = {}
        self.seen = {}

    def reset(self):
        self.records = {}
        self.seen = {}

    def take(self, name) -> bool:
        k = self.sample_every.get(name, 1)
        n = self.seen[name] = self.seen.get(name, 0) + 1
        return k <= 1 or ((n * 2654435761) >> 7) % k == 0


timing = _Timing()


class timed:
    """`with timed("name"):` brackets the enclosed launches with HIP events on the CURRENT stream
    (the stream the kernels are launched on) when timing is enabled; otherwise it is free."""

    __slots__ = ("name", "start", "units", "nbytes", "steps")

    def __init__(self, name, units=1, nbytes=0):
        self.name = name
        self.units = units      # kernel launches of the dominant kernel inside the bracket
        self.nbytes = nbytes    # algorithmic bytes moved inside the bracket
        self.steps = units      # dependent timesteps inside the bracket (== units unless one launch covers several)
        self.start = None

    def __enter__(self):
        if timing.enabled and timing.take(self.name):
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if self.start is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            timing.records.setdefault(self.name, []).append((self.start, end, self.units, self.nbytes, self.steps))
        return False


def timing_summary():
    """{name: (brackets, total_ms, kernel launches, algorithmic bytes, timesteps)} scaled from the sampled brackets
    to all of them; synchronises."""
    torch.cuda.synchronize()
    out = {}
    for k, v in timing.records.items():
        scale = timing.seen.get(k, len(v)) / max(len(v), 1)
        out[k] = (timing.seen.get(k, len(v)), scale * sum(r[0].elapsed_time(r[1]) for r in v),
                  scale * sum(r[2] for r in v), scale * sum(r[3] for r in v), scale * sum(r[4] for r in v))
    return out
