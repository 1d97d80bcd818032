"""ctypes binding of libpfotgn.so (the C ABI declared in include/pfotgn.h).

There is no CPU fallback: if the library is missing, loading raises; if no HIP device is present,
every compute entry raises ``RuntimeError`` before anything is launched.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PFOTGN_LIB", os.path.join(_HERE, "lib", "libpfotgn.so"))   # override: kernel A/B builds

c_i32p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())
_VP = C.c_void_p

MAX_LAYERS = 4


class TgnConfig(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_edges_p1", C.c_int32), ("D", C.c_int32), ("Ef", C.c_int32),
                ("n_layers", C.c_int32), ("n_heads", C.c_int32), ("use_memory", C.c_int32), ("max_roots", C.c_int32),
                ("max_neighbors", C.c_int32), ("max_batch", C.c_int32)]


class TgnLayerLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("wq", "wk", "wv", "b_in", "wo", "bo", "w1", "b1", "w2", "b2")]


class TgnLayout(C.Structure):
    _fields_ = [("time_w", C.c_int64), ("time_b", C.c_int64), ("gru_w_ih", C.c_int64), ("gru_w_hh", C.c_int64),
                ("gru_b_ih", C.c_int64), ("gru_b_hh", C.c_int64), ("layer", TgnLayerLayout * MAX_LAYERS),
                ("total", C.c_int64)]


class TgnState(C.Structure):
    _fields_ = [(n, _VP) for n in ("indptr", "adj_nbr", "adj_eidx", "adj_ts", "node_feat", "edge_feat", "memory",
                                   "last_update", "msg_table", "msg_time", "has_msg", "params", "pcache")] + [("pcache_valid", C.c_int32)]


class TgnBatch(C.Structure):
    _fields_ = [("roots", _VP), ("root_ts", _VP), ("R", C.c_int32), ("K", C.c_int32), ("uniform", C.c_int32),
                ("draws", C.POINTER(_VP)), ("seed", C.c_uint64), ("offset", C.c_uint64), ("dropout_p", C.c_float),
                ("training", C.c_int32), ("extra_nodes", _VP), ("n_extra", C.c_int32), ("offset_dev", _VP),
                ("deterministic", C.c_int32), ("prepared", C.c_int32),
                ("upd_src", _VP), ("upd_dst", _VP), ("upd_ts", _VP), ("upd_eidx", _VP), ("upd_B", C.c_int32),
                ("dropout_keep", C.POINTER(_VP)), ("mid_event", _VP), ("defer_join", C.c_int32), ("seg_in_forward", C.c_int32), ("mid_event_late", C.c_int32)]


class TgnDebug(C.Structure):
    _fields_ = [(n, _VP) for n in ("n_touched", "touched_ids", "h0_table", "slot", "n_core", "l1_ctx", "l1_dh1", "l1_dW1ovT",
                                   "gru_dgi", "gru_msg_rows")] + [("Cp", C.c_int32)]


# name -> (restype, argtypes); every symbol of include/pfotgn.h
PROTOTYPES = {
    "pfo_abi_version": (C.c_int, []),
    "pfo_last_error": (C.c_char_p, []),
    "pfo_tnbr_sample": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, _VP, _VP, C.c_int64, C.c_int32, C.c_int32, _VP,
                                  C.c_uint64, C.c_uint64, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pfo_neg_draw": (C.c_int, [_VP, C.c_int32, _VP, _VP, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_uint64,
                               C.c_uint64, _VP, _VP]),
    "pfo_neg_draw_dev": (C.c_int, [_VP, C.c_int32, _VP, _VP, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_uint64,
                                   C.c_uint64, _VP, _VP, _VP]),
    "pfo_mv_select": (C.c_int, [_VP, C.c_int32, C.c_int32, C.c_int32, _VP, _VP, C.c_int32, _VP, _VP, C.c_int32,
                                C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32, _VP, _VP, _VP,
                                _VP, _VP]),
    "pfo_time_encode": (C.c_int, [_VP, C.c_int64, _VP, _VP, C.c_int32, _VP, _VP]),
    "pfo_segment_sum": (C.c_int, [_VP, C.c_int32, _VP, C.c_int32, _VP, _VP, _VP, C.c_int64, _VP, C.c_int32, C.c_int32, _VP, _VP,
                                  _VP]),
    "pfo_attn_dropout_mask": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_float, _VP, _VP]),
    "pfo_gemm_f32": (C.c_int, [_VP, C.c_int64, C.c_int32, _VP, C.c_int64, C.c_int32, _VP, C.c_int64, _VP, C.c_int32,
                               C.c_int32, C.c_int32, C.c_int32, _VP, C.c_int64, _VP]),
    "pfo_gemm_bf16x3_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "pfo_gemm_bf16x3": (C.c_int, [_VP, C.c_int64, _VP, C.c_int64, C.c_int32, _VP, C.c_int64, _VP, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, _VP, C.c_int64, _VP]),
    "pfo_roots_assemble": (C.c_int, [_VP, _VP, _VP, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32,
                                     _VP, _VP, _VP]),
    "pfo_bpr_loss": (C.c_int, [_VP, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_float, _VP,
                               _VP, _VP, _VP]),
    "pfo_bpr_loss_fused": (C.c_int, [_VP, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_float, _VP,
                                     _VP, _VP, _VP, _VP]),
    "pfo_bpr_loss_parts": (C.c_int, [_VP, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_float, _VP,
                                     _VP, _VP]),
    "pfo_rank_metrics": (C.c_int, [_VP, C.c_int64, C.c_int32, C.c_int32, _VP, _VP, _VP, _VP]),
    "pfo_adam_step": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                                _VP]),
    "pfo_csr_build_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64]),
    "pfo_csr_build": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, C.c_int64, _VP, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    "pfo_csr_append": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, _VP, _VP, _VP, _VP, C.c_int64, _VP, _VP, _VP, _VP, _VP]),
    "pfo_adam_step_ranges": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int32), C.c_float, C.c_float, C.c_float, C.c_float, _VP]),
    "pfo_adam_step_ranges_dev": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_int32), _VP, C.c_float, C.c_float, C.c_float, C.c_float, _VP]),
    "pfo_tgn_param_layout": (C.c_int, [C.POINTER(TgnConfig), C.POINTER(TgnLayout)]),
    "pfo_tgn_workspace_bytes": (C.c_int64, [C.POINTER(TgnConfig)]),
    "pfo_tgn_adam_side": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int32), C.c_float, C.c_float, C.c_float, C.c_float]),
    "pfo_tgn_adam_side_bucket": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int32), C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32]),
    "pfo_tgn_side_stream": (_VP, []),
    "pfo_tgn_join": (C.c_int, [_VP]),
    "pfo_tgn_pcache_bytes": (C.c_int64, [C.POINTER(TgnConfig)]),
    "pfo_tgn_refresh": (C.c_int, [C.POINTER(TgnConfig), C.POINTER(TgnState), _VP]),
    "pfo_tgn_forward": (C.c_int, [C.POINTER(TgnConfig), C.POINTER(TgnState), C.POINTER(TgnBatch), _VP, _VP, _VP]),
    "pfo_tgn_prepare": (C.c_int, [C.POINTER(TgnConfig), C.POINTER(TgnState), C.POINTER(TgnBatch), _VP, _VP]),
    "pfo_tgn_backward": (C.c_int, [C.POINTER(TgnConfig), C.POINTER(TgnState), C.POINTER(TgnBatch), _VP, _VP, _VP, _VP]),
    "pfo_tgn_backward_ev": (C.c_int, [C.POINTER(TgnConfig), C.POINTER(TgnState), C.POINTER(TgnBatch), _VP, _VP, _VP, C.c_int32,
                                      _VP, _VP, C.c_int64, _VP, _VP]),
    "pfo_tgn_grad_split": (C.c_int, [C.POINTER(TgnConfig), C.POINTER(C.c_int64)]),
    "pfo_tgn_update_state": (C.c_int, [C.POINTER(TgnConfig), C.POINTER(TgnState), _VP, _VP, _VP, _VP, C.c_int32, _VP,
                                       _VP]),
    "pfo_tgn_debug_views": (C.c_int, [C.POINTER(TgnConfig), _VP, C.POINTER(TgnDebug)]),
    "pfo_prof_enable": (C.c_int, [C.c_int32]),
    "pfo_marks_enable": (C.c_int, [C.c_int32]),
    "pfo_mark": (C.c_int, [C.c_char_p, _VP]),
    "pfo_marks_dump": (C.c_int64, [C.c_char_p, C.c_int64]),
    "pfo_prof_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pfo_shader_clock": (C.c_int, [C.POINTER(C.c_double), C.c_int32]),
}

PROF_KINDS = ["gemm_nt", "gemm_nn", "gemm_tn", "gemm_devm", "attn_fwd", "attn_bwd", "sampler", "gemm_bx", "gemm_tn_bx", "gemm_bx_skinny",
              "attn_bwd_runs", "gru_fused", "gemm_multi", "segsum", "tn_reduce", "gru_gates_bwd", "gemm_tn_bx8"]


_MARK_NAMES = {}


def mark(name):
    """A milestone on the current stream (no-op unless ``marks_enable(True)``); the C side keeps the pointer, so the bytes
    object of every name is kept alive here."""
    b = _MARK_NAMES.get(name)
    if b is None:
        b = _MARK_NAMES[name] = C.c_char_p(name.encode())
    load().pfo_mark(b, stream_ptr())


def marks_enable(on):
    call("pfo_marks_enable", 1 if on else 0)


def marks_dump():
    buf = C.create_string_buffer(1 << 16)
    load().pfo_marks_dump(buf, len(buf))
    return buf.value.decode()


_PROF_ON = [False]


def prof_enable(on):
    _PROF_ON[0] = bool(on)
    call("pfo_prof_enable", 1 if on else 0)


def prof_is_on():
    return _PROF_ON[0]


CLOCK_KERNELS = ["attn_fwd", "attn_bwd_runs", "gemm_tn_bx"]


def shader_clock(reset=False):
    """GHz the first wavefront of the three stamped kernels ran at since the last reset (0.0: not launched)."""
    out = (C.c_double * len(CLOCK_KERNELS))()
    call("pfo_shader_clock", out, 1 if reset else 0)
    return {k: out[i] for i, k in enumerate(CLOCK_KERNELS)}


def prof_collect():
    n = len(PROF_KINDS)
    ms, work, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_int64 * n)()
    call("pfo_prof_collect", ms, work, cnt)
    return {k: dict(ms=ms[i], work=work[i], count=cnt[i]) for i, k in enumerate(PROF_KINDS)}

_lib = None


class PfoError(RuntimeError):
    pass


def load():
    """Loads the shared library (raises if it has not been built: run ``python -m pfotgnrec_amd.build``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PfoError("libpfotgn.so not found at %s - build it with `python -m pfotgnrec_amd.build` "
                           "(there is no CPU fallback)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise PfoError("%s failed (%d): %s" % (name, rc, lib.pfo_last_error().decode()))


_GPU_SEEN = [False]


def require_gpu(device=None):
    # (torch.cuda.is_available() reads the environment on every call: ~3 us, a dozen times per step - asked once)
    if not _GPU_SEEN[0]:
        import torch
        if not torch.cuda.is_available():
            raise PfoError("a HIP device (MI355X) is required: the hot path has no CPU implementation")
        _GPU_SEEN[0] = True
    if device is not None:
        t = getattr(device, "type", None)
        if t is None:
            import torch
            t = torch.device(device).type
        if t != "cuda":
            raise PfoError("tensors must live on a HIP device, got %s" % device)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


# torch.cuda.current_stream() builds a Stream object through several Python layers (~8 us); the step asks for the current
# stream's handle a dozen times.  The raw handle comes from one C call; Stream OBJECTS (for Event.record / Stream.wait_event)
# are kept per handle - torch's streams live in a pool and are never destroyed, an external stream is its handle.
_RAW = [None, None]
_STREAM_OBJECTS = {}


def _raw_fns():
    import torch
    torch.cuda.current_stream()                         # (initialises the runtime the first time)
    _RAW[0], _RAW[1] = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
    return _RAW[0]


def stream_ptr():
    get = _RAW[0] or _raw_fns()
    return get(_RAW[1]())


def current_stream():
    """The current stream of the current device as a (cached) torch Stream object."""
    get = _RAW[0] or _raw_fns()
    dev = _RAW[1]()
    key = (dev, get(dev))                               # (the default stream's handle is 0 on every device)
    s = _STREAM_OBJECTS.get(key)
    if s is None:
        import torch
        s = _STREAM_OBJECTS[key] = torch.cuda.current_stream()
    return s


class on_stream:
    """``with torch.cuda.stream(s):`` for a stream of the CURRENT device, without the context manager's Python layers
    (two C calls each way); another device's stream takes torch's own context."""
    __slots__ = ("s", "prev", "ctx")

    def __init__(self, stream):
        self.s, self.prev, self.ctx = stream, None, None

    def __enter__(self):
        import torch
        s = self.s
        if (_RAW[1] or (_raw_fns() and _RAW[1]))() != s.device_index:
            self.ctx = torch.cuda.stream(s)
            return self.ctx.__enter__()
        self.prev = torch._C._cuda_getCurrentStream(s.device_index)
        torch._C._cuda_setStream(stream_id=s.stream_id, device_index=s.device_index, device_type=s.device_type)
        return None

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        import torch
        p = self.prev
        torch._C._cuda_setStream(stream_id=p[0], device_index=p[1], device_type=p[2])
        return False
