"""ctypes binding of libladiff_hip.so (C ABI in include/ladiff_hip.h).

This is the stub a maintainer of the reference would add: the reference is pure
Python, so its "FFI" for the hot path is a `ctypes.CDLL` with `tensor.data_ptr()` arguments and the
current HIP stream.  There is NO CPU fallback: if the library is missing, or a call gets a non-GPU
tensor, it raises.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_size_t, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libladiff_hip.so")

MAX_LATENTS = 8
MAX_FRAMES = 224
COEF_STRIDE = 8
ACT = {"none": 0, "relu": 1, "gelu": 2, "silu": 3, "qgelu": 4, "lrelu": 5}


class LadiffHipError(RuntimeError):
    pass


SPLIT_MODES = ("f16x3", "bf16x3", "split")     # names accepted for "matrix products as hi + lo pairs of 16-bit halves"


def is_split(precision):
    """True for the split arithmetic mode.  WHICH 16-bit type the halves are is a property of the loaded build (`split_mode_name()`:
    fp16 pairs = "f16x3" in the product, bf16 pairs = "bf16x3" in a -DLADIFF_SPLIT_BF16 build); every name selects that one path."""
    if precision == "fp32":
        return False
    if precision in SPLIT_MODES:
        return True
    raise ValueError(f'precision must be "fp32" or one of {SPLIT_MODES}, got {precision!r}')


def split_mode_name():
    return "f16x3" if lib().ladiff_split_format() == 1 else "bf16x3"


_SIGNATURES = {
    "ladiff_version": (c_int, []),
    "ladiff_split_format": (c_int, []),
    "ladiff_error_string": (c_char_p, [c_int]),
    "ladiff_denoiser_num_params": (c_int, []),
    "ladiff_denoiser_param_name": (c_char_p, [c_int]),
    "ladiff_decoder_num_params": (c_int, []),
    "ladiff_decoder_param_name": (c_char_p, [c_int]),
    "ladiff_gemm": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                            c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ladiff_gemm_resident": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                     c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ladiff_gemm_split": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                  c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ladiff_mlp_ln_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int, c_void_p]),
    "ladiff_split_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ladiff_combine_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ladiff_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ladiff_timestep_sinusoid": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "ladiff_decoder_self_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ladiff_self_attention_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ladiff_decoder_cross_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ladiff_denoiser_tables_floats": (c_size_t, [c_int]),
    "ladiff_denoiser_text_cache_floats": (c_size_t, [c_int, c_int, c_int]),
    "ladiff_denoiser_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ladiff_linear_cross_attention_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ladiff_linear_cross_attention": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                              c_void_p, c_size_t, c_void_p]),
    "ladiff_denoiser_time_tables": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ladiff_denoiser_text_cache": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t,
                                           c_void_p]),
    "ladiff_denoiser_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                        c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ladiff_cfg_scheduler_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int,
                                          c_int, c_void_p]),
    "ladiff_advance_step": (c_int, [c_void_p, c_void_p]),
    "ladiff_init_latents": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p]),
    "ladiff_finalize_latents": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ladiff_sampler_create": (c_int, [ctypes.POINTER(c_void_p)]),
    "ladiff_sampler_destroy": (c_int, [c_void_p]),
    "ladiff_sampler_set_loop": (c_int, [c_void_p, c_int]),
    "ladiff_sampler_loop_ms": (c_int, [c_void_p, ctypes.POINTER(c_float)]),
    "ladiff_sampler_set_window_timing": (c_int, [c_void_p, c_int]),
    "ladiff_sampler_window_ms": (c_int, [c_void_p, ctypes.POINTER(c_float), ctypes.POINTER(c_int)]),
    "ladiff_sampler_last_loop": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "ladiff_reverse_status": (c_int, [c_void_p, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "ladiff_reverse_status_offset_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ladiff_debug_set_stage_plan": (c_int, [c_int]),
    "ladiff_debug_set_poll_pause": (c_int, [c_int, c_int]),
    "ladiff_debug_set_stage_delay": (c_int, [c_int, c_int]),
    "ladiff_debug_set_pacing": (c_int, [c_int, c_int]),
    "ladiff_debug_set_loop_thresholds": (c_int, [c_int, c_int]),
    "ladiff_debug_set_graph_epoch_rule": (c_int, [c_int]),
    "ladiff_debug_graph_instantiations": (c_int, []),
    "ladiff_sampler_set_fault": (c_int, [c_void_p, c_int, c_int]),
    "ladiff_sampler_set_noise_generator": (c_int, [c_void_p, c_uint64, ctypes.c_uint32, c_int]),
    "ladiff_noise_fill": (c_int, [c_uint64, ctypes.c_uint32, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ladiff_reverse_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ladiff_diffusion_reverse": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_int, c_int, c_int, c_int,
                                         c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "ladiff_encoder_num_params": (c_int, []),
    "ladiff_encoder_param_name": (c_char_p, [c_int]),
    "ladiff_encoder_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ladiff_vae_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ladiff_clip_num_params": (c_int, []),
    "ladiff_clip_param_name": (c_char_p, [c_int]),
    "ladiff_clip_workspace_bytes": (c_size_t, [c_int, c_int]),
    "ladiff_clip_text_encode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                        c_size_t, c_void_p]),
    "ladiff_clip_workspace_bytes_ragged": (c_size_t, [c_int, c_int]),
    "ladiff_clip_text_encode_ragged": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                               c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ladiff_t2m_movement_num_params": (c_int, []),
    "ladiff_t2m_movement_param_name": (c_char_p, [c_int]),
    "ladiff_t2m_motion_num_params": (c_int, []),
    "ladiff_t2m_motion_param_name": (c_char_p, [c_int]),
    "ladiff_t2m_text_num_params": (c_int, []),
    "ladiff_t2m_text_param_name": (c_char_p, [c_int]),
    "ladiff_t2m_movement_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ladiff_t2m_motion_workspace_bytes": (c_size_t, [c_int, c_int]),
    "ladiff_t2m_text_workspace_bytes": (c_size_t, [c_int, c_int]),
    "ladiff_t2m_movement_encode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                           c_void_p]),
    "ladiff_t2m_motion_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ladiff_t2m_text_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                       c_void_p]),
    "ladiff_feats2joints": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ladiff_decoder_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ladiff_vae_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                  c_void_p, c_size_t, c_void_p]),
    "ladiff_debug_set_stage_waves": (c_int, [c_int]),
    "ladiff_debug_set_handoff": (c_int, [c_int]),
    "ladiff_debug_set_xcd_local": (c_int, [c_int]),
    "ladiff_debug_set_decoder_fusion": (c_int, [c_int]),
    "ladiff_debug_set_mlp_variant": (c_int, [c_int]),
    "ladiff_reverse_plan": (c_int, [c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ladiff_decoder_graph_create": (c_int, [ctypes.POINTER(c_void_p)]),
    "ladiff_decoder_graph_destroy": (c_int, [c_void_p]),
    "ladiff_vae_decode_graphed": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                          c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ladiff_vae_decode_ragged": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                         c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
}

EXPORTS = tuple(_SIGNATURES)
_lib = None
LIB_PATH_BF16 = os.path.join(_HERE, "libladiff_hip_bf16.so")      # the same library built with -DLADIFF_SPLIT_BF16 (build.build_all)


def select_split_format(fmt):
    """Process-wide choice of the split arithmetic's operand pairs, BEFORE the library is first used: "fp16" (the default build: 22
    significant bits, each half saturates at +-65504 - every GEMM operand of the shipped networks is a LayerNorm output, an activation
    of one, a softmax probability or a latent; 241 at most on the random-init weights, tests/test_operand_range.py) or "bf16" (16
    bits, fp32's exponent range: for weights whose activations leave fp16's range).  One format per process: the S-format weight
    copies and every kernel agree on it by construction."""
    global LIB_PATH, _lib
    if fmt not in ("fp16", "bf16"):
        raise ValueError(f'split format must be "fp16" or "bf16", got {fmt!r}')
    path = LIB_PATH_BF16 if fmt == "bf16" else os.path.join(_HERE, "libladiff_hip.so")
    if _lib is not None and path != LIB_PATH:
        raise LadiffHipError("select_split_format() after the library has been used: weight tables of the other format exist")
    LIB_PATH = path


def lib():
    """The loaded library; raises LadiffHipError when it has not been built (python -m ladiff_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LadiffHipError(f"{LIB_PATH} not found: build it with `python -m ladiff_amd.build` "
                                 "(the HIP library is the only implementation of this path; there is no CPU fallback)")
        cdll = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(cdll, name)   # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = cdll
    return _lib


def check(rc):
    if rc != 0:
        raise LadiffHipError(f"libladiff_hip: {lib().ladiff_error_string(rc).decode()} (code {rc})")


def ptr(t, dtype=torch.float32):
    """Device pointer of a contiguous GPU tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise LadiffHipError("libladiff_hip works on GPU tensors only; got a CPU tensor (no CPU fallback exists)")
    if t.dtype != dtype:
        raise LadiffHipError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise LadiffHipError("expected a contiguous tensor")
    return t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def param_names(kind):
    l = lib()
    n = getattr(l, f"ladiff_{kind}_num_params")()
    return [getattr(l, f"ladiff_{kind}_param_name")(i).decode() for i in range(n)]


class WeightTable:
    """Array of device pointers in the order the library expects, built from a state dict.  `generation` is unique per
    table object: the library keys its captured graphs on it, because a rebuilt table can land on the addresses (host
    array and device copies) a freed one used."""
    _next_generation = 1

    def __init__(self, kind, tensors, n_names=None, no_split=(), pad_rows=None):
        self.generation = WeightTable._next_generation
        WeightTable._next_generation += 1
        names = param_names(kind)
        if n_names is not None:      # a prefix of the table (CLIP with fewer than 12 layers); the tail stays NULL
            names = names[:n_names]
        self.no_split = set(no_split)
        # name -> multiple: in the S-format table these tensors are padded with zero rows (matrices) / zeros (vectors) up to a multiple
        # of that many output features (the decoder's final_layer: 263 / 251 features -> 384 / 256, whole 128-column GEMM tiles)
        self.pad_rows = dict(pad_rows or {})
        self.names = names
        self.tensors = []   # keep the fp32 contiguous GPU tensors alive
        for n in names:
            if n not in tensors:
                raise LadiffHipError(f"missing weight {n}")
            t = tensors[n].detach()
            if not t.is_cuda:
                raise LadiffHipError(f"weight {n} is on {t.device}; move the module to the GPU first")
            self.tensors.append(t.to(torch.float32).contiguous())
        self.n_total = len(param_names(kind))
        self.array = (c_void_p * self.n_total)(*[t.data_ptr() for t in self.tensors])
        self.key = tuple((tensors[n].data_ptr(), tensors[n]._version) for n in names)
        self._split = None

    def split_array(self):
        """Second pointer table for the f16x3 path: S-format copies of the weight matrices (built once, on the GPU)."""
        if self._split is None:
            L = lib()
            self.split_tensors = []
            for n, t in zip(self.names, self.tensors):
                mult = self.pad_rows.get(n)
                if mult and t.shape[0] % mult:
                    padded = torch.zeros(((t.shape[0] + mult - 1) // mult * mult,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                    padded[:t.shape[0]] = t
                    t = padded
                if t.dim() == 2 and t.shape[1] % 64 == 0 and n not in self.no_split:
                    s = torch.empty_like(t)
                    check(L.ladiff_split_rows(t.data_ptr(), s.data_ptr(), t.shape[0], t.shape[1], stream_ptr()))
                else:
                    s = t
                self.split_tensors.append(s)
            self._split = (c_void_p * self.n_total)(*[t.data_ptr() for t in self.split_tensors])
        return self._split

    @staticmethod
    def key_of(kind_names, tensors):
        return tuple((tensors[n].data_ptr(), tensors[n]._version) for n in kind_names)


_INT_CACHE = {}


def device_ints(values, device):
    """int32 device tensor of a short host list (lengths, latent counts).  A pageable host-to-device copy blocks the host
    until the stream has drained, which would serialise the host behind a whole sampling pass; the values go through a
    pinned buffer with a non-blocking copy, and recently used lists (batches repeat their lengths) are served from a
    small cache.  The cache is keyed on the current stream as well: the copy is ordered on the stream it was issued on, so
    a hit from another stream would have no dependency on it (and the block would return to the wrong allocator pool)."""
    if torch.device(device).type != "cuda":
        raise LadiffHipError("libladiff_hip works on GPU tensors only; got a CPU tensor (no CPU fallback exists)")
    key = (tuple(int(v) for v in values), str(device), torch.cuda.current_stream(device).cuda_stream)
    t = _INT_CACHE.get(key)
    if t is None:
        if len(_INT_CACHE) >= 64:
            _INT_CACHE.pop(next(iter(_INT_CACHE)))
        host = torch.tensor(key[0], dtype=torch.int32).pin_memory()
        t = host.to(device, non_blocking=True)
        t._ladiff_host = host                       # keep the pinned source alive until the copy has run
        _INT_CACHE[key] = t
    return t


def workspace(nbytes, device):
    return torch.empty((int(nbytes) + 3) // 4, dtype=torch.float32, device=device)
