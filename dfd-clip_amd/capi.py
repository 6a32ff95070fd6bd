"""ctypes binding of libdfdclip_hip.so (C ABI declared in include/dfdclip.h).

There is no CPU or PyTorch fallback: if the library is missing or a call fails, this module
raises.  Tensors are passed as raw device pointers; every call enqueues on
`torch.cuda.current_stream()` and returns without synchronising.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_void_p

import torch

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libdfdclip_hip.so")

F32, BF16 = 0, 1
EPI_BIAS, EPI_BIAS_QUICKGELU, EPI_BIAS_RESIDUAL, EPI_PATCH_EMBED, EPI_QKV_EXPORT, EPI_RESIDUAL_POS = range(6)
ABI_VERSION = 15

_DTYPE = {torch.float32: F32, torch.bfloat16: BF16}
FP8 = 2           # ABI code of OCP e4m3; stored in uint8 / torch.float8_e4m3fn tensors
FP8_MAX = 448.0   # largest finite e4m3 value
FP8_MIN_ROWS = 1024  # dfd_gemm_fp8 serves M >= 1024 (include/dfdclip.h)


class DfdError(RuntimeError):
    pass


GEMM_STREAM_OUT = 1
GEMM_SPARE_IF_FREE = 2  # spare_cus is honoured only where it costs this shape no extra round of tiles


class GemmExtra(Structure):
    _fields_ = [("pos", c_void_p), ("cls", c_void_p), ("k_export", c_void_p), ("v_export", c_void_p),
                ("tokens", c_int32), ("frames_per_clip", c_int32), ("residual", c_void_p), ("qkv_first", c_int32),
                ("drop_rng", c_void_p), ("drop_site", c_uint32), ("drop_p", c_float), ("flags", c_uint32)]


class DropoutDesc(Structure):
    _fields_ = [("rng_state", c_void_p), ("site", c_uint32), ("p", c_float)]


class Dropout:
    """One dropout layer of one step: `rng` = device int64[2] {seed, step}, `site` = which layer, `p`."""
    __slots__ = ("rng", "site", "p")

    def __init__(self, rng, site, p):
        assert rng.is_cuda and rng.dtype == torch.int64 and rng.numel() == 2 and 0.0 <= p < 1.0
        self.rng, self.site, self.p = rng, int(site), float(p)

    def desc(self):
        return DropoutDesc(self.rng.data_ptr(), self.site, self.p)


def _drop(d):
    return ctypes.byref(d.desc()) if d is not None and d.p > 0 else None


# name -> (restype, argtypes); mirrors include/dfdclip.h one to one
SIGNATURES = {
    "dfd_last_error": (c_char_p, []),
    "dfd_abi_version": (c_int, []),
    "dfd_gemm_last_path": (c_int, []),
    "dfd_gemm_set_variant": (c_int, [c_int]),
    "dfd_dropout": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, POINTER(DropoutDesc), c_void_p]),
    "dfd_device_check": (c_int, []),
    "dfd_layernorm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_float, c_float, c_void_p]),
    "dfd_layernorm2": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_float,
                               c_float, c_void_p]),
    "dfd_add_layernorm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64,
                                  c_int, c_int64, c_int, c_float, c_float, c_void_p]),
    "dfd_patchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dfd_preprocess_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float),
                                  c_void_p, c_int, c_int, c_int, c_void_p]),
    "dfd_preprocess_geometry": (c_int, [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "dfd_gemm": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_int,
                         POINTER(GemmExtra), c_int64, c_int, c_int, c_void_p]),
    "dfd_gemm_fp8": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_float, c_int,
                             POINTER(GemmExtra), c_int64, c_int, c_int, c_void_p]),
    "dfd_gemm_at_b_workspace": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "dfd_gemm_at_b": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "dfd_adapter_norm_gelu": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                      c_void_p]),
    "dfd_attention_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "dfd_linear_rows": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "dfd_linear_rows_t_workspace": (c_size_t, [c_int, c_int, c_int]),
    "dfd_linear_rows_t": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p,
                                  c_void_p]),
    "dfd_decoder_attn_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dfd_decoder_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dfd_decoder_attn_modes_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                           c_int, c_int, c_void_p]),
    "dfd_decoder_attn_modes_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                           c_int, c_int, c_void_p]),
    "dfd_decoder_attn_bwd_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dfd_decoder_attn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                     c_int, c_void_p]),
    "dfd_adapter_norm_gelu_bwd_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dfd_adapter_norm_gelu_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "dfd_linear_rows_bwd_weight": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dfd_transpose_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dfd_sgd_blocks": (c_int64, [c_int64, c_int, c_int, c_int]),
    "dfd_sgd_step": (c_int, [c_void_p, c_int, c_int64, c_float, c_float, c_float, c_int, c_void_p]),
    "dfd_layernorm_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_float, c_int, c_void_p]),
    "dfd_quickgelu": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, POINTER(DropoutDesc), c_void_p]),
    "dfd_head_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                             c_void_p]),
    "dfd_head_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                             c_int, c_float, POINTER(DropoutDesc), c_void_p]),
}

_lib = None


def load_library(path=None):
    """dlopen the kernel library and bind every declared symbol.  Raises DfdError when the
    file is absent (run `python -c "import __graft_entry__ as g; g.build()"` or
    `python dfd-clip_amd/build.py`), a symbol is missing or the ABI version differs."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise DfdError(f"{path} not found: the HIP kernel library is not built and there is no fallback path")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise DfdError(f"{path} does not export {name}")
        fn.restype = res
        fn.argtypes = args
    if lib.dfd_abi_version() != ABI_VERSION:
        raise DfdError(f"{path}: ABI version {lib.dfd_abi_version()} != {ABI_VERSION}")
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise DfdError(f"{what} failed ({rc}): {load_library().dfd_last_error().decode()}")


def _stream():
    # torch's current stream on the current device, as a raw hipStream_t: ~0.3 us, where
    # torch.cuda.current_stream().cuda_stream costs ~8 us (700+ launches per train step)
    return c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DfdError("HIP kernels need device tensors; got a CPU tensor (there is no CPU path)")


def _out_dtype(t):
    """ABI dtype code of an output tensor: one-byte tensors (uint8 / float8_e4m3fn storage) are e4m3."""
    return FP8 if t.element_size() == 1 else _DTYPE[t.dtype]


def layernorm(x, gamma, beta, out, eps=1e-5, out_inv_scale=0.0):
    """out[rows, cols] = LayerNorm(x[rows, cols]); x f32; out f32 (may alias x), bf16, or e4m3 bytes of the result
    times `out_inv_scale`."""
    _dev(x, gamma, beta, out)
    assert x.dtype == torch.float32 and x.dim() == 2 and out.shape == x.shape
    assert x.stride(1) == 1 and out.stride(1) == 1
    _check(load_library().dfd_layernorm(_ptr(x), x.stride(0), _ptr(gamma), _ptr(beta), _ptr(out), out.stride(0),
                                        _out_dtype(out), x.shape[0], x.shape[1], eps, float(out_inv_scale), _stream()), "dfd_layernorm")
    return out


def layernorm2(x, gamma_a, beta_a, gamma_b, beta_b, out, eps=1e-5, out_inv_scale=0.0):
    """x <- LayerNorm_a(x) in place (f32), out = LayerNorm_b(x): one pass over the rows."""
    _dev(x, gamma_a, beta_a, gamma_b, beta_b, out)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and out.stride(1) == 1 and out.shape == x.shape
    _check(load_library().dfd_layernorm2(_ptr(x), x.stride(0), _ptr(gamma_a), _ptr(beta_a), _ptr(gamma_b), _ptr(beta_b), _ptr(out),
                                         out.stride(0), _out_dtype(out), x.shape[0], x.shape[1], eps, float(out_inv_scale), _stream()),
           "dfd_layernorm2")
    return out


def add_layernorm(x, delta, gamma, beta, out, eps=1e-5, delta2=None, store_x=True, out_inv_scale=0.0):
    """v = (x + delta) [+ delta2] (x f32, deltas f32 / bf16); out = LayerNorm(v) in f32, bf16 or e4m3 (times
    `out_inv_scale`); x <- v when `store_x`."""
    _dev(x, delta, gamma, beta, out, delta2)
    assert x.dtype == torch.float32 and x.dim() == 2 and out.shape == x.shape and delta.shape == x.shape
    assert x.stride(1) == 1 and out.stride(1) == 1 and delta.stride(1) == 1
    assert delta2 is None or (delta2.dtype == delta.dtype and delta2.shape == x.shape and delta2.stride() == delta.stride())
    _check(load_library().dfd_add_layernorm(_ptr(x), x.stride(0), _ptr(delta), _ptr(delta2), delta.stride(0), _DTYPE[delta.dtype],
                                            int(bool(store_x)), _ptr(gamma), _ptr(beta), _ptr(out), out.stride(0), _out_dtype(out),
                                            x.shape[0], x.shape[1], eps, float(out_inv_scale), _stream()), "dfd_add_layernorm")
    return out


def patchify(frames, out, res, patch):
    _dev(frames, out)
    assert frames.dtype == torch.float32 and frames.is_contiguous() and out.is_contiguous()
    n = frames.shape[0]
    _check(load_library().dfd_patchify(_ptr(frames), _ptr(out), _DTYPE[out.dtype], n, res, patch, out.shape[1], _stream()),
           "dfd_patchify")
    return out


def preprocess_geometry(in_h, in_w, res):
    """(resized_h, resized_w, crop_top, crop_left) of `preprocess_u8`."""
    v = [c_int() for _ in range(4)]
    _check(load_library().dfd_preprocess_geometry(in_h, in_w, res, *[ctypes.byref(i) for i in v]), "dfd_preprocess_geometry")
    return tuple(i.value for i in v)


def preprocess_u8(frames, out, res, patch, mean, std, antialias=True, patch_rows=True):
    """uint8 frames [n,3,H,W] -> normalised patch rows [n*P, kpad] (`patch_rows`) or frames
    [n,3,res,res]; the device-side `Detector._transform` (reference `src/models.py:756-768`)."""
    _dev(frames, out)
    assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[1] == 3
    assert frames.is_contiguous() and out.is_contiguous()
    n, _, h, w = frames.shape
    m3 = (c_float * 3)(*[float(v) for v in mean])
    s3 = (c_float * 3)(*[float(v) for v in std])
    if patch_rows:
        assert out.dim() == 2 and out.shape[0] == n * (res // patch) ** 2
        kpad = out.shape[1]
    else:
        assert tuple(out.shape) == (n, 3, res, res)
        kpad = 0
    _check(load_library().dfd_preprocess_u8(_ptr(frames), n, h, w, res, patch, int(bool(antialias)), m3, s3, _ptr(out),
                                            _DTYPE[out.dtype], 1 if patch_rows else 0, kpad, _stream()), "dfd_preprocess_u8")
    return out


_profile = {"epilogue": None, "events": [], "base": None}


def profile_gemm(epilogue=None):
    """Time every `gemm` launch with this epilogue from now on: a HIP event pair is recorded on
    the launch stream around the kernel (bench.py's roofline leg).  None switches it off."""
    _profile["epilogue"] = epilogue
    _profile["events"] = []
    _profile["base"] = None
    if epilogue is not None:
        _profile["base"] = torch.cuda.Event(enable_timing=True)
        _profile["base"].record()


def profile_gemm_collect():
    """(start_ms, end_ms, FLOPs) of each timed launch since `profile_gemm`, on one time base (launches
    may sit on different streams and overlap); call after a device sync."""
    base = _profile["base"]
    spans = [(base.elapsed_time(a), base.elapsed_time(b), flops) for a, b, flops in _profile["events"]]
    profile_gemm(None)
    return spans


def gemm(a, w, c, bias=None, epilogue=EPI_BIAS, m=None, pos=None, cls=None, k_export=None, v_export=None, tokens=0,
         frames_per_clip=0, residual=None, qkv_first=0, drop=None, stream_out=False, spare_cus=0, tile_blocks=0, spare_if_free=False):
    """c = epilogue(a[M,K] @ w[N,K]^T).  `m` limits the rows used (buffers may be over-allocated)."""
    _dev(a, w, c, bias, pos, cls, k_export, v_export, residual)
    assert a.dtype == w.dtype and a.stride(1) == 1 and w.stride(1) == 1 and c.stride(1) == 1
    M = a.shape[0] if m is None else m
    N, K = w.shape
    assert a.shape[1] == K
    assert residual is None or (residual.dtype == c.dtype and residual.stride(0) == c.stride(0))
    assert drop is None or epilogue == EPI_RESIDUAL_POS
    extra = GemmExtra(_ptr(pos).value, _ptr(cls).value, _ptr(k_export).value, _ptr(v_export).value, tokens, frames_per_clip,
                      _ptr(residual).value, qkv_first, drop.rng.data_ptr() if drop is not None and drop.p > 0 else None,
                      drop.site if drop is not None else 0, drop.p if drop is not None else 0.0, (GEMM_STREAM_OUT if stream_out else 0) | (GEMM_SPARE_IF_FREE if spare_if_free else 0) | ((int(spare_cus) & 0xff) << 8) | ((int(tile_blocks) & 0xf) << 16))
    timed = _profile["epilogue"] == epilogue
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _check(load_library().dfd_gemm(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _DTYPE[a.dtype], _ptr(c), c.stride(0),
                                   _DTYPE[c.dtype], _ptr(bias), epilogue, ctypes.byref(extra), M, N, K, _stream()), "dfd_gemm")
    if timed:
        e1.record()
        _profile["events"].append((e0, e1, 2.0 * M * N * K))
    return c


def gemm_fp8(a, w, c, col_scale, bias=None, epilogue=EPI_BIAS, m=None, out_inv_scale=0.0, pos=None, k_export=None, v_export=None,
             tokens=0, frames_per_clip=0, qkv_first=0, stream_out=False, spare_cus=0, spare_if_free=False):
    """c = epilogue((a[M,K] @ w[N,K]^T) * col_scale + bias) on e4m3 operands (uint8 / float8_e4m3fn storage);
    c bf16, or e4m3 bytes of result * out_inv_scale."""
    _dev(a, w, c, col_scale, bias, pos, k_export, v_export)
    assert a.element_size() == 1 and w.element_size() == 1 and a.stride(1) == 1 and w.stride(1) == 1 and c.stride(1) == 1
    assert col_scale.dtype == torch.float32 and col_scale.is_contiguous()
    M = a.shape[0] if m is None else m
    N, K = w.shape
    assert a.shape[1] == K and col_scale.numel() == N
    c_dtype = FP8 if c.element_size() == 1 else _DTYPE[c.dtype]
    extra = GemmExtra(_ptr(pos).value, None, _ptr(k_export).value, _ptr(v_export).value, tokens, frames_per_clip, None, qkv_first,
                      None, 0, 0.0, (GEMM_STREAM_OUT if stream_out else 0) | (GEMM_SPARE_IF_FREE if spare_if_free else 0) | ((int(spare_cus) & 0xff) << 8))
    timed = _profile["epilogue"] == epilogue
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _check(load_library().dfd_gemm_fp8(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(c), c.stride(0), c_dtype, _ptr(col_scale),
                                       _ptr(bias), float(out_inv_scale), epilogue, ctypes.byref(extra), M, N, K, _stream()),
           "dfd_gemm_fp8")
    if timed:
        e1.record()
        _profile["events"].append((e0, e1, 2.0 * M * N * K))
    return c


def gemm_set_variant(variant):
    """0 = every GEMM kernel eligible, tiles dealt statically (default); 1 = skip the ping-pong kernel (tests / A-B runs);
    3 = the ping-pong kernel hands its tiles out dynamically (per-XCD counters).  Per thread.  Returns the previous value."""
    return load_library().dfd_gemm_set_variant(int(variant))


def gemm_last_path():
    """256 when this thread's last gemm ran on the tuned 256x256 bf16 kernel, 128 for the general kernel."""
    return load_library().dfd_gemm_last_path()


def gemm_at_b_workspace_bytes(R, Ma, Nb, dtype):
    return load_library().dfd_gemm_at_b_workspace(R, Ma, Nb, _DTYPE[dtype])


def gemm_at_b(a, b, c, workspace):
    """c[Ma, Nb] (f32) = a[R, Ma]^T @ b[R, Nb]."""
    _dev(a, b, c, workspace)
    assert a.dtype == b.dtype and a.stride(1) == 1 and b.stride(1) == 1 and c.is_contiguous() and c.dtype == torch.float32
    R, Ma = a.shape
    Nb = b.shape[1]
    assert b.shape[0] == R and tuple(c.shape) == (Ma, Nb)
    assert workspace.numel() * workspace.element_size() >= gemm_at_b_workspace_bytes(R, Ma, Nb, a.dtype)
    _check(load_library().dfd_gemm_at_b(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _DTYPE[a.dtype], _ptr(c), R, Ma, Nb,
                                        _ptr(workspace), _stream()), "dfd_gemm_at_b")
    return c


def adapter_norm_gelu(a, y, weight, bias, frames, patches, x, joint, eps=1e-5):
    """y = GELU(LayerNorm(a)) on [frames, patches, x]; mode `joint`: 0 per row, 1 statistics over (patches, x),
    2 = LayerNorm(GELU(a)) per row.  a may be f32 with a bf16 y."""
    _dev(a, y, weight, bias)
    assert (a.dtype == y.dtype or a.dtype == torch.float32) and a.is_contiguous() and y.is_contiguous()
    assert weight.is_contiguous() and bias.is_contiguous()
    _check(load_library().dfd_adapter_norm_gelu(_ptr(a), _DTYPE[a.dtype], _ptr(y), _DTYPE[y.dtype], _ptr(weight), _ptr(bias), frames,
                                                patches, x, int(joint), eps, _stream()), "dfd_adapter_norm_gelu")
    return y


def adapter_norm_gelu_bwd_workspace_bytes(frames, patches, x, joint):
    return load_library().dfd_adapter_norm_gelu_bwd_workspace(frames, patches, x, int(joint))


def adapter_norm_gelu_bwd(a, dy, da, weight, bias, dweight, dbias, workspace, frames, patches, x, joint, eps=1e-5):
    _dev(a, dy, da, weight, bias, dweight, dbias, workspace)
    assert dy.dtype == da.dtype and (a.dtype == dy.dtype or a.dtype == torch.float32)
    assert a.is_contiguous() and dy.is_contiguous() and da.is_contiguous()
    _check(load_library().dfd_adapter_norm_gelu_bwd(_ptr(a), _DTYPE[a.dtype], _ptr(dy), _ptr(da), _DTYPE[dy.dtype], _ptr(weight), _ptr(bias),
                                                    _ptr(dweight), _ptr(dbias), _ptr(workspace), frames, patches, x, int(joint),
                                                    eps, _stream()), "dfd_adapter_norm_gelu_bwd")
    return da


def attention_fwd(qkv, out, n_frames, tokens, heads, head_dim=64):
    _dev(qkv, out)
    assert qkv.dtype == out.dtype and qkv.stride(1) == 1 and out.stride(1) == 1
    _check(load_library().dfd_attention_fwd(_ptr(qkv), qkv.stride(0), _ptr(out), out.stride(0), _DTYPE[qkv.dtype], n_frames,
                                            tokens, heads, head_dim, head_dim ** -0.5, _stream()), "dfd_attention_fwd")
    return out


def linear_rows(x, w, bias, y, epilogue=EPI_BIAS):
    _dev(x, w, bias, y)
    assert x.dtype == torch.float32 and w.dtype == torch.float32 and y.dtype == torch.float32
    assert x.stride(1) == 1 and y.stride(1) == 1 and w.is_contiguous()
    B, K = x.shape
    N = w.shape[0]
    _check(load_library().dfd_linear_rows(_ptr(x), x.stride(0), _ptr(w), _ptr(bias), _ptr(y), y.stride(0), epilogue, B, N, K,
                                          _stream()), "dfd_linear_rows")
    return y


def linear_rows_t_workspace_bytes(B, N, K):
    return load_library().dfd_linear_rows_t_workspace(B, N, K)


def linear_rows_t(x, wt, bias, y, workspace, epilogue=EPI_BIAS, residual=None):
    """y[B,N] = epilogue(x[B,K] @ wt[K,N] + bias): forward with wt = weight^T, data gradient with wt = weight.
    EPI_BIAS_RESIDUAL adds `residual` [B,N] (None: y itself, in place)."""
    _dev(x, wt, bias, y, workspace, residual)
    assert residual is None or (residual.dtype == torch.float32 and residual.stride(1) == 1 and residual.shape == y.shape)
    assert x.dtype == torch.float32 and wt.dtype == torch.float32 and y.dtype == torch.float32
    assert x.stride(1) == 1 and y.stride(1) == 1 and wt.is_contiguous()
    B, K = x.shape
    N = wt.shape[1]
    assert wt.shape[0] == K and workspace.numel() * workspace.element_size() >= linear_rows_t_workspace_bytes(B, N, K)
    _check(load_library().dfd_linear_rows_t(_ptr(x), x.stride(0), _ptr(wt), _ptr(bias), _ptr(residual),
                                            residual.stride(0) if residual is not None else 0, _ptr(y), y.stride(0), epilogue, B, N, K,
                                            _ptr(workspace), _stream()), "dfd_linear_rows_t")
    return y


def decoder_attn_workspace_bytes(B, heads, d, splits):
    return load_library().dfd_decoder_attn_workspace(B, heads, d, splits)


class KvLayoutDesc(ctypes.Structure):
    """dfd_kv_layout_t"""
    _fields_ = [("row_stride", c_int64), ("frame_stride", c_int64), ("pos", c_void_p)]


def _kv_layout(x, other, pos, B, T, patches, D):
    """Keys / values as the dense export [B*T*patches, D] (contiguous, any leading shape) or as a strided view
    [B*T, patches, D] of a larger activation (e.g. qkv[:, 1:, D:2*D] of the encoder's [frames, tokens, 3*D]);
    `pos` [T, D] f32: temporal positional embedding added on the fly.  Returns (layout or None, keep-alive)."""
    if other is not None:
        assert other.dtype == x.dtype and other.shape == x.shape and other.stride() == x.stride()
    if x.is_contiguous() and pos is None:
        assert x.numel() == B * T * patches * D
        return None, None
    if x.is_contiguous():
        rs, fs = D, patches * D
    else:
        assert x.dim() == 3 and x.shape == (B * T, patches, D) and x.stride(2) == 1, "strided K/V must be [B*T, patches, D] views"
        fs, rs = x.stride(0), x.stride(1)
    if pos is not None:
        assert pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape == (T, D) and pos.device == x.device
    lay = KvLayoutDesc(rs, fs, pos.data_ptr() if pos is not None else None)
    return ctypes.byref(lay), lay


def decoder_attn_fwd(q, k, v, frame_mask, mix, stats, workspace, splits, B, T, patches, heads, d=64, mix_softmax=None,
                     ext_weights=None, pos=None):
    _dev(q, k, v, frame_mask, mix, stats, workspace, mix_softmax, ext_weights)
    assert q.dtype == torch.float32 and q.is_contiguous()
    assert frame_mask.dtype == torch.uint8 and frame_mask.is_contiguous()
    assert ext_weights is None or (ext_weights.is_contiguous() and ext_weights.numel() == B * heads * T * patches)
    lay, _keep = _kv_layout(k, v, pos, B, T, patches, heads * d)
    _check(load_library().dfd_decoder_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _DTYPE[k.dtype], lay, _ptr(frame_mask), _ptr(ext_weights),
                                               _ptr(mix), _ptr(mix_softmax), _ptr(stats), _ptr(workspace), splits, B, T, patches,
                                               heads, d, _stream()),
           "dfd_decoder_attn_fwd")
    return mix


ATTN_MODE_BITS = {"frame": 1, "temporal": 2}


def decoder_attn_modes_fwd(q, k, frame_mask, modes, scores, weights, B, T, patches, heads, d=64, pos=None):
    """attn_mode softmax branch: scores [B,heads,S] and grouped-softmax weights [B,heads,S]."""
    _dev(q, k, frame_mask, scores, weights)
    assert q.is_contiguous() and scores.is_contiguous() and weights.is_contiguous()
    assert scores.numel() == B * heads * T * patches == weights.numel()
    lay, _keep = _kv_layout(k, None, pos, B, T, patches, heads * d)
    _check(load_library().dfd_decoder_attn_modes_fwd(_ptr(q), _ptr(k), _DTYPE[k.dtype], lay, _ptr(frame_mask), modes, _ptr(scores),
                                                     _ptr(weights), B, T, patches, heads, d, _stream()),
           "dfd_decoder_attn_modes_fwd")
    return weights


def decoder_attn_modes_bwd(scores, v, dmix, modes, dwv_ws, dscores, B, T, patches, heads, d=64, pos=None):
    _dev(scores, v, dmix, dwv_ws, dscores)
    assert scores.is_contiguous() and dmix.is_contiguous()
    assert dwv_ws.numel() >= B * heads * T * patches and dscores.numel() == B * heads * T * patches
    lay, _keep = _kv_layout(v, None, pos, B, T, patches, heads * d)
    _check(load_library().dfd_decoder_attn_modes_bwd(_ptr(scores), _ptr(v), _DTYPE[v.dtype], lay, _ptr(dmix), modes, _ptr(dwv_ws),
                                                     _ptr(dscores), B, T, patches, heads, d, _stream()),
           "dfd_decoder_attn_modes_bwd")
    return dscores


def head_fwd(x, gamma, beta, proj, feat, raw, logits, eps=1e-5, drop=None):
    _dev(x, gamma, beta, proj, feat, raw, logits)
    B, D = x.shape
    assert proj.is_contiguous() and proj.shape[0] == D and feat.is_contiguous()
    _check(load_library().dfd_head_fwd(_ptr(x), x.stride(0), _ptr(gamma), _ptr(beta), _ptr(proj), _ptr(feat), _ptr(raw),
                                       _ptr(logits), B, D, proj.shape[1], eps, _drop(drop), _stream()), "dfd_head_fwd")
    return logits


def dropout(x, out, drop):
    """out = dropout(x) elementwise (element index = flat position); also its own backward.  In place allowed."""
    _dev(x, out)
    assert x.is_contiguous() and out.is_contiguous() and x.numel() == out.numel() and drop is not None
    _check(load_library().dfd_dropout(_ptr(x), _DTYPE[x.dtype], _ptr(out), _DTYPE[out.dtype], x.numel(),
                                      ctypes.byref(drop.desc()), _stream()), "dfd_dropout")
    return out


# ---- decoder backward ------------------------------------------------------------------------

def decoder_attn_bwd_workspace_bytes(B, T, heads, d=64):
    return load_library().dfd_decoder_attn_bwd_workspace(B, T, heads, d)


def decoder_attn_bwd(q, k, v, frame_mask, dmix, mix_softmax, stats, dq, dpos, workspace, B, T, patches, heads, d=64, dk=None,
                     dv=None, ext_weights=None, ext_dscores=None, pos=None):
    _dev(q, k, v, frame_mask, dmix, mix_softmax, stats, dq, dpos, workspace, dk, dv, ext_weights, ext_dscores)
    assert q.is_contiguous() and dmix.is_contiguous()
    assert mix_softmax is None or mix_softmax.is_contiguous()
    lay, _keep = _kv_layout(k, v, pos, B, T, patches, heads * d)
    _check(load_library().dfd_decoder_attn_bwd(_ptr(q), _ptr(k), _ptr(v), _DTYPE[k.dtype], lay, _ptr(frame_mask), _ptr(dmix),
                                               _ptr(mix_softmax), _ptr(stats), _ptr(ext_weights), _ptr(ext_dscores), _ptr(dq),
                                               _ptr(dpos), _ptr(dk), _ptr(dv),
                                               _DTYPE[dk.dtype] if dk is not None else F32, _ptr(workspace), B, T, patches,
                                               heads, d, _stream()), "dfd_decoder_attn_bwd")
    return dq


def linear_rows_bwd_weight(dy, x, dw, db=None):
    _dev(dy, x, dw, db)
    assert dy.stride(1) == 1 and x.stride(1) == 1 and dw.is_contiguous()
    B, N = dy.shape
    K = x.shape[1]
    assert tuple(dw.shape) == (N, K)
    _check(load_library().dfd_linear_rows_bwd_weight(_ptr(dy), dy.stride(0), _ptr(x), x.stride(0), _ptr(dw), _ptr(db), B, N, K,
                                                     _stream()), "dfd_linear_rows_bwd_weight")
    return dw


def transpose(src, dst):
    _dev(src, dst)
    assert src.is_contiguous() and dst.is_contiguous() and src.dtype == torch.float32
    R, C = src.shape
    _check(load_library().dfd_transpose_f32(_ptr(src), _ptr(dst), R, C, _stream()), "dfd_transpose_f32")
    return dst


def sgd_blocks(numel, rows=0, cols=0, mirrored=False):
    return int(load_library().dfd_sgd_blocks(int(numel), int(rows), int(cols), 1 if mirrored else 0))


def sgd_step(table, n, total_blocks, lr, momentum, weight_decay, first_step):
    """`table`: device int64 tensor [n, 7] laid out as dfd_sgd_param (p, g, buf, mirror, numel, rows | cols << 32, first_block)."""
    _dev(table)
    assert table.dtype == torch.int64 and table.is_contiguous() and table.shape == (n, 7)
    _check(load_library().dfd_sgd_step(_ptr(table), int(n), int(total_blocks), float(lr), float(momentum), float(weight_decay),
                                       1 if first_step else 0, _stream()), "dfd_sgd_step")


def layernorm_bwd(x, gamma, dy, dx, dgamma, dbeta, xhat_ws, accumulate_dx=False, eps=1e-5):
    _dev(x, gamma, dy, dx, dgamma, dbeta, xhat_ws)
    rows, cols = x.shape
    _check(load_library().dfd_layernorm_bwd(_ptr(x), x.stride(0), _ptr(gamma), _ptr(dy), dy.stride(0), _ptr(dx), dx.stride(0),
                                            _ptr(dgamma), _ptr(dbeta), _ptr(xhat_ws), rows, cols, eps, int(accumulate_dx),
                                            _stream()), "dfd_layernorm_bwd")
    return dx


def quickgelu(u, out, du=None, drop=None):
    _dev(u, out, du)
    assert u.is_contiguous() and out.is_contiguous() and (du is None or du.is_contiguous())
    _check(load_library().dfd_quickgelu(_ptr(u), _ptr(du), _ptr(out), u.numel(), _drop(drop), _stream()), "dfd_quickgelu")
    return out


def head_bwd(raw, dlogits, proj, feat, dfeat_ext, dz, dfeat, dproj):
    _dev(raw, dlogits, proj, feat, dfeat_ext, dz, dfeat, dproj)
    B, od = raw.shape
    D = proj.shape[0]
    assert dlogits.is_contiguous() and proj.is_contiguous() and feat.is_contiguous()
    _check(load_library().dfd_head_bwd(_ptr(raw), _ptr(dlogits), _ptr(proj), _ptr(feat), _ptr(dfeat_ext), _ptr(dz), _ptr(dfeat),
                                       _ptr(dproj), B, D, od, _stream()), "dfd_head_bwd")
    return dfeat
