"""ctypes binding of libthinkdiff_hip.so (the C ABI in include/thinkdiff_hip.h).

torch is used here only as the owner of device memory and streams: every call passes raw device
pointers + sizes to the C ABI.  There is deliberately NO fallback: if the library is missing the
import of any op raises, so a GPU run can never silently take an eager/CPU path.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libthinkdiff_hip.so")

ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_SILU = 0, 1, 2, 3

_lib = None


class ThinkDiffHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise loudly when the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ThinkDiffHipError(
                f"{LIB_PATH} not found: build it with `make -C thinkdiff-mlre_amd` "
                "(or __graft_entry__.build()); there is no fallback path")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.td_last_error.restype = ctypes.c_char_p
        _declare(_lib)
    return _lib


def _declare(L):
    vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
    sig = {
        "td_abi_version": [],
        "td_linear_bf16": [vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp, i64, vp],
        "td_linear_split_bf16": [vp, i64, vp, vp, vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, vp],
        "td_attention_bf16": [vp, i64, i64, vp, vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, i32, i32, f32, i32, vp],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = ctypes.c_int
    return sig


def check(status):
    if status != 0:
        raise ThinkDiffHipError(f"libthinkdiff_hip error {status}: {lib().td_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda, "thinkdiff_hip ops take device tensors only"
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rows(t):
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.bfloat16, (t.shape, t.stride(), t.dtype)
    return t.stride(0)


def linear(x, w, bias=None, act=ACT_NONE, gate=None, res=None, out=None):
    """out = act(x @ w.T + bias) * gate + res   (2-D bf16 tensors, row stride may exceed width)."""
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and w.is_contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    check(lib().td_linear_bf16(ptr(x), _rows(x), ptr(w), ptr(bias), ptr(out), _rows(out), M, N, K, act,
                               ptr(gate), ptr(res), _rows(res) if res is not None else 0, stream_ptr()))
    return out


def linear_split(x, w, bias, out0, act0, out1, act1, n_split):
    M, K = x.shape
    N = w.shape[0]
    check(lib().td_linear_split_bf16(ptr(x), _rows(x), ptr(w), ptr(bias), ptr(out0), _rows(out0), act0,
                                     ptr(out1), _rows(out1), act1, M, N, K, n_split, stream_ptr()))
    return out0, out1


def attention(q, k, v, out, Hq, Hkv, scale=None, causal=False):
    """q:[B,Sq,>=Hq*128] k,v:[B,Skv,>=Hkv*128] (views into projection outputs) -> out:[B,Sq,>=Hq*128]."""
    assert q.dim() == 3 and k.dim() == 3 and v.dim() == 3 and out.dim() == 3
    B, Sq, _ = q.shape
    Skv = k.shape[1]
    for t in (q, k, v, out):
        assert t.dtype == torch.bfloat16 and t.stride(2) == 1
    assert k.stride() == v.stride()
    if scale is None:
        scale = 128 ** -0.5
    check(lib().td_attention_bf16(ptr(q), q.stride(1), q.stride(0), ptr(k), ptr(v), k.stride(1), k.stride(0),
                                  ptr(out), out.stride(1), out.stride(0), B, Sq, Skv, Hq, Hkv, 128,
                                  float(scale), int(causal), stream_ptr()))
    return out
