"""ctypes binding of libmuscle_hip.so — the C ABI declared in include/muscle_hip.h.

The product path has no fallback: if the library is missing or a kernel reports an error this
module raises.  `lib()` loads lazily so that CPU-only hosts can still import the package
(state_dict handling, host logic, symbol-export tests).
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MUSCLE_HIP_LIB names another build of the SAME ABI (lab A/B runs, tools/dbg/lib_ab.sh): nothing is ever copied over the
# in-tree library, and the override announces itself on stderr so a lab build cannot pass for the product silently.
LIB_PATH = os.environ.get("MUSCLE_HIP_LIB") or os.path.join(_HERE, "libmuscle_hip.so")

HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "muscle_hip.h")

LONG_RETURNS = set()      # entry points declared `long mx_...` (byte counts)
_C = {"p": ctypes.c_void_p, "i": ctypes.c_int, "l": ctypes.c_long, "f": ctypes.c_float, "d": ctypes.c_double}


def parse_header(path: str = HEADER_PATH) -> Dict[str, str]:
    """{entry point: argument codes} from include/muscle_hip.h — the header is the single source of
    truth for the ABI; p = pointer, i = int, l = long, f = float, d = double."""
    import re
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    sigs: Dict[str, str] = {}
    for m in re.finditer(r"\b(int|long)\s+(mx_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        if ret == "long":
            LONG_RETURNS.add(name)
        codes = ""
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    codes += "p"
                else:
                    codes += {"int": "i", "long": "l", "float": "f", "double": "d"}[a.split()[0]]
        sigs[name] = codes
    return sigs


_lib: Optional[ctypes.CDLL] = None


class MuscleHipError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MuscleHipError(
                f"{LIB_PATH} not found: build it with `python -m muscle_amd._build` "
                "(or __graft_entry__.build()); there is no fallback path")
        if os.environ.get("MUSCLE_HIP_LIB"):
            import sys
            print(f"[muscle_amd] MUSCLE_HIP_LIB: loading {LIB_PATH} instead of the in-tree library", file=sys.stderr)
        L = ctypes.CDLL(LIB_PATH)
        L.mx_last_error.restype = ctypes.c_char_p
        L.mx_version.restype = ctypes.c_int
        from ._build import abi_hash
        try:
            L.mx_abi_hash.restype = ctypes.c_int
            built = L.mx_abi_hash()
        except AttributeError:
            built = -1
        if built != abi_hash(HEADER_PATH):
            raise MuscleHipError(
                f"{LIB_PATH} was built against a different include/muscle_hip.h (ABI hash {built} != {abi_hash(HEADER_PATH)}): "
                "rebuild it with `python -m muscle_amd._build` (or __graft_entry__.build())")
        for name, codes in parse_header().items():
            fn = getattr(L, name)
            fn.restype = ctypes.c_long if name in LONG_RETURNS else ctypes.c_int
            fn.argtypes = [_C[c] for c in codes]      # incl. the pure-host mx_*_parts() helpers
        _lib = L
    return _lib


_fns: Dict[str, object] = {}


def call(name: str, *args):
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        raise MuscleHipError(f"{name} failed (rc={rc}): {lib().mx_last_error().decode()}")


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MuscleHipError("muscle_amd kernels need CUDA (ROCm) tensors; there is no CPU path")
    if not t.is_contiguous():
        raise MuscleHipError("muscle_amd kernels need contiguous tensors")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> int:
    """The current HIP stream of the current device as an integer handle (follows torch.cuda.stream(...) contexts and graph
    capture).  torch.cuda.current_stream().cuda_stream builds a Stream object per call: 2.6 us, ~1700 times per B7 step."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
