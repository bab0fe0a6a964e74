"""ctypes binding of the C-ABI library ``csrc/libmsmd_hip.so`` (declared in ``include/msmd_hip.h``).

The header is the single source of truth: prototypes are parsed from it, so the
Python argtypes can never drift from the C declarations.  There is NO fallback:
if the library is missing or a symbol cannot be resolved, import of the product
modules fails loudly (the product path never routes through the CPU oracle).
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "msmd_hip.h")
# MSMD_LIB selects another build of the SAME C ABI (developers: the experimental library, make -C csrc EXP=1)
LIB_PATH = os.environ.get("MSMD_LIB") or os.path.join(_HERE, "csrc", "libmsmd_hip.so")

_CTYPE = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float,
    "msmd_stream_t": ctypes.c_void_p, "unsigned": ctypes.c_uint, "double": ctypes.c_double,
}


def parse_header(path: str = HEADER):
    """Return {name: [argtype, ...]} for every ``int msmd_*(...)`` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    RESTYPE.clear()
    for m in re.finditer(r"\b(int|long)\s+(msmd_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(2), m.group(3).strip()
        RESTYPE[name] = _CTYPE[m.group(1)]
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    types.append(ctypes.c_void_p)
                elif a.startswith("double") or a.startswith("const double"):
                    types.append(ctypes.c_double)
                else:
                    base = a.replace("const ", "").split(" ")[0]
                    types.append(_CTYPE[base])
        protos[name] = types
    return protos


RESTYPE = {}


class MsmdLibraryError(RuntimeError):
    pass


_lib = None
PROTOS = None


def load(path: str = LIB_PATH):
    """Load the library once and attach argtypes/restype to every declared entry point."""
    global _lib, PROTOS
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise MsmdLibraryError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ubisoft-laforge-msmd_amd/csrc` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the product path.")
    try:
        import torch  # noqa: F401  (load torch's HIP runtime first so both sides share ONE libamdhip64)
    except Exception:  # pragma: no cover - the symbol-export test may run without torch
        pass
    lib = ctypes.CDLL(path)
    PROTOS = parse_header()
    for name, argtypes in PROTOS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            if os.environ.get("MSMD_LIB") and os.environ.get("MSMD_LIB_ALLOW_MISSING") == "1":
                continue     # developers' same-box A/B against an OLDER build (tools/ab_lib*.sh): calls of newer entries then fail
            raise MsmdLibraryError(f"{path} does not export {name} declared in {HEADER}") from e
        fn.argtypes = argtypes
        fn.restype = RESTYPE[name]
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        raise MsmdLibraryError(f"{what} failed with hipError {code}")
