"""
ctypes binding of libhydra_hip.so (the C ABI declared in include/hydra_hip.h).

The prototypes are read from the header itself, so the Python binding, the header
and the library cannot drift apart: ``declared_symbols()`` lists what the header
declares and ``load()`` binds every one of them (and raises if any is missing).

There is NO CPU fallback: if the library has not been built, ``load()`` raises
``RuntimeError`` -- build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C hydra_amd/csrc``.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "hydra_hip.h")
#: measurement / test hooks, declared apart from the product boundary
DIAG_HEADER = os.path.join(os.path.dirname(_HERE), "include", "hydra_hip_diag.h")
LIBRARY = os.environ.get("HYDRA_HIP_LIBRARY", os.path.join(_HERE, "libhydra_hip.so"))

_CTYPE = {
    "int": ctypes.c_int,
    "long long": ctypes.c_longlong,
    "size_t": ctypes.c_size_t,
    "double": ctypes.c_double,
}

_lib = None
_protos = None


class HydraError(RuntimeError):
    """A libhydra_hip call returned a non-zero status."""


def _parse_header(path=HEADER):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(hydra_[a-z0-9_A-Z]+)\s*\(([^)]*)\)\s*;",
                         text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    base, _, nm = a.rpartition("*")
                    base = base.replace("const", "").strip()
                    if base == "int" and nm.strip().endswith("_host"):
                        argtypes.append(ctypes.POINTER(ctypes.c_int))
                    else:
                        argtypes.append(ctypes.c_void_p)
                    argnames.append(nm.strip())
                else:
                    base, _, nm = a.rpartition(" ")
                    argtypes.append(_CTYPE[base.replace("const", "").strip()])
                    argnames.append(nm.strip())
        if "*" in ret:
            restype = ctypes.c_char_p
        else:
            restype = _CTYPE[ret.replace("const", "").strip()]
        protos[name] = (restype, argtypes, argnames)
    return protos


def prototypes():
    global _protos
    if _protos is None:
        _protos = _parse_header()
        _protos.update(_parse_header(DIAG_HEADER))
    return _protos


def declared_symbols(product_only=False):
    if product_only:
        return sorted(_parse_header().keys())
    return sorted(prototypes().keys())


def _load_hip_runtime():
    """libhydra_hip.so is linked WITHOUT its own HIP runtime (-no-hip-rt): a process must hold
    exactly one libamdhip64, and PyTorch-ROCm ships its own.  Import torch first and make its
    runtime's symbols global so ours resolve against it."""
    import torch
    rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        ctypes.CDLL(rt, mode=ctypes.RTLD_GLOBAL)
    else:                                   # a torch build that uses the system ROCm
        ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL)


def load():
    """Load the library and bind every declared entry point (cached)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBRARY):
        raise RuntimeError(
            f"{LIBRARY} not found: the HIP library has not been built. hydra_amd has no CPU "
            "fallback; run `make -C hydra_amd/csrc` (or __graft_entry__.build()) first.")
    _load_hip_runtime()
    lib = ctypes.CDLL(LIBRARY)
    for name, (restype, argtypes, _) in prototypes().items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise RuntimeError(f"{LIBRARY} does not export {name} declared in {HEADER}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error():
    msg = load().hydra_last_error()
    return msg.decode() if msg else ""


def call(name, *args):
    """Call an int-returning entry point; raise HydraError on a non-zero status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise HydraError(f"{name} failed ({rc}): {last_error()}")
    return rc


def query(name, *args):
    """Call a size_t / int query that cannot fail."""
    return getattr(load(), name)(*args)
