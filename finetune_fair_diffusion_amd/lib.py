"""ctypes loader for libfairdiff_hip.so (the C-ABI in include/fairdiff_hip.h).

The product path has NO fallback: if the shared library is missing this module raises,
and every op in ``ops.py`` raises on a non-zero return code.  Prototypes (argtypes) are
derived from the public header so the Python side cannot drift from the C-ABI.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# The 16-bit working dtype of the process ("wd" of SURVEY 8a) is fixed before the package is imported: FD_DTYPE=fp16 (default; the
# reference's mixed_precision fp16) or bf16 (BASELINE configs[4]).  It selects the library -- same sources, same C-ABI, built twice
# (csrc/Makefile) -- and ``ops.F16``, the torch dtype every activation / frozen weight is held in.
WORKING_DTYPE = {"fp16": "fp16", "f16": "fp16", "half": "fp16", "bf16": "bf16", "bfloat16": "bf16"}[os.environ.get("FD_DTYPE", "fp16").lower()]
_DEFAULT_LIB = "libfairdiff_hip.so" if WORKING_DTYPE == "fp16" else "libfairdiff_hip_bf16.so"
LIB_PATH = os.environ.get("FAIRDIFF_LIB") or os.path.join(_HERE, _DEFAULT_LIB)   # override: A/B builds of the same ABI
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "fairdiff_hip.h")


ABI_VERSION = 4      # FD_ABI_VERSION of include/fairdiff_hip.h; load() refuses a library whose fd_version() differs


class _Desc(ctypes.Structure):
    """Descriptor structs of the C-ABI start with ``struct_size`` = sizeof as the caller sees it (the library refuses a mismatch): filled in at
    construction, also for every element of a ctypes array of descriptors (``(T * n)()`` constructs its elements without calling __init__, so
    arrays go through ``T.array(n)``)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = ctypes.sizeof(type(self))

    @classmethod
    def array(cls, n):
        arr = (cls * n)()
        for i in range(n):
            arr[i].struct_size = ctypes.sizeof(cls)
        return arr


class GemmDesc(_Desc):
    """Mirror of ``fd_gemm_desc`` (include/fairdiff_hip.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("A", ctypes.c_void_p), ("lda", ctypes.c_int64),
        ("B", ctypes.c_void_p), ("ldb", ctypes.c_int64),
        ("A2", ctypes.c_void_p), ("lda2", ctypes.c_int64),
        ("B2", ctypes.c_void_p), ("ldb2", ctypes.c_int64),
        ("C", ctypes.c_void_p), ("ldc", ctypes.c_int64),
        ("bias", ctypes.c_void_p),
        ("rowbias", ctypes.c_void_p), ("ld_rowbias", ctypes.c_int64), ("rows_per_batch", ctypes.c_int32),
        ("residual", ctypes.c_void_p), ("ldr", ctypes.c_int64),
        ("alpha", ctypes.c_float),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("K2", ctypes.c_int32),
        ("act", ctypes.c_int32), ("out_dtype", ctypes.c_int32),
        ("batch", ctypes.c_int32), ("sA", ctypes.c_int64), ("sB", ctypes.c_int64), ("sC", ctypes.c_int64), ("sR", ctypes.c_int64),
        ("conv", ctypes.c_int32), ("conv_mode", ctypes.c_int32), ("Bn", ctypes.c_int32), ("H", ctypes.c_int32),
        ("W", ctypes.c_int32), ("Cin", ctypes.c_int32), ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_int64),
        ("gn_stats", ctypes.c_void_p),
        ("colscale", ctypes.c_float), ("colscale_cols", ctypes.c_int32),
    ]


class WgradDesc(_Desc):
    """Mirror of ``fd_wgrad_desc`` (include/fairdiff_hip.h)."""
    _fields_ = [("struct_size", ctypes.c_int32), ("X", ctypes.c_void_p), ("ldx", ctypes.c_int64), ("T", ctypes.c_void_p), ("ldt", ctypes.c_int64),
                ("G", ctypes.c_void_p), ("g_stride_n", ctypes.c_int64), ("g_stride_r", ctypes.c_int64),
                ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("R", ctypes.c_int32), ("scale", ctypes.c_float)]


class LoraRefreshDesc(_Desc):
    """Mirror of ``fd_lora_refresh_desc`` (include/fairdiff_hip.h)."""
    _fields_ = [("struct_size", ctypes.c_int32), ("down", ctypes.c_void_p), ("up", ctypes.c_void_p), ("d16", ctypes.c_void_p), ("ld_d16", ctypes.c_int64),
                ("dT16", ctypes.c_void_p), ("ld_dT16", ctypes.c_int64), ("u16", ctypes.c_void_p), ("ld_u16", ctypes.c_int64),
                ("uT16", ctypes.c_void_p), ("ld_uT16", ctypes.c_int64), ("r", ctypes.c_int32), ("rp", ctypes.c_int32),
                ("K", ctypes.c_int32), ("N", ctypes.c_int32), ("scale", ctypes.c_float)]


class CrossBlockDesc(_Desc):
    """Mirror of ``fd_cross_block_desc`` (include/fairdiff_hip.h)."""
    _fields_ = [("struct_size", ctypes.c_int32), ("x", ctypes.c_void_p), ("ln2_gamma", ctypes.c_void_p), ("ln2_beta", ctypes.c_void_p), ("ln2_eps", ctypes.c_float),
                ("wq", ctypes.c_void_p), ("k", ctypes.c_void_p), ("vt", ctypes.c_void_p), ("L", ctypes.c_int32), ("Lp", ctypes.c_int32),
                ("wo", ctypes.c_void_p), ("bo", ctypes.c_void_p), ("ln3_gamma", ctypes.c_void_p), ("ln3_beta", ctypes.c_void_p), ("ln3_eps", ctypes.c_float),
                ("y", ctypes.c_void_p), ("yn", ctypes.c_void_p), ("yn_stats", ctypes.c_void_p),
                ("M", ctypes.c_int32), ("C", ctypes.c_int32), ("heads", ctypes.c_int32), ("rows_per_sample", ctypes.c_int32), ("kv_div", ctypes.c_int32),
                ("scale", ctypes.c_float),
                ("lora_q_down", ctypes.c_void_p), ("ld_q_down", ctypes.c_int64), ("lora_q_up", ctypes.c_void_p), ("ld_q_up", ctypes.c_int64),
                ("lora_o_down", ctypes.c_void_p), ("ld_o_down", ctypes.c_int64), ("lora_o_up", ctypes.c_void_p), ("ld_o_up", ctypes.c_int64),
                ("lora_rp", ctypes.c_int32),
                ("n2_out", ctypes.c_void_p), ("ln2_stats", ctypes.c_void_p), ("q_out", ctypes.c_void_p), ("tq_out", ctypes.c_void_p), ("o_out", ctypes.c_void_p),
                ("lse_out", ctypes.c_void_p), ("to_out", ctypes.c_void_p), ("q_prescaled", ctypes.c_int32)]


_CTYPE = {"int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float}


def parse_header(path=HEADER_PATH):
    """Return {name: (restype, [argtypes])} for every ``int fd_*(...)`` / ``const char* fd_*`` prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(fd_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "fd_gemm_desc" in a:
                    argtypes.append(ctypes.POINTER(GemmDesc))
                elif "fd_cross_block_desc" in a:
                    argtypes.append(ctypes.POINTER(CrossBlockDesc))
                elif "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    toks = a.replace("const ", "").split()
                    argtypes.append(_CTYPE[toks[0]])
        protos[name] = (ctypes.c_char_p if ret != "int" else ctypes.c_int, argtypes)
    return protos


def load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is required (no fallback path). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C finetune_fair_diffusion_amd/csrc`.")
    # torch bundles its own HIP runtime: import it FIRST so libfairdiff_hip.so binds to the runtime torch's
    # streams and device memory live in (loading the system libamdhip64 first leaves two runtimes in one process
    # and every launch then fails with "no ROCm-capable device").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (ret, argtypes) in parse_header().items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = ret
        fn.argtypes = argtypes
    if lib.fd_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} reports ABI revision {lib.fd_version()} but this package binds revision {ABI_VERSION} of include/fairdiff_hip.h: rebuild it "
                           "(`make -C finetune_fair_diffusion_amd/csrc`)")
    built = lib.fd_working_dtype().decode()
    if built != WORKING_DTYPE:
        raise RuntimeError(f"{LIB_PATH} was built for {built} but this process runs with FD_DTYPE={WORKING_DTYPE}")
    info = lib.fd_build_info().decode()
    if "packed_fp32=off" not in info and os.environ.get("FD_ALLOW_PACKED_FP32") is None:
        raise RuntimeError(f"{LIB_PATH} was built with packed-fp32 VALU code ({info}): on gfx950 such kernels returned wrong lanes whenever kernels of several "
                           "streams shared a SIMD (DESIGN.md, round 4).  Rebuild with csrc/Makefile's flags (FD_ALLOW_PACKED_FP32=1 loads it anyway, for measurement).")
    return lib


def torch_working_dtype():
    import torch
    return torch.float16 if WORKING_DTYPE == "fp16" else torch.bfloat16


_lib = None


def get():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib
