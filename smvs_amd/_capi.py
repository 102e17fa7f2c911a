"""ctypes loader for the C ABI (include/smvs_hip.h -> csrc/libsmvs_hip.so).

There is no CPU fallback: if the HIP library is missing or no GPU is visible
the product path raises.  (The CPU oracle lives under oracle/ and is test
infrastructure; this package never imports it.)
"""
import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libsmvs_hip.so")
HEADER = os.path.join(HERE, "..", "include", "smvs_hip.h")

_lib = None


class SmvsError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("smvs_hip status %d: %s" % (status, text))
        self.status = status


class LoopParams(C.Structure):
    _fields_ = [("regularization", C.c_double),
                ("light_surf_regularization", C.c_double),
                ("full_optimization", C.c_int),
                ("max_newton_steps", C.c_int),
                ("cg_max_iterations", C.c_int),
                ("cg_q_tolerance", C.c_double),
                ("active_threshold", C.c_double),
                ("full_opt_threshold", C.c_double),
                ("use_lighting", C.c_int),
                ("lighting", C.c_double * 16),
                ("reset_active", C.c_int)]


class LoopStats(C.Structure):
    _fields_ = [("newton_steps", C.c_int),
                ("linear_iterations", C.c_int),
                ("active_patch_steps", C.c_longlong),
                ("final_active_nodes", C.c_int),
                ("nan_break", C.c_int)]


K_NAMES = ["patch", "assemble", "cg_spmv", "cg_update", "cg_init",
           "reactivate", "misc", "cg_resident"]


def declared_symbols():
    """Entry points declared in include/smvs_hip.h."""
    with open(HEADER) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smvs_[a-z_0-9]+)\s*\(", text)))


def load():
    """Load libsmvs_hip.so (raises if it was not built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmvsError(-2, "HIP extension %s is missing: run "
                        "`python -c 'import __graft_entry__ as g; g.build()'`"
                        % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.smvs_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(status):
    if status != 0:
        raise SmvsError(status, load().smvs_last_error().decode())
