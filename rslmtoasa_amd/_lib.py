"""ctypes binding of librsrec.so (C ABI declared in include/rsrec.h).

The library is built in-tree (rslmtoasa_amd/csrc/Makefile, or __graft_entry__.build()).  There is no fallback:
if the shared object is missing this module raises, and rsrec_create fails when no gfx950 device is usable.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RSREC_LIB: development override (kernel timing probes build variant libraries); there is still no fallback
LIB_PATH = os.environ.get("RSREC_LIB") or os.path.join(_HERE, "librsrec.so")

ERR_ARG, ERR_DEVICE, ERR_DIVERGED, ERR_EIG = 1, 2, 3, 4

_lib = None

_SIGNATURES = {
    "rsrec_version": (C.c_int, []),
    "rsrec_device_count": (C.c_int, []),
    "rsrec_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "rsrec_destroy": (C.c_int, [C.c_void_p]),
    "rsrec_set_lattice": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "rsrec_set_positions": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rsrec_set_hamiltonian": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6),
    "rsrec_assemble_blocks": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "rsrec_block_lanczos": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "rsrec_block_lanczos_seeded": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "rsrec_block_lanczos_local_axis": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "rsrec_pack_diag": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rsrec_orbital_moments": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]),
    "rsrec_pack_moments": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "rsrec_comm_unique_id": (C.c_int, [C.c_char_p]),
    "rsrec_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p]),
    "rsrec_comm_init_file": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_double]),
    "rsrec_allreduce_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rsrec_comm_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rsrec_comm_destroy": (C.c_int, [C.c_void_p]),
    "rsrec_terminator": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6),
    "rsrec_scalar_density": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 4),
    "rsrec_block_ldos": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5),
    "rsrec_kubo_moments": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double] + [C.c_void_p] * 5),
    "rsrec_apply_operator": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double]),
    "rsrec_zsqr": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "rsrec_chebyshev_green": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "rsrec_block_green": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_int] + [C.c_void_p] * 5),
    "rsrec_chebyshev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "rsrec_chebyshev_seeded": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "rsrec_scalar_lanczos": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rsrec_site_partition": (None, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rsrec_last_error": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "rsrec_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_long]),
    "rsrec_get_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
}


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("librsrec.so not built: run `make -C rslmtoasa_amd/csrc` (or __graft_entry__.build()); "
                               "there is no CPU fallback for the recursion engine")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class RsrecError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rsrec error %d: %s" % (code, msg))
        self.code = code
