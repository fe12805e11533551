"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADERS = {
    "spadot_ot.h": "libspadot_ot.so",
    "spadot_model.h": "libspadot_model.so",
}

LIBOT_SYMBOLS = [  # the 15 extern "C" names of the reference's libot.so (ot_func.cpp:938-1373)
    "dummy_float", "dummy_double", "primal_float", "primal_double", "dual_float", "dual_double",
    "compute_duality_gap_float", "compute_duality_gap_double", "update_k_float", "update_k_double",
    "update_R_float", "update_R_double", "step1_process_double", "update_process_double",
]


def declared_functions(header_path):
    src = open(header_path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
    names = re.findall(r"\b([A-Za-z_]\w*)\s*\([^;{]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined",)))


@pytest.mark.parametrize("header", sorted(HEADERS))
def test_library_exports_every_declared_symbol(header):
    hp = os.path.join(ROOT, "include", header)
    if not os.path.exists(hp):
        pytest.skip(f"{header} not present yet")
    so = os.path.join(ROOT, "spadot_amd", "csrc", HEADERS[header])
    assert os.path.exists(so), f"{so} missing: run python -m spadot_amd.csrc.build"
    import torch  # noqa: F401  (binds the library to torch's HIP runtime, as the product does)
    lib = ctypes.CDLL(so)
    names = declared_functions(hp)
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in {header} but not exported: {missing}"


def test_libot_drop_in_names_present():
    import torch  # noqa: F401
    lib = ctypes.CDLL(os.path.join(ROOT, "spadot_amd", "csrc", "libspadot_ot.so"))
    for n in LIBOT_SYMBOLS:
        assert hasattr(lib, n), n
    # the 15th libot.so export is a C++-mangled helper of the empty ot_ctx stub (ot_ctx.hpp:5-13),
    # not part of the C ABI the ctypes shim binds


def test_reference_shim_mirror_has_same_wrapper_names():
    from spadot_amd.utils.OT_loss import ot_func
    for n in ("dummy_c", "primal_c", "dual_c", "compute_duality_gap_c", "update_K_c", "update_R_c",
              "update_a_b_c", "step1_process_c", "update_process_c"):
        assert callable(getattr(ot_func, n)), n


def test_missing_library_is_a_loud_error(tmp_path, monkeypatch):
    from spadot_amd import _lib
    monkeypatch.setattr(_lib, "_CSRC", str(tmp_path))
    monkeypatch.setattr(_lib, "_CACHE", {})
    with pytest.raises(_lib.NativeLibraryMissing):
        _lib._load("libspadot_ot.so")


# ---------------------------------------------------------------------------------------------------------------
# header prototypes against the ctypes declarations (the round-3 k_sgemm_small fault was a call whose
# Python side and C side disagreed about the parameter list; this is the check that catches that without a GPU)

def prototypes(header_path):
    """{name: [type class per parameter]} for every function prototype of a header; classes: 'ptr', 'int', 'll', 'double',
    'float'."""
    src = open(header_path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
    out = {}
    for name, params in re.findall(r"\b([A-Za-z_]\w*)\s*\(([^;{]*)\)\s*;", src):
        params = params.strip()
        kinds = []
        if params and params != "void":
            for prm in params.split(","):
                prm = " ".join(prm.split())
                if "*" in prm or "[" in prm:
                    kinds.append("ptr")
                elif re.search(r"\blong long\b", prm) or "int64_t" in prm or "size_t" in prm:
                    kinds.append("ll")
                elif re.search(r"\bdouble\b", prm):
                    kinds.append("double")
                elif re.search(r"\bfloat\b", prm):
                    kinds.append("float")
                elif re.search(r"\b(int|unsigned)\b", prm):
                    kinds.append("int")
                else:
                    kinds.append("?" + prm)
        out[name] = kinds
    return out


def ctype_class(t):
    if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents") or getattr(t, "_type_", None) == "P":
        return "ptr"
    if isinstance(t, type) and issubclass(t, ctypes._Pointer):
        return "ptr"
    return {ctypes.c_int: "int", ctypes.c_uint: "int", ctypes.c_longlong: "ll", ctypes.c_ulonglong: "ll", ctypes.c_size_t: "ll",
            ctypes.c_double: "double", ctypes.c_float: "float"}.get(t, "?" + repr(t))


@pytest.mark.parametrize("header,loader", [("spadot_model.h", "model_lib"), ("spadot_ot.h", "ot_lib")])
def test_every_spadot_entry_has_argtypes_that_match_its_prototype(header, loader):
    from spadot_amd import _lib
    lib = getattr(_lib, loader)()
    protos = {n: k for n, k in prototypes(os.path.join(ROOT, "include", header)).items() if n.startswith("spadot_")}
    assert len(protos) >= 10
    bad = []
    for name, kinds in sorted(protos.items()):
        fn = getattr(lib, name)
        if name.endswith("_version"):
            continue
        if fn.argtypes is None:
            bad.append(f"{name}: declared in {header}, no argtypes in _lib.py")
            continue
        got = [ctype_class(t) for t in fn.argtypes]
        if got != kinds:
            bad.append(f"{name}: header {kinds} != argtypes {got}")
        elif not isinstance(fn, _lib._Checked):
            bad.append(f"{name}: not sealed (a call with surplus arguments would go through)")
    assert not bad, "\n".join(bad)


def test_a_call_with_the_wrong_argument_count_is_refused_before_it_reaches_the_library():
    from spadot_amd import _lib
    lib = _lib.model_lib()
    n = len(lib.spadot_sgemm_small.argtypes)
    with pytest.raises(TypeError, match="spadot_sgemm_small takes"):
        lib.spadot_sgemm_small(*([0] * (n - 4)))        # the pre-batch call form against the batched entry
    with pytest.raises(TypeError):
        lib.spadot_sgemm_small(*([0] * (n + 1)))
