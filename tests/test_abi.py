"""CPU tests of the drop-in boundary: the C-ABI library loads and exports exactly what
include/ife_hip.h declares; no compute call is made here (there is no GPU)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ife_hip.h")


def header_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ife_[a-z0-9_]+)\s*\(", txt)))


def test_header_is_plain_c():
    """extern "C", plain pointers and sizes: compiles as C with gcc."""
    src = '#include "ife_hip.h"\nint main(void){ return IFE_NUM_FEATURES == 8 ? 0 : 1; }\n'
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-x", "c", "-", "-o", "/dev/null"], input=src.encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()


def test_library_exports_every_declared_symbol(ife):
    lib = ife.load_library()
    declared = header_functions()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib, name), "libife_hip.so does not export %s" % name
    assert sorted(ife.EXPORTS) == declared
    assert lib.ife_abi_version() == 1


def test_binding_enums_match_header(ife):
    txt = open(HEADER).read()
    def val(name):
        m = re.search(r"\b%s\s*=\s*(-?\d+)" % name, txt)
        assert m, name
        return int(m.group(1))
    assert (ife.OK, ife.E_ARG, ife.E_SIZE, ife.E_HIP, ife.E_NOMEM, ife.E_STATE) == tuple(
        val(n) for n in ("IFE_OK", "IFE_E_ARG", "IFE_E_SIZE", "IFE_E_HIP", "IFE_E_NOMEM",
                         "IFE_E_STATE"))
    assert (ife.F32, ife.I16, ife.U8, ife.U16) == tuple(
        val(n) for n in ("IFE_F32", "IFE_I16", "IFE_U8", "IFE_U16"))
    assert (ife.OPT_TRIG_MODE, ife.OPT_DSCALE_MODE, ife.OPT_PROFILE, ife.OPT_ZCHUNK,
            ife.OPT_IIR_BLOCK, ife.OPT_IIR_CKPT, ife.OPT_IIR_FMA, ife.OPT_FUSED_DIVIDE, ife.OPT_CONST_LINES) == tuple(val(n) for n in (
                "IFE_OPT_TRIG_MODE", "IFE_OPT_DSCALE_MODE", "IFE_OPT_PROFILE", "IFE_OPT_ZCHUNK",
                "IFE_OPT_IIR_BLOCK", "IFE_OPT_IIR_CKPT", "IFE_OPT_IIR_FMA", "IFE_OPT_FUSED_DIVIDE", "IFE_OPT_CONST_LINES"))
    names = re.search(r"#define IFE_FEATURE_NAMES\s*\\\s*\{(.*?)\}", txt, re.S).group(1)
    assert tuple(re.findall(r'"(\w+)"', names)) == ife.FEATURE_NAMES


def test_no_cpu_fallback(ife):
    """Without a usable gfx950 device the product refuses to run (it never routes to the
    oracle or any CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the refusal path cannot be shown here")
    with pytest.raises(ife.IfeError) as e:
        ife.Context(0)
    assert e.value.code == ife.E_HIP and "no CPU path" in str(e.value)


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "image-feature-extraction_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".cxx", "Makefile", ".txt")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "libife_oracle" not in txt and \
                    "ife_oracle.h" not in txt, os.path.join(dirpath, f)
