"""Syntax check (g++ -fsyntax-only) of the drop-in files that are written against the reference's headers, the Photoshop SDK and libheif:
the two adapters under integration/ and the harness of the reference pin under oracle/.  Declaration-only headers stand in for the two
SDKs this image lacks (tests/compilecheck/README.md): NO object is produced, nothing here is a build of the reference, and the pin
(oracle/_ref, tests/test_ref_pin.py) stays skipped.  Skips where the reference checkout or g++ is absent (e.g. on the GPU box)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFSRC = "/root/reference/src/common"
DECL = os.path.join(ROOT, "tests", "compilecheck", "decl")
FILES = ["integration/WriteHeifImage_gpu.cpp", "integration/ReadHeifImage_gpu.cpp", "oracle/ref_harness.cpp", "oracle/ref_transfer_wrap.cpp"]


def _lcms_include():
    for prefix in ("/opt/conda", "/usr", "/usr/local"):
        if os.path.exists(os.path.join(prefix, "include", "lcms2.h")):
            return os.path.join(prefix, "include")
    return None


@pytest.mark.parametrize("path", FILES)
@pytest.mark.parametrize("fused", [False, True])
def test_drop_in_file_passes_a_syntax_check(path, fused):
    if not os.path.exists(os.path.join(REFSRC, "WriteHeifImage.h")):
        pytest.skip("no reference checkout at /root/reference (its headers are read where they lie)")
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not installed")
    lcms = _lcms_include()
    if lcms is None:
        pytest.skip("lcms2.h not installed (ColorProfileConversion.h includes it)")
    if fused and not path.startswith("integration/Write"):
        pytest.skip("AVIFGPU_FUSED_YCBCR only changes the save adapter")
    # the declaration-only directory goes LAST among the quoted-include paths and is never on a Makefile's path
    cmd = [gxx, "-std=c++20", "-fsyntax-only", "-Wall", "-Wno-unused", "-I" + REFSRC, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "integration"),
           "-I" + lcms, "-I" + DECL] + (["-DAVIFGPU_FUSED_YCBCR=1"] if fused else []) + [os.path.join(ROOT, path)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "error" not in r.stdout, r.stdout[-4000:]


def test_declaration_headers_are_on_no_build_path():
    """Guard rail: no Makefile of the repository may see tests/compilecheck (the pin and the adapters build against REAL headers only)."""
    for mk in ("oracle/Makefile", "integration/Makefile", "avif-format_amd/Makefile"):
        assert "compilecheck" not in open(os.path.join(ROOT, mk)).read(), mk
    assert not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref.so")) or os.environ.get("PSSDK"), "oracle/_ref exists without a real SDK?"
