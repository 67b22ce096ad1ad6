"""The ICC parser reads untrusted bytes (formatRecord->iCCprofileData): truncations and corrupted tag tables / counts / offsets
must come back as error codes (or as a transform, if the damage is harmless) -- never as a crash or an out-of-bounds read.
Runs the four host-side entry points on a few thousand seeded mutations of valid profiles (CPU only; run under the test process,
so a segfault fails the suite loudly)."""
import ctypes
import os

import numpy as np
import pytest

import harness

pkg = harness.pkg
V = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "icc_vectors.npz"))
PROFILES = [V[k].tobytes() for k in sorted(V.files) if k.endswith(".icc")]


def _call_all(lib, blob):
    n = len(blob)
    buf = ctypes.create_string_buffer(blob, n) if n else ctypes.create_string_buffer(1)
    codes = [lib.avifgpu_icc_detect(buf, n),
             lib.avifgpu_icc_prepare(buf, n, pkg.ICC_TARGET_REC2020_LINEAR, ctypes.byref(pkg.IccTransform())),
             lib.avifgpu_icc_prepare(buf, n, pkg.ICC_TARGET_SRGB_FLOAT, ctypes.byref(pkg.IccTransform())),
             lib.avifgpu_icc_prepare_shaper8(buf, n, ctypes.byref(pkg.IccShaper8()))]
    return codes


def test_truncations_are_rejected_or_harmless():
    lib = pkg.load()
    for icc in PROFILES:
        for n in list(range(0, 200, 7)) + list(range(200, len(icc), 37)):
            _call_all(lib, icc[:n])


def test_corrupted_tag_tables_do_not_crash():
    lib = pkg.load()
    rng = np.random.default_rng(77)
    clut = pkg.IccClut16()
    for icc in PROFILES:
        count = int.from_bytes(icc[128:132], "big")
        table_end = 132 + 12 * count
        for trial in range(400):
            b = bytearray(icc)
            kind = trial % 5
            if kind == 0:                                   # random bytes inside the tag table (signatures, offsets, sizes)
                for _ in range(int(rng.integers(1, 6))):
                    b[int(rng.integers(128, table_end))] = int(rng.integers(0, 256))
            elif kind == 1:                                 # huge / zero tag count
                b[128:132] = int(rng.choice([0, 1, 0xffffffff, 0x7fffffff, count + 1000])).to_bytes(4, "big")
            elif kind == 2:                                 # offsets / sizes pointing outside the buffer
                i = 132 + 12 * int(rng.integers(0, count)) + int(rng.choice([4, 8]))
                b[i:i + 4] = int(rng.choice([0xffffffff, 0xfffffff0, len(icc) - 1, len(icc), 0x80000000])).to_bytes(4, "big")
            elif kind == 3:                                 # curve entry counts / parametric function types
                for _ in range(int(rng.integers(1, 4))):
                    b[int(rng.integers(table_end, len(icc)))] = int(rng.choice([0, 1, 0x7f, 0xff]))
            else:                                           # random tail garbage
                pos = int(rng.integers(table_end, len(icc)))
                b[pos:] = bytes(rng.integers(0, 256, size=len(icc) - pos, dtype=np.uint8))
            blob = bytes(b)
            _call_all(lib, blob)
            if trial % 8 == 0:                              # the 16-bit table builder is slower (35937 nodes): sample it
                lib.avifgpu_icc_prepare_clut16(ctypes.create_string_buffer(blob, len(blob)), len(blob), ctypes.byref(clut))


def test_valid_profiles_still_parse():
    lib = pkg.load()
    for icc in PROFILES:
        assert lib.avifgpu_icc_detect(icc, len(icc)) >= 0
        assert lib.avifgpu_icc_prepare_shaper8(icc, len(icc), ctypes.byref(pkg.IccShaper8())) == 0
