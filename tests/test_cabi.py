"""The C-ABI library loads on a machine without a GPU and exports every entry point the header
declares; the ctypes mirror of the structs has the header's sizes.  No compute call is made."""
import ctypes
import importlib
import os
import re

import pytest

from tests.helpers import ROOT

pkg = importlib.import_module("monocular-visual-odometry_amd")


def header_functions():
    text = open(os.path.join(ROOT, "include", "picp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(picp_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(pkg._lib.LIB_PATH):
        pytest.fail(f"{pkg._lib.LIB_PATH} not built: run __graft_entry__.build()")
    lib = pkg.load_library()
    names = header_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(pkg._lib.EXPORTED_SYMBOLS) == names      # the Python binding covers the whole header
    assert lib.picp_version() == 100


def test_struct_layouts_match_header():
    L = pkg._lib
    assert ctypes.sizeof(L.CameraStruct) == 11 * 8
    assert ctypes.sizeof(L.ScheduleStruct) == 8 + 6 * 8 + 8
    assert ctypes.sizeof(L.ResultStruct) == 16 * 8 + 16 + 8
    text = open(os.path.join(ROOT, "include", "picp_hip.h")).read()
    consts = {k: int(v) for k, v in re.findall(r"#define (PICP_[A-Z0-9_]+) (\d+)\b", text)}
    assert consts["PICP_HIST_STRIDE"] == L.HIST_STRIDE and consts["PICP_STATE_DOUBLES"] == L.STATE_DOUBLES
    assert consts["PICP_S1_DOUBLES"] == L.S1_DOUBLES and consts["PICP_S2_DOUBLES"] == L.S2_DOUBLES
    for name, val in [("PICP_H_T", L.H_T), ("PICP_H_ERROR", L.H_ERROR), ("PICP_H_NUM_INLIERS", L.H_NUM_INLIERS),
                      ("PICP_H_KERNEL_THR", L.H_KERNEL_THR), ("PICP_H_DAMPING", L.H_DAMPING),
                      ("PICP_H_N_INSIDE", L.H_N_INSIDE), ("PICP_H_DONE", L.H_DONE), ("PICP_H_DX", L.H_DX),
                      ("PICP_H_B", L.H_B), ("PICP_H_H", L.H_H)]:
        assert consts[name] == val, name


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.PicpError):
        pkg.get_context(0)
    with pytest.raises(pkg.PicpError):
        pkg.data_association_on_appearance({"position": [[0, 0]], "appearance": [[1.0]]},
                                           {"position": [[0, 0, 1]], "appearance": [[1.0]]})


def test_header_compiles_as_plain_c(tmp_path):
    """include/picp_hip.h is the boundary other host languages bind: it must be valid C99 on its own."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("no gcc")
    src = tmp_path / "use_header.c"
    src.write_text('#include "picp_hip.h"\nint main(void) { picp_result r; picp_schedule s; picp_camera c;\n'
                   '  (void)r; (void)s; (void)c; return sizeof(picp_result) == 152 ? 0 : 1; }\n')
    header_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    done = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", header_dir, str(src)],
                          capture_output=True, text=True)
    assert done.returncode == 0, done.stderr
