"""The C-ABI library loads on a GPU-less machine and exports exactly what include/apgd_hip.h declares."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "apgd_hip.h")
HEADERS = [HEADER, os.path.join(ROOT, "include", "convnext_hip.h")]


def declared_functions():
    src = "\n".join(open(h).read() for h in HEADERS)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+((?:apgd|cnx)_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


@pytest.fixture(scope="module")
def R():
    import revisiting_at_amd as R
    if not os.path.exists(R._lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "revisiting-at_amd", "csrc")], check=True)
    return R


def test_header_declares_the_expected_entry_points():
    d = declared_functions()
    for name in ("apgd_hip_version", "apgd_init_f32", "apgd_linf_step_f32", "apgd_l2_step_f32", "apgd_loss_pred",
                 "apgd_state_update", "apgd_track_rows", "apgd_check_imgs_f32", "cnx_dwconv7x7_nhwc",
                 "cnx_dwconv7x7_wgrad_nhwc", "cnx_layernorm_fwd", "cnx_layernorm_bwd"):
        assert name in d


def test_library_exports_every_declared_symbol(R):
    lib = R._lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in apgd_hip.h but not exported"
    assert lib.apgd_hip_version() == 10800
    assert lib.apgd_hip_strerror(0) == b"ok" and b"NULL" in lib.apgd_hip_strerror(-1)
    assert lib.apgd_l2_parts() == 64


def test_python_prototypes_match_header(R):
    d = declared_functions()
    assert set(R._lib.PROTOTYPES) == set(d)
    for name, (_, args) in R._lib.PROTOTYPES.items():
        assert len(args) == d[name], name


def test_no_torch_or_cxx_types_in_the_abi():
    src = "\n".join(open(h).read() for h in HEADERS)
    assert 'extern "C"' in src and "at::" not in src and "torch" not in src.lower().replace("pytorch", "")
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "revisiting-at_amd", "libapgd_hip.so")],
                         capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(declared_functions()) <= exported


def test_missing_library_fails_loudly(R, tmp_path):
    with pytest.raises(R._lib.ApgdHipError):
        R._lib.load(str(tmp_path / "nope.so"))
