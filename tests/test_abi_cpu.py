"""CPU-only: the C-ABI library loads and exports every symbol include/mtgv.h declares; the ctypes table
covers the header; the host classes refuse to run without a GPU (no silent fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "mtgv.h")


@pytest.fixture(scope="module")
def libpath():
    from mtgv import native

    if not os.path.exists(native.LIB_PATH):
        subprocess.run([sys.executable, os.path.join(ROOT, "mtg-vision_amd", "build.py")], check=True)
    return native.LIB_PATH


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"MTGV_API\s+[\w\s\*]+?\b(mtgv_\w+)\s*\(", src)))


def test_header_symbols_exported(libpath):
    names = _declared()
    assert len(names) >= 40
    lib = ctypes.CDLL(libpath)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"not exported: {missing}"


def test_ctypes_table_matches_header(libpath):
    from mtgv import native

    assert sorted(native.SIGNATURES) == _declared()
    L = native.lib()
    assert L.mtgv_version() >= 100
    assert L.mtgv_device_count() >= 0


def test_status_convention_without_gpu(libpath):
    from mtgv import native

    L = native.lib()
    # null arguments -> status 1 (AssertionError), message available; no device needed
    rc = L.mtgv_encoder_create(None, None)
    assert rc == 1 and b"null" in L.mtgv_last_error()
    with pytest.raises(AssertionError):
        native.check(rc)
    assert L.mtgv_nms_workspace_bytes(2, 8400) == 2 * 16384 * 7 * 4
    assert L.mtgv_warp_workspace_bytes(3) == 3 * 9 * 8


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mtgv import spec
    from mtgv.encoder import Encoder
    from mtgv.matcher import Matcher

    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        Encoder(spec.encoder_config("cnvnxt2ae_nano"))
    with pytest.raises(RuntimeError):
        Matcher(768)


def test_product_does_not_import_oracle():
    """the oracle is test infrastructure: nothing under mtg-vision_amd/ may import it"""
    pkg = os.path.join(ROOT, "mtg-vision_amd")
    bad = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_gemm_precision_selection_without_gpu(libpath):
    """default f16x3; MTGV_GEMM_PREC picks the initial mode; a bad value or a bad code is an AssertionError-class status"""
    code = (
        "import sys; sys.path[:0] = [%r]\n"
        "from mtgv import native\n"
        "print(native.get_gemm_precision())\n"
        "native.set_gemm_precision('f32'); print(native.get_gemm_precision())\n"
        "print(native.lib().mtgv_set_gemm_precision(5))\n"
    ) % os.path.join(ROOT, "mtg-vision_amd")

    def run(env_val):
        env = {k: v for k, v in os.environ.items() if k != "MTGV_GEMM_PREC"}
        if env_val is not None:
            env["MTGV_GEMM_PREC"] = env_val
        return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)

    r = run(None)
    assert r.returncode == 0 and r.stdout.split() == ["f16x3", "f32", "1"], r.stdout + r.stderr
    assert run("f32").stdout.split()[0] == "f32"
    bad = run("fp8")
    assert bad.returncode != 0 and "AssertionError" in bad.stderr and "MTGV_GEMM_PREC" in bad.stderr
