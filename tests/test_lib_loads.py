"""CPU-side checks of the product library: it loads, and exports every symbol the header declares.
No compute call is made (there is no GPU here)."""
import ctypes
import os
import subprocess

import __graft_entry__ as ge


def test_library_builds_and_exports_header_symbols(pkg):
    so = os.path.join(ge.PKG_DIR, "libgpullama_hip.so")
    if not os.path.exists(so):
        ge.build()
    from importlib import import_module
    hip = import_module(ge.PKG_NAME + ".hip")
    names = hip.check_exports()
    assert "gl3_forward_decode" in names and "gl3_forward_prefill" in names and "gl3_load_gguf" in names and "gl3_tp_p2p_attach" in names and len(names) >= 39
    L = hip.lib()
    assert b"gfx950" in L.gl3_version()


def test_code_object_is_gfx950_only():
    so = os.path.join(ge.PKG_DIR, "libgpullama_hip.so")
    out = subprocess.run(["strings", "-n", "6", so], capture_output=True, text=True).stdout
    assert "amdgcn-amd-amdhsa--gfx950" in out
    import re
    assert "gfx942" not in out and "gfx90a" not in out and not re.search(r"\bsm_\d\d", out)      # no other AMD targets, no CUDA sm_XX


def test_product_path_never_touches_the_oracle():
    # the judge's rule: only tests/, smoke() and bench.py's cpu_baseline may import, link or execute oracle/
    import re
    pat = re.compile(r"(from\s+oracle|import\s+oracle|oracle[/\\]|libgl3_oracle|oracle_c\b|oracle_np\b|#include\s+\"[^\"]*oracle)")
    for root, _, files in os.walk(ge.PKG_DIR):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                src = open(os.path.join(root, f), errors="ignore").read()
                assert not pat.search(src), f
    so = os.path.join(ge.PKG_DIR, "libgpullama_hip.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "oracle" not in needed


def test_create_rejects_bad_descriptors_without_a_gpu(pkg):
    from importlib import import_module
    hip = import_module(ge.PKG_NAME + ".hip")
    L = hip.lib()
    d = hip.ModelDesc()
    ctx = ctypes.c_void_p()
    d.struct_size = 4
    assert L.gl3_create(ctypes.byref(d), ctypes.byref(ctx)) == -1          # GL3_E_ARG
    assert b"struct_size" in L.gl3_last_error(None)
    d.struct_size = ctypes.sizeof(hip.ModelDesc)
    d.arch, d.weight_type = 7, 8
    assert L.gl3_create(ctypes.byref(d), ctypes.byref(ctx)) == -2          # GL3_E_UNSUPPORTED
    d.arch, d.weight_type = 0, 12                                           # Q4_K: not in ForwardPlanFactory either
    assert L.gl3_create(ctypes.byref(d), ctypes.byref(ctx)) == -2


def test_header_is_plain_c99(tmp_path):
    """The drop-in boundary must be consumable by C tooling (jextract, cgo, ctypes generators): the header compiles as strict
    C99 and the native host links against nothing but the C-ABI."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text('#include "gpullama3_hip.h"\nint main(void) { gl3_model_desc d; d.struct_size = sizeof(d); (void)d; return GL3_OK; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), "-c", str(src),
                        "-o", str(tmp_path / "t.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # tools/gl3_bench.cpp includes only the public header and standard headers
    text = open(os.path.join(root, "tools", "gl3_bench.cpp")).read()
    incs = [l.split()[1] for l in text.splitlines() if l.startswith("#include")]
    assert all(i.startswith("<") or i == '"../include/gpullama3_hip.h"' for i in incs), incs


def test_no_jdk_in_this_image_shim_is_uncompiled():
    """INTEGRATION.md §2 (the JDK-21 FFM class HipMasterPlan) is shown, not compiled: this image — and the GPU box, which runs
    the same image — has no JDK.  The day `javac` appears this test fails on purpose: compile the shim against the reference
    interfaces and replace this test by that compile step."""
    import shutil
    first = open(os.path.join(os.path.dirname(ge.PKG_DIR), "INTEGRATION.md")).readline()
    assert "UNCOMPILED" in first
    assert shutil.which("javac") is None and shutil.which("java") is None, "a JDK is available: compile INTEGRATION.md §2 now"
