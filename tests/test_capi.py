"""CPU tests of the C-ABI boundary: the shared library loads, exports every symbol include/msplat.h
declares, fails loudly without a GPU (no CPU fallback), and the product never touches oracle/."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from splatapult_amd import SplatRenderer, _capi
from tests.conftest import ROOT, has_gpu


def declared_symbols(headers=("msplat.h", "msplat_debug.h")):
    names = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(msplat_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_the_integrator_header_stays_small_and_free_of_debug_entry_points():
    """msplat.h is what a maintainer binding Sort / Render reads; taps and probes live in msplat_debug.h"""
    lines = open(os.path.join(ROOT, "include", "msplat.h")).read().splitlines()
    assert len(lines) <= 300, len(lines)
    assert not [n for n in declared_symbols(("msplat.h",)) if n.startswith("msplat_debug_") or "probe" in n]


def test_library_exports_every_declared_symbol():
    L = C.CDLL(_capi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "libmsplat.so does not export " + n
    bound = {n for n, _, _ in _capi.SYMBOLS}
    assert set(names) == bound, "ctypes binding and header disagree: %s" % (set(names) ^ bound)


def test_signatures_are_plain_c_no_framework_types():
    src = open(os.path.join(ROOT, "include", "msplat.h")).read() + open(os.path.join(ROOT, "include", "msplat_debug.h")).read()
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    assert "torch" not in code and "std::" not in code and "glm" not in code and "hipStream" not in code
    assert 'extern "C"' in code


def test_version_and_struct_sizes():
    L = _capi.lib()
    assert b"msplat" in L.msplat_version_string()
    assert C.sizeof(_capi.Config) == 72 and _capi.Config.rank_mode.offset == 48 and _capi.Config.spatial_order.offset == 56
    assert _capi.Config.two_pass.offset == 64
    assert C.sizeof(_capi.AttrOffsets) == 64
    assert C.sizeof(_capi.Stats) == 64 and C.sizeof(_capi.Timings) == 32


@pytest.mark.skipif(has_gpu(), reason="a GPU is present")
def test_create_fails_loudly_without_a_gpu():
    L = _capi.lib()
    h = C.c_void_p()
    rc = L.msplat_create(C.byref(h), None)
    assert rc == _capi.ERR_NO_DEVICE and not h.value
    msg = L.msplat_last_error(None).decode()
    assert "no HIP device" in msg and "no CPU fallback" in msg
    r = SplatRenderer()
    import numpy as np
    assert r.Init(np.zeros((4, 61), np.float32)) is False          # Init -> false after logging, like the reference
    assert "no CPU fallback" in r.last_error()


def test_bad_arguments_are_rejected_without_a_context():
    L = _capi.lib()
    assert L.msplat_create(None, None) == _capi.ERR_INVALID_ARG
    cfg = _capi.Config()
    cfg.struct_size = 7
    h = C.c_void_p()
    assert L.msplat_create(C.byref(h), C.byref(cfg)) == _capi.ERR_INVALID_ARG
    # the config struct grows at its end: the size of the struct before rank_mode was added is still accepted ...
    cfg = _capi.Config()
    cfg.struct_size = 48
    cfg.t_epsilon = -1.0
    rc = L.msplat_create(C.byref(h), C.byref(cfg))
    assert rc in (_capi.OK, _capi.ERR_NO_DEVICE)            # past the argument checks either way
    if rc == _capi.OK:
        L.msplat_destroy(h)
    # ... and so is the r3 struct, which ended before spatial_order (AUTO is then assumed)
    cfg = _capi.Config()
    cfg.struct_size = 56
    cfg.t_epsilon = -1.0
    rc = L.msplat_create(C.byref(h), C.byref(cfg))
    assert rc in (_capi.OK, _capi.ERR_NO_DEVICE)
    if rc == _capi.OK:
        L.msplat_destroy(h)
    # ... and the struct of the first r4 builds, which ended before two_pass
    cfg = _capi.Config()
    cfg.struct_size = 64
    cfg.t_epsilon = -1.0
    rc = L.msplat_create(C.byref(h), C.byref(cfg))
    assert rc in (_capi.OK, _capi.ERR_NO_DEVICE)
    if rc == _capi.OK:
        L.msplat_destroy(h)
    # ... a shorter size, a larger one, an unknown rank_mode, spatial_order or two_pass are not
    for size, mode, spatial, two in ((44, 0, 0, 0), (80, 0, 0, 0), (C.sizeof(_capi.Config), 7, 0, 0), (C.sizeof(_capi.Config), 0, 3, 0),
                                     (C.sizeof(_capi.Config), 0, 0, 3)):
        cfg = _capi.Config()
        cfg.struct_size = size
        cfg.rank_mode = mode
        cfg.spatial_order = spatial
        cfg.two_pass = two
        h = C.c_void_p()
        assert L.msplat_create(C.byref(h), C.byref(cfg)) == _capi.ERR_INVALID_ARG and not h.value
    assert L.msplat_sort(None, None, None, None, None) == _capi.ERR_INVALID_ARG
    assert L.msplat_render(None, None, None, None, None, None, 0, 0) == _capi.ERR_INVALID_ARG
    assert L.msplat_cloud_import_ply(None, b"x") == _capi.ERR_INVALID_ARG
    L.msplat_destroy(None)
    L.msplat_cloud_destroy(None)


def test_product_never_imports_or_links_the_oracle():
    pkg = os.path.join(ROOT, "splatapult_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                code = "\n".join(l for l in text.splitlines()
                                 if not l.strip().startswith(("#", "//", "*", '"""')) and "oracle is test" not in l)
                assert "import oracle" not in code and "from oracle" not in code and "liboracle" not in code, f
    # and the shared library has no dependency on it
    import subprocess
    out = subprocess.run(["ldd", _capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "torch" not in out


def test_cpp_shim_compiles_with_plain_gxx(tmp_path):
    """the SplatRenderer/GaussianCloud C++ surface (splatapult_amd/host/msplat_host.hpp) needs only g++ + the C ABI"""
    import subprocess
    exe = str(tmp_path / "example_render")
    libdir = os.path.dirname(_capi.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-I", ROOT, os.path.join(ROOT, "splatapult_amd", "host", "example_render.cpp"),
           "-L", libdir, "-lmsplat", "-Wl,-rpath," + libdir, "-o", exe]
    subprocess.run(cmd, check=True, cwd=ROOT)
    if not has_gpu():
        p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "test.ply"), str(tmp_path / "o.f32"), "64", "48"],
                           capture_output=True, text=True)
        assert p.returncode == 1 and "no CPU fallback" in p.stderr


def test_frame_arguments_are_marshalled_through_one_buffer():
    """renderer._FrameArgs (r3): the four per-call arrays go through one preallocated float buffer whose pointers are made
    once; sizes are still checked, lists and (4, 4) arrays are accepted, values arrive unchanged"""
    import ctypes as C
    from splatapult_amd.renderer import _FrameArgs
    a = _FrameArgs()
    cam = np.arange(16, dtype=np.float32).reshape(4, 4)
    proj = [float(v) for v in range(100, 116)]
    pc, pp, pv, pn = a.load(cam, proj, [0, 0, 640, 480], (0.1, 1000.0))
    assert [pc[i] for i in range(16)] == list(range(16))
    assert [pp[i] for i in range(16)] == list(range(100, 116))
    assert [pv[i] for i in range(4)] == [0.0, 0.0, 640.0, 480.0]
    assert abs(pn[0] - 0.1) < 1e-7 and pn[1] == 1000.0
    base = C.addressof(pc.contents)
    for p, off in ((pp, 64), (pv, 128), (pn, 144)):
        assert C.addressof(p.contents) == base + off
    again = a.load(cam.T.copy(), proj, [0, 0, 8, 8], (1.0, 2.0))
    assert C.addressof(again[0].contents) == base and again[0][1] == 4.0       # same buffer, new values
    for bad in ((cam[:3], proj, [0, 0, 1, 1], (0.1, 1.0)), (cam, proj[:15], [0, 0, 1, 1], (0.1, 1.0)),
                (cam, proj, [0, 0, 1], (0.1, 1.0)), (cam, proj, [0, 0, 1, 1], (0.1,)), (cam, proj, [0, 0, 1, 1], 3.0)):
        with pytest.raises(AssertionError):
            a.load(*bad)


def test_every_kernel_the_host_registers_exists_in_the_device_code(tmp_path):
    """The host half of libmsplat.so registers its kernels by mangled name; the first launch of a kernel whose name the device
    compilation mangled differently aborts the process ("Cannot find Symbol", seen r5 with an unnamed enum in a kernel's
    signature).  Checked without a GPU: every registered name has its kernel descriptor in the gfx950 code object."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not all(os.path.exists(os.path.join(llvm, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")):
        pytest.skip("ROCm llvm tools not present")
    lib = os.path.join(ROOT, "splatapult_amd", "lib", "libmsplat.so")
    work = str(tmp_path)
    shutil.copy(lib, os.path.join(work, "lib.so"))
    fb, dev = os.path.join(work, "fb.bin"), os.path.join(work, "dev.elf")
    subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, os.path.join(work, "lib.so")], check=True)
    subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fb,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + dev], check=True)
    syms = subprocess.run([os.path.join(llvm, "llvm-readelf"), "-s", "-W", dev], check=True, capture_output=True, text=True).stdout
    device = set(re.findall(r"(\S+)\.kd\b", syms))
    ro = subprocess.run([os.path.join(llvm, "llvm-readelf"), "-p", ".rodata", os.path.join(work, "lib.so")], check=True,
                        capture_output=True, text=True).stdout
    host = set(re.findall(r"\]\s+(_ZN6msplat\S+)", ro))
    assert len(host) > 40, len(host)                       # the registration strings were found at all
    missing = sorted(h for h in host if h not in device)
    assert not missing, missing
