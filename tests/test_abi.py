"""CPU-side checks of the drop-in boundary: the shared library builds, loads, exports every symbol
include/gmapper_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os, re, subprocess
import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def gm():
    from shrimp_amd import gmapper
    if not os.path.exists(gmapper.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "shrimp_amd", "csrc"), "-j8"], check=True, capture_output=True)
    return gmapper


def declared_functions():
    src = open(os.path.join(ROOT, "include", "gmapper_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set()
    for m in re.finditer(r"\b([a-z_][a-z0-9_]*)\s*\(", src):
        n = m.group(1)
        if n.startswith(("gm_", "sw_", "post_sw")) and not n.endswith("_t"):
            names.add(n)
    return sorted(names)


def test_every_declared_symbol_is_exported(gm):
    L = gm.lib()
    decl = declared_functions()
    assert len(decl) >= 25
    missing = [n for n in decl if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(gm.EXPORTS) == decl, set(gm.EXPORTS) ^ set(decl)


def test_symbols_have_c_linkage(gm):
    out = subprocess.run(["nm", "-D", "--defined-only", gm.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for n in declared_functions():
        assert n in syms, n          # unmangled: the reference's own names (sw_vector, sw_full_ls, ...) link as C


def test_params_default_match_reference_defaults(gm):
    p = gm.default_params()
    assert (p.match_score, p.mismatch_score, p.a_gap_open_score, p.a_gap_extend_score, p.b_gap_open_score, p.b_gap_extend_score) == (10, -15, -33, -7, -33, -3)
    assert (p.window_len, p.window_overlap, p.window_gen_threshold, p.sw_vect_threshold, p.sw_full_threshold) == (140.0, 90.0, 55.0, 50.0, 50.0)
    assert (p.match_mode, p.num_outputs, p.num_tmp_outputs, p.anchor_width, p.region_bits, p.region_overlap) == (2, 10, 30, 8, 11, 50)


def test_no_cpu_fallback_without_gpu(gm):
    L = gm.lib()
    if L.gm_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(gm.GmError, match="no HIP device"):
        gm.Index([np.zeros(1000, dtype=np.uint8)])
    with pytest.raises(gm.GmError):
        gm.sw_vector_setup(140, 100, -33, -7, -33, -3, 10, -15)


def test_sw_vector_setup_range_check(gm):
    """match * qrlen >= 32768 is rejected (the reference exit(1)s, sw-vector.c:393-398) before any device use."""
    rc = gm.lib().sw_vector_setup(1400, 4000, -33, -7, -33, -3, 10, -15, 0, True)
    assert rc == -4


def test_reference_objects_link_against_the_library(gm, tmp_path):
    """INTEGRATION.md section A, literally: the reference's own objects (gmapper/*.o and common/*.o as oracle/Makefile.ref compiles them from
    /root/reference) minus sw-vector.o, sw-gapless.o, sw-full-ls.o, sw-full-cs.o and sw-post.o link against libgmapper_hip.so without an unresolved symbol --
    the library exports the C++-linkage names those objects reference (ref: common/util.h:8-10 has extern "C" commented out).  Link only:
    nothing runs here (no GPU), and nothing of the reference travels in source form."""
    objdir = os.path.join(ROOT, "oracle", "_ref", "obj")
    if not os.path.isdir(os.path.join(objdir, "gmapper")):
        if not os.path.isdir("/root/reference/gmapper"):
            pytest.skip("reference objects not built and /root/reference absent (GPU box)")
        subprocess.run(["make", "-f", os.path.join("oracle", "Makefile.ref"), "-j8"], cwd=ROOT, check=True, capture_output=True)
    import glob
    dropped = {"sw-vector.o", "sw-gapless.o", "sw-full-ls.o", "sw-full-cs.o", "sw-post.o"}
    objs = sorted(glob.glob(os.path.join(objdir, "gmapper", "*.o"))) + [o for o in sorted(glob.glob(os.path.join(objdir, "common", "*.o"))) if os.path.basename(o) not in dropped]
    assert len(objs) == 5 + 9, objs
    exe = str(tmp_path / "gmapper-seams")
    r = subprocess.run(["g++", "-fopenmp", "-o", exe, *objs, "-L" + os.path.dirname(gm.LIB_PATH), "-lgmapper_hip", "-Wl,-rpath," + os.path.dirname(gm.LIB_PATH),
                        "-Wl,-rpath-link,/opt/rocm/lib", "-lm", "-lz", "-lstdc++", "-lrt"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # the symbols the dropped objects used to define are now undefined in the program and defined (mangled, as the objects spell them) by the library
    und = subprocess.run(["nm", "-u", exe], capture_output=True, text=True, check=True).stdout
    lib = subprocess.run(["nm", "-D", "--defined-only", gm.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for sym in ("_Z9sw_vectorPjiiS_iS_ib", "_Z15sw_vector_setupiiiiiiiiib", "_Z10sw_gaplessPjiS_iiiS_ib", "_Z16sw_gapless_setupiib", "_Z16sw_gapless_statsPmS_S_", "_Z10sw_full_lsPjiiS_iiiP15sw_full_resultsbP6anchorii", "_Z16sw_full_ls_statsPmS_Pd",
                "_Z10sw_full_csPjiiS_iiiP15sw_full_resultsbbP6anchoriiPi", "_Z16sw_full_cs_statsPmS_Pd", "_Z13post_sw_setupiddddddbbiib", "_Z7post_swPjiPcP15sw_full_results",
                "_Z13post_sw_statsPmS_Pd", "_Z15post_sw_cleanupv"):
        assert sym in und, sym
        assert sym in lib, sym


def test_seam_driver_links_by_the_mangled_names(gm, tmp_path):
    """tests/seam_driver.cpp (the program the GPU parity test runs over every known-answer record) declares the seams with C++ linkage, as the reference's
    headers do: it must link against the library, resolving the Itanium-mangled names to the library's exports.  Link only (no GPU here)."""
    exe = str(tmp_path / "seam_driver")
    libdir = os.path.dirname(gm.LIB_PATH)
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "seam_driver.cpp"), "-L" + libdir, "-lgmapper_hip", "-Wl,-rpath," + libdir,
                        "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    und = subprocess.run(["nm", "-u", exe], capture_output=True, text=True, check=True).stdout
    lib = subprocess.run(["nm", "-D", "--defined-only", gm.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for sym in ("_Z9sw_vectorPjiiS_iS_ib", "_Z10sw_gaplessPjiS_iiiS_ib", "_Z10sw_full_lsPjiiS_iiiP15sw_full_resultsbP6anchorii",
                "_Z10sw_full_csPjiiS_iiiP15sw_full_resultsbbP6anchoriiPi", "_Z7post_swPjiPcP15sw_full_results", "_Z13post_sw_setupiddddddbbiib"):
        assert sym in und and sym in lib, sym


def test_nothing_built_from_the_reference_travels_to_the_gpu_box():
    """oracle/_ref/ (the reference compiled by oracle/Makefile.ref) is kept out of history AND out of the gpurun snapshot"""
    assert "oracle/_ref/" in open(os.path.join(ROOT, ".gitignore")).read().split()
    assert "oracle/_ref/" in open(os.path.join(ROOT, ".gpurunignore")).read().split()


def test_struct_sizes_match_the_python_mirrors():
    """gm_abi_sizeof: the ctypes mirrors in shrimp_amd/gmapper.py have the size the library was compiled with (a shorter mirror would let gm_params_default write past it)"""
    import ctypes as C
    from shrimp_amd import gmapper as gm
    L = gm.lib(); L.gm_abi_sizeof.restype = C.c_int; L.gm_abi_sizeof.argtypes = [C.c_int]
    for which, cls in enumerate((gm.Params, gm.PairOpts, gm.MapStats, gm.MergeOptions)):
        assert L.gm_abi_sizeof(which) == C.sizeof(cls), (cls.__name__, L.gm_abi_sizeof(which), C.sizeof(cls))
    assert L.gm_abi_sizeof(99) == -1
