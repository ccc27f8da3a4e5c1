"""Compile proof of the drop-in claim: the reference application's own main.cpp
(R/main.cpp:23-149 -- `camera` tweaks :64-65, the constructor :66, init_voxel_cone_tracing() :68, Render() :90,
the camera callbacks :101-149) is syntax-checked UNCHANGED against voxel-cone-tracing_amd/host/Voxel_Cone_Tracing.h
in place of the reference's header.  GLFW / GLEW / GL names are only declared (tests/ref_gl_decls.h): no window
system exists here.  Runs in this container only -- the reference tree does not travel to the GPU box."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MAIN = "/root/reference/Voxel_Cone_Tracing_Final/main.cpp"


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference tree is not present on this box")
def test_reference_main_cpp_compiles_against_the_facade(tmp_path):
    # a quoted #include searches the including file's directory first, so main.cpp is copied (at run time, into
    # a scratch directory) next to a one-line Voxel_Cone_Tracing.h that forwards to the facade
    shutil.copy(REF_MAIN, tmp_path / "main.cpp")
    (tmp_path / "Voxel_Cone_Tracing.h").write_text(
        f'#include "{ROOT}/tests/ref_gl_decls.h"\n'
        f'#include "{ROOT}/voxel-cone-tracing_amd/host/Voxel_Cone_Tracing.h"\n')
    # -Dmain=ref_main: the reference declares `void main()` (MSVC accepts it, ISO C++ does not)
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Dmain=ref_main", "-w", str(tmp_path / "main.cpp")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
