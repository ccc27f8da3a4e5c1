"""Independent cross-check of the oracle's gather + composite against the reference's own shader TEXT.

The reference's cone-trace fragment shader (S/VoxelConeTracing.fs) is read from /root/reference at run time -- it is
never stored in this repository -- rewritten mechanically into C++ (qualifiers dropped, GLSL array constructors ->
braces, multi-component swizzles -> calls, `discard` -> a flag) and compiled against tests/glsl_shim.h, a minimal
GLSL stand-in written for this test.  `main()` of the shader then runs on 10^4 random G-buffer pixels next to
oracle/vct_oracle.cpp vcto_shade_pixel on the same inputs.

What it shows: the oracle's restatement of trace.fs:82-107 (march), :165-228 (frame, 6 + 1 cones, composite, the .rrra
rule, discard) agrees with the shader as written -- operation by operation but for rounding (the stand-in uses the GLSL
definitions of normalize / reflect / inverse and no fused multiply-adds; the oracle fixes its own fp32 order).
What it does NOT show by itself: `textureLod` here is the oracle's own sampler and the material / height / shadow
samplers are per-pixel constants, so this test alone pins no output value.  (Round 5: the reference's shaders DO run in
this container, unmodified, on Mesa llvmpipe -- oracle/ref_gl.c -- and tests/test_ref_gl.py holds the oracle to what
they produce; this text-level cross-check stays as a second, driver-independent reading of the same shader.)
Runs in the build container only: the reference tree does not travel to the GPU box."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FS = "/root/reference/Voxel_Cone_Tracing_Final/Shader/VoxelConeTracing.fs"
ORACLE = os.path.join(ROOT, "oracle", "libvct_oracle.so")

HARNESS = r'''
ShimState g_shim;
}  // namespace glsl
extern "C" int vcto_shade_pixel(const void* p, const unsigned char* chain, const float gb[23], float out[4],
                                unsigned char steps[7], float cones[28]);
using namespace glsl;
// gbin [n][23] (bump normal slots ignored: CalcBumpNormal of the shader fills them), spec_raw [n][4], shadow_k [n];
// outputs: shader colour [n][4], oracle colour [n][4], steps of both [n][7], the G-buffer as fed to the oracle [n][23]
extern "C" int run_pixels(const void* params, const unsigned char* chain, int V, float G, const float cam[3],
                          const float light[3], float ambient, float shininess, int n, const float* gbin,
                          const float* spec_raw, const int* shadow_k, float* out_shader, float* out_oracle,
                          int* steps_shader, int* steps_oracle, float* gb_used, int* discarded) {
    VoxelGridWorldSize = G; VoxelDimensions = V; ambientFactor = ambient; Shininess = shininess; Opacity = 1.0f;
    LightDirection = vec3(light[0], light[1], light[2]);
    ShadowMapSize = 4096; HeightTextureSize = vec2(64.0, 64.0);
    DiffuseTexture.which = 0; SpecularTexture.which = 1; MaskTexture.which = 2; HeightTexture.which = 3; ShadowMap.which = 4;
    for (int i = 0; i < n; ++i) {
        const float* g = gbin + 23 * i;
        Position_world = vec3(g[0], g[1], g[2]);
        Normal_world = vec3(g[3], g[4], g[5]);
        Tangent_world = vec3(g[6], g[7], g[8]);
        BiTangent_world = vec3(g[9], g[10], g[11]);
        CameraDirection_world = vec3(cam[0], cam[1], cam[2]) - Position_world;      // trace.vs:34
        Position_depth = vec4(0.5, 0.5, 0.5, 1.0);
        tex = vec2(0.25, 0.75);
        g_shim = ShimState();
        g_shim.params = params; g_shim.chain = chain;
        g_shim.diffuse = vec4(g[15], g[16], g[17], g[18]);
        g_shim.specular = vec4(spec_raw[4 * i], spec_raw[4 * i + 1], spec_raw[4 * i + 2], spec_raw[4 * i + 3]);
        g_shim.height = 0.5f;
        g_shim.shadow_pass = shadow_k[i];
        color = vec4(-1.0);
        shader_main();
        discarded[i] = g_shim.discarded ? 1 : 0;
        out_shader[4 * i] = color.x; out_shader[4 * i + 1] = color.y; out_shader[4 * i + 2] = color.z; out_shader[4 * i + 3] = color.w;
        for (int k = 0; k < 7; ++k) steps_shader[7 * i + k] = g_shim.cone_steps[k];
        if (g_shim.ncones > 7) steps_shader[7 * i] = -1;
        // the oracle's G-buffer: the same pixel with the shader's own bump normal, the resolved specular colour
        // (trace.fs:210) and the shader's shadow term as INPUTS (SURVEY.md 8 a5)
        float gb[23];
        for (int k = 0; k < 23; ++k) gb[k] = g[k];
        const mat3 TBN = inverse(transpose(mat3(Tangent_world, BiTangent_world, Normal_world)));
        const vec3 N = CalcBumpNormal(TBN);
        gb[12] = N.x; gb[13] = N.y; gb[14] = N.z;
        const vec4 sp = g_shim.specular;
        const bool has_gb = length(sp.gb()) > 0.0f;
        gb[19] = sp.x; gb[20] = has_gb ? sp.y : sp.x; gb[21] = has_gb ? sp.z : sp.x;
        g_shim.shadow_calls = 0;
        gb[22] = PCF_Shadow_Mapping(0.002f);
        for (int k = 0; k < 23; ++k) gb_used[23 * i + k] = gb[k];
        unsigned char st[7];
        vcto_shade_pixel(params, chain, gb, out_oracle + 4 * i, st, nullptr);
        for (int k = 0; k < 7; ++k) steps_oracle[7 * i + k] = st[k];
    }
    return 0;
}
'''


def glsl_to_cpp(src):
    """The mechanical rewrites; anything else in the text is compiled as it stands."""
    out = []
    for line in src.splitlines():
        if line.lstrip().startswith("#version"):
            continue
        line = re.sub(r"^\s*(in|out|uniform)\s+", "", line)                  # storage qualifiers -> plain globals
        out.append(line)
    s = "\n".join(out)
    s = re.sub(r"=\s*float\[\]\s*\(([^;]*)\)\s*;", r"= {\1};", s)           # float[](...)  -> {...}
    s = re.sub(r"=\s*vec3\[\]\s*\(([^;]*)\)\s*;", r"= {\1};", s, flags=re.S)  # vec3[](...)   -> {...}
    s = re.sub(r"\.(rgb|gb|rrra|xy)\b", r".\1()", s)                        # multi-component swizzles (reads only)
    s = re.sub(r"\bdiscard\s*;", "{ g_shim.discarded = true; return; }", s)
    s = re.sub(r"\bvoid\s+main\s*\(\s*\)", "void shader_main()", s)
    # instrumentation: count the textureLod calls of each Voxel_Cone_Tracing invocation (= its executed steps)
    s, k = re.subn(r"(vec4\s+Voxel_Cone_Tracing\s*\([^)]*\)\s*\{)", r"\1 shim_cone_begin();", s)
    assert k == 1
    return s


@pytest.mark.skipif(not os.path.exists(REF_FS), reason="the reference tree is not present on this box")
def test_oracle_agrees_with_the_reference_shader_text(tmp_path):
    import synth
    from oracle import pyoracle
    cpp = tmp_path / "shader_gen.cpp"
    cpp.write_text(f'#include "{ROOT}/tests/glsl_shim.h"\nnamespace glsl {{\n' + glsl_to_cpp(open(REF_FS).read()) + HARNESS)
    so = tmp_path / "libshader_check.so"
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-w", "-o", str(so), str(cpp),
                        ORACLE, f"-Wl,-rpath,{os.path.dirname(ORACLE)}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    lib = C.CDLL(str(so))
    for V, seed in ((32, 11), (64, 12)):
        n = 5000
        l0 = synth.noise_volume(V, seed=seed)
        chain = np.ascontiguousarray(pyoracle.build_mips(l0))
        planes = synth.random_gbuffer(n, seed=seed, discard_frac=0.05)           # [23][n]
        gb = np.ascontiguousarray(planes.T.astype(np.float32))                   # [n][23]
        rng = np.random.default_rng(seed)
        spec_raw = rng.uniform(0.0, 1.0, (n, 4)).astype(np.float32)
        spec_raw[rng.random(n) < 0.3, 1:3] = 0.0                                 # red-only maps: the .rrra rule (trace.fs:210)
        shadow_k = rng.integers(0, 26, n).astype(np.int32)
        p = pyoracle.default_params(V)
        outs = np.zeros((n, 4), np.float32); outo = np.zeros((n, 4), np.float32)
        ss = np.zeros((n, 7), np.int32); so_ = np.zeros((n, 7), np.int32)
        gbu = np.zeros((n, 23), np.float32); disc = np.zeros(n, np.int32)
        cam = np.array(p.camera_pos, np.float32); light = np.array(p.light_dir, np.float32)
        rc = lib.run_pixels(C.byref(p), chain.ctypes.data_as(C.c_void_p), C.c_int(V), C.c_float(p.G),
                            cam.ctypes.data_as(C.c_void_p), light.ctypes.data_as(C.c_void_p), C.c_float(p.ambient_factor),
                            C.c_float(p.shininess), C.c_int(n), gb.ctypes.data_as(C.c_void_p),
                            spec_raw.ctypes.data_as(C.c_void_p), shadow_k.ctypes.data_as(C.c_void_p),
                            outs.ctypes.data_as(C.c_void_p), outo.ctypes.data_as(C.c_void_p),
                            ss.ctypes.data_as(C.c_void_p), so_.ctypes.data_as(C.c_void_p),
                            gbu.ctypes.data_as(C.c_void_p), disc.ctypes.data_as(C.c_void_p))
        assert rc == 0
        live = disc == 0
        # discard: the shader discards exactly the pixels the oracle does not shade (alpha < 0.5, trace.fs:169-172)
        assert np.array_equal(disc == 1, gb[:, 18] < np.float32(0.5))
        assert live.sum() > 0.9 * n
        # executed steps, cone by cone
        same = (ss[live] == so_[live]).all(axis=1)
        print(f"V={V}: {live.sum()} live pixels, step counts differ on {np.count_nonzero(~same)}, steps {so_[live].sum()}")
        assert same.mean() > 0.999, f"step counts differ on {np.count_nonzero(~same)} of {live.sum()} pixels"
        a, b = outs[live][same], outo[live][same]
        rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
        assert rel.max() <= 1e-5, rel.max()
        # (a pixel whose alpha lands within an ulp of MAX_ALPHA may take one step more in one of the two evaluations --
        # the stand-in has no fused multiply-add, the oracle's accumulation has: still within the frame tolerance)
        if (~same).any():
            assert synth.rel_l2(outs[live][~same], outo[live][~same]) <= 1e-3
        assert so_[live].sum() > 20 * live.sum()                                # the march really ran


# ---- the voxelization shaders (S/Voxelization.gs: dominant axis + projection; S/Voxelization.fs: voxel index, PCF,
# stored value) against the oracle's fragment-level restatements (vcto_dominant_axis, vcto_voxel_proj,
# vcto_frag_to_voxel, vcto_pcf25) -------------------------------------------------------------------------------------
REF_VOX_FS = "/root/reference/Voxel_Cone_Tracing_Final/Shader/Voxelization.fs"
REF_VOX_GS = "/root/reference/Voxel_Cone_Tracing_Final/Shader/Voxelization.gs"

VOX_FS_HARNESS = r'''
ShimState g_shim;
}  // namespace glsl
using namespace glsl;
// frag [n][8]: gl_FragCoord x y z, axis, DepthCoord x y z, (pad); albedo [n][4]; out: voxel [n][3], value [n][4]
extern "C" int run_fragments(int V, int S, const float* depth, int n, const float* frag, const float* albedo, int* voxel,
                             float* value) {
    VoxelDimensions = V; ShadowMapSize = S;
    DiffuseTexture.which = 0; ShadowMap.which = 4;
    for (int i = 0; i < n; ++i) {
        const float* f = frag + 8 * i;
        g_shim = ShimState();
        g_shim.shadow_depth = depth; g_shim.shadow_S = S;
        g_shim.diffuse = vec4(albedo[4 * i], albedo[4 * i + 1], albedo[4 * i + 2], albedo[4 * i + 3]);
        gl_FragCoord = vec4(f[0], f[1], f[2], 1.0);
        gs.axis = (int)f[3];
        gs.DepthCoord = vec4(f[4], f[5], f[6], 1.0);          // an orthographic light: w = 1 (vox.vs:18-19)
        gs.TexCoord = vec2(0.25, 0.75);
        shader_main();
        if (g_shim.stores != 1) return 1 + i;
        voxel[3 * i] = g_shim.stored_pos.x; voxel[3 * i + 1] = g_shim.stored_pos.y; voxel[3 * i + 2] = g_shim.stored_pos.z;
        value[4 * i] = g_shim.stored_value.x; value[4 * i + 1] = g_shim.stored_value.y;
        value[4 * i + 2] = g_shim.stored_value.z; value[4 * i + 3] = g_shim.stored_value.w;
    }
    return 0;
}
'''

VOX_GS_HARNESS = r'''
ShimState g_shim;
static int g_emitted;
static float g_out[3][4];
static int g_axis_out;
void EmitVertex() { if (g_emitted < 3) { g_out[g_emitted][0] = gl_Position.x; g_out[g_emitted][1] = gl_Position.y;
                                        g_out[g_emitted][2] = gl_Position.z; g_out[g_emitted][3] = gl_Position.w; }
                    g_axis_out = axis; ++g_emitted; }
void EndPrimitive() {}
void shader_main();
}  // namespace glsl
using namespace glsl;
// tri [n][9] world-space positions; proj [3][16] column-major ProjX / ProjY / ProjZ; out: axis [n], clip [n][3][4]
extern "C" int run_triangles(const float* proj, int n, const float* tri, int* axis_out, float* clip) {
    mat4* P[3] = {&ProjX, &ProjY, &ProjZ};
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 4; ++c) P[a]->c[c] = vec4(proj[16 * a + 4 * c], proj[16 * a + 4 * c + 1], proj[16 * a + 4 * c + 2], proj[16 * a + 4 * c + 3]);
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) gl_in[k].gl_Position = vec4(tri[9 * i + 3 * k], tri[9 * i + 3 * k + 1], tri[9 * i + 3 * k + 2], 1.0);
        g_emitted = 0;
        shader_main();
        if (g_emitted != 3) return 1 + i;
        axis_out[i] = g_axis_out;
        for (int k = 0; k < 3; ++k) for (int c = 0; c < 4; ++c) clip[12 * i + 4 * k + c] = g_out[k][c];
    }
    return 0;
}
'''


def vox_glsl_to_cpp(src, geometry):
    """Mechanical rewrites for the two voxelization stages: layout qualifiers and `flat` dropped, interface blocks ->
    a struct instance (named block) or plain globals (unnamed block), swizzles -> calls, `.r` of a fetch -> `.x`."""
    s = "\n".join(ln for ln in src.splitlines() if not ln.lstrip().startswith("#version"))
    s = re.sub(r"^\s*layout\s*\([^)]*\)\s*(in|out)\s*;", "", s, flags=re.M)           # layout (triangles) in;
    s = re.sub(r"layout\s*\([^)]*\)\s*", "", s)                                        # uniform layout(RGBA8) image3D ...
    s = re.sub(r"\bflat\s+", "", s)
    if geometry:
        s = re.sub(r"^\s*in\s+Vertex\s*\{([^}]*)\}\s*vertices\s*\[\s*\]\s*;", r"struct Vertex {\1} vertices[3];", s, flags=re.M)
        s = re.sub(r"^\s*out\s+Vertex_GS\s*\{([^}]*)\}\s*;", r"\1", s, flags=re.M)  # unnamed block: its members are globals
        s = "GlInArray gl_in; vec4 gl_Position; void EmitVertex(); void EndPrimitive();\n" + s
    else:
        s = re.sub(r"^\s*in\s+Vertex_GS\s*\{([^}]*)\}\s*gs\s*;", r"struct Vertex_GS {\1} gs;", s, flags=re.M)
        s = "vec4 gl_FragCoord;\n" + s
    s = re.sub(r"^\s*(in|out|uniform)\s+", "", s, flags=re.M)
    s = re.sub(r"\.(xyz|xy|rgb)\b", r".\1()", s)
    s = re.sub(r"\)\.r\b", ").x", s)
    s = re.sub(r"\bvoid\s+main\s*\(\s*\)", "void shader_main()", s)
    return s


def _build(tmp_path, name, text):
    cpp = tmp_path / (name + ".cpp")
    cpp.write_text(f'#include "{ROOT}/tests/glsl_shim.h"\nnamespace glsl {{\n' + text)
    so = tmp_path / ("lib" + name + ".so")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-w", "-o", str(so), str(cpp),
                        ORACLE, f"-Wl,-rpath,{os.path.dirname(ORACLE)}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return C.CDLL(str(so))


@pytest.mark.skipif(not os.path.exists(REF_VOX_FS), reason="the reference tree is not present on this box")
def test_oracle_agrees_with_the_reference_voxelization_shaders(tmp_path):
    """S/Voxelization.gs main() (dominant axis, projection by ProjX / ProjY / ProjZ) and S/Voxelization.fs main() (voxel
    index from gl_FragCoord, 25-tap PCF / 25, imageStore of albedo * shadow) run as C++ next to the oracle's fragment-level
    restatements.  Same caveats as above: the shadow sampler under the shader is the oracle's bilinear fetch, the diffuse
    sampler a per-fragment constant (the executed-shader check of these stages is tests/test_ref_gl.py)."""
    from oracle import pyoracle
    lib = pyoracle._lib if hasattr(pyoracle, "_lib") else C.CDLL(ORACLE)
    rng = np.random.default_rng(5)
    # ---- geometry stage ----
    gs = _build(tmp_path, "vox_gs", vox_glsl_to_cpp(open(REF_VOX_GS).read(), True) + VOX_GS_HARNESS)
    G = 150.0
    proj = np.zeros((3, 16), np.float32)
    lib.vcto_voxel_proj.argtypes = [C.c_float, C.c_int, C.c_void_p]
    for a in range(3):
        lib.vcto_voxel_proj(C.c_float(G), a + 1, proj[a].ctypes.data_as(C.c_void_p))
    n = 20000
    tri = rng.uniform(-70.0, 70.0, (n, 9)).astype(np.float32)
    tri[: n // 4, 3:] = tri[: n // 4, :3].repeat(2).reshape(-1, 3, 2).transpose(0, 2, 1).reshape(-1, 6) \
        + rng.uniform(-2.0, 2.0, (n // 4, 6)).astype(np.float32)                    # a quarter small triangles
    # axis-aligned ones: ties between the components of the normal (vox.gs:34-39 resolves them X, then Y, then Z)
    tri[n // 4: n // 4 + 300] = np.round(tri[n // 4: n // 4 + 300] / 8.0) * 8.0
    axis_s = np.zeros(n, np.int32); clip = np.zeros((n, 3, 4), np.float32)
    rc = gs.run_triangles(proj.ctypes.data_as(C.c_void_p), C.c_int(n), tri.ctypes.data_as(C.c_void_p),
                          axis_s.ctypes.data_as(C.c_void_p), clip.ctypes.data_as(C.c_void_p))
    assert rc == 0
    lib.vcto_dominant_axis.argtypes = [C.c_void_p] * 3
    lib.vcto_dominant_axis.restype = C.c_int
    axis_o = np.array([lib.vcto_dominant_axis(t[0:3].ctypes.data_as(C.c_void_p), t[3:6].ctypes.data_as(C.c_void_p),
                                              t[6:9].ctypes.data_as(C.c_void_p)) for t in tri], np.int32)
    degenerate = ~np.isfinite(clip).all(axis=(1, 2))
    assert degenerate.mean() < 0.02
    assert np.array_equal(axis_s[~degenerate], axis_o[~degenerate])
    assert set(np.unique(axis_o)) == {1, 2, 3}
    # the projection the shader applied is the oracle's matrix of that axis
    hom = np.concatenate([tri.reshape(n, 3, 3), np.ones((n, 3, 1), np.float32)], axis=2).astype(np.float64)
    want = np.einsum("nrc,nkc->nkr", proj[axis_o - 1].reshape(n, 4, 4).transpose(0, 2, 1).astype(np.float64), hom)
    ok = ~degenerate
    assert np.abs(clip[ok] - want[ok]).max() <= 1e-4 * max(1.0, np.abs(want[ok]).max())
    # ---- fragment stage ----
    fs = _build(tmp_path, "vox_fs", vox_glsl_to_cpp(open(REF_VOX_FS).read(), False) + VOX_FS_HARNESS)
    lib.vcto_frag_to_voxel.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
    lib.vcto_pcf25.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float]
    lib.vcto_pcf25.restype = C.c_int
    for V, S in ((64, 64), (256, 128)):
        m = 6000
        # a shadow map with structure at every scale: blocks of depth + noise, so that windows straddle edges
        depth = (rng.integers(0, 4, (S // 8, S // 8)).repeat(8, 0).repeat(8, 1) * 0.2 + 0.1
                 + rng.uniform(0.0, 0.02, (S, S))).astype(np.float32)
        frag = np.zeros((m, 8), np.float32)
        frag[:, 0] = rng.integers(0, V, m) + 0.5                               # pixel centres of the V x V voxelization raster
        frag[:, 1] = rng.integers(0, V, m) + 0.5
        frag[:, 2] = rng.uniform(0.0, 1.0, m)
        frag[: m // 20, 2] = rng.integers(0, V, m // 20) / np.float32(V)       # depths on voxel boundaries
        frag[:, 3] = rng.integers(1, 4, m)
        frag[:, 4:6] = rng.uniform(-0.05, 1.05, (m, 2))                        # shadow coordinates, some past the rim (clamp to edge)
        frag[:, 6] = rng.uniform(0.0, 1.0, m)
        albedo = rng.uniform(0.0, 1.0, (m, 4)).astype(np.float32)
        voxel = np.zeros((m, 3), np.int32); value = np.zeros((m, 4), np.float32)
        rc = fs.run_fragments(C.c_int(V), C.c_int(S), depth.ctypes.data_as(C.c_void_p), C.c_int(m),
                              frag.ctypes.data_as(C.c_void_p), albedo.ctypes.data_as(C.c_void_p),
                              voxel.ctypes.data_as(C.c_void_p), value.ctypes.data_as(C.c_void_p))
        assert rc == 0
        wv = np.zeros((m, 3), np.int32); cnt = np.zeros(m, np.int32)
        for i in range(m):
            lib.vcto_frag_to_voxel(V, int(frag[i, 3]), C.c_float(frag[i, 0]), C.c_float(frag[i, 1]), C.c_float(frag[i, 2]),
                                   wv[i].ctypes.data_as(C.c_void_p))
            cnt[i] = lib.vcto_pcf25(depth.ctypes.data_as(C.c_void_p), S, frag[i, 4:7].ctypes.data_as(C.c_void_p), C.c_float(0.002))
        assert np.array_equal(voxel, wv)
        assert 0 < (cnt == 0).sum() and 0 < (cnt == 25).sum() and ((cnt > 0) & (cnt < 25)).sum() > m // 20
        shadow = cnt.astype(np.float32) / np.float32(25.0)
        want_rgb = albedo[:, :3] * shadow[:, None]
        assert np.abs(value[:, :3] - want_rgb).max() <= 1e-6 and (value[:, 3] == 1.0).all()
