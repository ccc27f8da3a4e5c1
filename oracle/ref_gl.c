/*
 * ref_gl.c -- oracle/_ref: the REFERENCE'S OWN SHADERS executed in this container.
 *
 * TEST INFRASTRUCTURE ONLY (like everything under oracle/).  Built into
 * oracle/_ref/libvct_refgl.so by oracle/Makefile (`make -C oracle ref`); only
 * tests/golden/make_ref_golden.py and the `not gpu` tests that re-check the committed
 * fixtures load it.  It is never built or loaded on the GPU box (no /root/reference there).
 *
 * What runs: the seven GLSL files of /root/reference/Voxel_Cone_Tracing_Final/Shader/
 * (Shadow.vs/.fs, Voxelization.vs/.gs/.fs, VoxelConeTracing.vs/.fs), READ AT RUN TIME
 * from that directory and handed to the GL compiler UNMODIFIED -- nothing of them is
 * copied into this repository.  The GL implementation is Mesa's llvmpipe (OpenGL 4.5 core,
 * GLSL 4.50) which this image ships as /usr/lib/x86_64-linux-gnu/dri/swrast_dri.so; the
 * context is created without X / EGL / OSMesa by loading the DRI driver directly
 * (DRI_SWRast + DRI_Core extensions, a do-nothing DRI_SWRastLoader for the "window").
 *
 * What this file itself is: the HOST side of the reference path, restated as plain C GL
 * calls because the reference's host (Voxel_Cone_Tracing.h, Model.h, Mesh.h, Shader.h)
 * needs GLFW, GLEW, glm, assimp and a Windows tool-chain and is unbuildable here.  Every
 * function cites the reference lines whose GL call sequence it follows; matrices arrive
 * from the caller (column-major, like glm::value_ptr).
 *
 *   R = /root/reference/Voxel_Cone_Tracing_Final
 *   refgl_init                R/Shader.h (compile + link), R/Voxel_Cone_Tracing.h:70-72, R/main.cpp:52-58
 *   refgl_volume_*            R/Voxel_Cone_Tracing.h:107-126, :248
 *   refgl_texture_*           R/Model.h:141-181
 *   refgl_mesh_*              R/Mesh.h:49-82 (setup_Mesh), :84-119 (Draw_Mesh)
 *   refgl_shadow_*            R/Voxel_Cone_Tracing.h:79-105, :192-211 (DrawDepthTexture)
 *   refgl_draw_voxel_texture  R/Voxel_Cone_Tracing.h:213-250 (DrawVoxelTexture)
 *   refgl_render              R/Voxel_Cone_Tracing.h:146-190 (Render)
 *   refgl_trace_points        the same program as refgl_render, fed one GL_POINT per G-buffer pixel
 */
#define _GNU_SOURCE
#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GLFUNCS(X)                                                                                  \
    X(PFNGLGETSTRINGPROC, GetString) X(PFNGLGETERRORPROC, GetError) X(PFNGLCREATESHADERPROC, CreateShader) \
    X(PFNGLSHADERSOURCEPROC, ShaderSource) X(PFNGLCOMPILESHADERPROC, CompileShader)                 \
    X(PFNGLGETSHADERIVPROC, GetShaderiv) X(PFNGLGETSHADERINFOLOGPROC, GetShaderInfoLog)             \
    X(PFNGLCREATEPROGRAMPROC, CreateProgram) X(PFNGLATTACHSHADERPROC, AttachShader)                 \
    X(PFNGLLINKPROGRAMPROC, LinkProgram) X(PFNGLGETPROGRAMIVPROC, GetProgramiv)                     \
    X(PFNGLGETPROGRAMINFOLOGPROC, GetProgramInfoLog) X(PFNGLUSEPROGRAMPROC, UseProgram)             \
    X(PFNGLDELETESHADERPROC, DeleteShader)                                                          \
    X(PFNGLGETUNIFORMLOCATIONPROC, GetUniformLocation) X(PFNGLUNIFORM1IPROC, Uniform1i)             \
    X(PFNGLUNIFORM1FPROC, Uniform1f) X(PFNGLUNIFORM2FPROC, Uniform2f) X(PFNGLUNIFORM3FPROC, Uniform3f) \
    X(PFNGLUNIFORMMATRIX4FVPROC, UniformMatrix4fv) X(PFNGLGENTEXTURESPROC, GenTextures)             \
    X(PFNGLBINDTEXTUREPROC, BindTexture) X(PFNGLTEXPARAMETERIPROC, TexParameteri)                   \
    X(PFNGLTEXIMAGE2DPROC, TexImage2D) X(PFNGLTEXIMAGE3DPROC, TexImage3D)                           \
    X(PFNGLGENERATEMIPMAPPROC, GenerateMipmap) X(PFNGLGETTEXIMAGEPROC, GetTexImage)                 \
    X(PFNGLGETTEXLEVELPARAMETERIVPROC, GetTexLevelParameteriv)                                      \
    X(PFNGLACTIVETEXTUREPROC, ActiveTexture) X(PFNGLBINDIMAGETEXTUREPROC, BindImageTexture)         \
    X(PFNGLGENFRAMEBUFFERSPROC, GenFramebuffers) X(PFNGLBINDFRAMEBUFFERPROC, BindFramebuffer)       \
    X(PFNGLFRAMEBUFFERTEXTUREPROC, FramebufferTexture) X(PFNGLDRAWBUFFERPROC, DrawBuffer)           \
    X(PFNGLREADBUFFERPROC, ReadBuffer) X(PFNGLCHECKFRAMEBUFFERSTATUSPROC, CheckFramebufferStatus)   \
    X(PFNGLDELETEFRAMEBUFFERSPROC, DeleteFramebuffers) X(PFNGLDELETETEXTURESPROC, DeleteTextures)   \
    X(PFNGLVIEWPORTPROC, Viewport) X(PFNGLCLEARCOLORPROC, ClearColor) X(PFNGLCLEARPROC, Clear)      \
    X(PFNGLENABLEPROC, Enable) X(PFNGLDISABLEPROC, Disable) X(PFNGLDEPTHFUNCPROC, DepthFunc)        \
    X(PFNGLCULLFACEPROC, CullFace) X(PFNGLGENVERTEXARRAYSPROC, GenVertexArrays)                     \
    X(PFNGLBINDVERTEXARRAYPROC, BindVertexArray) X(PFNGLGENBUFFERSPROC, GenBuffers)                 \
    X(PFNGLBINDBUFFERPROC, BindBuffer) X(PFNGLBUFFERDATAPROC, BufferData)                           \
    X(PFNGLDELETEBUFFERSPROC, DeleteBuffers) X(PFNGLDELETEVERTEXARRAYSPROC, DeleteVertexArrays)     \
    X(PFNGLENABLEVERTEXATTRIBARRAYPROC, EnableVertexAttribArray)                                    \
    X(PFNGLVERTEXATTRIBPOINTERPROC, VertexAttribPointer) X(PFNGLDRAWELEMENTSPROC, DrawElements)     \
    X(PFNGLDRAWARRAYSPROC, DrawArrays) X(PFNGLREADPIXELSPROC, ReadPixels) X(PFNGLFINISHPROC, Finish) \
    X(PFNGLMEMORYBARRIERPROC, MemoryBarrier) X(PFNGLPIXELSTOREIPROC, PixelStorei)                   \
    X(PFNGLPOINTSIZEPROC, PointSize)

#define X(T, N) static T gl##N;
GLFUNCS(X)
#undef X

#define MAX_MESH 64
#define MAX_TEX 64
#define MAX_MESH_TEX 8

typedef struct {
    GLuint vao, vbo, ibo;
    int nidx;
    int ntex;
    GLuint tex[MAX_MESH_TEX];
    int type[MAX_MESH_TEX]; /* 0 texture_diffuse, 1 texture_specular, 2 texture_normal (bound, never named), 3 texture_height */
    int tw[MAX_MESH_TEX], th[MAX_MESH_TEX];
} ref_mesh;

static struct {
    int ready;
    int win_w, win_h;
    void* dri;
    const __DRIcoreExtension* core;
    const __DRIswrastExtension* swrast;
    __DRIscreen* screen;
    __DRIcontext* ctx;
    __DRIdrawable* draw;
    GLuint prog_shadow, prog_voxelize, prog_trace;
    GLuint volume;
    int V;
    GLuint depth_fbo, depth_tex;
    int S;
    ref_mesh mesh[MAX_MESH];
    int nmesh;
    GLuint tex[MAX_TEX];
    int tex_w[MAX_TEX], tex_h[MAX_TEX];
    int ntex;
    char log[16384];
} R;

static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(R.log, sizeof R.log, fmt, ap);
    va_end(ap);
    return -1;
}

static int gl_check(const char* where) {
    GLenum e = glGetError();
    if (e != GL_NO_ERROR) return fail("%s: GL error 0x%04x", where, e);
    return 0;
}

/* ---- the "window system": a drawable of fixed size whose contents go nowhere ---- */
static void ld_get_drawable_info(__DRIdrawable* d, int* x, int* y, int* w, int* h, void* priv) {
    (void)d; (void)priv;
    *x = 0; *y = 0; *w = R.win_w; *h = R.win_h;
}
static void ld_put_image(__DRIdrawable* d, int op, int x, int y, int w, int h, char* data, void* priv) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)priv;
}
static void ld_get_image(__DRIdrawable* d, int x, int y, int w, int h, char* data, void* priv) {
    (void)d; (void)x; (void)y; (void)priv;
    memset(data, 0, (size_t)w * h * 4);
}
static const __DRIswrastLoaderExtension loader_ext = {
    .base = {__DRI_SWRAST_LOADER, 1},
    .getDrawableInfo = ld_get_drawable_info,
    .putImage = ld_put_image,
    .getImage = ld_get_image,
};
static const __DRIextension* loader_exts[] = {&loader_ext.base, NULL};

static char* read_file(const char* dir, const char* name) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fail("cannot open %s", path); return NULL; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* s = (char*)malloc((size_t)n + 1);
    if (fread(s, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(s); fail("short read %s", path); return NULL; }
    s[n] = 0;
    fclose(f);
    return s;
}

/* R/Shader.h: glCreateShader / glShaderSource / glCompileShader per stage, glAttachShader, glLinkProgram.
 * The source string is the file's bytes, untouched. */
static GLuint build_program(const char* dir, const char* vs, const char* fs, const char* gs) {
    const char* names[3] = {vs, fs, gs};
    GLenum kinds[3] = {GL_VERTEX_SHADER, GL_FRAGMENT_SHADER, GL_GEOMETRY_SHADER};
    GLuint prog = glCreateProgram();
    for (int i = 0; i < 3; ++i) {
        if (!names[i]) continue;
        char* src = read_file(dir, names[i]);
        if (!src) return 0;
        GLuint sh = glCreateShader(kinds[i]);
        const GLchar* p = src;
        glShaderSource(sh, 1, &p, NULL);
        glCompileShader(sh);
        free(src);
        GLint ok = 0;
        glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
        if (!ok) {
            char msg[8192];
            glGetShaderInfoLog(sh, sizeof msg, NULL, msg);
            fail("compile %s: %s", names[i], msg);
            return 0;
        }
        glAttachShader(prog, sh);
        glDeleteShader(sh);
    }
    glLinkProgram(prog);
    GLint ok = 0;
    glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) {
        char msg[8192];
        glGetProgramInfoLog(prog, sizeof msg, NULL, msg);
        fail("link %s+%s: %s", vs, fs, msg);
        return 0;
    }
    return prog;
}

const char* refgl_log(void) { return R.log; }

const char* refgl_string(int which) {
    if (!R.ready) return "";
    GLenum n = which == 0 ? GL_VERSION : which == 1 ? GL_RENDERER : which == 2 ? GL_SHADING_LANGUAGE_VERSION : GL_VENDOR;
    return (const char*)glGetString(n);
}

/* Creates the GL 4.3 core context and builds the reference's three programs from shader_dir.
 * win_w x win_h: size of the default framebuffer (the reference's GLFW window, R/main.cpp:30-38); the
 * voxelization pass draws into it (VCT.h:209,218), so it must be at least V x V. */
int refgl_init(const char* shader_dir, int win_w, int win_h) {
    if (R.ready) return 0;
    R.win_w = win_w;
    R.win_h = win_h;
    const char* drv = getenv("REFGL_DRI_DRIVER");
    if (!drv) drv = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";
    R.dri = dlopen(drv, RTLD_NOW | RTLD_GLOBAL);
    if (!R.dri) return fail("dlopen %s: %s", drv, dlerror());
    const __DRIextension** (*get_exts)(void) =
        (const __DRIextension** (*)(void))dlsym(R.dri, __DRI_DRIVER_GET_EXTENSIONS "_swrast");
    if (!get_exts) return fail("no %s_swrast in %s", __DRI_DRIVER_GET_EXTENSIONS, drv);
    const __DRIextension** exts = get_exts();
    for (int i = 0; exts[i]; ++i) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) R.core = (const __DRIcoreExtension*)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) R.swrast = (const __DRIswrastExtension*)exts[i];
    }
    if (!R.core || !R.swrast || R.swrast->base.version < 4) return fail("DRI_Core / DRI_SWRast v4 missing");
    const __DRIconfig** configs = NULL;
    R.screen = R.swrast->createNewScreen2(0, loader_exts, exts, &configs, NULL);
    if (!R.screen || !configs || !configs[0]) return fail("createNewScreen2 failed");
    /* a config with a depth buffer (the reference's window has one: main.cpp:55 enables the depth test) */
    const __DRIconfig* cfg = configs[0];
    for (int i = 0; configs[i]; ++i) {
        unsigned depth = 0, rgba = 0, dbl = 0;
        R.core->getConfigAttrib(configs[i], __DRI_ATTRIB_DEPTH_SIZE, &depth);
        R.core->getConfigAttrib(configs[i], __DRI_ATTRIB_BUFFER_SIZE, &rgba);
        R.core->getConfigAttrib(configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &dbl);
        if (depth == 24 && rgba == 32 && dbl) { cfg = configs[i]; break; }
    }
    uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 3};
    unsigned err = 0;
    R.ctx = R.swrast->createContextAttribs(R.screen, __DRI_API_OPENGL_CORE, cfg, NULL, 2, attribs, &err, NULL);
    if (!R.ctx) return fail("createContextAttribs(core 4.3) failed, error %u", err);
    R.draw = R.swrast->createNewDrawable(R.screen, cfg, NULL);
    if (!R.draw) return fail("createNewDrawable failed");
    if (!R.core->bindContext(R.ctx, R.draw, R.draw)) return fail("bindContext failed");

    void* glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) return fail("dlopen libglapi.so.0: %s", dlerror());
    void* (*gpa)(const char*) = (void* (*)(const char*))dlsym(glapi, "_glapi_get_proc_address");
    if (!gpa) return fail("no _glapi_get_proc_address");
#define X(T, N)                                               \
    gl##N = (T)gpa("gl" #N);                                  \
    if (!gl##N) return fail("GL entry point gl" #N " missing");
    GLFUNCS(X)
#undef X
    R.ready = 1;

    /* VCT.h:70-72 */
    R.prog_voxelize = build_program(shader_dir, "Voxelization.vs", "Voxelization.fs", "Voxelization.gs");
    if (!R.prog_voxelize) return -1;
    R.prog_shadow = build_program(shader_dir, "Shadow.vs", "Shadow.fs", NULL);
    if (!R.prog_shadow) return -1;
    R.prog_trace = build_program(shader_dir, "VoxelConeTracing.vs", "VoxelConeTracing.fs", NULL);
    if (!R.prog_trace) return -1;

    /* main.cpp:55-58 */
    glEnable(GL_DEPTH_TEST);
    glDepthFunc(GL_LESS);
    glEnable(GL_CULL_FACE);
    glCullFace(GL_BACK);
    glPixelStorei(GL_PACK_ALIGNMENT, 1);
    glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    return gl_check("refgl_init");
}

/* ---------------------------------------------------------------- volume (VCT.h:107-126) ---- */
int refgl_volume_create(int V) {
    if (!R.ready) return fail("not initialised");
    if (R.volume) glDeleteTextures(1, &R.volume);
    R.V = V;
    glGenTextures(1, &R.volume);                                                    /* :109 */
    glBindTexture(GL_TEXTURE_3D, R.volume);                                         /* :110 */
    glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_MIN_FILTER, GL_LINEAR_MIPMAP_LINEAR); /* :111 */
    glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);               /* :112 */
    size_t n = (size_t)4 * V * V * V;
    GLubyte* data = (GLubyte*)calloc(n, 1);                                         /* :114-116 */
    glTexImage3D(GL_TEXTURE_3D, 0, GL_RGBA8, V, V, V, 0, GL_RGBA, GL_UNSIGNED_BYTE, data); /* :118 */
    free(data);
    glGenerateMipmap(GL_TEXTURE_3D);                                                /* :125 */
    return gl_check("refgl_volume_create");
}

/* Replaces one level's texels (fixtures that isolate the sampler from the mip builder upload every level). */
int refgl_volume_set_level(int level, const uint8_t* texels) {
    if (!R.volume) return fail("no volume");
    int N = R.V >> level;
    if (N < 1) return fail("level %d out of range", level);
    glBindTexture(GL_TEXTURE_3D, R.volume);
    glTexImage3D(GL_TEXTURE_3D, level, GL_RGBA8, N, N, N, 0, GL_RGBA, GL_UNSIGNED_BYTE, texels);
    return gl_check("refgl_volume_set_level");
}

int refgl_volume_generate_mipmap(void) { /* VCT.h:246-248 */
    if (!R.volume) return fail("no volume");
    glActiveTexture(GL_TEXTURE6);
    glBindTexture(GL_TEXTURE_3D, R.volume);
    glGenerateMipmap(GL_TEXTURE_3D);
    glActiveTexture(GL_TEXTURE0);
    return gl_check("refgl_volume_generate_mipmap");
}

int refgl_volume_get_level(int level, uint8_t* out) {
    if (!R.volume) return fail("no volume");
    glBindTexture(GL_TEXTURE_3D, R.volume);
    glGetTexImage(GL_TEXTURE_3D, level, GL_RGBA, GL_UNSIGNED_BYTE, out);
    return gl_check("refgl_volume_get_level");
}

/* wrap: 0 leaves the GL default (GL_REPEAT -- the reference never sets it), 1 sets CLAMP_TO_EDGE (the
 * build's switchable option; no reference code) */
int refgl_volume_set_wrap(int clamp) {
    if (!R.volume) return fail("no volume");
    glBindTexture(GL_TEXTURE_3D, R.volume);
    GLint m = clamp ? GL_CLAMP_TO_EDGE : GL_REPEAT;
    glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_WRAP_S, m);
    glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_WRAP_T, m);
    glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_WRAP_R, m);
    return gl_check("refgl_volume_set_wrap");
}

/* ------------------------------------------------------------- textures (Model.h:141-181) ---- */
/* channels 1/3/4 like stb_image's n (Model.h:160-165).  Returns a handle >= 0. */
int refgl_texture_create(int w, int h, int channels, const uint8_t* data) {
    if (!R.ready) return fail("not initialised");
    if (R.ntex >= MAX_TEX) return fail("too many textures");
    GLenum format = channels == 1 ? GL_RED : channels == 3 ? GL_RGB : GL_RGBA;
    GLuint id;
    glGenTextures(1, &id);                                                          /* :148 */
    glBindTexture(GL_TEXTURE_2D, id);                                               /* :167 */
    glTexImage2D(GL_TEXTURE_2D, 0, format, w, h, 0, format, GL_UNSIGNED_BYTE, data);/* :168 */
    glGenerateMipmap(GL_TEXTURE_2D);                                                /* :169 */
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_REPEAT);                   /* :171 */
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_REPEAT);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR_MIPMAP_LINEAR);
    if (gl_check("refgl_texture_create")) return -1;
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR_MIPMAP_LINEAR); /* :174 -- GL_INVALID_ENUM, state unchanged */
    GLenum e = glGetError();
    if (e != GL_INVALID_ENUM) return fail("Model.h:174 expected GL_INVALID_ENUM, got 0x%04x", e);
    R.tex[R.ntex] = id;
    R.tex_w[R.ntex] = w;
    R.tex_h[R.ntex] = h;
    return R.ntex++;
}

/* fp32 RGBA, NEAREST, no mip chain: the per-pixel material tables of refgl_trace_points (NOT a reference
 * texture; it only carries G-buffer values into the reference's texture() calls unfiltered). */
int refgl_texture_create_f32(int w, int h, const float* rgba) {
    if (!R.ready) return fail("not initialised");
    if (R.ntex >= MAX_TEX) return fail("too many textures");
    GLuint id;
    glGenTextures(1, &id);
    glBindTexture(GL_TEXTURE_2D, id);
    glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, w, h, 0, GL_RGBA, GL_FLOAT, rgba);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    if (gl_check("refgl_texture_create_f32")) return -1;
    R.tex[R.ntex] = id;
    R.tex_w[R.ntex] = w;
    R.tex_h[R.ntex] = h;
    return R.ntex++;
}

/* Reads one level of a texture's glGenerateMipmap chain back as RGBA8 (pins the 2-D box filter). */
int refgl_texture_get_level(int tex, int level, uint8_t* out, int* w, int* h) {
    if (tex < 0 || tex >= R.ntex) return fail("bad texture handle");
    glBindTexture(GL_TEXTURE_2D, R.tex[tex]);
    GLint lw = 0, lh = 0;
    glGetTexLevelParameteriv(GL_TEXTURE_2D, level, GL_TEXTURE_WIDTH, &lw);
    glGetTexLevelParameteriv(GL_TEXTURE_2D, level, GL_TEXTURE_HEIGHT, &lh);
    if (w) *w = lw;
    if (h) *h = lh;
    if (out && lw > 0 && lh > 0) glGetTexImage(GL_TEXTURE_2D, level, GL_RGBA, GL_UNSIGNED_BYTE, out);
    return gl_check("refgl_texture_get_level");
}

/* ---------------------------------------------------------------- meshes (Mesh.h:49-82) ---- */
/* verts: [nvert][14] = Position(3) Normal(3) TexCoords(2) Tangents(3) Bi_Tangents(3) -- struct Vertex, Mesh.h:12-19.
 * tex_handles/types: the mesh's `textures` vector in the order Model.h:126-136 builds it
 * (type 0 texture_diffuse, 1 texture_specular, 2 texture_normal, 3 texture_height). */
int refgl_mesh_create(const float* verts, int nvert, const uint32_t* indices, int nidx, const int* tex_handles,
                      const int* tex_types, int ntex) {
    if (!R.ready) return fail("not initialised");
    if (R.nmesh >= MAX_MESH) return fail("too many meshes");
    if (ntex > MAX_MESH_TEX) return fail("too many mesh textures");
    ref_mesh* m = &R.mesh[R.nmesh];
    const GLsizei stride = 14 * sizeof(float);
    glGenVertexArrays(1, &m->vao);
    glGenBuffers(1, &m->vbo);
    glGenBuffers(1, &m->ibo);
    glBindVertexArray(m->vao);
    glBindBuffer(GL_ARRAY_BUFFER, m->vbo);
    glBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)nvert * stride, verts, GL_STATIC_DRAW);
    glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, m->ibo);
    glBufferData(GL_ELEMENT_ARRAY_BUFFER, (GLsizeiptr)nidx * sizeof(uint32_t), indices, GL_STATIC_DRAW);
    glEnableVertexAttribArray(0);
    glVertexAttribPointer(0, 3, GL_FLOAT, GL_FALSE, stride, (void*)0);
    glEnableVertexAttribArray(1);
    glVertexAttribPointer(1, 3, GL_FLOAT, GL_FALSE, stride, (void*)(3 * sizeof(float)));
    glEnableVertexAttribArray(2);
    glVertexAttribPointer(2, 2, GL_FLOAT, GL_FALSE, stride, (void*)(6 * sizeof(float)));
    glEnableVertexAttribArray(3);
    glVertexAttribPointer(3, 3, GL_FLOAT, GL_FALSE, stride, (void*)(8 * sizeof(float)));
    glEnableVertexAttribArray(4);
    glVertexAttribPointer(4, 3, GL_FLOAT, GL_FALSE, stride, (void*)(11 * sizeof(float)));
    glBindVertexArray(0);
    m->nidx = nidx;
    m->ntex = ntex;
    for (int i = 0; i < ntex; ++i) {
        if (tex_handles[i] < 0 || tex_handles[i] >= R.ntex) return fail("bad texture handle");
        m->tex[i] = R.tex[tex_handles[i]];
        m->type[i] = tex_types[i];
        m->tw[i] = R.tex_w[tex_handles[i]];
        m->th[i] = R.tex_h[tex_handles[i]];
    }
    if (gl_check("refgl_mesh_create")) return -1;
    return R.nmesh++;
}

/* Mesh::Draw_Mesh (Mesh.h:84-119), GL_TRIANGLES from the index buffer. */
static void draw_mesh(const ref_mesh* m, GLuint prog, GLenum mode) {
    glUniform1f(glGetUniformLocation(prog, "Shininess"), 20.0f);
    glUniform1f(glGetUniformLocation(prog, "Opacity"), 1.0f);
    for (int i = 0; i < m->ntex; ++i) {
        glActiveTexture(GL_TEXTURE0 + i);
        if (m->type[i] == 0) {
            glUniform1i(glGetUniformLocation(prog, "DiffuseTexture"), i);
            glUniform2f(glGetUniformLocation(prog, "DiffuseTextureSize"), (float)m->tw[i], (float)m->th[i]);
        } else if (m->type[i] == 1) {
            glUniform1i(glGetUniformLocation(prog, "SpecularTexture"), i);
            glUniform2f(glGetUniformLocation(prog, "SpecularTextureSize"), (float)m->tw[i], (float)m->th[i]);
        } else if (m->type[i] == 3) {
            glUniform1i(glGetUniformLocation(prog, "HeightTexture"), i);
            glUniform2f(glGetUniformLocation(prog, "HeightTextureSize"), (float)m->tw[i], (float)m->th[i]);
        }
        glBindTexture(GL_TEXTURE_2D, m->tex[i]);
    }
    glBindVertexArray(m->vao);
    if (mode == GL_TRIANGLES)
        glDrawElements(GL_TRIANGLES, m->nidx, GL_UNSIGNED_INT, 0);
    glBindVertexArray(0);
    glActiveTexture(GL_TEXTURE0);
    /* uniforms the programs do not declare give location -1: glUniform* on -1 is a silent no-op */
}

static void set_mat4(GLuint prog, const char* name, const float* m) {
    glUniformMatrix4fv(glGetUniformLocation(prog, name), 1, GL_FALSE, m);
}

/* ---------------------------------------------------- shadow map (VCT.h:79-105, 192-211) ---- */
int refgl_shadow_create(int S) {
    if (!R.ready) return fail("not initialised");
    if (R.depth_tex) glDeleteTextures(1, &R.depth_tex);
    if (R.depth_fbo) glDeleteFramebuffers(1, &R.depth_fbo);
    R.S = S;
    glGenFramebuffers(1, &R.depth_fbo);                                   /* :79-80 */
    glBindFramebuffer(GL_FRAMEBUFFER, R.depth_fbo);
    glGenTextures(1, &R.depth_tex);                                       /* :88-91 */
    glBindTexture(GL_TEXTURE_2D, R.depth_tex);
    glTexImage2D(GL_TEXTURE_2D, 0, GL_DEPTH_COMPONENT24, S, S, 0, GL_DEPTH_COMPONENT, GL_FLOAT, 0);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);     /* :93-96 */
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    glFramebufferTexture(GL_FRAMEBUFFER, GL_DEPTH_ATTACHMENT, R.depth_tex, 0); /* :98 */
    glDrawBuffer(GL_NONE);                                                /* :99 */
    if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) return fail("depth FBO incomplete");
    glBindFramebuffer(GL_FRAMEBUFFER, 0);                                 /* :136 */
    return gl_check("refgl_shadow_create");
}

/* DrawDepthTexture.  depth_mvp = DepthViewProjectionMatrix * mMat (VCT.h:204-205). */
int refgl_draw_depth_texture(const float* depth_mvp, const int* meshes, int nmesh) {
    if (!R.depth_fbo) return fail("no shadow map");
    glEnable(GL_CULL_FACE);                                               /* :194-195 */
    glEnable(GL_DEPTH_TEST);
    glBindFramebuffer(GL_FRAMEBUFFER, R.depth_fbo);                       /* :197-200 */
    glViewport(0, 0, R.S, R.S);
    glClearColor(0, 0, 0, 1);
    glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);
    glUseProgram(R.prog_shadow);                                          /* :202 */
    set_mat4(R.prog_shadow, "DepthModelViewProjectionMatrix", depth_mvp); /* :205 */
    for (int i = 0; i < nmesh; ++i) {                                     /* :207 */
        if (meshes[i] < 0 || meshes[i] >= R.nmesh) return fail("bad mesh handle");
        draw_mesh(&R.mesh[meshes[i]], R.prog_shadow, GL_TRIANGLES);
    }
    glBindFramebuffer(GL_FRAMEBUFFER, 0);                                 /* :209-210 */
    glViewport(0, 0, R.win_w, R.win_h);
    glFinish();
    return gl_check("refgl_draw_depth_texture");
}

int refgl_shadow_get(float* depth) {
    if (!R.depth_tex) return fail("no shadow map");
    glBindTexture(GL_TEXTURE_2D, R.depth_tex);
    glGetTexImage(GL_TEXTURE_2D, 0, GL_DEPTH_COMPONENT, GL_FLOAT, depth);
    return gl_check("refgl_shadow_get");
}

/* Replaces the shadow map's contents (fixtures that isolate the PCF from the depth raster). */
int refgl_shadow_set(const float* depth) {
    if (!R.depth_tex) return fail("no shadow map");
    glBindTexture(GL_TEXTURE_2D, R.depth_tex);
    glTexImage2D(GL_TEXTURE_2D, 0, GL_DEPTH_COMPONENT24, R.S, R.S, 0, GL_DEPTH_COMPONENT, GL_FLOAT, depth);
    return gl_check("refgl_shadow_set");
}

/* ------------------------------------------------ DrawVoxelTexture (VCT.h:213-250) ---- */
int refgl_draw_voxel_texture(const float* model, const float* depth_mvp, const float* projx, const float* projy,
                             const float* projz, const int* meshes, int nmesh, int generate_mipmap) {
    if (!R.volume) return fail("no volume");
    if (!R.depth_tex) return fail("no shadow map");
    if (R.V > R.win_w || R.V > R.win_h) return fail("window smaller than the voxel grid (VCT.h:218 draws into it)");
    GLuint p = R.prog_voxelize;
    glBindFramebuffer(GL_FRAMEBUFFER, 0);
    glDisable(GL_CULL_FACE);                                              /* :215-216 */
    glDisable(GL_DEPTH_TEST);
    glViewport(0, 0, R.V, R.V);                                           /* :218-220 */
    glClearColor(1.0f, 1.0f, 1.0f, 1.0f);
    glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);
    glUseProgram(p);                                                      /* :222 */
    glUniform1i(glGetUniformLocation(p, "VoxelDimensions"), R.V);         /* :224 */
    set_mat4(p, "ProjX", projx);                                          /* :226-228 */
    set_mat4(p, "ProjY", projy);
    set_mat4(p, "ProjZ", projz);
    glActiveTexture(GL_TEXTURE0 + 5);                                     /* :230-232 */
    glBindTexture(GL_TEXTURE_2D, R.depth_tex);
    glUniform1i(glGetUniformLocation(p, "ShadowMap"), 5);
    glBindImageTexture(6, R.volume, 0, GL_TRUE, 0, GL_WRITE_ONLY, GL_RGBA8); /* :234-235 */
    glUniform1i(glGetUniformLocation(p, "VoxelTexture"), 6);
    set_mat4(p, "ModelMatrix", model);                                    /* :240-243 */
    set_mat4(p, "DepthModelViewProjectionMatrix", depth_mvp);
    glUniform1i(glGetUniformLocation(p, "ShadowMapSize"), R.S);
    for (int i = 0; i < nmesh; ++i) {                                     /* :245 */
        if (meshes[i] < 0 || meshes[i] >= R.nmesh) return fail("bad mesh handle");
        draw_mesh(&R.mesh[meshes[i]], p, GL_TRIANGLES);
    }
    glMemoryBarrier(GL_ALL_BARRIER_BITS); /* the reference issues none; llvmpipe's stores are already visible */
    if (generate_mipmap) {
        glActiveTexture(GL_TEXTURE6);                                     /* :246-248 */
        glBindTexture(GL_TEXTURE_3D, R.volume);
        glGenerateMipmap(GL_TEXTURE_3D);
    }
    glViewport(0, 0, R.win_w, R.win_h);                                   /* :249 */
    glActiveTexture(GL_TEXTURE0);
    glEnable(GL_CULL_FACE);
    glEnable(GL_DEPTH_TEST);
    glFinish();
    return gl_check("refgl_draw_voxel_texture");
}

/* ---------------------------------------------------------------- Render (VCT.h:146-190) ---- */
typedef struct refgl_frame_params {
    float camera_pos[3];     /* :167 */
    float light_dir[3];      /* :168 */
    float grid_world_size;   /* :169 */
    int32_t voxel_dim;       /* :170 */
    float ambient_factor;    /* :171 */
    float model[16];         /* :183-184 */
    float model_view[16];    /* :185 */
    float projection[16];    /* :186 */
    float depth_mvp[16];     /* :187 */
} refgl_frame_params;

static void set_frame_uniforms(GLuint p, const refgl_frame_params* fp) {
    glUseProgram(p);                                                                          /* :164 */
    glUniform3f(glGetUniformLocation(p, "CameraPosition"), fp->camera_pos[0], fp->camera_pos[1], fp->camera_pos[2]);
    glUniform3f(glGetUniformLocation(p, "LightDirection"), fp->light_dir[0], fp->light_dir[1], fp->light_dir[2]);
    glUniform1f(glGetUniformLocation(p, "VoxelGridWorldSize"), fp->grid_world_size);
    glUniform1i(glGetUniformLocation(p, "VoxelDimensions"), fp->voxel_dim);
    glUniform1f(glGetUniformLocation(p, "ambientFactor"), fp->ambient_factor);
    glUniform1i(glGetUniformLocation(p, "ShadowMapSize"), R.S);                               /* :172 */
    glActiveTexture(GL_TEXTURE0 + 5);                                                         /* :174-176 */
    glBindTexture(GL_TEXTURE_2D, R.depth_tex);
    glUniform1i(glGetUniformLocation(p, "ShadowMap"), 5);
    glActiveTexture(GL_TEXTURE0 + 6);                                                         /* :178-180 */
    glBindTexture(GL_TEXTURE_3D, R.volume);
    glUniform1i(glGetUniformLocation(p, "VoxelTexture"), 6);
    set_mat4(p, "ModelMatrix", fp->model);                                                    /* :184-187 */
    set_mat4(p, "ModelViewMatrix", fp->model_view);
    set_mat4(p, "ProjectionMatrix", fp->projection);
    set_mat4(p, "DepthModelViewProjectionMatrix", fp->depth_mvp);
}

/* An RGBA32F + DEPTH24 framebuffer in place of the reference's RGBA8 window, so that the shader's output can be
 * read unquantised (the north star's output buffer is RGBA16F for the same reason). */
static int make_fbo(int W, int H, GLuint* fbo, GLuint* col, GLuint* dep) {
    glGenFramebuffers(1, fbo);
    glBindFramebuffer(GL_FRAMEBUFFER, *fbo);
    glGenTextures(1, col);
    glBindTexture(GL_TEXTURE_2D, *col);
    glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, W, H, 0, GL_RGBA, GL_FLOAT, 0);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glFramebufferTexture(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, *col, 0);
    glGenTextures(1, dep);
    glBindTexture(GL_TEXTURE_2D, *dep);
    glTexImage2D(GL_TEXTURE_2D, 0, GL_DEPTH_COMPONENT24, W, H, 0, GL_DEPTH_COMPONENT, GL_FLOAT, 0);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glFramebufferTexture(GL_FRAMEBUFFER, GL_DEPTH_ATTACHMENT, *dep, 0);
    glDrawBuffer(GL_COLOR_ATTACHMENT0);
    glReadBuffer(GL_COLOR_ATTACHMENT0);
    if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) return fail("frame FBO incomplete");
    return 0;
}

/* out_rgba [H][W][4] fp32, row 0 = bottom row of the GL window; out_depth [H][W] optional. */
int refgl_render(int W, int H, const refgl_frame_params* fp, const int* meshes, int nmesh, float* out_rgba,
                 float* out_depth) {
    if (!R.volume || !R.depth_tex) return fail("volume / shadow map missing");
    GLuint fbo, col, dep;
    if (make_fbo(W, H, &fbo, &col, &dep)) return -1;
    glEnable(GL_CULL_FACE);                                               /* :150-151 */
    glEnable(GL_DEPTH_TEST);
    glViewport(0, 0, W, H);                                               /* :154 */
    if (fp->ambient_factor < 0.5f)                                        /* :156-159 */
        glClearColor(0.5f, 0.5f, 0.5f, 1.0f);
    else
        glClearColor(1.0f, 1.0f, 1.0f, 1.0f);
    glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);                   /* main.cpp:81 */
    set_frame_uniforms(R.prog_trace, fp);
    for (int i = 0; i < nmesh; ++i) {                                     /* :189 */
        if (meshes[i] < 0 || meshes[i] >= R.nmesh) return fail("bad mesh handle");
        draw_mesh(&R.mesh[meshes[i]], R.prog_trace, GL_TRIANGLES);
    }
    glFinish();
    glReadPixels(0, 0, W, H, GL_RGBA, GL_FLOAT, out_rgba);
    if (out_depth) glReadPixels(0, 0, W, H, GL_DEPTH_COMPONENT, GL_FLOAT, out_depth);
    glBindFramebuffer(GL_FRAMEBUFFER, 0);
    glDeleteFramebuffers(1, &fbo);
    glDeleteTextures(1, &col);
    glDeleteTextures(1, &dep);
    glViewport(0, 0, R.win_w, R.win_h);
    return gl_check("refgl_render");
}

/* The reference's VoxelConeTracing program fed one GL_POINT per G-buffer pixel, so that arbitrary (random, discarded,
 * degenerate) per-pixel inputs reach the UNMODIFIED fragment shader:
 *   attributes  Position = P, Normal / Tangent / BiTangent = the raw world vectors, TexCoord = the pixel's own texel
 *               centre of the W x H material tables; ModelMatrix = identity, so the vertex shader's varyings are
 *               exactly the G-buffer fields (trace.vs:27-34);
 *   placement   ProjectionMatrix = identity and a per-point ModelViewMatrix = translate(ndc(pixel centre) - P): the
 *               only quantity that is not a reference input, and it feeds gl_Position alone (trace.vs:25);
 *   materials   DiffuseTexture / SpecularTexture / HeightTexture = W x H fp32 NEAREST tables (albedo rgba, spec rgba,
 *               height in .r) made with refgl_texture_create_f32; HeightTextureSize = (W, H);
 *   shadow      the ShadowMap + DepthModelViewProjectionMatrix of fp (PCF runs as written, trace.fs:132-163).
 * verts: [npix][14] in struct Vertex order (TexCoords ignored: the pixel's own table coordinate is used). */
int refgl_trace_points(int W, int H, const refgl_frame_params* fp, const float* verts, int tex_albedo, int tex_spec,
                       int tex_height, float* out_rgba) {
    if (!R.volume || !R.depth_tex) return fail("volume / shadow map missing");
    if (tex_albedo < 0 || tex_albedo >= R.ntex || tex_spec < 0 || tex_spec >= R.ntex || tex_height < 0 ||
        tex_height >= R.ntex)
        return fail("bad texture handle");
    int npix = W * H;
    float* v = (float*)malloc((size_t)npix * 14 * sizeof(float));
    memcpy(v, verts, (size_t)npix * 14 * sizeof(float));
    for (int i = 0; i < npix; ++i) {
        v[i * 14 + 6] = ((float)(i % W) + 0.5f) / (float)W;
        v[i * 14 + 7] = ((float)(i / W) + 0.5f) / (float)H;
    }
    GLuint fbo, col, dep, vao, vbo;
    if (make_fbo(W, H, &fbo, &col, &dep)) { free(v); return -1; }
    const GLsizei stride = 14 * sizeof(float);
    glGenVertexArrays(1, &vao);
    glGenBuffers(1, &vbo);
    glBindVertexArray(vao);
    glBindBuffer(GL_ARRAY_BUFFER, vbo);
    glBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)npix * stride, v, GL_STATIC_DRAW);
    const int offs[5] = {0, 3, 6, 8, 11}, comps[5] = {3, 3, 2, 3, 3};
    for (int a = 0; a < 5; ++a) {
        glEnableVertexAttribArray(a);
        glVertexAttribPointer(a, comps[a], GL_FLOAT, GL_FALSE, stride, (void*)(offs[a] * sizeof(float)));
    }
    glDisable(GL_CULL_FACE);
    glDisable(GL_DEPTH_TEST);
    glViewport(0, 0, W, H);
    if (fp->ambient_factor < 0.5f)
        glClearColor(0.5f, 0.5f, 0.5f, 1.0f);
    else
        glClearColor(1.0f, 1.0f, 1.0f, 1.0f);
    glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);
    set_frame_uniforms(R.prog_trace, fp);
    GLuint p = R.prog_trace;
    glUniform1f(glGetUniformLocation(p, "Shininess"), 20.0f);             /* Mesh.h:86-87 */
    glUniform1f(glGetUniformLocation(p, "Opacity"), 1.0f);
    glActiveTexture(GL_TEXTURE0);
    glBindTexture(GL_TEXTURE_2D, R.tex[tex_albedo]);
    glUniform1i(glGetUniformLocation(p, "DiffuseTexture"), 0);
    glActiveTexture(GL_TEXTURE1);
    glBindTexture(GL_TEXTURE_2D, R.tex[tex_spec]);
    glUniform1i(glGetUniformLocation(p, "SpecularTexture"), 1);
    glActiveTexture(GL_TEXTURE2);
    glBindTexture(GL_TEXTURE_2D, R.tex[tex_height]);
    glUniform1i(glGetUniformLocation(p, "HeightTexture"), 2);
    glUniform2f(glGetUniformLocation(p, "HeightTextureSize"), (float)R.tex_w[tex_height], (float)R.tex_h[tex_height]);
    glActiveTexture(GL_TEXTURE0);
    glPointSize(1.0f);
    GLint loc_mv = glGetUniformLocation(p, "ModelViewMatrix");
    for (int i = 0; i < npix; ++i) {
        const float* P = &v[i * 14];
        float ndc_x = (2.0f * ((float)(i % W) + 0.5f)) / (float)W - 1.0f;
        float ndc_y = (2.0f * ((float)(i / W) + 0.5f)) / (float)H - 1.0f;
        float mv[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, ndc_x - P[0], ndc_y - P[1], 0, 1};
        glUniformMatrix4fv(loc_mv, 1, GL_FALSE, mv);
        glDrawArrays(GL_POINTS, i, 1);
    }
    glFinish();
    glReadPixels(0, 0, W, H, GL_RGBA, GL_FLOAT, out_rgba);
    glBindVertexArray(0);
    glBindFramebuffer(GL_FRAMEBUFFER, 0);
    glDeleteFramebuffers(1, &fbo);
    glDeleteTextures(1, &col);
    glDeleteTextures(1, &dep);
    glDeleteBuffers(1, &vbo);
    glDeleteVertexArrays(1, &vao);
    glEnable(GL_CULL_FACE);
    glEnable(GL_DEPTH_TEST);
    glViewport(0, 0, R.win_w, R.win_h);
    free(v);
    return gl_check("refgl_trace_points");
}
