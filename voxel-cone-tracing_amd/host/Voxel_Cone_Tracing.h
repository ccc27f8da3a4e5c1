// Voxel_Cone_Tracing.h -- call-compatible facade over the MI355X C ABI (include/vct.h).
//
// Stands where the reference's header-only orchestrator stands (R/Voxel_Cone_Tracing.h:11-252): a
// caller written against the reference -- `camera` global, `Voxel_Cone_Tracing(w, h, window)`,
// `init_voxel_cone_tracing()`, `Render()` per frame (R/main.cpp:64-68,90) -- compiles against this
// header unchanged in those lines.  The type, member and method names are the reference's public
// surface; nothing else is shared: there is no GL, no GLSL and no glm here.  Where the reference
// binds GL objects and issues draws, each method forwards to the C ABI:
//
//   init_voxel_cone_tracing()  VCT.h:67-140   -> vct_create, mesh upload, DrawDepthTexture, DrawVoxelTexture
//   DrawDepthTexture()         VCT.h:192-211  -> vct_render_shadow_map (GPU depth raster from the light)
//   DrawVoxelTexture()         VCT.h:213-250  -> vct_voxelize + vct_inject_light + vct_build_mips
//   Render()                   VCT.h:146-190  -> vct_render_gbuffer (GPU raster of the main draw's vertex and
//                                                non-cone fragment work) + vct_trace_current -> RGBA16F frame
// Every stage runs on the GPU and a frame never leaves HBM until Frame() is read.
//
// Differences a caller can observe, all forced by running headless on a compute GPU:
//   * `GLFWwindow` is an opaque forward declaration; Render() does not query the window size.
//   * The frame lands in an RGBA16F host buffer (`Frame()`), not in GL framebuffer 0.
//   * `VoxelDimensions` / `VoxelGridWorldSize` are plain members (the reference declares them const,
//     which pins it to 128^3): set them before init_voxel_cone_tracing().
//   * `model_path` replaces the reference's hard-coded absolute Windows path (VCT.h:77; the reference
//     ships no assets): "procedural:atrium", "procedural:atrium-textured", "procedural:bistro", "procedural:cornell", or the path of
//     a Wavefront .obj (+ .mtl with map_Kd / map_Ks / map_bump as PPM or TGA).
//   * Errors keep the reference's print-and-continue behaviour (VCT.h:101-105) and are also
//     readable through `last_status` / vct_last_error(ctx).
#ifndef VOXEL_CONE_TRACING_FACADE_H_
#define VOXEL_CONE_TRACING_FACADE_H_

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/vct.h"
#include "vct_host.h"

struct GLFWwindow;   // opaque: the headless build never dereferences it (the reference does once, VCT.h:149)

// ---- the few glm spellings the reference's public fields use -----------------------------------
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    vec3 operator+(const vec3& o) const { return vec3(x + o.x, y + o.y, z + o.z); }
    vec3 operator-(const vec3& o) const { return vec3(x - o.x, y - o.y, z - o.z); }
    vec3 operator*(float s) const { return vec3(x * s, y * s, z * s); }
    vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
    vec3& operator-=(const vec3& o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
};
struct mat4 {          // column-major like glm: m[col*4 + row]
    float m[16];
    mat4() { memset(m, 0, sizeof(m)); m[0] = m[5] = m[10] = m[15] = 1.0f; }
};
namespace vct_facade {
inline float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(const vec3& a, const vec3& b) {
    return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline vec3 normalize(const vec3& a) { const float l = sqrtf(dot(a, a)); return a * (1.0f / l); }
inline float radians(float deg) { return deg * 0.017453292519943295f; }
inline mat4 mul(const mat4& a, const mat4& b) {
    mat4 r;
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += a.m[k * 4 + row] * b.m[c * 4 + k];
            r.m[c * 4 + row] = s;
        }
    return r;
}
inline mat4 ortho(float l, float r, float b, float t, float n, float f) {
    mat4 o;
    o.m[0] = 2.0f / (r - l); o.m[5] = 2.0f / (t - b); o.m[10] = -2.0f / (f - n);
    o.m[12] = -(r + l) / (r - l); o.m[13] = -(t + b) / (t - b); o.m[14] = -(f + n) / (f - n);
    return o;
}
inline mat4 lookAt(const vec3& eye, const vec3& center, const vec3& up) {
    const vec3 f = normalize(center - eye), s = normalize(cross(f, up)), u = cross(s, f);
    mat4 o;
    o.m[0] = s.x; o.m[4] = s.y; o.m[8] = s.z;
    o.m[1] = u.x; o.m[5] = u.y; o.m[9] = u.z;
    o.m[2] = -f.x; o.m[6] = -f.y; o.m[10] = -f.z;
    o.m[12] = -dot(s, eye); o.m[13] = -dot(u, eye); o.m[14] = dot(f, eye);
    return o;
}
}  // namespace vct_facade

// ---- Camera: the fly camera main.cpp drives (R/Camera.h:28-145), same members and defaults -----
enum Camera_Direction { FORWARD, BACKWARD, LEFT, RIGHT, UP, DOWN };

class Camera {
public:
    vec3 position, Front, Up, Right, WorldUp;
    float Yaw, Pitch;
    float MovementSpeed, MouseSensitivity, Zoom;

    explicit Camera(vec3 position_ = vec3(0.0f, 0.0f, 0.0f), vec3 Up_ = vec3(0.0f, 1.0f, 0.0f),
                    float Yaw_ = -90.0f, float Pitch_ = 0.0f)
        : position(position_), Front(0.0f, 0.0f, -1.0f), Up(0.0f, 1.0f, 0.0f), WorldUp(Up_),
          Yaw(Yaw_), Pitch(Pitch_), MovementSpeed(2.6f), MouseSensitivity(0.1f), Zoom(45.0f) {
        UpdateCamera();
    }
    mat4 GetViewMatrix() const { return vct_facade::lookAt(position, position + Front, Up); }
    void ProcessKeyBoard(Camera_Direction direction, float& deltaTime) {
        const float v = MovementSpeed * deltaTime;
        switch (direction) {
            case FORWARD: position += Front * v; break;
            case BACKWARD: position -= Front * v; break;
            case LEFT: position -= Right * v; break;
            case RIGHT: position += Right * v; break;
            case UP: position += WorldUp * v; break;
            case DOWN: position -= WorldUp * v; break;
        }
    }
    void ProcessMouseMovement(float& xOffset, float& yOffset, bool constrainPitch = true) {
        xOffset *= MouseSensitivity;
        yOffset *= MouseSensitivity;
        Yaw += xOffset;
        Pitch += yOffset;
        if (constrainPitch) Pitch = fminf(fmaxf(Pitch, -89.0f), 89.0f);
        UpdateCamera();
    }
    void ProcessMouseScroll(float yOffset) { Zoom = fminf(fmaxf(Zoom - yOffset, 1.0f), 45.0f); }
    void UpdateCamera() {
        using namespace vct_facade;
        const float cy = cosf(radians(Yaw)), sy = sinf(radians(Yaw));
        const float cp = cosf(radians(Pitch)), sp = sinf(radians(Pitch));
        Front = normalize(vec3(cy * cp, sp, sy * cp));
        Right = normalize(cross(Front, WorldUp));
        Up = normalize(cross(Right, Front));
    }
};

// the header-defined global main.cpp mutates (VCT.h:8)
inline Camera camera(vec3(0.0f, 4.0f, 0.0f));

// ---- Model: scene container (R/Model.h).  Procedural stand-ins, see the header comment. --------
struct Model {
    vcth_scene* scene = nullptr;
    std::string path;
    Model() {}
    explicit Model(const std::string& p) { Load(p); }
    Model(const Model&) = delete;
    Model& operator=(const Model&) = delete;
    ~Model() { if (scene) vcth_scene_destroy(scene); }
    bool Load(const std::string& p) {
        if (scene) { vcth_scene_destroy(scene); scene = nullptr; }
        path = p;
        char err[256] = "";
        if (p == "procedural:cornell") scene = vcth_scene_create(0, 1.0f, 1234u);
        else if (p == "procedural:atrium") scene = vcth_scene_create(1, 1.0f, 1234u);
        else if (p == "procedural:atrium-textured") scene = vcth_scene_create(2, 1.0f, 1234u);
        else if (p == "procedural:bistro") scene = vcth_scene_create(3, 1.0f, 1234u);        // Bistro-exterior-class street, 2.8 M triangles
        else if (p.size() > 9 && p.compare(p.size() - 9, 9, ".vctscene") == 0) scene = vcth_scene_load_cache(p.c_str(), err);
        else scene = vcth_scene_load_obj(p.c_str(), err);             // R/Model.h:39-61
        if (!scene) printf("ERROR::MODEL: cannot load '%s' (%s; or use procedural:atrium | procedural:cornell)\n",
                           p.c_str(), err);
        return scene != nullptr;
    }
};

struct Voxel_Cone_Tracing {
    // Global properties                                                        VCT.h:14-17
    vec3 lightDirection = vec3(0.0f, 1.0f, 0.25f);
    int VoxelDimensions = 128;
    float VoxelGridWorldSize = 150.0f;

    GLFWwindow* window = nullptr;
    int screen_width = 1280;                                                 // VCT.h:24-25
    int screen_height = 720;

    unsigned ShadowMapSize = 4096;                                           // VCT.h:35

    mat4 DepthViewProjectionMatrix, ProjX, ProjY, ProjZ;                     // VCT.h:42-45
    Model model;                                                             // VCT.h:48
    std::string model_path = "procedural:atrium";

    bool ShowDiffuse = true, ShowIndirectDiffuse = true, ShowSpecular = true,
         ShowIndirectSpecular = true, ShowAmbientOcclusion = true;           // VCT.h:51 (never uploaded there either)
    float AmbientFactor = 0.1f;                                              // VCT.h:53
    // Multi-GPU (no reference counterpart, R/main.cpp drives one GL context): one process per GPU, each with
    // its own Voxel_Cone_Tracing; set Rank / World / Device and the 128-byte RCCL id (vct_comm_get_unique_id
    // on rank 0, shared out of band) before init.  Render() then rasterises and traces only this rank's
    // slab of tile rows and rank 0's Frame() receives the whole frame through ONE ncclGather.
    int Rank = 0, World = 1, Device = -1;
    unsigned char CommId[VCT_COMM_ID_BYTES] = {0};
    // Extension: the reference never rebuilds the shadow map or the volume after init (a changed lightDirection only
    // moves the direct term, VCT.h:168).  With DynamicLight every Render() is one whole GI pass for the current
    // lightDirection (vct_gi_pass: shadow map, voxelize + inject + mips beside the G-buffer raster, trace).
    bool DynamicLight = false;
    // The reference's own voxelization semantics (S/Voxelization.vs/.gs/.fs as written: dominant-axis raster at pixel
    // centres, last writer wins, VCT.h:213-250) instead of the north-star default (conservative + averaged); set
    // before init or before calling DrawVoxelTexture() again.
    bool ReferenceVoxelization = false;
    // 2: consecutive Render() calls alternate between two frame slots of the context (vct_set_frames_in_flight: own
    // stream, G-buffer and frame each), so frame k + 1's raster and trace start while frame k's trace drains -- what the
    // GL driver does with the reference's frames (R/main.cpp:77-94 never waits for one).  Frame() returns the frame of
    // the last Render().  Same pixels.  What it buys is small since the facade issues its frames without timing events
    // (vct_set_trace_timing: those were most of the gap a second slot hides): configs[1] 0.70 -> 0.69 ms per Render(),
    // erratic at configs[4].  Frames that are whole GI passes (DynamicLight) stay on one slot.  Set before init.
    int FramesInFlight = 1;
    int Bounces = 1;    // 2 = re-inject the lit voxels once (the "2 bounces" of the reference's README.md:16,
                        // which its code does not implement: VCT.h:138-139 injects once); set before init

    // what replaces the GL object names (Depth_FBO, Depth_Texture, VoxelTexture)
    vct_ctx* ctx = nullptr;
    int last_status = VCT_OK;
    // Like the reference's Render() (VCT.h:146-190: draw into framebuffer 0, never read back) a frame stays in HBM;
    // Frame() copies it to the host when -- and only when -- somebody asks for it.
    mutable std::vector<uint16_t> FrameRGBA16F;      // screen_width * screen_height * 4 halves
    mutable bool frame_on_host = false;
    unsigned frame_no = 0;                           // Render() calls so far (FramesInFlight: selects the slot)

    Voxel_Cone_Tracing() {}
    Voxel_Cone_Tracing(int screen_width_, int screen_height_, GLFWwindow*& window_)
        : window(window_), screen_width(screen_width_), screen_height(screen_height_) {}
    Voxel_Cone_Tracing(const Voxel_Cone_Tracing&) = delete;
    Voxel_Cone_Tracing& operator=(const Voxel_Cone_Tracing&) = delete;
    ~Voxel_Cone_Tracing() { if (ctx) vct_destroy(ctx); }

    void init_voxel_cone_tracing() {
        using namespace vct_facade;
        vct_config cfg;
        vct_default_config(&cfg);
        cfg.voxel_dim = VoxelDimensions;
        cfg.grid_world_size = VoxelGridWorldSize;
        cfg.width = screen_width;
        cfg.height = screen_height;
        cfg.shadow_map_size = (int32_t)ShadowMapSize;
        cfg.ambient_factor = AmbientFactor;
        cfg.voxel_attributes = Bounces >= 2 ? 1 : 0;
        cfg.device = Device;
        if (!check(vct_create(&cfg, &ctx), "vct_create")) return;
        if (World > 1 || Rank != 0 || CommId[0] || CommId[1])
            if (!check(vct_comm_init(ctx, CommId, Rank, World), "vct_comm_init")) return;
        if (FramesInFlight == 2)
            if (!check(vct_set_frames_in_flight(ctx, 2), "vct_set_frames_in_flight")) return;
        // a frame loop: no timing events around the trace launches (they cost a launch ~7 us; vct_last_trace_ms wants
        // vct_set_trace_timing(ctx, 1) before the frame it is to time)
        if (!check(vct_set_trace_timing(ctx, 0), "vct_set_trace_timing")) return;
        if (!model.Load(model_path)) { last_status = VCT_ERR_INVALID; return; }

        // VCT.h:84-86 and :128-134 (the projections are kept as public data; the HIP voxelizer maps
        // world -> voxel directly, which is what the three of them amount to: SURVEY.md a8)
        const float L[3] = {lightDirection.x, lightDirection.y, lightDirection.z};
        vcth_light_view_proj(L, DepthViewProjectionMatrix.m);
        const float G = VoxelGridWorldSize, h = G * 0.5f;
        const mat4 o = ortho(-h, h, -h, h, h, G * 1.5f);
        ProjX = mul(o, lookAt(vec3(G, 0, 0), vec3(0, 0, 0), vec3(0, 1, 0)));
        ProjY = mul(o, lookAt(vec3(0, G, 0), vec3(0, 0, 0), vec3(0, 0, -1)));
        ProjZ = mul(o, lookAt(vec3(0, 0, G), vec3(0, 0, 0), vec3(0, 1, 0)));

        const int32_t ntri = vcth_scene_num_triangles(model.scene), nmat = vcth_scene_num_materials(model.scene);
        std::vector<float> pos((size_t)ntri * 9), albedo((size_t)nmat * 4), specular((size_t)nmat * 3);
        std::vector<int32_t> material((size_t)ntri);
        vcth_scene_get(model.scene, pos.data(), material.data(), albedo.data(), specular.data());
        if (!check(vct_upload_triangles(ctx, pos.data(), material.data(), ntri, albedo.data(), nmat),
                   "vct_upload_triangles")) return;
        std::vector<float> nrm((size_t)ntri * 9), tan((size_t)ntri * 9), bit((size_t)ntri * 9);
        vcth_scene_get_frames(model.scene, nrm.data(), tan.data(), bit.data());
        if (!check(vct_upload_mesh_attributes(ctx, nrm.data(), tan.data(), bit.data(), specular.data()),
                   "vct_upload_mesh_attributes")) return;
        const int32_t ntex = vcth_scene_num_textures(model.scene);        // R/Model.h:126-136: material textures
        if (ntex > 0) {
            std::vector<float> uv((size_t)ntri * 6);
            vcth_scene_get_uvs(model.scene, uv.data());
            std::vector<std::vector<uint8_t>> texels((size_t)ntex);
            std::vector<const uint8_t*> ptr((size_t)ntex);
            std::vector<int32_t> tw((size_t)ntex), th((size_t)ntex), mat_tex((size_t)nmat * 3);
            for (int32_t i = 0; i < ntex; ++i) {
                vcth_scene_texture_info(model.scene, i, &tw[(size_t)i], &th[(size_t)i]);
                texels[(size_t)i].resize((size_t)tw[(size_t)i] * th[(size_t)i] * 4);
                vcth_scene_get_texture(model.scene, i, texels[(size_t)i].data());
                ptr[(size_t)i] = texels[(size_t)i].data();
            }
            vcth_scene_get_material_textures(model.scene, mat_tex.data());
            if (!check(vct_upload_mesh_uvs(ctx, uv.data()), "vct_upload_mesh_uvs")) return;
            if (!check(vct_upload_textures(ctx, ptr.data(), tw.data(), th.data(), ntex, mat_tex.data()),
                       "vct_upload_textures")) return;
        }
        DrawDepthTexture();     // VCT.h:138
        DrawVoxelTexture();     // VCT.h:139
    }

    void Render() {
        if (!ctx || !model.scene) return;
        vct_set_ambient_factor(ctx, AmbientFactor);
        const float cam[3] = {camera.position.x, camera.position.y, camera.position.z};   // VCT.h:167
        const float L[3] = {lightDirection.x, lightDirection.y, lightDirection.z};         // VCT.h:168
        vct_set_camera_position(ctx, cam);
        vct_set_light_direction(ctx, L);
        vcth_camera hc;
        memcpy(hc.position, cam, sizeof(cam));
        hc.yaw = camera.Yaw; hc.pitch = camera.Pitch; hc.zoom = camera.Zoom;
        hc.z_near = 0.1f; hc.z_far = 1000.0f;                                              // VCT.h:162
        float vp[16];
        vcth_camera_view_proj(&hc, screen_width, screen_height, vp);                       // VCT.h:161-163
        // frame k in slot k & 1 -- unless every Render() is a whole GI pass (DynamicLight): passes that rewrite the chain run
        // one after the other whatever the slot and would only pay the hand-over between hardware queues (include/vct.h:
        // 0.842 ms per pass on alternating slots against 0.818 on one), so they stay on the slot that is selected
        int32_t row0 = 0, row1 = 0;
        const bool rank_ctx = vct_comm_slab(ctx, &row0, &row1) == VCT_OK;
        const bool whole_pass = !rank_ctx && DynamicLight && Bounces < 2;
        int32_t slots = 1;
        if (!whole_pass && vct_get_frames_in_flight(ctx, &slots, nullptr, nullptr) == VCT_OK && slots == 2)
            if (!check(vct_select_frame_slot(ctx, (int32_t)(frame_no++ & 1u)), "vct_select_frame_slot")) return;
        if (rank_ctx) {                                          // multi-GPU: this rank's slab, one gather
            if (!check(vct_render_gbuffer_rows(ctx, vp, row0, row1), "vct_render_gbuffer_rows")) return;
            if (!check(vct_frame_step(ctx), "vct_frame_step")) return;
            frame_on_host = false;
            return;
        }
        if (whole_pass) {
            vcth_light_view_proj(L, DepthViewProjectionMatrix.m);                          // VCT.h:84-86, per frame
            const int32_t mode = ReferenceVoxelization ? VCT_VOX_REFERENCE : VCT_VOX_CONSERVATIVE_AVG;
            if (!check(vct_gi_pass(ctx, DepthViewProjectionMatrix.m, vp, mode), "vct_gi_pass")) return;
            frame_on_host = false;
            return;
        }
        if (!check(vct_render_gbuffer(ctx, vp), "vct_render_gbuffer")) return;
        if (!check(vct_trace_resident_rows(ctx, 0, (screen_height + 7) / 8), "vct_trace_resident_rows")) return;
        frame_on_host = false;
    }

    // Block until the GPU has finished the frames issued so far (glFinish in GL terms); no copy.
    void Finish() {
        if (!ctx) return;
        int32_t row0 = 0, row1 = 0;
        if (vct_comm_slab(ctx, &row0, &row1) == VCT_OK) check(vct_comm_sync(ctx), "vct_comm_sync");
        else check(vct_synchronize(ctx), "vct_synchronize");
    }

    void DrawDepthTexture() {
        if (!ctx || !model.scene) return;
        check(vct_render_shadow_map(ctx, DepthViewProjectionMatrix.m), "vct_render_shadow_map");
    }

    void DrawVoxelTexture() {
        if (!ctx) return;
        if (!check(vct_voxelize(ctx, ReferenceVoxelization ? VCT_VOX_REFERENCE : VCT_VOX_CONSERVATIVE_AVG), "vct_voxelize")) return;
        if (!check(vct_inject_light(ctx), "vct_inject_light")) return;
        if (!check(vct_build_mips(ctx), "vct_build_mips")) return;          // VCT.h:248
        if (Bounces >= 2) check(vct_bounce(ctx), "vct_bounce");
    }

    // The last frame as RGBA16F on the host (row 0 = bottom row of the GL window): waits for the GPU and downloads on
    // the first call after a Render().  On a multi-GPU run rank 0 holds the gathered frame; other ranks get nullptr.
    const uint16_t* Frame() const {
        if (!ctx) return nullptr;
        if (!frame_on_host) {
            FrameRGBA16F.resize((size_t)screen_width * screen_height * 4);
            int32_t row0 = 0, row1 = 0;
            int rc;
            if (vct_comm_slab(ctx, &row0, &row1) == VCT_OK) {
                if (Rank != 0) { vct_comm_sync(ctx); return nullptr; }
                rc = vct_comm_download_frame(ctx, FrameRGBA16F.data());
            } else {
                rc = vct_download_frame(ctx, FrameRGBA16F.data());
            }
            if (rc != VCT_OK) { printf("ERROR::VCT::Frame: %s\n", vct_last_error(ctx)); return nullptr; }
            frame_on_host = true;
        }
        return FrameRGBA16F.data();
    }

private:
    bool check(int rc, const char* what) {
        last_status = rc;
        if (rc != VCT_OK) printf("ERROR::VCT::%s: %s\n", what, vct_last_error(ctx));
        return rc == VCT_OK;
    }
};

#endif
