/*
 * vct_host.h -- host-side input stages around the GI hot path (libvct_host.so, plain C++, no HIP).
 *
 * These stand where the reference's asset loader and its two non-GI draws stand:
 *   - scene:       R/Model.h + R/Mesh.h (assimp import, `Vertex` = position/normal/uv/tangent/
 *                  bitangent, R/Mesh.h:12-19).  There are no assets in the reference repo and no
 *                  network here, so scenes are procedural (Cornell box, Sponza-class atrium) with a
 *                  flat albedo/specular colour per material in place of textures.
 *   - shadow map:  DrawDepthTexture (VCT.h:192-211, S/Shadow.vs): depth-only orthographic raster
 *                  from the light, 24-bit depth, back faces culled.
 *   - G-buffer:    the vertex + fixed-function part of the main draw (S/VoxelConeTracing.vs:23-37,
 *                  perspective raster, depth test LESS, back-face cull: R/main.cpp:55-58) plus the
 *                  per-fragment inputs of S/VoxelConeTracing.fs that are not cone tracing: bump
 *                  normal (:110-128 with a flat height map), material colours (:167,:209-210) and
 *                  the 25-tap PCF shadow term with its 0.111 scale (:132-163).
 * They produce the inputs of vct_voxelize / vct_trace (include/vct.h); they are not part of the
 * measured hot path and run on the CPU (SURVEY.md 8(f) rows f1-f3 list their GPU versions as next).
 */
#ifndef VCT_HOST_H_
#define VCT_HOST_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vcth_scene vcth_scene;

typedef struct vcth_camera {
    float position[3];   /* Camera.h: position, default (0,4,0)  VCT.h:8 */
    float yaw, pitch;    /* degrees; defaults -90, 0             Camera.h:21-22 */
    float zoom;          /* fov in degrees, default 45           Camera.h:47 */
    float z_near, z_far; /* 0.1, 1000                            VCT.h:162 */
} vcth_camera;

void vcth_default_camera(vcth_camera* cam);

/* kind: 0 = Cornell box (~40 tris), 1 = atrium (Sponza-class; `detail` scales tessellation,
 * detail = 1.0 gives ~262k triangles).  Model-space coordinates = world / 0.05 (VCT.h:183). */
vcth_scene* vcth_scene_create(int kind, float detail, uint32_t seed);
/* Wavefront OBJ (+ MTL) reader -- stands where the reference's assimp import stands (R/Model.h:39-61:
 * triangulate, smooth normals, tangent space).  Reads v / vn / vt / f (polygons are fan-triangulated,
 * negative indices allowed), usemtl + mtllib (Kd -> albedo, Ks -> specular, d -> albedo alpha); missing
 * normals are generated area-weighted per position, tangents from the UVs when present else from the
 * normal.  Textures are not read: materials are flat colours (SURVEY.md A.7 allows this).  Coordinates
 * are taken as MODEL space (the orchestrator scales by 0.05, VCT.h:183).  Returns NULL on failure;
 * `error` (optional, >= 256 bytes) receives the reason. */
vcth_scene* vcth_scene_load_obj(const char* path, char* error);
void vcth_scene_destroy(vcth_scene* s);
int32_t vcth_scene_num_triangles(const vcth_scene* s);
int32_t vcth_scene_num_materials(const vcth_scene* s);
/* pos [ntri*9], material [ntri], albedo [nmat*4], specular [nmat*3]; any may be NULL. */
void vcth_scene_get(const vcth_scene* s, float* pos, int32_t* material, float* albedo,
                    float* specular);

/* Per-vertex frame (R/Mesh.h:12-19): normal, tangent, bitangent [ntri*9] each; any may be NULL. */
void vcth_scene_get_frames(const vcth_scene* s, float* normal, float* tangent, float* bitangent);
/* VCT.h:161-163: perspective(radians(Zoom), w/h, near, far) * camera.GetViewMatrix(), column-major. */
void vcth_camera_view_proj(const vcth_camera* cam, int32_t width, int32_t height, float out_vp[16]);

/* VCT.h:84-86: DepthViewProjectionMatrix = ortho(-120,120,-120,120,-100,100) * lookAt(L,0,+Y),
 * column-major. */
void vcth_light_view_proj(const float light_dir[3], float out_vp[16]);
/* VCT.h:192-211: depth [size*size] in [0,1], 24-bit quantised, cleared to 1. */
void vcth_render_shadow_map(const vcth_scene* s, float model_scale, const float light_vp[16],
                            int32_t size, float* depth);

/* Fills the 23-plane linear G-buffer (include/vct.h VCT_GB_*), planes [23][w*h].  Pixels without
 * a fragment get albedo.a = 0.  Row 0 is the bottom row of the GL window (y up).
 * shadow_depth may be NULL (shadow_value = 25 * 0.111, fully lit). */
void vcth_render_gbuffer(const vcth_scene* s, float model_scale, const vcth_camera* cam,
                         int32_t width, int32_t height, const float* shadow_depth,
                         int32_t shadow_size, const float light_vp[16], float* planes);

#ifdef __cplusplus
}
#endif
#endif
