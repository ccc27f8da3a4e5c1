/*
 * vct_host.h -- host-side scene + camera helpers around the GI hot path (libvct_host.so, plain C++, no HIP).
 *
 * These stand where the reference's asset loader and camera stand:
 *   - scene:   R/Model.h + R/Mesh.h (assimp import, `Vertex` = position / normal / uv / tangent / bitangent,
 *              R/Mesh.h:12-19; diffuse / specular / height textures, R/Model.h:126-136).  There are no assets
 *              in the reference repo and no network here, so the built-in scenes are procedural (Cornell box,
 *              Sponza-class atrium -- flat-coloured or with procedural texture maps); a Wavefront OBJ + MTL
 *              (+ PPM / TGA maps) can be loaded by path.
 *   - camera:  R/Camera.h defaults and the matrices of VCT.h:84-86 (light) and :161-163 (view-projection).
 * The shadow-map and G-buffer stages themselves run on the GPU (vct_render_shadow_map / vct_render_gbuffer,
 * csrc/vct_raster.hip); their CPU checkers live in oracle/ and are not part of this library.
 */
#ifndef VCT_HOST_H_
#define VCT_HOST_H_

#include <stddef.h>
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
 * detail = 1.0 gives ~262k triangles; flat colour per material), 2 = the same atrium with procedural
 * texture maps (checker floor + red-only specular map, brick + height map, stone + noisy height map,
 * bronze specular map, cloth with alpha cut-outs), 3 = Bistro-exterior-class street (BASELINE.json configs[4]; detail
 * 1.0 gives ~2.8 M triangles: two rows of facades with windows / balconies / awnings, terraces with tables and chairs,
 * lamps and strings of lights, 14 trees whose crowns are alpha-tested leaf cards with random orientations -- about
 * 40 % of the triangles -- every surface textured, noisy height maps).  Model-space coordinates = world / 0.05
 * (VCT.h:183). */
vcth_scene* vcth_scene_create(int kind, float detail, uint32_t seed);
/* Wavefront OBJ (+ MTL) reader -- stands where the reference's assimp import stands (R/Model.h:39-61:
 * triangulate, smooth normals, tangent space).  Reads v / vn / vt / f (polygons are fan-triangulated,
 * negative indices allowed), usemtl + mtllib (Kd -> albedo, Ks -> specular, d -> albedo alpha); missing
 * normals are generated area-weighted per position, tangents from the UVs when present else from the
 * normal.  map_Kd / map_Ks / map_bump name the diffuse / specular / height textures (R/Model.h:126-136);
 * binary PPM and uncompressed TGA files are decoded, a map that cannot be read leaves the flat colour.  Coordinates
 * are taken as MODEL space (the orchestrator scales by 0.05, VCT.h:183).  Returns NULL on failure;
 * `error` (optional, >= 256 bytes) receives the reason. */
vcth_scene* vcth_scene_load_obj(const char* path, char* error);
/* On-disk cache of a scene (triangles, frames, texture coordinates, materials, decoded texture maps): what
 * R/Model.h:39-61 + :141-226 rebuild from the .obj and its images on every start.  vcth_scene_save returns 0 on
 * success; vcth_scene_load_cache returns NULL (reason in `error`, optional, >= 256 bytes) for a missing,
 * truncated, inconsistent or differently versioned file. */
int vcth_scene_save(const vcth_scene* s, const char* path);
vcth_scene* vcth_scene_load_cache(const char* path, char* error);
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
/* Texture coordinates [ntri*6] (attribute 2, R/Mesh.h:72-73) and the material textures: RGBA8, row 0 at
 * v = 0; mat_tex [nmat*3] = diffuse / specular / height texture index or -1 (flat colour / flat height). */
void vcth_scene_get_uvs(const vcth_scene* s, float* uv);
int32_t vcth_scene_num_textures(const vcth_scene* s);
void vcth_scene_texture_info(const vcth_scene* s, int32_t i, int32_t* width, int32_t* height);
void vcth_scene_get_texture(const vcth_scene* s, int32_t i, uint8_t* rgba);
void vcth_scene_get_material_textures(const vcth_scene* s, int32_t* mat_tex);

/* The decoder behind map_Kd / map_Ks / map_bump (host/vct_image.h: PNG, baseline JPEG, BMP, TGA raw / RLE, PPM / PGM --
 * the containers stb_image serves the reference with, R/Model.h:141-226).  Two calls: rgba == NULL returns the size;
 * then rgba [h][w][4] with `capacity` bytes of room, row 0 = BOTTOM row (width / height are reported again: check them).
 * Returns 0, -1 when the file is unreadable / unsupported / corrupt, -2 when the image does not fit `capacity` (the
 * file changed between the two calls). */
int32_t vcth_image_load(const char* path, int32_t* width, int32_t* height, uint8_t* rgba, size_t capacity);

#ifdef __cplusplus
}
#endif
#endif
