/*
 * vct.h -- C ABI of the MI355X-native voxel-cone-tracing GI path (libvct_amd.so).
 *
 * This is the drop-in boundary.  The reference has no plugin / FFI interface: its boundary is
 * the header-only `struct Voxel_Cone_Tracing` (R/Voxel_Cone_Tracing.h:11-252) that main.cpp
 * calls directly (R/main.cpp:66,68,90).  The C++ facade of the same name in
 * voxel-cone-tracing_amd/host/Voxel_Cone_Tracing.h keeps those member names and forwards to the
 * entry points below; each entry point cites the reference code it replaces
 * (R = Voxel_Cone_Tracing_Final, S = R/Shader).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a
 * negative vct_status; vct_last_error() returns the message of the last failure (per context,
 * or the creation failure when ctx == NULL).  Host pointers are caller-owned; all HBM is
 * context-owned and released by vct_destroy().  One host thread per context, one context per
 * GPU; every kernel runs on the context's HIP stream.  There is no CPU fallback: creation
 * fails if no gfx950 device is usable.
 */
#ifndef VCT_H_
#define VCT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VCT_ABI_VERSION 7

typedef enum vct_status {
    VCT_OK = 0,
    VCT_ERR_INVALID = -1,     /* bad argument / call order */
    VCT_ERR_DEVICE = -2,      /* HIP runtime error (message in vct_last_error) */
    VCT_ERR_NO_DEVICE = -3,   /* no usable GPU */
    VCT_ERR_NOMEM = -4
} vct_status;

/* G-buffer: 23 fp32 planes per pixel (92 B) -- the per-fragment varyings and material fetches of
 * S/VoxelConeTracing.vs:10-15 and S/VoxelConeTracing.fs:3-9,167,209 made explicit. */
enum {
    VCT_GB_POSITION = 0,   /* Position_world xyz            trace.vs:27 */
    VCT_GB_NORMAL = 3,     /* Normal_world xyz (raw)        trace.vs:31 */
    VCT_GB_TANGENT = 6,    /* Tangent_world xyz (raw)       trace.vs:32 */
    VCT_GB_BITANGENT = 9,  /* BiTangent_world xyz (raw)     trace.vs:33 */
    VCT_GB_BUMP_N = 12,    /* bump normal N (unit)          trace.fs:177 */
    VCT_GB_ALBEDO = 15,    /* matColor rgba                 trace.fs:167 */
    VCT_GB_SPECULAR = 19,  /* specColor rgb                 trace.fs:209-210 */
    VCT_GB_SHADOW = 22,    /* shadow_value                  trace.fs:186 */
    VCT_GB_PLANES = 23
};

typedef enum vct_gb_layout {
    VCT_GB_LINEAR = 0,     /* planes[k][y*width + x] */
    VCT_GB_TILED = 1       /* device layout: [tile][plane][64], tile = 8x8 px, lane = (y&7)*8+(x&7),
                              tiles row-major over ceil(w/8) x ceil(h/8) */
} vct_gb_layout;

typedef enum vct_mem {
    VCT_MEM_HOST = 0,
    VCT_MEM_DEVICE = 1
} vct_mem;

typedef struct vct_gbuffer {
    const float* planes;
    int32_t width, height;
    int32_t layout;        /* vct_gb_layout */
    int32_t location;      /* vct_mem */
} vct_gbuffer;

/* Configuration = the reference's compile-time constants and public fields
 * (R/Voxel_Cone_Tracing.h:14-53, S/VoxelConeTracing.fs:43-46), runtime-settable. */
typedef struct vct_config {
    int32_t abi_version;       /* VCT_ABI_VERSION */
    int32_t device;            /* HIP device ordinal; -1 = current device */
    int32_t voxel_dim;         /* VoxelDimensions: power of two in [8,1024]   VCT.h:16 */
    float grid_world_size;     /* VoxelGridWorldSize = 150                     VCT.h:17 */
    int32_t width, height;     /* screen_width/height                          VCT.h:24-25 */
    int32_t shadow_map_size;   /* ShadowMapSize = 4096                         VCT.h:35 */
    float model_scale;         /* ModelMatrix = scale(0.05)                    VCT.h:183,240 */
    float ambient_factor;      /* AmbientFactor = 0.1                          VCT.h:53 */
    float shininess;           /* Shininess = 20                               Mesh.h:86 */
    float max_distance;        /* MAX_DISTANCE = 75                            trace.fs:43 */
    float max_alpha;           /* MAX_ALPHA = 0.95                             trace.fs:44 */
    float tan_diffuse;         /* 0.577                                        trace.fs:198 */
    float tan_specular;        /* 0.07                                         trace.fs:218 */
    int32_t wrap_repeat;       /* 1 = GL_REPEAT (VCT.h:110-113 leaves the GL default) */
    int32_t debug_outputs;     /* 1 = also keep per-cone step counts and raw cone vec4s */
    int32_t trace_variant;     /* 0 = default (cooperative sampler, tile split over 3 waves); A/B variants with
                                  identical results: 1 = per-lane sampler, 2 = one wave per tile.  3 = the default
                                  kernel with a one-multiply unorm8 decode and reciprocal-multiply divisions: NOT
                                  bit-exact (within the 1e-3 frame tolerance), never a default -- it exists to
                                  measure what the exactness costs (bench.py exactness_tax).  4 = the
                                  default kernel over a live-pixel compaction of 16x16 super-tiles (identical
                                  results; experiment, profiles/experiments/README.md) */
    int32_t voxel_attributes;  /* 1 = the voxelizer also keeps per-voxel mean albedo + face normal
                                  (needed by vct_bounce; 24 B/voxel of extra accumulators) */
    int32_t anisotropic_mips;  /* 1 = also keep six directional (pre-integrated) mip chains and sample
                                  levels >= 1 from them by cone direction (north-star option; the
                                  reference has one isotropic chain -- VCT.h:248 -- so the default 0 is
                                  what matches the shader transliteration) */
    int32_t texture_mipmaps;   /* 1 (default) = material textures get mip chains (glGenerateMipmap, Model.h:168) and are
                                  sampled LINEAR_MIPMAP_LINEAR with the implicit derivatives of texture() (Model.h:172;
                                  trace.fs:114-116,167,209, vox.fs:56); 0 = level 0, bilinear (rounds 1-2) */
} vct_config;

typedef struct vct_ctx vct_ctx;

typedef enum vct_voxelize_mode {
    VCT_VOX_CONSERVATIVE_AVG = 0,   /* north-star: conservative overlap + atomic integer average */
    VCT_VOX_REFERENCE = 1           /* S/Voxelization.*: pixel-centre raster, last triangle wins */
} vct_voxelize_mode;

int vct_default_config(vct_config* cfg);

/* Replaces the ctor + resource creation of init_voxel_cone_tracing (VCT.h:57-65,107-126):
 * allocates the brick mip chain (zero-filled, like VCT.h:115-119) and frame buffers in HBM. */
int vct_create(const vct_config* cfg, vct_ctx** out_ctx);
void vct_destroy(vct_ctx* ctx);
const char* vct_last_error(const vct_ctx* ctx);
int vct_get_config(const vct_ctx* ctx, vct_config* cfg);

/* Per-frame uniforms (VCT.h:167-168,171). */
int vct_set_camera_position(vct_ctx* ctx, const float pos[3]);
int vct_set_light_direction(vct_ctx* ctx, const float dir[3]);
int vct_set_ambient_factor(vct_ctx* ctx, float ambient);
int vct_set_cone_apertures(vct_ctx* ctx, float tan_diffuse, float tan_specular);
/* config.trace_variant of the following traces (0 .. 4, see vct_config).  No reference counterpart: the variants are
 * measurement alternatives of the one cone trace of S/VoxelConeTracing.fs:82-107,165-228; 0 is the exact default. */
int vct_set_trace_variant(vct_ctx* ctx, int32_t variant);
/* Footprint records (no reference counterpart; a layout option of the 8^3-brick Morton chain for HBM-bound volumes):
 * on != 0 keeps, beside the chain, one 32-byte record per texel of the levels >= 1 holding the 8 texels of the
 * trilinear footprint anchored there (GL_REPEAT folded in), so that an incoherent (per-lane) level sample is one
 * 32-byte fetch instead of eight 4-byte ones from two to four cache lines.  Costs 8 x the bytes of those levels
 * (1.14 x level 0: 77 MB at 256^3, 4.9 GB at 1024^3) and a dense rebuild after every mip build.  Same frame, bit for
 * bit.  Pays where the chain does not fit the caches (dense random 1024^3 chain: trace 5.61 -> 2.86 ms); a scene that
 * touches a thin shell of its grid stays cache-resident and gains nothing (street at 1024^3 / 4K: 2.72 -> 2.70 ms).
 * Default off (VCT_FOOTPRINT_RECORDS=1 in the environment turns it on at vct_create).  The clamp-to-edge sampler,
 * the anisotropic chains and the second-bounce chain keep per-texel gathers. */
int vct_set_footprint_records(vct_ctx* ctx, int32_t on);

/* Scene upload -- replaces Model/Mesh VBO setup (R/Mesh.h:49-82) for the two attributes the
 * voxelizer reads (vox.vs:3-4).  pos: [ntri][3][3] model-space fp32; material: [ntri];
 * albedo: [nmat][4] flat per-material albedo (stands in for DiffuseTexture, vox.fs:56). */
int vct_upload_triangles(vct_ctx* ctx, const float* pos, const int32_t* material, int32_t ntri,
                         const float* albedo, int32_t nmat);
/* Shadow map produced by the depth pass (VCT.h:192-211): size*size fp32 depths in [0,1] plus the
 * column-major DepthViewProjectionMatrix (VCT.h:84-86).  depth == NULL detaches it (PCF = 1). */
int vct_upload_shadow_map(vct_ctx* ctx, const float* depth, int32_t size, const float light_vp[16]);

/* ---- raster input stages on the GPU ------------------------------------------
 * Per-vertex frame of the uploaded triangles (R/Mesh.h:12-19 normal / tangent / bitangent, attribs
 * 1,3,4 of S/VoxelConeTracing.vs) and the per-material specular colour (trace.fs:209): normal,
 * tangent, bitangent [ntri][3][3] model space, specular [nmat][3].  Call after vct_upload_triangles. */
int vct_upload_mesh_attributes(vct_ctx* ctx, const float* normal, const float* tangent,
                               const float* bitangent, const float* specular);
/* Texture coordinates of the uploaded triangles (attribute 2, R/Mesh.h:72-73): uv [ntri][3][2]. */
int vct_upload_mesh_uvs(vct_ctx* ctx, const float* uv);
/* Material textures (R/Model.h:126-136,141-226; bound per draw at R/Mesh.h:91-108): ntex RGBA8 images
 * (rgba8[i]: height[i] * width[i] * 4 bytes, row 0 at v = 0) and, per material, the index of its
 * DiffuseTexture / SpecularTexture / HeightTexture or -1 (mat_tex [nmat][3]; -1 keeps the flat colour of
 * vct_upload_triangles / vct_upload_mesh_attributes, resp. a flat height map).  texture(sampler, uv) is
 * restated as GL_REPEAT, mip-mapped (config.texture_mipmaps: box-filtered chain built on the GPU at upload,
 * LINEAR_MIPMAP_LINEAR / LINEAR, lambda from the differences of uv inside the fragment's 2x2 quad; the rules an
 * OpenGL implementation is free to choose are written down in oracle/vct_oracle.h) or level 0 bilinear.  Used by
 * vct_voxelize (albedo fetch, vox.fs:56) and vct_render_gbuffer (matColor + alpha test trace.fs:167-172,
 * CalcBumpNormal :110-128, specColor :209-210) once vct_upload_mesh_uvs has been called too.  ntex = 0
 * detaches them.  Call after vct_upload_triangles. */
int vct_upload_textures(vct_ctx* ctx, const uint8_t* const* rgba8, const int32_t* width, const int32_t* height,
                        int32_t ntex, const int32_t* mat_tex);
/* DrawDepthTexture (VCT.h:192-211, S/Shadow.vs/.fs): rasterises the uploaded triangles from the light
 * (column-major DepthViewProjectionMatrix, VCT.h:84-86) into the context's shadow map of
 * config.shadow_map_size^2 24-bit depths -- the map vct_voxelize and vct_render_gbuffer then read. */
int vct_render_shadow_map(vct_ctx* ctx, const float light_vp[16]);
int vct_download_shadow_map(vct_ctx* ctx, float* depth);
/* The vertex + fixed-function part of Render (VCT.h:161-189, S/VoxelConeTracing.vs, depth test LESS,
 * back faces culled) and the non-cone per-fragment inputs of S/VoxelConeTracing.fs (bump normal,
 * material colours, PCF shadow term): fills the resident tiled G-buffer from the uploaded mesh for
 * the column-major view-projection matrix (VCT.h:161-163).  Follow with vct_trace_resident or
 * vct_trace_current. */
int vct_render_gbuffer(vct_ctx* ctx, const float view_proj[16]);
/* The same for tile rows [tile_row0, tile_row1) only (scissored raster + shading of those tiles): what a
 * multi-GPU rank runs for its slab.  Other rows of the resident G-buffer keep their old content. */
int vct_render_gbuffer_rows(vct_ctx* ctx, const float view_proj[16], int32_t tile_row0, int32_t tile_row1);
/* Linear planes [23][h*w] of the resident G-buffer. */
int vct_download_gbuffer(vct_ctx* ctx, float* planes);
/* Trace the resident G-buffer (vct_render_gbuffer, or the last vct_trace upload) and return the frame
 * like vct_trace. */
int vct_trace_current(vct_ctx* ctx, void* out_rgba16f, int32_t out_location);

/* DrawVoxelTexture (VCT.h:213-245) -> vox.vs / vox.gs / vox.fs: voxelize the uploaded triangles
 * into per-voxel integer accumulators (mode selects coverage + resolve rule).  In the north-star mode only the light
 * is evaluated per pass: the fragment list, every fragment's barycentrics and -- with textures -- its albedo depend on
 * mesh, grid and textures alone and are kept per context (12-24 B per fragment); the first pass after
 * vct_upload_mesh_uvs / vct_upload_textures rebuilds the albedo (and may return VCT_ERR_NOMEM for it). */
int vct_voxelize(vct_ctx* ctx, int32_t mode);
/* vox.fs:88: resolve the accumulators into radiance level 0 (rgb = albedo * PCF shadow, a = 1). */
int vct_inject_light(vct_ctx* ctx);
/* glGenerateMipmap (VCT.h:126,248): 2x2x2 box, requantised per level, over the brick chain. */
int vct_build_mips(vct_ctx* ctx);

/* Second bounce (north-star, BASELINE.json config 3; the reference's README claims 2 bounces but its
 * code injects once -- VCT.h:138-139 -- so the definition is this build's, oracle/vct_oracle.h):
 * every occupied voxel gathers 6 diffuse cones from the current (bounce-0) chain along its stored
 * normal and adds albedo * occlusion-weighted irradiance; the result becomes level 0 of a second
 * chain, its mips are built, and vct_trace reads that chain until the next vct_inject_light.
 * Needs config.voxel_attributes = 1 and vct_voxelize + vct_inject_light + vct_build_mips first. */
int vct_bounce(vct_ctx* ctx);
/* Per-voxel attributes of the last resolve (config.voxel_attributes = 1): V^3 * 4 bytes each, linear
 * voxel order; albedo rgb (a = 255 where occupied), normal xyz biased by +128 (w = 255 where occupied). */
int vct_download_voxel_attributes(vct_ctx* ctx, uint8_t* albedo, uint8_t* normal);

/* Volumes built elsewhere (fixtures, oracle-built volumes).  Linear layout: level k has
 * N = V>>k texels per side, texel (x,y,z) at ((z*N+y)*N+x)*4, levels concatenated. */
int vct_upload_volume_rgba8(vct_ctx* ctx, const uint8_t* level0_linear);
int vct_upload_chain_rgba8(vct_ctx* ctx, const uint8_t* chain_linear);
int vct_download_chain_rgba8(vct_ctx* ctx, uint8_t* chain_linear);
/* config.anisotropic_mips = 1: the six directional chains built by the last vct_build_mips / vct_bounce,
 * [6][chain_texels - V^3][4] bytes, direction = 2*axis + (0: towards +axis, 1: towards -axis), level k
 * of a direction at texel offset level_offset(k) - V^3, linear layout. */
int vct_download_aniso_rgba8(vct_ctx* ctx, uint8_t* aniso_linear);
size_t vct_chain_texels(int32_t voxel_dim);

/* Render (VCT.h:146-190) -> trace.fs:165-228.  Traces the G-buffer through the brick chain and
 * writes the frame as RGBA16F, row-major width*height*4 halves (8 B / px).  out location follows
 * out_location (vct_mem).  A HOST G-buffer is uploaded (and tiled) first. */
int vct_trace(vct_ctx* ctx, const vct_gbuffer* gb, void* out_rgba16f, int32_t out_location);
/* Screen-tile slab [tile_row0, tile_row1) of the same frame (multi-GPU sharding): only those
 * 8-pixel tile rows are traced; out addresses the full frame and only the slab's rows are
 * written. */
int vct_trace_slab(vct_ctx* ctx, const vct_gbuffer* gb, int32_t tile_row0, int32_t tile_row1,
                   void* out_rgba16f, int32_t out_location);
/* Re-run the trace kernel on the G-buffer already resident from the last vct_trace (no upload,
 * no download); used for timing.  stream work only, asynchronous. */
int vct_trace_resident(vct_ctx* ctx);
/* Same for the tile-row slab [tile_row0, tile_row1) of the resident G-buffer; later
 * vct_trace_resident calls repeat this slab. */
int vct_trace_resident_rows(vct_ctx* ctx, int32_t tile_row0, int32_t tile_row1);
/* every `stride`-th tile row of [tile_row0, tile_row1), starting with tile_row0 (the rows rank tile_row0 of `stride`
 * ranks traces under interleaved slabs, vct_comm_set_interleaved); the pixels land at their own place in the frame.
 * (No reference counterpart: the reference draws one full-screen quad on one GPU, R/main.cpp:77-94.) */
int vct_trace_resident_strided(vct_ctx* ctx, int32_t tile_row0, int32_t tile_row1, int32_t stride);
/* One whole GI pass for a light AND a camera that moved -- init_voxel_cone_tracing's DrawDepthTexture +
 * DrawVoxelTexture (VCT.h:138-139) followed by Render (VCT.h:146-190) -- issued as one call:
 *   { shadow map -> voxelize(mode) -> inject -> mips }  ||  { G-buffer visibility -> (shadow map ready) -> shade }  -> trace.
 * The main draw's visibility raster needs nothing of this pass and its shading kernel only READS the shadow map, so
 * the G-buffer stage runs on a second HIP stream beside the shadow pass and the voxel stages (none of them fills the
 * GPU on its own; the two raster passes have their own work lists); the trace waits for both.  The
 * frame is bit-identical to vct_render_shadow_map, vct_voxelize, vct_inject_light, vct_build_mips,
 * vct_render_gbuffer, vct_trace_resident called in that order.  Asynchronous (vct_synchronize).
 * On a rank of a multi-GPU frame (after vct_comm_init) the same call runs the rank's share: the G-buffer stream is
 * scissored to the rank's slab (vct_render_gbuffer_rows) and the pass ends with vct_frame_step -- slab trace + the
 * frame's one gather -- instead of the full-frame trace; collective like vct_frame_step (vct_comm_sync to wait). */
int vct_gi_pass(vct_ctx* ctx, const float light_vp[16], const float view_proj[16], int32_t voxelize_mode);
/* Redirect the trace kernel's RGBA16F output to caller-owned HBM (full-frame addressing: pixel (x,y)
 * at ((y*width + x) * 4) halves from `rgba16f_dev`); NULL restores the context-owned frame.  A slab
 * rank passes its gather buffer minus the slab's first row so the kernel writes the gather buffer
 * directly (no device-to-device copy per frame).  The caller keeps the memory alive. */
int vct_set_frame_target(vct_ctx* ctx, void* rgba16f_dev);
/* Host copy of the frame the trace kernels write (the context-owned frame, or the target set above with
 * full-frame addressing): width*height*4 halves.  Rows no trace has written keep their old content. */
int vct_download_frame(vct_ctx* ctx, void* out_rgba16f_host);
int vct_synchronize(vct_ctx* ctx);

/* ---- two frames in flight (round 6) -----------------------------------------------------------------
 * The reference's Render() (VCT.h:146-190) issues GL commands; the driver starts frame k + 1 while frame k
 * drains -- nothing in R/main.cpp:77-94 waits for a frame.  A HIP stream does wait: each whole-frame trace
 * launch pays ~20 us of ramp + drain and a dispatch gap before the next kernel of its stream (4-5 % of a
 * 0.61 ms frame).  With n = 2 the context owns two FRAME SLOTS -- stream, G-buffer, RGBA16F frame, step
 * counts, timing events each (190 MB + 17 MB more at 1080p) -- and vct_select_frame_slot(ctx, k & 1) before
 * frame k's vct_render_gbuffer / vct_trace_resident puts consecutive frames on alternate streams: frame
 * k + 1's raster and trace start while frame k's trace drains.  Every entry point works on the selected
 * slot (its G-buffer, its frame, its raster scratch, its vct_last_* values); the chain, shadow map and mesh are
 * shared and ordered by events inside the library:
 * a stage that writes shared state (uploads, vct_render_shadow_map, vct_inject_light, vct_build_mips,
 * vct_bounce, vct_gi_pass) first waits for everything the other slot has in flight, and the other slot's
 * next work waits for it (one cross-stream wait per slot selection: later producers of the same selection
 * skip theirs).  So passes that REWRITE the chain every frame -- vct_gi_pass with a moving light -- run one after
 * the other whatever the slot, and pay the cross-queue hand-over: 0.818 ms per pass on one slot, 0.842 on two
 * (configs[1]; configs[4] equal) -- two slots are for frames that share a chain (Render(): raster + trace).
 * Frames are bit-identical to the one-slot frames.  vct_synchronize waits for both
 * slots.  n = 1 (default) releases the second slot.  Not with config.debug_outputs or trace_variant 4.  A rank of a
 * multi-GPU frame may use it too: vct_frame_step traces on the selected slot's stream, so slab k + 1 starts while slab k
 * drains (a slab launch pays the same ~20 us as a whole frame: a third of an 8-way slab); vct_comm_sync waits for both.
 * Measured: with the round-5 kernel and the timing events of vct_last_trace_ms around every launch, configs[1] trace
 * 0.617 -> 0.591 ms per step, Render() 0.752 -> 0.725.  Most of that turned out to be the events themselves (~7 us of
 * dispatch gaps per launch, which a second stream hides): with vct_set_trace_timing(ctx, 0) one stream runs a
 * trace-only step in the kernel's own time (0.540 ms against 0.548 on two slots; a 1/8 slab step 0.112 either way) and
 * two slots keep 1-2 % only where frames are long (configs[4] trace 2.64 -> 2.58, configs[2] 2.42 -> 2.39, Render()
 * 0.70 -> 0.69) -- opt-in, default 1. */
int vct_set_frames_in_flight(vct_ctx* ctx, int32_t n);
/* streams_overlap: 1 when the second slot's stream was seen to run beside the first at set-up (HIP shares a few hardware
 * queues between a process' streams; the library probes candidates until one overlaps), 0: no such stream was found --
 * results are the same, there is just nothing gained. */
int vct_get_frames_in_flight(const vct_ctx* ctx, int32_t* n, int32_t* selected_slot, int32_t* streams_overlap);
int vct_select_frame_slot(vct_ctx* ctx, int32_t slot);

/* ---- multi-GPU: screen-tile slabs + ONE RCCL gather per frame (BASELINE.json config 4) -------------
 * No reference counterpart (R/main.cpp:77-94 drives one GL context).  One process and one context per
 * GPU; every rank holds the whole scene + chain (voxelize / inject / mips are replicated: cheaper than
 * broadcasting the chain), rasterises and traces only its slab of 8-pixel tile rows, and rank 0 receives
 * the frame through a single ncclGather of padded equal slabs (RCCL over xGMI; rccl.h ncclGather, in
 * place on the root).  Slab r = tile rows [r*per, min((r+1)*per, tiles_y)), per = ceil(tiles_y / world). */
#define VCT_COMM_ID_BYTES 128
int vct_slab_partition(int32_t height, int32_t world, int32_t rank, int32_t* tile_row0, int32_t* tile_row1,
                       int32_t* rows_per_rank);
/* ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to every rank out of band (file, pipe, MPI ...). */
int vct_comm_get_unique_id(void* id128);
/* ncclCommInitRank on the context's device; allocates the two gather buffers (the root's are whole
 * frames), a communication stream and events.  Collective: every rank must call it. */
/* All or nothing: on any failure (allocation, ncclCommInitRank) nothing stays attached to the context and the call
 * may be repeated.
 * VCT_COMM_MODE=direct in the environment (EXPERIMENTAL, default off; no reference counterpart): "direct slabs" -- no
 * RCCL.  The root publishes hipIpc handles of its two frame buffers in a POSIX shared-memory block named after the id,
 * every other rank maps them and its trace kernel stores its slab straight into the root's frame (8 B per pixel over
 * xGMI); the frame's exchange step is a pair of flags in that block (mapped into every rank's GPU) instead of a
 * collective: rank: wait "root is past frame f - 2" -> trace -> "slab f done"; root: trace -> wait for every slab.
 * Every wait has the communicator's deadline (counted in the device's own wall-clock rate) and polls the process' abort
 * word: vct_comm_destroy raises it, drains the streams that still hold queued waits / peer stores and only then closes
 * the mappings.  The root's frame buffers are fine-grained allocations (peers write them, the root's next kernel reads
 * them).  All other vct_comm_* / vct_frame_step calls work unchanged (equal,
 * load-aware and interleaved slabs); vct_comm_get_unique_id then returns 128 random bytes and vct_comm_info reports
 * version 0.  Ranks must be processes of one host.  Tested with several ranks on ONE GPU (tests/test_gpu_multi.py);
 * never run across GPUs -- no multi-GPU box was available to this build. */
int vct_comm_init(vct_ctx* ctx, const void* id128, int32_t rank, int32_t world);
int vct_comm_destroy(vct_ctx* ctx);
int vct_comm_slab(vct_ctx* ctx, int32_t* tile_row0, int32_t* tile_row1);
/* One frame, asynchronous: trace this rank's slab of the resident G-buffer straight into gather buffer k
 * (k alternates), then ONE ncclGather on the communication stream.  Two buffers let frame k+1 be traced while frame
 * k is gathered; measured on one GPU the two do NOT overlap (the gather's kernels queue behind the trace's waves), so
 * budget slab trace + ~23 us dependent dispatch + wire time per frame.  With two frame slots (vct_set_frames_in_flight)
 * the trace runs on the selected slot's stream: slab k + 1 starts while slab k drains (1/8-slab step on one GPU
 * 0.131 -> 0.120 ms).  Collective: every rank calls it once per frame. */
int vct_frame_step(vct_ctx* ctx);
/* Waits for this rank's trace and gather.  Compute still queued in front of the last step's exchange gets the
 * timeout to itself first (its "slab traced" event); then: a peer that died or hangs would keep every other rank inside the
 * collective forever: after the communicator's timeout (default 60 s; VCT_COMM_TIMEOUT_MS in the environment or
 * vct_comm_set_timeout_ms) or on an asynchronous RCCL error the communicator is ABORTED (ncclCommAbort) and the call
 * returns VCT_ERR_DEVICE; afterwards only vct_comm_destroy (then a new vct_comm_init) is accepted on it. */
int vct_comm_sync(vct_ctx* ctx);
int vct_comm_set_timeout_ms(vct_ctx* ctx, int32_t milliseconds);
/* Diagnostics (bench lines of N > 1 runs explain themselves with these): what RCCL reports about the communicator --
 * out[0] ncclCommCount, out[1] ncclCommUserRank, out[2] ncclCommCuDevice, out[3] ncclGetVersion (-1 where the loaded
 * RCCL lacks the entry point) -- and the device time of the last frame's exchange step alone (waits for it). */
int vct_comm_info(vct_ctx* ctx, int32_t out[4]);
int vct_comm_last_gather_ms(vct_ctx* ctx, float* ms);
/* Load-aware slabs (SURVEY.md 8e offers unequal assignment as an option): equal ROWS are not equal WORK -- rows
 * showing sky or near walls march fewer steps.  vct_last_row_steps returns the executed cone steps per 8-pixel tile
 * row of the last screen trace (rows outside a slab trace are 0; nrows = ceil(height / 8)); after summing the
 * ranks' histograms (any out-of-band reduction) vct_slab_partition_weighted cuts [0, tile_rows) into `world`
 * contiguous slabs of near-equal cost (starts[world + 1]), and vct_comm_set_slab_rows installs those boundaries on a
 * communicator (collective; NULL restores the equal partition).  Unequal slabs travel as one fused group of
 * ncclSend / ncclRecv, each slab straight to its rows of the root's frame -- still one exchange step per frame. */
int vct_last_row_steps(vct_ctx* ctx, uint64_t* rows, int32_t nrows);
int vct_slab_partition_weighted(const uint64_t* row_cost, int32_t tile_rows, int32_t world, int32_t* starts);
int vct_comm_set_slab_rows(vct_ctx* ctx, const int32_t* starts);
/* Collective: interleaved slabs -- tile row r belongs to rank r % world, so every rank samples the whole frame and the
 * slabs cost the same by construction (SURVEY.md 8e "interleaved tile assignment ... with a de-interleave after the
 * gather").  The frame still travels as ONE equal-count ncclGather; the root de-interleaves behind it.  on = 0 returns
 * to contiguous equal slabs.  A rank's G-buffer must cover the whole frame (vct_render_gbuffer; vct_gi_pass does). */
int vct_comm_set_interleaved(vct_ctx* ctx, int32_t on);
/* One-GPU check of the interleaved data path for any world size: strided + packed traces of every emulated rank, the
 * root's de-interleave, compared with the frame of one launch; *mismatches = differing pixels. */
int vct_selftest_interleaved(vct_ctx* ctx, int32_t world, uint64_t* mismatches);
/* Root only: the last gathered frame (device pointer valid until the next-but-one vct_frame_step) / a host copy. */
int vct_comm_frame(vct_ctx* ctx, void** rgba16f_dev, size_t* bytes);
int vct_comm_download_frame(vct_ctx* ctx, void* out_rgba16f_host);

/* Debug outputs of the last trace (config.debug_outputs = 1): steps [npix][7] uint8, cones
 * [npix][7][4] fp32, linear pixel order. */
int vct_download_steps(vct_ctx* ctx, uint8_t* steps);
int vct_download_cones(vct_ctx* ctx, float* cones);
/* Executed cone steps of the last trace (always counted). */
int vct_last_step_count(vct_ctx* ctx, uint64_t* steps);
/* Wave-level statistics of the last trace launch -- instrumented builds only (-DVCT_STATS=1; the
 * production library returns VCT_ERR_INVALID): [0] march-loop iterations executed by waves, [1] live
 * lanes summed over them (= executed cone steps), [2] level samples whose cooperative 4x4x4 block was
 * all zero (skipped), [3] served through the cooperative block, [4] served by the per-lane gather,
 * [5] live lanes in [4], [6] those of [4] whose live footprints would fit one block anchored at their minimum,
 * [7] blocks a greedy multi-anchor cover of [4] needs in total, [8..10] those of [4] it covers with <= 2 / 3 / 4
 * blocks, [11..15] reserved. */
int vct_last_trace_stats(vct_ctx* ctx, uint64_t out[16]);
/* Work-item counts behind the per-stage byte figures of bench.py (`stage_roofline`): [0] triangles uploaded,
 * [1] conservative fragments of the mesh at this grid size (the voxelizer's brick-sorted list), [2] reserved,
 * [3] brick slots = 8^3 bricks a fragment of the mesh can land in, [4] bricks level 0 shows after the last
 * resolve, [5] compute units reserved for the communication stream (VCT_COMM_RESERVED_CUS; 0 = none), [6] form of the
 * last main-draw visibility pass (0 none yet, 1 direct, 2 tile-binned: chosen per context by timing),
 * [7] work items of the voxelize pass (brick slots, the heavy ones cut into chunks of 4096 fragments).  Synchronises
 * the stream. */
int vct_get_stage_counts(vct_ctx* ctx, uint64_t out[8]);
/* Device time of the last trace kernel launch in milliseconds (HIP events on the ctx stream). */
int vct_last_trace_ms(vct_ctx* ctx, float* ms);
/* on = 0: march launches (trace, slab step, bounce) are no longer bracketed by the two timing events that
 * vct_last_trace_ms reads (it then reports VCT_ERR_INVALID for such a launch).  Default on.  The events cost a launch
 * ~7 us of dispatch gaps on this GPU -- configs[1], steps on one stream: 0.549 -> 0.542 ms; a 1/8 slab's 0.12 ms step
 * pays the same 7 us -- so a frame loop switches them off (the facade does; bench.py does for its timed region and
 * measures the kernel time in a separate loop with them on).  The exchange step of a multi-GPU frame keeps its own
 * pair (vct_comm_last_gather_ms) on the communication stream: leaving those out as well measured no different.
 * No reference counterpart (round 6). */
int vct_set_trace_timing(vct_ctx* ctx, int32_t on);
/* Raw handles for interop (torch tensors wrap these): HIP stream of the context and the
 * device pointers of the resident tiled G-buffer / RGBA16F frame. */
/* Self-test of the kernel's constant division (x / d as fma(x, r_hi, x * r_lo), r_hi + r_lo = 1/d to 48 bits):
 * runs it on the GPU over every fp32 x of its domain (x == +0 or 2^-100 <= |x| < inf, normal quotient)
 * next to the IEEE divide and returns the number of x whose quotient differs.  0 is the guarantee the
 * trace kernel relies on (vct_trace.hip shows why the march never leaves that domain in a way that
 * could change a result). */
int vct_selftest_const_divide(vct_ctx* ctx, float d, uint64_t* mismatches);
/* Round 6: the trace kernels fetch texels through typed-buffer loads (the level as an RGBA8 UNORM texel buffer) and take
 * the four floats the texture path returns instead of decoding the bytes themselves -- valid because that conversion
 * is bit for bit (float)c / 255.0f on gfx950.  This runs every byte value in every channel position through the same
 * load against the library's exact decode: *mismatches = channel values that differ (0 expected).  vct_create runs it
 * once per process and device and fails (VCT_ERR_DEVICE) on a device where the conversion is not exact. */
int vct_selftest_texel_buffer(vct_ctx* ctx, uint64_t* mismatches);
/* Self-test of the rasteriser's shared-reciprocal division (csrc/vct_raster.hip div_area: the barycentrics' x / area
 * as two FMA corrections of x * RN(1/area), correctly rounded by Markstein's theorem): `count` pseudo-random pairs of
 * the operands' form (integers of 1..53 bits scaled by 2^-16) next to the IEEE division; returns the number of pairs
 * whose quotients differ in any bit (0 expected; the raster stages' equality with the oracle rests on it). */
int vct_selftest_area_divide(vct_ctx* ctx, uint64_t seed, uint64_t count, uint64_t* mismatches);
int vct_get_stream(vct_ctx* ctx, void** hip_stream);
int vct_get_frame_device(vct_ctx* ctx, void** rgba16f_dev, size_t* bytes);

#ifdef __cplusplus
}
#endif
#endif
