// vct_ctx.h -- the context behind the C ABI (private to the library's translation units).
#ifndef VCT_CTX_H_
#define VCT_CTX_H_

#include <string>

#include "../../include/vct.h"
#include "vct_internal.h"

struct vct_comm;      // multi-GPU state (vct_multi.hip)

// A stream restricted to a range of the device's compute units (hipExtStreamCreateWithCUMask).  Used when
// VCT_COMM_RESERVED_CUS = k > 0 keeps the last k CUs away from the context's streams and hands exactly those to the
// multi-GPU step's communication stream: a running trace leaves another queue's kernels almost no wave slots
// (k_raster_mid 658 us instead of 60 beside a trace), and RCCL's gather kernel is such a queue.
hipError_t vct_create_masked_stream(hipStream_t* s, int device, int first_cu, int last_cu);

// Two frames in flight (vct_set_frames_in_flight, round 6).  A whole-frame trace launch pays ~20 us of ramp and drain
// (the last generation of workgroups leaves compute units idle, tools/quant_probe.py) plus the dispatch gap to the next
// kernel of its stream: 4-5 % of a 0.61 ms frame.  A renderer that starts frame k + 1 on a second stream while frame k
// drains gets that back (tools/pipe_probe.py: trace 0.626 -> 0.598 ms per frame, Render() 0.773 -> 0.738 at
// configs[1]) -- what the reference's GL driver does with consecutive frames of its command queue.  What a frame owns
// -- stream, G-buffer, output frame, per-tile step counts, timing events, "last launch" bookkeeping -- exists once per
// SLOT; vct_select_frame_slot swaps a slot's set into the context fields of the same names, so every entry point works
// on the selected slot unchanged; the main draw's raster scratch exists per slot too (set [2] below), so a frame's
// G-buffer pass needs nothing of the other frame's.  Everything else (chain, shadow map, mesh) is shared, ordered by
// events: a stage that WRITES shared state (uploads, shadow map, inject, mips, bounce) waits for everything the other
// slot has in flight (pipeline_join), and the next slot switch makes the other stream wait for that stage.
struct VctFrameSlot {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float* gb_tiled = nullptr;
    const float* gb_current = nullptr;
    uint16_t* frame = nullptr;
    uint16_t* frame_target = nullptr;
    uint32_t* tile_steps = nullptr;
    int last_row0 = 0, last_row1 = 0, last_row_stride = 1;
    bool have_trace = false, last_trace_compacted = false, last_was_screen_trace = false, have_gbuffer = false;
    bool last_trace_timed = false;      // the slot's last march launch was bracketed by its timing events
};

struct vct_ctx {
    vct_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    int reserved_cus = 0;             // VCT_COMM_RESERVED_CUS at vct_create: CUs kept for the communication stream
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;

    uint32_t* chain = nullptr;        // Morton chain (bounce 0: direct light)
    uint32_t* chain_b = nullptr;      // second chain (bounce 1), allocated by vct_bounce
    bool use_chain_b = false;         // the trace reads chain_b until the next vct_inject_light
    uint32_t* attr_albedo = nullptr;  // [V^3] resolved mean albedo, Morton order
    uint32_t* attr_normal = nullptr;  // [V^3] resolved mean normal (biased), Morton order
    bool mips_valid = true;           // levels >= 1 describe level 0 (a fresh chain is all zero)
    bool want_cells = false;          // vct_set_footprint_records / VCT_FOOTPRINT_RECORDS=1
    uint4* cells = nullptr;           // footprint records of the levels >= 1 of `chain` (32 B per texel of those levels)
    bool cells_valid = false;         // ... rebuilt by vct_build_mips / vct_upload_chain_rgba8
    bool attrs_valid = false;         // attr_albedo / attr_normal hold a resolve of the CURRENT mesh's pools (vct_bounce needs it)
    uint32_t* aniso = nullptr;        // [6][chain_texels - V^3] directional chains (cfg.anisotropic_mips)
    size_t chain_texels = 0;
    uint32_t* staging = nullptr;      // linear staging for up/downloads (size of level 0)
    int nlev = 0;

    float* gb_linear = nullptr;       // [23][w*h] staging
    float* gb_tiled = nullptr;        // [tiles][23][64]
    const float* gb_current = nullptr;   // tiled buffer the next resident trace reads
    uint16_t* frame = nullptr;        // RGBA16F [h][w][4]
    uint16_t* frame_target = nullptr; // caller-owned output (vct_set_frame_target) or null
    uint8_t* dbg_steps = nullptr;
    float* dbg_cones = nullptr;
    unsigned long long* step_counter = nullptr;   // [VCT_STEP_COUNTERS] atomic bank of the bounce kernels (memset before each bounce)
    uint32_t* tile_steps = nullptr;               // [tiles] executed steps per 8x8 tile of the screen trace
    unsigned long long* stats = nullptr;      // [8] march statistics of instrumented builds (VCT_STATS)
    uint32_t* vt_pix = nullptr;       // trace_variant 4: compaction list [tiles][64] + the virtual-tile counter behind it
    VctStep* steps_dev = nullptr;     // [2][VCT_MAX_STEPS]
    uint32_t* spread_lut = nullptr;   // [1024] spread3(i) << 2 (vct_trace.hip: dilated anchor coordinates by scalar load)
    int n_diffuse = 0, n_specular = 0;
    bool steps_dirty = true;
    bool fast_div = false;            // set by refresh_steps: constant divisors admit the FMA division
    int last_row0 = 0, last_row1 = 0;
    int last_row_stride = 1;          // the last screen trace took every last_row_stride-th tile row of [last_row0, last_row1)
    bool have_trace = false;
    // vct_set_trace_timing: bracket every march launch with the two timing events vct_last_trace_ms reads.  On by default
    // (every entry point keeps working); a frame loop switches it off -- the two events cost a launch ~7 us of dispatch
    // gaps on this GPU (one-stream step 0.549 -> 0.542 ms, a 1/8 slab's 0.12 ms step the same 7 us).
    bool time_traces = true, last_trace_timed = false;
    bool last_trace_compacted = false;    // ... of trace_variant 4: counts per virtual tile, no per-row histogram
    bool last_was_screen_trace = false;   // the step counters hold a screen trace (indexed by tile row), not a bounce
    bool have_gbuffer = false;        // a G-buffer is resident (uploaded by vct_trace or rendered)

    float cam[3] = {0.0f, 4.0f, 0.0f};        // VCT.h:8
    float light[3] = {0.0f, 1.0f, 0.25f};     // VCT.h:14

    // scene
    float* tri_pos = nullptr;
    int32_t* tri_mat = nullptr;
    int32_t* tri_alpha = nullptr;     // [ntri] alpha-test class per triangle (main draw), rebuilt when the mesh / textures change
    bool tri_alpha_dirty = true;
    float* mat_albedo = nullptr;
    int32_t ntri = 0, nmat = 0;
    uint32_t* shadow = nullptr;       // shadow-map words (vct_internal.h vct_shadow_depth), shadow_size^2
    int32_t shadow_size = 0;
    uint32_t shadow_ebase = 0;        // epoch base of the words the map currently shows
    uint2* shadow_tiles = nullptr;    // decoded (min, max) per dilated 8 x 8 tile of the current map (vct_launch_shadow_minmax)
    uint32_t shadow_passes = 0;       // shadow passes rasterised into this buffer since its last memset
    // raster input stages
    float* tri_nrm = nullptr;
    float* tri_tan = nullptr;
    float* tri_bit = nullptr;
    float* mat_specular = nullptr;
    // material textures (vct_upload_textures) + texture coordinates (vct_upload_mesh_uvs)
    float* tri_uv = nullptr;
    uint32_t* tex_texels = nullptr;
    VctTexDesc* tex_desc = nullptr;
    int32_t* mat_tex = nullptr;
    int32_t ntex = 0;
    // raster scratch.  Between passes every visibility word is all-ones and the counter set of the next pass is
    // zero: the kernels re-establish both themselves (vct_raster.hip run_visibility), so a pass launches no memset.
    // `raster_dirty` (a launch failed, or nothing is initialised yet) makes the next pass clear everything once.
    // Scratch SETS: [0] the shadow pass, [1] the main draw, [2] the main draw of the second frame slot (two frames in flight:
    // frame k + 1's G-buffer pass then needs nothing of frame k's and the two overlap, vct_capi.hip raster_set_of)
    unsigned long long* vis[2] = {nullptr, nullptr};   // 64-bit words of the main draw, per frame slot
    size_t vis_words[2] = {0, 0};
    // lists / counters / tile items exist twice, [0] for the shadow pass and [1] for the main draw, so that the main
    // draw's visibility raster can run on the second stream WHILE the shadow map is rasterised (vct_gi_pass)
    int32_t* raster_lists[3] = {nullptr, nullptr, nullptr};     // [2*ntri] wave list, [2*ntri] group list
    void* raster_recs[3] = {nullptr, nullptr, nullptr};         // [2*ntri] 96-byte set-up records handed from k_raster_vis to k_raster_mid
    uint32_t* raster_counts[3] = {nullptr, nullptr, nullptr};   // two sets of [tile work items, wave list, group list, pad]
    int raster_set[3] = {0, 0, 0};                        // the counter set the next pass of that kind uses
    bool raster_dirty[3] = {true, true, true};
    uint2* raster_items[3] = {nullptr, nullptr, nullptr};
    uint32_t raster_item_capacity[3] = {0, 0, 0};
    // tile-binned visibility (vct_raster.hip): scratch per pass kind ([0] shadow pass, [1] main draw), see VctRasterArgs
    // Which form the MAIN draw's visibility takes (the shadow pass is opaque and sparse: the direct form won every
    // measurement).  raster_mode 0 = auto: scenes without alpha-tested textures keep the direct form; otherwise the first
    // pass runs direct, the second -- over the same tile rows -- binned, both between events, and the faster one is kept
    // until the mesh or the textures change.  1 / 2 = VCT_RASTER_PATH=direct / binned (both passes), for A/B runs.
    int raster_mode = 0;
    int auto_rows[2] = {0, 0};         // tile rows of the timed sample pair
    int last_raster_form = 0;          // form of the last main-draw visibility pass: 1 direct, 2 tile-binned (vct_get_stage_counts [6])
    uint32_t bin_test_caps[2] = {0u, 0u};   // VCT_BIN_TEST_CAPS="records,entries": capacities REPORTED to the binned kernels (tests of the overflow paths)
    int auto_state = 0;                                // position in the sampling sequence (vct_capi.hip kAutoSeq); 6: all sampled
    int auto_choice = -1;                              // -1 undecided, 0 direct, 1 binned
    hipEvent_t ev_auto[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // begin / end of the four timed samples
    bool has_alpha_textures = false;
    void* bin_recs[3] = {nullptr, nullptr, nullptr};
    uint32_t bin_rec_cap[3] = {0, 0, 0};
    uint2* bin_entries[3] = {nullptr, nullptr, nullptr};
    uint32_t bin_entry_cap[3] = {0, 0, 0};
    uint32_t* bin_count[3] = {nullptr, nullptr, nullptr};       // count + cursor, [2 * bins * VCT_BIN_CSTRIDE]
    uint32_t bin_bins[3] = {0, 0, 0};
    uint4* bin_items[3] = {nullptr, nullptr, nullptr};
    uint32_t bin_item_cap[3] = {0, 0, 0};
    uint32_t* bin_huge[3] = {nullptr, nullptr, nullptr};        // huge list [VCT_BIN_HUGE_CAP] + two counter sets [16] behind it
    int bin_set[3] = {0, 0, 0};
    // second stream: vct_gi_pass runs the G-buffer raster beside the voxel stages
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_shadow = nullptr, ev_join = nullptr;
    float light_vp[16];
    unsigned long long* acc = nullptr;         // reference mode only, allocated on first use: [nslots][512][2] ((triangle + 1) << 32 | rgb)
    uint32_t* brick_slot = nullptr;            // [V^3/512] brick -> slot or VCT_NO_SLOT
    uint32_t nslots = 0;
    // voxelization plan (geometry only; built by vct_upload_triangles): the mesh's conservative fragments sorted by
    // brick slot, and the staging pool a pass resolves into; sparse-resolve state
    uint32_t* frag_sorted = nullptr;   // [n_frags] triangle << 9 | voxel inside the brick
    float2* frag_bary = nullptr;       // [n_frags] the fragment's barycentrics (geometry only: once per mesh, k_frag_geom)
    float* frag_alb = nullptr;         // [n_frags][3] the fragment's albedo (scenes with textures); built by the first
    bool frag_alb_dirty = true;        //   voxelize pass after the texture coordinates / textures changed
    uint32_t* tri_qnrm = nullptr;      // [ntri][3] quantised face normals (config.voxel_attributes)
    uint32_t n_frags = 0;
    uint32_t* slot_first = nullptr;    // [nslots + 1]
    uint32_t* slot_brick = nullptr;    // [nslots]
    void* vox_items = nullptr;         // [n_vox_items] uint4 work items of the voxelize pass (VctVoxParams::items)
    uint32_t n_vox_items = 0;
    uint32_t vox_chunk = VCT_VOX_CHUNK;
    unsigned long long* vox_acc2 = nullptr;       // HBM accumulators of the multi-chunk slots (+ attributes): chunks add, k_vox_resolve_multi resolves and re-zeroes
    unsigned long long* vox_acc2_attr = nullptr;
    uint32_t* vox_multi_slot = nullptr;          // [n_vox_multi] slot of every multi-chunk slot
    uint32_t n_vox_multi = 0;
    uint32_t* stage = nullptr;         // [nslots][512] RGBA8 of the pending north-star pass
    uint32_t* stage_albedo = nullptr;  // [nslots][512] (cfg.voxel_attributes)
    uint32_t* stage_normal = nullptr;
    uint32_t* plan = nullptr;          // [4] device counters used while planning
    uint32_t* brick_flags = nullptr;   // [V^3/512] touched in the pending pass
    uint32_t* brick_prev = nullptr;    // [V^3/512] touched in the pass level 0 currently shows
    uint32_t* mip_seen = nullptr;      // [V^3/512] bricks non-empty when the chain's mips were last built
    uint32_t* mip_seen_b = nullptr;    // same for the bounce chain
    uint32_t* bounce_list = nullptr;   // occupied-voxel list of the bounce (+1 counter word in front)
    uint32_t bounce_list_cap = 0;
    uint32_t* brick_over = nullptr;
    bool chain_sparse_ready = true;    // bricks outside mip_seen have all-zero ancestors (true for a fresh, zero-filled
                                       // chain; an upload clears it until a dense mip build over a resolved level 0)
    bool acc_pending = false;          // accumulators hold an unresolved voxelize pass
    int acc_mode = 0;                  // vct_voxelize_mode of that pass
    int32_t* ref_big = nullptr;        // reference mode: triangles left to the workgroup pass (+ counter)
    bool level0_dirty = false;         // level 0 was written by an upload: next resolve is dense
    vct_comm* comm = nullptr;          // multi-GPU slabs + gather (vct_comm_init)
    // frame slots (see VctFrameSlot): slots[cur_slot] is STALE while selected -- its live values are the context fields
    int frames_in_flight = 1;          // 1 or 2
    int cur_slot = 0;
    VctFrameSlot slots[2];
    hipEvent_t ev_xslot = nullptr;     // scratch event of the cross-slot waits
    bool slot_streams_overlap = false; // the second slot's stream was SEEN to run beside the first (vct_capi.hip streams_overlap)
    bool produced_since_switch = false;   // a stage that writes shared state ran on the selected slot's stream since the last switch
    // The other slot's stream receives work only while its slot is selected, so ONE join (drain) per selection orders
    // (waits for) everything it holds: later producers of the same selection skip theirs.  vct_gi_pass -- five producers
    // in a row -- paid five cross-queue waits per pass for nothing (0.860 ms against 0.817 on one slot).
    bool joined_since_switch = false, drained_since_switch = false;
};

// shared helpers (vct_capi.hip)
int vct_fail(vct_ctx* c, int code, const std::string& msg);
// trace kernel on the context stream, asynchronous; out_base (full-frame addressing) overrides the frame target when not null
// row_stride > 1: only every row_stride-th tile row from row0 on; pack_rows: those rows back to back in out_base (interleaved slabs)
int vct_launch_trace_rows(vct_ctx* c, int row0, int row1, uint16_t* out_base = nullptr, int row_stride = 1, bool pack_rows = false);
// a new stream that was SEEN to run beside `base` (HIP shares a few hardware queues between a process' streams; two
// streams on one queue execute in order): vct_capi.hip create_overlapping_stream
int vct_create_overlapping_stream(vct_ctx* c, hipStream_t base, hipStream_t* out, bool* overlaps);
int vct_tiles_x(const vct_ctx* c);
int vct_tiles_y(const vct_ctx* c);

#define HIP_TRY(c, expr)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return vct_fail((c), e_ == hipErrorOutOfMemory ? VCT_ERR_NOMEM : VCT_ERR_DEVICE, \
                            std::string(#expr) + ": " + hipGetErrorString(e_));              \
    } while (0)

#endif
