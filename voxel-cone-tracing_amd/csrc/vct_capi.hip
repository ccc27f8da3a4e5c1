// vct_capi.hip -- the C ABI declared in include/vct.h: context, HBM ownership, launch sequencing.
//
// Counterpart of the reference's orchestrator struct (R/Voxel_Cone_Tracing.h:11-252): where that
// owns GL object names and issues draws, this owns HBM buffers and issues HIP kernels on one
// stream.  No CPU fallback exists: every entry point that computes launches a kernel or fails.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "vct_ctx.h"
#include "vct_divisors.h"

bool vct_comm_rows(const vct_ctx* c, int* row0, int* row1);      // vct_multi.hip: slab of the attached communicator

namespace {

std::string g_create_error;

int fail(vct_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// glm::ortho / glm::lookAt(eye, origin, up) / mat4 product as the reference builds ProjX/Y/Z
// (VCT.h:128-134; glm defaults: right-handed, NDC z in [-1,1]); column-major, fp32, one rounding per
// operation -- the same operation order as the oracle's restatement.
void glm_ortho(float l, float r, float b, float t, float n, float f, float m[16]) {
    memset(m, 0, 64);
    m[0] = 2.0f / (r - l);
    m[5] = 2.0f / (t - b);
    m[10] = -2.0f / (f - n);
    m[12] = -(r + l) / (r - l);
    m[13] = -(t + b) / (t - b);
    m[14] = -(f + n) / (f - n);
    m[15] = 1.0f;
}
void glm_lookat_origin(const float eye[3], const float up[3], float m[16]) {
    auto norm3 = [](float v[3]) {
        const float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        v[0] = v[0] / l; v[1] = v[1] / l; v[2] = v[2] / l;
    };
    auto cross = [](const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
    };
    auto dot = [](const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    float f[3] = {0.0f - eye[0], 0.0f - eye[1], 0.0f - eye[2]}, s[3], u[3];
    norm3(f);
    cross(f, up, s);
    norm3(s);
    cross(s, f, u);
    memset(m, 0, 64);
    m[0] = s[0]; m[4] = s[1]; m[8] = s[2];
    m[1] = u[0]; m[5] = u[1]; m[9] = u[2];
    m[2] = -f[0]; m[6] = -f[1]; m[10] = -f[2];
    m[12] = -dot(s, eye); m[13] = -dot(u, eye); m[14] = dot(f, eye);
    m[15] = 1.0f;
}
void mat_mul(const float a[16], const float b[16], float o[16]) {      // column-major o = a * b
    float t[16];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += a[k * 4 + r] * b[c * 4 + k];
            t[c * 4 + r] = s;
        }
    memcpy(o, t, sizeof(t));
}

// The step sequence of trace.fs:90-104, evaluated with the reference's operation order:
//   dist = vs; while (dist < MAX) { diameter = max(vs, 2*t*dist); lod = log2(diameter/vs); ...
//   dist += diameter; }   and the [GL] textureLod level selection for that lod.
int build_steps(const vct_config& cfg, float tan_half, std::vector<VctStep>& out) {
    out.clear();
    const int maxl = vct_ilog2(cfg.voxel_dim);
    const float vs = cfg.grid_world_size / (float)cfg.voxel_dim;
    float dist = vs;
    while (dist < cfg.max_distance) {
        if ((int)out.size() >= VCT_MAX_STEPS) return -1;
        VctStep s;
        const float diameter = fmaxf(vs, 2.0f * tan_half * dist);
        const float lod = log2f(diameter / vs);
        s.dist = dist;
        s.occ_den = 1.0f + 0.03f * diameter;
        s.occ_rcp = 1.0f / s.occ_den;
        float lam = lod;
        if (!(lam > 0.0f)) {
            s.two_levels = 0; s.level = 0; s.level2 = 0; s.frac = 0.0f;
        } else {
            if (lam > (float)maxl) lam = (float)maxl;
            const float fl = floorf(lam);
            s.two_levels = 1;
            s.level = (int)fl;
            s.level2 = s.level + 1 > maxl ? maxl : s.level + 1;
            s.frac = lam - fl;
        }
        // frac == 0: the blend is fma(0, tri(level2), 1*tri(level)) = tri(level) exactly (texels are
        // finite and >= +0), so the second level need not be sampled.
        if (s.two_levels && s.frac == 0.0f) s.two_levels = 0;
        s.omf = 1.0f - s.frac;
        auto ref = [&](int level) {
            VctLevelRef r;
            const int lg = maxl - level;
            r.off = (uint32_t)vct_level_offset(cfg.voxel_dim, level);
            r.mask_x = 0x09249249u & (uint32_t)((1ull << (3 * lg)) - 1ull);
            r.fN = (float)(1 << lg);
            r.m = (1 << lg) - 1;
            return r;
        };
        s.l1 = ref(s.level);
        s.l2 = ref(s.level2);
        out.push_back(s);
        const float nd = dist + diameter;
        if (!(nd > dist)) return -1;   // would never terminate
        dist = nd;
    }
    return 0;
}

// The kernel divides by wave-uniform constants (half_G, the per-step occlusion denominators) with
// q = fma(x, r_hi, x * r_lo), r_hi + r_lo = 1/d to 48 bits (vct_trace.hip div_const) -- exact only for some divisors, so every divisor of
// a step table is first verified on the device against the IEEE divide over all fp32 inputs
// (divisors_verified below); structural preconditions: significand not all ones, d and 1/d normal.
// Anything else switches the kernel to the IEEE-divide instantiation.
bool divisor_ok(float d) {
    uint32_t b;
    memcpy(&b, &d, 4);
    const uint32_t e = (b >> 23) & 0xffu, m = b & 0x7fffffu;
    if (!(d > 0.0f) || e == 0xffu) return false;
    return m != 0x7fffffu && e >= 4 && e <= 250;   // d and 1/d both far from the subnormal range
}

// Is the kernel's constant division exact for divisor d (vct_trace.hip div_const<1>)?  The divisors of the BASELINE
// grids and apertures ship as a table (vct_divisors.h: verified on the device, and every entry re-verified by
// tests/test_gpu_parity.py::test_const_divide_exhaustive), so a fresh process pays nothing for them; any other
// divisor is checked exhaustively on the device the first time a step table uses it (k_divide_selftest, 2 ms per
// divisor, synchronous -- an aperture animated per frame pays it once per new divisor) and the verdict is cached for
// the life of the process.  The cache is shared by every context of the process (one host thread per context, so
// two GPUs' threads may race here): guarded by a mutex.
int divisor_verified(vct_ctx* c, float d, bool* ok) {
    static std::map<uint32_t, bool> cache;
    static std::mutex cache_lock;
    uint32_t bits;
    memcpy(&bits, &d, 4);
    const uint32_t* end = kVerifiedDivisors + sizeof(kVerifiedDivisors) / sizeof(kVerifiedDivisors[0]);
    if (std::binary_search(kVerifiedDivisors, end, bits)) { *ok = true; return VCT_OK; }
    {
        std::lock_guard<std::mutex> g(cache_lock);
        auto it = cache.find(bits);
        if (it != cache.end()) { *ok = it->second; return VCT_OK; }
    }
    HIP_TRY(c, hipMemsetAsync(c->stats, 0, 2 * sizeof(unsigned long long), c->stream));
    HIP_TRY(c, vct_launch_divide_selftest(d, c->stats, c->stream));
    unsigned long long bad = 1;
    HIP_TRY(c, hipMemcpyAsync(&bad, c->stats, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *ok = bad == 0ull;
    std::lock_guard<std::mutex> g(cache_lock);
    cache[bits] = *ok;
    return VCT_OK;
}

// ---- frame slots (vct_ctx.h VctFrameSlot) ----------------------------------------------------------------------------
// Does a candidate stream run BESIDE the context's stream?  HIP spreads a process' streams over a few hardware queues
// (four by default) and two streams that share one execute in order -- a second frame slot on such a stream buys nothing
// (measured: tools/pipe_probe.py with five streams alive, 0.632 ms per step against 0.592).  A spin kernel on the base
// stream stamps its end, a stamp kernel issued on the candidate right behind it stamps its start: the streams overlap
// iff the stamp's start precedes the spin's end.
__global__ void k_spin_stamp(long long ticks, unsigned long long* out) {
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    out[0] = wall_clock64();
}
__global__ void k_stamp(unsigned long long* out) { out[1] = wall_clock64(); }

int streams_overlap(vct_ctx* c, hipStream_t base, hipStream_t cand, bool* overlap) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
    HIP_TRY(c, hipMemsetAsync(c->stats, 0, 2 * sizeof(unsigned long long), base));
    HIP_TRY(c, hipStreamSynchronize(base));
    hipLaunchKernelGGL(k_spin_stamp, dim3(1), dim3(1), 0, base, (long long)khz * 3 / 10, c->stats);      // 0.3 ms
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, cand, c->stats);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(base));
    HIP_TRY(c, hipStreamSynchronize(cand));
    unsigned long long v[2] = {0, 0};
    HIP_TRY(c, hipMemcpy(v, c->stats, sizeof(v), hipMemcpyDeviceToHost));
    *overlap = v[1] != 0ull && v[1] < v[0];
    return VCT_OK;
}

// A new stream that demonstrably runs beside `base`: candidates are created until one overlaps (the rejected ones stay
// alive during the search so that the runtime hands out other hardware queues), at most 8; if none does, the last
// candidate is returned with *overlaps = false (correct all the same, nothing gained).  VCT_STREAM_PROBE=0: the first
// stream the runtime hands out, unprobed (A/B runs).
int create_overlapping_stream(vct_ctx* c, hipStream_t base, hipStream_t* out, bool* overlaps) {
    *out = nullptr;
    *overlaps = false;
    const char* pr = getenv("VCT_STREAM_PROBE");
    if (pr && pr[0] == '0') { HIP_TRY(c, hipStreamCreateWithFlags(out, hipStreamNonBlocking)); return VCT_OK; }
    hipStream_t rejected[8];
    int nrej = 0, rc = VCT_OK;
    while (nrej < 8) {
        hipStream_t cand = nullptr;
        const hipError_t e = hipStreamCreateWithFlags(&cand, hipStreamNonBlocking);
        if (e != hipSuccess) { rc = fail(c, VCT_ERR_DEVICE, std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(e)); break; }
        bool ov = false;
        rc = streams_overlap(c, base, cand, &ov);
        if (rc != VCT_OK) { (void)hipStreamDestroy(cand); break; }
        if (ov) { *out = cand; *overlaps = true; break; }
        rejected[nrej++] = cand;
    }
    if (rc == VCT_OK && !*out && nrej > 0) *out = rejected[--nrej];
    for (int k = 0; k < nrej; ++k) (void)hipStreamDestroy(rejected[k]);
    return rc;
}

void slot_save(const vct_ctx* c, VctFrameSlot& s) {
    s.stream = c->stream; s.ev0 = c->ev0; s.ev1 = c->ev1;
    s.gb_tiled = c->gb_tiled; s.gb_current = c->gb_current; s.frame = c->frame; s.frame_target = c->frame_target;
    s.tile_steps = c->tile_steps;
    s.last_row0 = c->last_row0; s.last_row1 = c->last_row1; s.last_row_stride = c->last_row_stride;
    s.have_trace = c->have_trace; s.last_trace_compacted = c->last_trace_compacted; s.last_trace_timed = c->last_trace_timed;
    s.last_was_screen_trace = c->last_was_screen_trace; s.have_gbuffer = c->have_gbuffer;
}
void slot_load(vct_ctx* c, const VctFrameSlot& s) {
    c->stream = s.stream; c->ev0 = s.ev0; c->ev1 = s.ev1;
    c->gb_tiled = s.gb_tiled; c->gb_current = s.gb_current; c->frame = s.frame; c->frame_target = s.frame_target;
    c->tile_steps = s.tile_steps;
    c->last_row0 = s.last_row0; c->last_row1 = s.last_row1; c->last_row_stride = s.last_row_stride;
    c->have_trace = s.have_trace; c->last_trace_compacted = s.last_trace_compacted; c->last_trace_timed = s.last_trace_timed;
    c->last_was_screen_trace = s.last_was_screen_trace; c->have_gbuffer = s.have_gbuffer;
}
// A stage that WRITES state both slots read (shadow map, chain, accumulators): on the GPU it waits for everything the
// other slot has in flight, and the next slot switch makes the other stream wait for it (produced_since_switch).
int pipeline_join(vct_ctx* c) {
    if (c->frames_in_flight < 2) return VCT_OK;
    c->produced_since_switch = true;
    if (c->joined_since_switch) return VCT_OK;      // (vct_ctx.h: nothing new can be on the other stream)
    HIP_TRY(c, hipEventRecord(c->ev_xslot, c->slots[1 - c->cur_slot].stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_xslot, 0));
    c->joined_since_switch = true;
    return VCT_OK;
}
// Uploads free and reallocate buffers the other slot's kernels may still read: the host waits for that slot.
int pipeline_drain(vct_ctx* c) {
    if (c->frames_in_flight < 2) return VCT_OK;
    c->produced_since_switch = true;
    if (c->drained_since_switch) return VCT_OK;
    HIP_TRY(c, hipStreamSynchronize(c->slots[1 - c->cur_slot].stream));
    c->drained_since_switch = c->joined_since_switch = true;      // (a finished stream needs no GPU-side wait either)
    return VCT_OK;
}
#define PIPE_TRY(call)                    \
    do {                                  \
        const int rc_ = (call);           \
        if (rc_) return rc_;              \
    } while (0)

int refresh_steps(vct_ctx* c) {
    if (!c->steps_dirty) return VCT_OK;
    PIPE_TRY(pipeline_drain(c));      // the other slot's trace may still read the table that is rewritten below
    std::vector<VctStep> d, s;
    if (build_steps(c->cfg, c->cfg.tan_diffuse, d) || build_steps(c->cfg, c->cfg.tan_specular, s))
        return fail(c, VCT_ERR_INVALID, "cone aperture needs more than VCT_MAX_STEPS march steps");
    c->n_diffuse = (int)d.size();
    c->n_specular = (int)s.size();
    // preconditions of the kernel's FMA division (vct_trace.hip div_const): admissible divisors, and
    // occlusion numerators bounded away from the underflow range (blend factors 0 or >= 2^-10,
    // 1 - alpha >= 2^-5 while a cone is live)
    bool ok = divisor_ok(c->cfg.grid_world_size * 0.5f) && (1.0f - c->cfg.max_alpha) >= 0x1p-5f;
    auto blend_ok = [](const VctStep& st) {
        if (!st.two_levels) return true;
        return st.frac >= 0x1p-10f && (1.0f - st.frac) >= 0x1p-10f;
    };
    for (const VctStep& st : d) ok = ok && divisor_ok(st.occ_den) && blend_ok(st);
    for (const VctStep& st : s) ok = ok && divisor_ok(st.occ_den) && blend_ok(st);
    if (ok) {      // every divisor of the tables passes the device's exhaustive check of the kernel's division
        std::vector<float> divs = {c->cfg.grid_world_size * 0.5f};
        for (const VctStep& st : d) divs.push_back(st.occ_den);
        for (const VctStep& st : s) divs.push_back(st.occ_den);
        for (float dv : divs) {
            bool good = false;
            int rc = divisor_verified(c, dv, &good);
            if (rc) return rc;
            if (!good) { ok = false; break; }
        }
    }
    c->fast_div = ok;
    // a table for the verified division carries what div_const<1> takes beside the reciprocal (VCT_DIV2: its low word)
    // in the divisor's place; the IEEE-divide kernels of an unverified table keep the divisor
    if (ok) {
        for (VctStep& st : d) st.occ_den = vct_div_aux(st.occ_den, st.occ_rcp);
        for (VctStep& st : s) st.occ_den = vct_div_aux(st.occ_den, st.occ_rcp);
    }
    HIP_TRY(c, hipMemcpyAsync(c->steps_dev, d.data(), d.size() * sizeof(VctStep),
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->steps_dev + VCT_MAX_STEPS, s.data(), s.size() * sizeof(VctStep),
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // d, s go out of scope
    c->steps_dirty = false;
    return VCT_OK;
}

int tiles_x(const vct_ctx* c) { return (c->cfg.width + VCT_TILE - 1) / VCT_TILE; }
int tiles_y(const vct_ctx* c) { return (c->cfg.height + VCT_TILE - 1) / VCT_TILE; }
size_t gb_tiled_floats(const vct_ctx* c) {
    return (size_t)tiles_x(c) * tiles_y(c) * VCT_GB_NPLANES * VCT_TILE_PIX;
}

// everything the march needs (shared by the screen trace and the bounce)
void fill_march_params(const vct_ctx* c, VctTraceParams& p, const uint32_t* chain) {
    memset(&p, 0, sizeof(p));
    p.chain = chain;
    for (int l = 0; l < c->nlev; ++l) p.level_off[l] = (uint32_t)vct_level_offset(c->cfg.voxel_dim, l);
    p.V = c->cfg.voxel_dim;
    p.nlev = c->nlev;
    p.G = c->cfg.grid_world_size;
    p.half_G = c->cfg.grid_world_size * 0.5f;                       // trace.fs:61
    p.vs = c->cfg.grid_world_size / (float)c->cfg.voxel_dim;        // trace.fs:90
    p.half_G_rcp = 1.0f / p.half_G;
    p.half_G_aux = c->fast_div ? vct_div_aux(p.half_G, p.half_G_rcp) : p.half_G;
    p.fast_div = c->fast_div ? 1 : 0;
    p.max_alpha = c->cfg.max_alpha;
    p.wrap_repeat = c->cfg.wrap_repeat;
    p.spread_lut = c->spread_lut;
    // records of `c->chain` only (the bounce chain has none), biased by the first level's offset
    p.cells_biased = (c->cells_valid && chain == c->chain)
                         ? (const char*)c->cells - ((size_t)vct_level_offset(c->cfg.voxel_dim, 1) << 5) : nullptr;
    p.steps_diffuse = c->steps_dev;
    p.steps_specular = c->steps_dev + VCT_MAX_STEPS;
    p.n_diffuse = c->n_diffuse;
    p.n_specular = c->n_specular;
    p.step_counter = c->step_counter;
    p.tile_steps = c->tile_steps;
#if defined(VCT_STATS) && VCT_STATS
    p.stats = c->stats;
#endif
}

// `out_base`: where the kernel writes (full-frame addressing); null = the caller's vct_set_frame_target or the
// context-owned frame.  vct_frame_step passes its gather buffer here instead of re-pointing c->frame_target, which
// on a non-root rank would leave a pointer BEFORE a one-slab allocation behind for every later full-frame call.
int launch_trace(vct_ctx* c, int row0, int row1, uint16_t* out_base = nullptr, int row_stride = 1, bool pack_rows = false) {
    // the reference rebuilds the mips right after every voxelization (VCT.h:248); tracing a chain whose
    // coarse levels describe an older level 0 would return wrong GI without any sign of it
    if (!c->mips_valid)
        return fail(c, VCT_ERR_INVALID, "trace: level 0 changed since the last vct_build_mips (call it first)");
    int rc = refresh_steps(c);
    if (rc) return rc;
    if (c->cfg.debug_outputs && (c->n_diffuse > 255 || c->n_specular > 255))
        return fail(c, VCT_ERR_INVALID, "debug_outputs keeps per-cone step counts as uint8: this aperture needs more than 255 steps");
    VctTraceParams p;
    fill_march_params(c, p, c->use_chain_b ? c->chain_b : c->chain);
    for (int i = 0; i < 3; ++i) { p.cam[i] = c->cam[i]; p.light[i] = c->light[i]; }
    p.ambient = c->cfg.ambient_factor;
    p.shininess = c->cfg.shininess;
    p.width = c->cfg.width;
    p.height = c->cfg.height;
    p.tiles_x = tiles_x(c);
    p.tiles_y = tiles_y(c);
    p.tile_row0 = row0;
    p.tile_row1 = row1;
    p.row_stride = row_stride;
    p.pack_rows = pack_rows ? 1 : 0;
    p.ntiles = ((row1 - row0 + (row_stride > 1 ? row_stride : 1) - 1) / (row_stride > 1 ? row_stride : 1)) * tiles_x(c);
    if ((row_stride > 1 || pack_rows) && (c->cfg.trace_variant == 1 || c->cfg.trace_variant == 2 || c->cfg.trace_variant == 4))
        return fail(c, VCT_ERR_INVALID, "interleaved tile rows need the default trace kernel (config.trace_variant 0 or 3)");
    p.spec_prio = ((row1 - row0) / (row_stride > 1 ? row_stride : 1)) * 2 <= tiles_y(c) ? 1 : 0;
    const int variant = c->cfg.trace_variant;
    p.gbuf = c->gb_current;
    p.aniso = c->cfg.anisotropic_mips ? c->aniso : nullptr;
    p.aniso_alt_slab = 128;     // k_trace_tile_split<ANISO>: slabs [level 1][level 2][-axis of 1][-axis of 2]
    p.aniso_stride = (uint32_t)(c->chain_texels - (size_t)c->cfg.voxel_dim * c->cfg.voxel_dim * c->cfg.voxel_dim);
    p.out = out_base ? out_base : (c->frame_target ? c->frame_target : c->frame);
    p.dbg_steps = c->cfg.debug_outputs ? c->dbg_steps : nullptr;
    p.dbg_cones = c->cfg.debug_outputs ? c->dbg_cones : nullptr;
#if defined(VCT_STATS) && VCT_STATS
    HIP_TRY(c, hipMemsetAsync(c->stats, 0, 16 * sizeof(unsigned long long), c->stream));
#endif
    if (variant == 4 && !c->cfg.anisotropic_mips) {       // live-pixel compaction (experiment): list + counter, zeroed per launch
        const size_t nt = (size_t)tiles_x(c) * tiles_y(c);
        if (!c->vt_pix) HIP_TRY(c, hipMalloc(&c->vt_pix, (nt * 64 + 4) * sizeof(uint32_t)));
        p.vt_pix = c->vt_pix;
        p.vt_count = c->vt_pix + nt * 64;
        HIP_TRY(c, hipMemsetAsync(p.vt_count, 0, sizeof(uint32_t), c->stream));
    }
    if (c->time_traces) HIP_TRY(c, hipEventRecord(c->ev0, c->stream));      // (vct_set_trace_timing)
    HIP_TRY(c, vct_launch_trace(p, variant, c->stream));       // an empty row range (a rank without rows) launches nothing
    if (c->time_traces) HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    c->last_trace_timed = c->time_traces;
    c->last_row0 = row0;
    c->last_row1 = row1;
    c->last_row_stride = row_stride > 1 ? row_stride : 1;
    c->have_trace = true;
    c->last_was_screen_trace = true;
    c->last_trace_compacted = variant == 4 && !c->cfg.anisotropic_mips;
    return VCT_OK;
}

// textures are used once both the maps and the texture coordinates are there
VctTextures textures_of(const vct_ctx* c) {
    VctTextures t;
    memset(&t, 0, sizeof(t));
    if (c->tex_texels && c->tri_uv && c->mat_tex) {
        t.texels = c->tex_texels; t.desc = c->tex_desc; t.mat_tex = c->mat_tex; t.uv = c->tri_uv; t.ntex = c->ntex;
        t.mips = c->cfg.texture_mipmaps ? 1 : 0;
    }
    return t;
}

VctVoxParams vox_params(const vct_ctx* c) {
    VctVoxParams p;
    memset(&p, 0, sizeof(p));
    p.V = c->cfg.voxel_dim;
    p.G = c->cfg.grid_world_size;
    p.model_scale = c->cfg.model_scale;
    p.pos = c->tri_pos;
    p.material = c->tri_mat;
    p.albedo = c->mat_albedo;
    p.ntri = c->ntri;
    p.shadow = c->shadow;
    p.shadow_tiles = c->shadow ? c->shadow_tiles : nullptr;
    p.shadow_ebase = c->shadow_ebase;
    p.shadow_size = c->shadow_size;
    memcpy(p.light_vp, c->light_vp, 64);
    p.acc = c->acc;
    p.brick_slot = c->brick_slot;
    p.frag_sorted = c->frag_sorted;
    p.frag_bary = c->frag_bary;
    p.frag_alb = nullptr;       // vct_voxelize attaches it (scenes with textures)
    p.tri_qnrm = c->tri_qnrm;
    p.slot_first = c->slot_first;
    p.slot_brick = c->slot_brick;
    p.items = (const uint4*)c->vox_items;
    p.nitems = c->n_vox_items;
    p.chunk = c->vox_chunk;
    p.acc2 = c->vox_acc2;
    p.acc2_attr = c->vox_acc2_attr;
    p.multi_slot = c->vox_multi_slot;
    p.nmulti = c->n_vox_multi;
    p.nslots = c->nslots;
    p.stage = c->stage;
    p.stage_albedo = c->stage_albedo;
    p.stage_normal = c->stage_normal;
    p.brick_flags = c->brick_flags;
    p.tex = textures_of(c);
    return p;
}

void glm_voxel_projections(const vct_ctx* c, float proj[48]) {
    // VCT.h:128-134: ortho(-G/2, G/2, -G/2, G/2, G/2, 3G/2) * lookAt(+-G on the axis) per dominant axis
    const float G = c->cfg.grid_world_size, h = G * 0.5f;
    float o[16], v[16];
    glm_ortho(-h, h, -h, h, h, G * 1.5f, o);
    const float eye[3][3] = {{G, 0, 0}, {0, G, 0}, {0, 0, G}};
    const float up[3][3] = {{0, 1, 0}, {0, 0, -1}, {0, 1, 0}};
    for (int a = 0; a < 3; ++a) {
        glm_lookat_origin(eye[a], up[a], v);
        mat_mul(o, v, proj + 16 * a);
    }
}

// Drops the voxelization plan of the previous mesh (slots, sorted fragments, work items, staging pool, pooled
// attributes, reference-mode accumulators and big-triangle list): everything indexed by the OLD mesh's triangles or
// slots.  Called by vct_upload_triangles BEFORE any step that can fail, so that a failed upload leaves the context
// without a plan (vct_voxelize then refuses) instead of with the old mesh's plan over the new mesh's triangles.
// Level 0 / brick_prev keep describing what the chain shows.
void drop_voxel_plan(vct_ctx* c) {
    void** old[] = {(void**)&c->acc, (void**)&c->attr_albedo, (void**)&c->attr_normal, (void**)&c->brick_slot,
                    (void**)&c->frag_sorted, (void**)&c->slot_first, (void**)&c->slot_brick, (void**)&c->stage,
                    (void**)&c->stage_albedo, (void**)&c->stage_normal, (void**)&c->vox_items, (void**)&c->vox_acc2,
                    (void**)&c->vox_acc2_attr, (void**)&c->vox_multi_slot, (void**)&c->ref_big, (void**)&c->frag_bary,
                    (void**)&c->frag_alb, (void**)&c->tri_qnrm};
    for (void** q : old) if (*q) { (void)hipFree(*q); *q = nullptr; }
    c->frag_alb_dirty = true;
    c->nslots = 0;
    c->n_frags = 0;
    c->n_vox_items = 0;
    c->n_vox_multi = 0;
    c->acc_pending = false;
    c->attrs_valid = false;     // the pooled attributes are indexed by the NEW mesh's slots: nothing resolved into them yet
}

// Per-mesh brick slots.  The bricks a fragment of this mesh can land in depend only on geometry, V and G (not on the
// light, the shadow map or the textures), so they are found ONCE per upload -- from the conservative voxelizer's
// fragment list `frags` and a mark-only run of the reference-mode voxelizer -- and every marked brick gets a slot; the
// fragment list is then sorted by slot (counting sort) and the staging pool of the north-star pass is allocated (+ the
// voxel attributes when config.voxel_attributes).  Everything is allocated into locals and committed only when every
// allocation succeeded, so a failed hipMalloc leaves the context without a plan (vct_voxelize then reports it) instead
// of half-initialised.
int build_voxel_slots(vct_ctx* c, const uint2* frags, uint32_t nfrags) {
    const size_t nvox = (size_t)c->cfg.voxel_dim * c->cfg.voxel_dim * c->cfg.voxel_dim;
    const uint32_t nbricks = (uint32_t)(nvox / 512);
    drop_voxel_plan(c);         // (vct_upload_triangles already did, before its first fallible step)
    uint32_t *mark = nullptr, *slot = nullptr, *count = nullptr, *cnt = nullptr, *cursor = nullptr;
    uint32_t *sorted = nullptr, *first = nullptr, *slot_brick = nullptr, *stage = nullptr, *stage_albedo = nullptr,
             *stage_normal = nullptr, *attr_albedo = nullptr, *attr_normal = nullptr, *multi_slot = nullptr, *tri_nrm = nullptr;
    float2* bary = nullptr;
    void* items = nullptr;
    unsigned long long *acc2 = nullptr, *acc2_attr = nullptr;
    auto cleanup = [&]() {
        void* tmp[] = {mark, slot, count, cnt, cursor, sorted, first, slot_brick, stage, stage_albedo, stage_normal,
                       attr_albedo, attr_normal, multi_slot, items, acc2, acc2_attr, bary, tri_nrm};
        for (void* q : tmp) if (q) (void)hipFree(q);
    };
#define POOL_TRY(expr)                                                                                  \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            cleanup();                                                                                  \
            return fail(c, e_ == hipErrorOutOfMemory ? VCT_ERR_NOMEM : VCT_ERR_DEVICE,                  \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                             \
        }                                                                                               \
    } while (0)
    POOL_TRY(hipMalloc(&mark, (size_t)nbricks * sizeof(uint32_t)));
    POOL_TRY(hipMalloc(&slot, (size_t)nbricks * sizeof(uint32_t)));
    POOL_TRY(hipMalloc(&count, sizeof(uint32_t)));
    POOL_TRY(hipMemsetAsync(mark, 0, (size_t)nbricks * sizeof(uint32_t), c->stream));
    if (!c->ref_big) POOL_TRY(hipMalloc(&c->ref_big, ((size_t)c->ntri + 1) * sizeof(int32_t)));
    POOL_TRY(vct_launch_frag_mark(frags, nfrags, mark, c->stream));                  // north-star mode's fragments
    VctVoxParams p = vox_params(c);
    p.mark_only = 1;
    p.brick_mark = mark;
    p.shadow = nullptr;
    glm_voxel_projections(c, p.proj);
    POOL_TRY(vct_launch_voxelize_reference(p, c->ref_big + 1, c->ref_big, c->stream));   // reference mode's (mark only)
    POOL_TRY(vct_launch_assign_slots(mark, slot, count, nbricks, c->stream));
    uint32_t nslots = 0;
    POOL_TRY(hipMemcpyAsync(&nslots, count, sizeof(nslots), hipMemcpyDeviceToHost, c->stream));
    POOL_TRY(hipStreamSynchronize(c->stream));
    // counting sort of the fragments by slot: per-slot counts -> offsets (host prefix sum: once per mesh) -> scatter
    const size_t ns = nslots ? nslots : 1u;
    POOL_TRY(hipMalloc(&cnt, ns * sizeof(uint32_t)));
    POOL_TRY(hipMalloc(&cursor, ns * sizeof(uint32_t)));
    POOL_TRY(hipMalloc(&first, (ns + 1) * sizeof(uint32_t)));
    POOL_TRY(hipMalloc(&slot_brick, ns * sizeof(uint32_t)));
    POOL_TRY(hipMalloc(&sorted, (size_t)(nfrags ? nfrags : 1u) * sizeof(uint32_t)));
    POOL_TRY(hipMemsetAsync(cnt, 0, ns * sizeof(uint32_t), c->stream));
    POOL_TRY(hipMemsetAsync(cursor, 0, ns * sizeof(uint32_t), c->stream));
    POOL_TRY(hipMemsetAsync(slot_brick, 0, ns * sizeof(uint32_t), c->stream));
    POOL_TRY(vct_launch_slot_bricks(slot, nbricks, slot_brick, c->stream));      // every slot names its brick
    POOL_TRY(vct_launch_frag_count(frags, nfrags, slot, cnt, c->stream));
    std::vector<uint32_t> hcnt(ns, 0u), hfirst(ns + 1, 0u);
    POOL_TRY(hipMemcpyAsync(hcnt.data(), cnt, ns * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    POOL_TRY(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < ns; ++i) hfirst[i + 1] = hfirst[i] + hcnt[i];
    POOL_TRY(hipMemcpyAsync(first, hfirst.data(), (ns + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    POOL_TRY(vct_launch_frag_scatter(frags, nfrags, slot, first, cursor, sorted, slot_brick, c->stream));
    // the barycentrics of every fragment (geometry only): the pass reads them instead of re-deriving the triangle set-up
    POOL_TRY(hipMalloc(&bary, (size_t)(nfrags ? nfrags : 1u) * sizeof(float2)));
    {
        VctVoxParams q = vox_params(c);
        q.frag_sorted = sorted; q.slot_first = first; q.slot_brick = slot_brick; q.nslots = nslots;
        POOL_TRY(vct_launch_frag_geom(q, bary, nullptr, c->stream));
        if (c->cfg.voxel_attributes) {
            POOL_TRY(hipMalloc(&tri_nrm, (size_t)(c->ntri > 0 ? c->ntri : 1) * 3 * sizeof(uint32_t)));
            POOL_TRY(vct_launch_tri_nrm(q, tri_nrm, c->stream));
        }
    }
    // Work items of the pass, the heaviest slots first, slots above VCT_VOX_CHUNK fragments cut into chunks.  Fragments
    // per slot are uneven (atrium at 256^3: 425 on average, 1,918 at most; the street at 256^3: 4,138 and 23,128): with
    // one workgroup per slot in slot order the pass ended when an unluckily late heavy slot did.  Longest-first alone:
    // atrium 0.068 -> 0.049 ms, street at 1024^3 1.41 -> 1.27 ms; cutting the few very heavy slots as well: street at
    // 256^3 0.64 -> 0.31 ms.  (Smaller chunks cost more in accumulator atomics than they balance: 1024: 1.43 ms.)
    std::vector<uint32_t> order(ns);
    for (size_t i = 0; i < ns; ++i) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hcnt[a] > hcnt[b]; });
    std::vector<uint32_t> hitems;
    hitems.reserve(ns * 4 + 64);
    std::vector<uint32_t> hmulti;
    uint32_t nmulti = 0;
    const uint32_t CH = VCT_VOX_CHUNK;
    c->vox_chunk = CH;
    if (nslots)
        for (uint32_t sl : order) {
            const uint32_t chunks = hcnt[sl] ? (hcnt[sl] + CH - 1u) / CH : 1u;
            const uint32_t mi = chunks > 1u ? nmulti++ : 0xffffffffu;
            if (chunks > 1u) hmulti.push_back(sl);
            for (uint32_t ch = 0; ch < chunks; ++ch) {      // (slot, first fragment, fragments, multi index or ~0)
                const uint32_t nfr = hcnt[sl] - ch * CH < CH ? hcnt[sl] - ch * CH : CH;
                hitems.push_back(sl); hitems.push_back(hfirst[sl] + ch * CH); hitems.push_back(nfr); hitems.push_back(mi);
            }
        }
    const uint32_t nitems = (uint32_t)(hitems.size() / 4);
    POOL_TRY(hipMalloc(&items, (size_t)(nitems ? nitems : 1u) * 16));
    if (nitems) POOL_TRY(hipMemcpyAsync(items, hitems.data(), (size_t)nitems * 16, hipMemcpyHostToDevice, c->stream));
    const size_t nm = nmulti ? nmulti : 1u;
    POOL_TRY(hipMalloc(&acc2, nm * 512 * 2 * sizeof(unsigned long long)));
    POOL_TRY(hipMemsetAsync(acc2, 0, nm * 512 * 2 * sizeof(unsigned long long), c->stream));
    POOL_TRY(hipMalloc(&multi_slot, nm * sizeof(uint32_t)));
    if (nmulti) POOL_TRY(hipMemcpyAsync(multi_slot, hmulti.data(), (size_t)nmulti * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    if (c->cfg.voxel_attributes) {
        POOL_TRY(hipMalloc(&acc2_attr, nm * 512 * 3 * sizeof(unsigned long long)));
        POOL_TRY(hipMemsetAsync(acc2_attr, 0, nm * 512 * 3 * sizeof(unsigned long long), c->stream));
    }
    // staging pool of a pass (+ attributes): written in full by every pass, never cleared
    const size_t pool_vox = ns * 512;
    POOL_TRY(hipMalloc(&stage, pool_vox * 4));
    POOL_TRY(hipMemsetAsync(stage, 0, pool_vox * 4, c->stream));
    if (c->cfg.voxel_attributes) {
        POOL_TRY(hipMalloc(&stage_albedo, pool_vox * 4));
        POOL_TRY(hipMalloc(&stage_normal, pool_vox * 4));
        POOL_TRY(hipMalloc(&attr_albedo, pool_vox * 4));
        POOL_TRY(hipMalloc(&attr_normal, pool_vox * 4));
        POOL_TRY(hipMemsetAsync(stage_albedo, 0, pool_vox * 4, c->stream));
        POOL_TRY(hipMemsetAsync(stage_normal, 0, pool_vox * 4, c->stream));
        POOL_TRY(hipMemsetAsync(attr_albedo, 0, pool_vox * 4, c->stream));
        POOL_TRY(hipMemsetAsync(attr_normal, 0, pool_vox * 4, c->stream));
    }
    const size_t flag_bytes = (size_t)nbricks * sizeof(uint32_t);
    uint32_t** flags[] = {&c->brick_flags, &c->brick_prev, &c->mip_seen};
    for (uint32_t** f : flags)
        if (!*f) {
            POOL_TRY(hipMalloc(f, flag_bytes));
            POOL_TRY(hipMemsetAsync(*f, 0, flag_bytes, c->stream));
        }
    POOL_TRY(hipMemsetAsync(c->brick_flags, 0, flag_bytes, c->stream));
    POOL_TRY(hipStreamSynchronize(c->stream));
#undef POOL_TRY
    (void)hipFree(mark); (void)hipFree(count); (void)hipFree(cnt); (void)hipFree(cursor);
    c->attr_albedo = attr_albedo; c->attr_normal = attr_normal;
    c->brick_slot = slot;
    c->nslots = nslots;
    c->frag_sorted = sorted; c->n_frags = nfrags; c->slot_first = first; c->slot_brick = slot_brick;
    c->frag_bary = bary; c->tri_qnrm = tri_nrm; c->frag_alb_dirty = true;
    c->stage = stage; c->stage_albedo = stage_albedo; c->stage_normal = stage_normal;
    c->vox_items = items; c->n_vox_items = nitems; c->vox_acc2 = acc2; c->vox_acc2_attr = acc2_attr; c->vox_multi_slot = multi_slot; c->n_vox_multi = nmulti;
    return VCT_OK;
}

}  // namespace

int vct_fail(vct_ctx* c, int code, const std::string& msg) { return fail(c, code, msg); }

hipError_t vct_create_masked_stream(hipStream_t* s, int device, int first_cu, int last_cu) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return e;
    const int ncu = prop.multiProcessorCount;
    std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
    for (int cu = first_cu < 0 ? 0 : first_cu; cu < last_cu && cu < ncu; ++cu) mask[(size_t)cu / 32] |= 1u << (cu % 32);
    return hipExtStreamCreateWithCUMask(s, (uint32_t)mask.size(), mask.data());
}
int vct_launch_trace_rows(vct_ctx* c, int row0, int row1, uint16_t* out_base, int row_stride, bool pack_rows) {
    return launch_trace(c, row0, row1, out_base, row_stride, pack_rows);
}
int vct_create_overlapping_stream(vct_ctx* c, hipStream_t base, hipStream_t* out, bool* overlaps) {
    return create_overlapping_stream(c, base, out, overlaps);
}
int vct_tiles_x(const vct_ctx* c) { return tiles_x(c); }
int vct_tiles_y(const vct_ctx* c) { return tiles_y(c); }
void vct_comm_release(vct_ctx* c);      // vct_multi.hip

extern "C" {

int vct_default_config(vct_config* cfg) {
    if (!cfg) return VCT_ERR_INVALID;
    memset(cfg, 0, sizeof(*cfg));
    cfg->abi_version = VCT_ABI_VERSION;
    cfg->device = -1;
    cfg->voxel_dim = 128;            // VCT.h:16
    cfg->grid_world_size = 150.0f;   // VCT.h:17
    cfg->width = 1280;               // VCT.h:24
    cfg->height = 720;               // VCT.h:25
    cfg->shadow_map_size = 4096;     // VCT.h:35
    cfg->model_scale = 0.05f;        // VCT.h:183
    cfg->ambient_factor = 0.1f;      // VCT.h:53
    cfg->shininess = 20.0f;          // Mesh.h:86
    cfg->max_distance = 75.0f;       // trace.fs:43
    cfg->max_alpha = 0.95f;          // trace.fs:44
    cfg->tan_diffuse = 0.577f;       // trace.fs:198
    cfg->tan_specular = 0.07f;       // trace.fs:218
    cfg->wrap_repeat = 1;
    cfg->debug_outputs = 0;
    cfg->trace_variant = 0;
    cfg->voxel_attributes = 0;
    cfg->anisotropic_mips = 0;
    cfg->texture_mipmaps = 1;        // Model.h:168,172: glGenerateMipmap + LINEAR_MIPMAP_LINEAR
    return VCT_OK;
}

size_t vct_chain_texels(int32_t V) {
    if (!is_pow2(V)) return 0;
    return (size_t)vct_level_offset(V, vct_ilog2(V) + 1);
}

int vct_create(const vct_config* cfg, vct_ctx** out) {
    if (!cfg || !out) return fail(nullptr, VCT_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->abi_version != VCT_ABI_VERSION)
        return fail(nullptr, VCT_ERR_INVALID, "vct_config.abi_version mismatch");
    if (!is_pow2(cfg->voxel_dim) || cfg->voxel_dim < 8 || cfg->voxel_dim > 1024)
        return fail(nullptr, VCT_ERR_INVALID, "voxel_dim must be a power of two in [8,1024]");
    if (cfg->width <= 0 || cfg->height <= 0 || !(cfg->grid_world_size > 0.0f))
        return fail(nullptr, VCT_ERR_INVALID, "bad frame size or grid size");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, VCT_ERR_NO_DEVICE,
                    "no HIP device: this library has no CPU path (MI355X / gfx950 required)");
    vct_ctx* c = new vct_ctx();
    c->cfg = *cfg;
    int dev = cfg->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) { delete c; return fail(nullptr, VCT_ERR_INVALID, "device ordinal out of range"); }
    c->device = dev;
    c->cfg.device = dev;
#define CREATE_TRY(expr)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            std::string m = std::string(#expr) + ": " + hipGetErrorString(e_);               \
            vct_destroy(c);                                                                  \
            return fail(nullptr, e_ == hipErrorOutOfMemory ? VCT_ERR_NOMEM : VCT_ERR_DEVICE, m); \
        }                                                                                    \
    } while (0)
    CREATE_TRY(hipSetDevice(dev));
    {
        // VCT_STREAM_PRIORITY = high | low: experiments with two contexts sharing a GPU (tools/overlap_probe.py)
        int lo = 0, hi = 0;
        const char* pr = getenv("VCT_STREAM_PRIORITY");
        // VCT_COMM_RESERVED_CUS = k: the context's streams leave the device's last k compute units alone; the multi-GPU
        // step's communication stream gets exactly those (vct_ctx.h vct_create_masked_stream)
        const char* rc_ = getenv("VCT_COMM_RESERVED_CUS");
        c->reserved_cus = rc_ ? atoi(rc_) : 0;
        hipDeviceProp_t prop;
        CREATE_TRY(hipGetDeviceProperties(&prop, dev));
        if (c->reserved_cus < 0 || c->reserved_cus >= prop.multiProcessorCount) c->reserved_cus = 0;
        if (c->reserved_cus > 0)
            CREATE_TRY(vct_create_masked_stream(&c->stream, dev, 0, prop.multiProcessorCount - c->reserved_cus));
        else if (pr && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess)
            CREATE_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, pr[0] == 'h' ? hi : lo));
        else
            CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    CREATE_TRY(hipEventCreate(&c->ev0));
    CREATE_TRY(hipEventCreate(&c->ev1));
    {
        // VCT_RASTER_PATH=binned | direct: the tile-binned visibility (round 4) or the direct form of rounds 2-3 (one
        // device-scope atomicMin per covered pixel) for both raster passes -- same results.  Unset: chosen per scene by
        // measurement (vct_ctx.h raster_mode).
        const char* rp = getenv("VCT_RASTER_PATH");
        c->raster_mode = !rp ? 0 : (rp[0] == 'b' ? 2 : (rp[0] == 'd' ? 1 : 0));
        for (int k = 0; k < 8; ++k) CREATE_TRY(hipEventCreate(&c->ev_auto[k]));
        if (const char* fr = getenv("VCT_FOOTPRINT_RECORDS")) c->want_cells = fr[0] == '1';     // vct_set_footprint_records
        // VCT_BIN_TEST_CAPS="records,entries": the binned kernels are told these (smaller) capacities, so that a test can
        // drive the overflow paths -- sub-triangles rasterised in place, the merge by atomicMin -- on a small scene
        if (const char* tc = getenv("VCT_BIN_TEST_CAPS")) {
            unsigned a = 0, b = 0;
            if (sscanf(tc, "%u,%u", &a, &b) == 2) { c->bin_test_caps[0] = a; c->bin_test_caps[1] = b; }
        }
    }
    const int V = cfg->voxel_dim;
    c->nlev = vct_ilog2(V) + 1;
    c->chain_texels = vct_chain_texels(V);
    CREATE_TRY(hipMalloc(&c->chain, c->chain_texels * 4));
    CREATE_TRY(hipMemsetAsync(c->chain, 0, c->chain_texels * 4, c->stream));   // VCT.h:115-119
    const size_t npix = (size_t)cfg->width * cfg->height;
    CREATE_TRY(hipMalloc(&c->gb_tiled, gb_tiled_floats(c) * sizeof(float)));
    CREATE_TRY(hipMemsetAsync(c->gb_tiled, 0, gb_tiled_floats(c) * sizeof(float), c->stream));
    if (cfg->anisotropic_mips) {
        const size_t n = 6 * (c->chain_texels - (size_t)V * V * V);
        CREATE_TRY(hipMalloc(&c->aniso, n * 4));
        CREATE_TRY(hipMemsetAsync(c->aniso, 0, n * 4, c->stream));
    }
    CREATE_TRY(hipMalloc(&c->frame, npix * 8));
    CREATE_TRY(hipMemsetAsync(c->frame, 0, npix * 8, c->stream));
    CREATE_TRY(hipMalloc(&c->step_counter, VCT_STEP_COUNTERS * sizeof(unsigned long long)));
    CREATE_TRY(hipMemsetAsync(c->step_counter, 0, VCT_STEP_COUNTERS * sizeof(unsigned long long), c->stream));
    {
        const size_t nw = (size_t)tiles_x(c) * tiles_y(c);
        CREATE_TRY(hipMalloc(&c->tile_steps, nw * sizeof(uint32_t)));
        CREATE_TRY(hipMemsetAsync(c->tile_steps, 0, nw * sizeof(uint32_t), c->stream));
    }
    CREATE_TRY(hipMalloc(&c->stats, 16 * sizeof(unsigned long long)));
    CREATE_TRY(hipMemsetAsync(c->stats, 0, 16 * sizeof(unsigned long long), c->stream));
    CREATE_TRY(hipMalloc(&c->steps_dev, 2 * VCT_MAX_STEPS * sizeof(VctStep)));
    {
        std::vector<uint32_t> lut(1024);
        for (uint32_t i = 0; i < 1024u; ++i) lut[i] = vct_spread3(i) << 2;
        CREATE_TRY(hipMalloc(&c->spread_lut, lut.size() * sizeof(uint32_t)));
        CREATE_TRY(hipMemcpy(c->spread_lut, lut.data(), lut.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (cfg->debug_outputs) {
        CREATE_TRY(hipMalloc(&c->dbg_steps, npix * 7));
        CREATE_TRY(hipMalloc(&c->dbg_cones, npix * 28 * sizeof(float)));
        CREATE_TRY(hipMemsetAsync(c->dbg_steps, 0, npix * 7, c->stream));
        CREATE_TRY(hipMemsetAsync(c->dbg_cones, 0, npix * 28 * sizeof(float), c->stream));
    }
    CREATE_TRY(hipStreamSynchronize(c->stream));
#undef CREATE_TRY
    memset(c->light_vp, 0, sizeof(c->light_vp));
    c->light_vp[0] = c->light_vp[5] = c->light_vp[10] = c->light_vp[15] = 1.0f;
    c->gb_current = c->gb_tiled;
    {
        // once per process and device: the texture path's UNORM8 conversion must be the exact decode the trace kernels
        // count on (it is on gfx950).  No other path is compiled in: a device where it is not fails here, loudly.
        static std::mutex lock;
        static std::map<int, unsigned long long> verdict;
        std::lock_guard<std::mutex> g(lock);
        auto it = verdict.find(dev);
        if (it == verdict.end()) {
            uint64_t bad = 0;
            const int rc = vct_selftest_texel_buffer(c, &bad);
            if (rc != VCT_OK) { const std::string m = c->err; vct_destroy(c); return fail(nullptr, rc, m); }
            it = verdict.emplace(dev, bad).first;
        }
        if (it->second) {
            vct_destroy(c);
            return fail(nullptr, VCT_ERR_DEVICE, "this device's typed-buffer loads do not convert UNORM8 to exactly c / 255 (" +
                        std::to_string(it->second) + " of 4096 channel values differ): rebuild with -DVCT_HW_UNORM=0");
        }
    }
    *out = c;
    return VCT_OK;
}

void vct_destroy(vct_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    vct_comm_release(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->frames_in_flight > 1) {            // the slot that is not selected: its set is not in the context fields freed below
        VctFrameSlot& o = c->slots[1 - c->cur_slot];
        if (o.stream) (void)hipStreamSynchronize(o.stream);
        void* ob[] = {o.gb_tiled, o.frame, o.tile_steps};
        for (void* b : ob) if (b) (void)hipFree(b);
        if (o.ev0) (void)hipEventDestroy(o.ev0);
        if (o.ev1) (void)hipEventDestroy(o.ev1);
        if (o.stream) (void)hipStreamDestroy(o.stream);
    }
    if (c->ev_xslot) (void)hipEventDestroy(c->ev_xslot);
    void* bufs[] = {c->chain, c->cells, c->shadow_tiles, c->staging, c->gb_linear, c->gb_tiled, c->frame, c->dbg_steps,
                    c->dbg_cones, c->step_counter, c->tile_steps, c->stats, c->vt_pix, c->steps_dev, c->spread_lut, c->tri_pos,
                    c->tri_mat, c->tri_alpha, c->mat_albedo, c->shadow, c->acc, c->brick_slot, c->frag_sorted, c->slot_first, c->slot_brick, c->vox_items, c->vox_acc2, c->vox_acc2_attr, c->vox_multi_slot, c->stage,
                    c->stage_albedo, c->stage_normal, c->plan, c->frag_bary, c->frag_alb, c->tri_qnrm,
                    c->aniso, c->ref_big, c->brick_flags, c->brick_prev, c->mip_seen, c->mip_seen_b, c->bounce_list, c->brick_over, c->chain_b, c->attr_albedo, c->attr_normal,
                    c->tri_nrm, c->tri_tan, c->tri_bit, c->mat_specular, c->tri_uv, c->tex_texels, c->tex_desc, c->mat_tex, c->vis[0], c->vis[1], c->raster_lists[0], c->raster_lists[1], c->raster_lists[2],
                    c->raster_counts[0], c->raster_counts[1], c->raster_counts[2], c->raster_items[0], c->raster_items[1], c->raster_items[2],
                    c->raster_recs[0], c->raster_recs[1], c->raster_recs[2]};
    for (void* b : bufs) if (b) (void)hipFree(b);
    for (int k = 0; k < 3; ++k) {
        void* bb[] = {c->bin_recs[k], c->bin_entries[k], c->bin_count[k], c->bin_items[k], c->bin_huge[k]};
        for (void* b : bb) if (b) (void)hipFree(b);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_shadow) (void)hipEventDestroy(c->ev_shadow);
    for (int k = 0; k < 8; ++k) if (c->ev_auto[k]) (void)hipEventDestroy(c->ev_auto[k]);
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* vct_last_error(const vct_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int vct_get_config(const vct_ctx* c, vct_config* cfg) {
    if (!c || !cfg) return VCT_ERR_INVALID;
    *cfg = c->cfg;
    return VCT_OK;
}

int vct_set_camera_position(vct_ctx* c, const float pos[3]) {
    if (!c || !pos) return VCT_ERR_INVALID;
    memcpy(c->cam, pos, 12);
    return VCT_OK;
}

int vct_set_light_direction(vct_ctx* c, const float dir[3]) {
    if (!c || !dir) return VCT_ERR_INVALID;
    memcpy(c->light, dir, 12);
    return VCT_OK;
}

int vct_set_ambient_factor(vct_ctx* c, float a) {
    if (!c) return VCT_ERR_INVALID;
    c->cfg.ambient_factor = a;
    return VCT_OK;
}

int vct_set_cone_apertures(vct_ctx* c, float td, float ts) {
    if (!c) return VCT_ERR_INVALID;
    if (!(td > 0.0f) || !(ts > 0.0f)) return fail(c, VCT_ERR_INVALID, "aperture must be > 0");
    c->cfg.tan_diffuse = td;
    c->cfg.tan_specular = ts;
    c->steps_dirty = true;
    return VCT_OK;
}

int vct_set_footprint_records(vct_ctx* c, int32_t on);     // (defined next to vct_build_mips)

int vct_set_trace_variant(vct_ctx* c, int32_t variant) {
    if (!c) return VCT_ERR_INVALID;
    if (variant < 0 || variant > 4) return fail(c, VCT_ERR_INVALID, "vct_set_trace_variant: 0 .. 4");
    if (variant == 4 && c->frames_in_flight > 1)
        return fail(c, VCT_ERR_INVALID, "vct_set_trace_variant: variant 4 keeps per-context scratch (vct_set_frames_in_flight(ctx, 1) first)");
    c->cfg.trace_variant = variant;
    return VCT_OK;
}

// ---- scene -----------------------------------------------------------------------------

int vct_upload_triangles(vct_ctx* c, const float* pos, const int32_t* material, int32_t ntri,
                         const float* albedo, int32_t nmat) {
    if (!c) return VCT_ERR_INVALID;
    if (!pos || !material || !albedo || ntri <= 0 || nmat <= 0)
        return fail(c, VCT_ERR_INVALID, "vct_upload_triangles: null or empty input");
    for (int32_t i = 0; i < ntri; ++i)
        if (material[i] < 0 || material[i] >= nmat)
            return fail(c, VCT_ERR_INVALID, "vct_upload_triangles: material index out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    PIPE_TRY(pipeline_drain(c));
    if (c->tri_pos) { (void)hipFree(c->tri_pos); c->tri_pos = nullptr; }
    if (c->tri_mat) { (void)hipFree(c->tri_mat); c->tri_mat = nullptr; }
    if (c->tri_alpha) { (void)hipFree(c->tri_alpha); c->tri_alpha = nullptr; }
    c->tri_alpha_dirty = true;
    if (c->mat_albedo) { (void)hipFree(c->mat_albedo); c->mat_albedo = nullptr; }
    for (int k = 0; k < 3; ++k) {
        if (c->raster_lists[k]) { (void)hipFree(c->raster_lists[k]); c->raster_lists[k] = nullptr; }   // sized by ntri
        if (c->raster_recs[k]) { (void)hipFree(c->raster_recs[k]); c->raster_recs[k] = nullptr; }
        void** bb[] = {&c->bin_recs[k], (void**)&c->bin_entries[k], (void**)&c->bin_items[k]};
        for (void** b : bb) if (*b) { (void)hipFree(*b); *b = nullptr; }
        c->bin_rec_cap[k] = c->bin_entry_cap[k] = c->bin_item_cap[k] = 0u;
    }
    c->auto_state = 0; c->auto_choice = -1;       // the raster form is measured again for the new mesh
    drop_voxel_plan(c);     // before anything below can fail: the old plan indexes the old mesh
    float** frames[4] = {&c->tri_nrm, &c->tri_tan, &c->tri_bit, &c->tri_uv};       // belong to the old mesh
    for (float** f : frames) if (*f) { (void)hipFree(*f); *f = nullptr; }
    if (c->mat_tex) { (void)hipFree(c->mat_tex); c->mat_tex = nullptr; }           // indexed by the old materials
    if (ntri >= (1 << 23))
        return fail(c, VCT_ERR_INVALID, "vct_upload_triangles: more than 2^23 - 1 triangles (the voxelizer's fragment "
                                        "entries carry 23 bits of triangle index)");
    HIP_TRY(c, hipMalloc(&c->tri_pos, (size_t)ntri * 9 * sizeof(float)));
    HIP_TRY(c, hipMalloc(&c->tri_mat, (size_t)ntri * sizeof(int32_t)));
    HIP_TRY(c, hipMalloc(&c->mat_albedo, (size_t)nmat * 4 * sizeof(float)));
    if (!c->plan) HIP_TRY(c, hipMalloc(&c->plan, 4 * sizeof(uint32_t)));
    HIP_TRY(c, hipMemcpyAsync(c->tri_pos, pos, (size_t)ntri * 9 * sizeof(float),
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->tri_mat, material, (size_t)ntri * sizeof(int32_t),
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->mat_albedo, albedo, (size_t)nmat * 4 * sizeof(float),
                              hipMemcpyHostToDevice, c->stream));
    c->ntri = ntri;
    c->nmat = nmat;
    // Voxelization plan (geometry only): the exact conservative fragments of every triangle -- count, allocate, fill;
    // triangles with a huge bounding box are enumerated by a workgroup each -- then sorted by brick (build_voxel_slots).
    VctVoxParams p = vox_params(c);
    uint32_t counts[4] = {0, 0, 0, 0};
    int32_t* big_tmp = nullptr;
    uint2* frags = nullptr;
    auto plan_fail = [&](hipError_t e, const char* what) {
        if (big_tmp) (void)hipFree(big_tmp);
        if (frags) (void)hipFree(frags);
        return fail(c, e == hipErrorOutOfMemory ? VCT_ERR_NOMEM : VCT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    };
#define PLAN_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return plan_fail(e_, #expr); } while (0)
    PLAN_TRY(hipMalloc(&big_tmp, (size_t)ntri * sizeof(int32_t)));
    PLAN_TRY(hipMemsetAsync(c->plan, 0, 4 * sizeof(uint32_t), c->stream));
    PLAN_TRY(vct_launch_vox_plan(p, c->plan, nullptr, big_tmp, false, c->stream));
    PLAN_TRY(hipMemcpyAsync(counts, c->plan, sizeof(counts), hipMemcpyDeviceToHost, c->stream));
    PLAN_TRY(hipStreamSynchronize(c->stream));
    const uint32_t n_small = counts[0];
    const int n_big = (int)counts[1];
    PLAN_TRY(vct_launch_vox_plan_big(p, big_tmp, n_big, c->plan, nullptr, false, c->stream));
    PLAN_TRY(hipMemcpyAsync(counts, c->plan, sizeof(counts), hipMemcpyDeviceToHost, c->stream));
    PLAN_TRY(hipStreamSynchronize(c->stream));
    const unsigned long long total = (unsigned long long)n_small + counts[2];
    // counts[3]: one of the 32-bit device counters wrapped while counting (k_vox_plan / k_vox_plan_big flag it) -- the
    // totals above would then look small
    if (total >= (1ull << 32) || counts[3] != 0u) {
        (void)hipFree(big_tmp);
        return fail(c, VCT_ERR_INVALID, "vct_upload_triangles: the mesh has 2^32 or more conservative fragments at this grid size");
    }
    PLAN_TRY(hipMalloc(&frags, (size_t)(total ? total : 1ull) * sizeof(uint2)));
    const uint32_t restart[4] = {0u, 0u, n_small, 0u};          // small triangles fill [0, n_small), the big ones behind
    PLAN_TRY(hipMemcpyAsync(c->plan, restart, sizeof(restart), hipMemcpyHostToDevice, c->stream));
    PLAN_TRY(vct_launch_vox_plan(p, c->plan, frags, nullptr, true, c->stream));
    PLAN_TRY(vct_launch_vox_plan_big(p, big_tmp, n_big, c->plan, frags, true, c->stream));
    PLAN_TRY(hipStreamSynchronize(c->stream));
#undef PLAN_TRY
    (void)hipFree(big_tmp);
    big_tmp = nullptr;
    const int rc = build_voxel_slots(c, frags, (uint32_t)total);
    (void)hipFree(frags);
    return rc;
}

static size_t shadow_tile_count(int S) { const size_t nb = ((size_t)S + 7) >> 3; return nb * nb; }

int vct_upload_shadow_map(vct_ctx* c, const float* depth, int32_t size, const float light_vp[16]) {
    if (!c) return VCT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    PIPE_TRY(pipeline_drain(c));
    if (c->shadow) { (void)hipFree(c->shadow); c->shadow = nullptr; c->shadow_size = 0; }
    if (c->shadow_tiles) { (void)hipFree(c->shadow_tiles); c->shadow_tiles = nullptr; }
    if (!depth) return VCT_OK;
    if (size <= 0 || !light_vp) return fail(c, VCT_ERR_INVALID, "vct_upload_shadow_map: bad size");
    // the map lives as shadow-map words (vct_internal.h): depths clamped to [0, 1] like a GL depth texture, epoch 0;
    // a later vct_render_shadow_map starts a fresh epoch cycle over this buffer (shadow_passes = 0: memset first)
    const size_t n = (size_t)size * size;
    float* tmp = nullptr;
    HIP_TRY(c, hipMalloc(&c->shadow, n * sizeof(uint32_t)));
    hipError_t e = hipMalloc(&tmp, n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(tmp, depth, n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = vct_launch_shadow_encode(tmp, c->shadow, n, 0u, c->stream);
    if (e == hipSuccess) e = hipMalloc(&c->shadow_tiles, shadow_tile_count(size) * sizeof(uint2));
    if (e == hipSuccess) e = vct_launch_shadow_minmax(c->shadow, 0u, size, c->shadow_tiles, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) {
        (void)hipFree(c->shadow); c->shadow = nullptr;
        if (c->shadow_tiles) { (void)hipFree(c->shadow_tiles); c->shadow_tiles = nullptr; }
        HIP_TRY(c, e);
    }
    c->shadow_size = size;
    c->shadow_ebase = 0u;
    c->shadow_passes = 0u;
    memcpy(c->light_vp, light_vp, 64);
    return VCT_OK;
}

// ---- raster input stages ------------------------------------------------------------------

// Scratch of one raster pass on stream `s`: `pixels` 64-bit visibility words (main draw) or 32-bit ones (depth_only).
// scratch set of the main draw: [1], or [2] in the second frame slot (vct_ctx.h)
static int raster_set_of(const vct_ctx* c) { return c->frames_in_flight > 1 && c->cur_slot == 1 ? 2 : 1; }

static int raster_args(vct_ctx* c, int side_w, int side_h, bool depth_only, bool binned, hipStream_t s, VctRasterArgs& a) {
    if (!c->tri_pos) return fail(c, VCT_ERR_INVALID, "no triangles uploaded");
    const size_t pixels = (size_t)side_w * side_h;
    const int k = depth_only ? 0 : raster_set_of(c);
    const int vs = k == 2 ? 1 : 0;                 // visibility words of this frame slot
    if (!depth_only && c->vis_words[vs] < pixels) {
        if (c->vis[vs]) { (void)hipFree(c->vis[vs]); c->vis[vs] = nullptr; c->vis_words[vs] = 0; }
        HIP_TRY(c, hipMalloc(&c->vis[vs], pixels * sizeof(unsigned long long)));
        c->vis_words[vs] = pixels;
        c->raster_dirty[k] = true;
    }
    const uint32_t bins = (uint32_t)(((size_t)side_w + 15) / 16 * (((size_t)side_h + 15) / 16));
    if (binned) {
        // Scratch of the tile-binned form (vct_raster.hip).  One 160-byte record per visible sub-triangle (back faces are
        // always culled, a near-clipped triangle is two sub-triangles: ntri + 4096 holds every scene that is not mostly
        // near-clipped) and one 8-byte entry per (sub-triangle, 16x16 bin) overlap -- 3.3 per visible sub-triangle on the
        // Bistro-class street at 4K -- plus the bins' counters.  Whatever does not fit takes the huge list or is
        // rasterised in place (k_bin_setup), so these are sizes, not limits.  ~200 B per triangle and pass kind,
        // replicated per rank of a multi-GPU frame.
        const uint32_t want_recs = (uint32_t)c->ntri + 4096u;
        const size_t want_ent_sz = (size_t)c->ntri * 4 + (size_t)bins * 4 + 65536;
        const uint32_t want_ent = (uint32_t)(want_ent_sz < 0x7fffffffull ? want_ent_sz : 0x7fffffffull);
        if (c->bin_rec_cap[k] < want_recs) {
            if (c->bin_recs[k]) { (void)hipFree(c->bin_recs[k]); c->bin_recs[k] = nullptr; c->bin_rec_cap[k] = 0u; }
            HIP_TRY(c, hipMalloc(&c->bin_recs[k], (size_t)want_recs * 160));
            c->bin_rec_cap[k] = want_recs;
        }
        if (c->bin_entry_cap[k] < want_ent) {
            if (c->bin_entries[k]) { (void)hipFree(c->bin_entries[k]); c->bin_entries[k] = nullptr; c->bin_entry_cap[k] = 0u; }
            HIP_TRY(c, hipMalloc(&c->bin_entries[k], ((size_t)want_ent + 1) * sizeof(uint2)));      // + the spare entry k_bin_fill's idle lanes write
            c->bin_entry_cap[k] = want_ent;
        }
        const uint32_t want_items = bins + c->bin_entry_cap[k] / 512u + 1u;      // sum over bins of ceil(entries / slice)
        if (c->bin_item_cap[k] < want_items) {
            if (c->bin_items[k]) { (void)hipFree(c->bin_items[k]); c->bin_items[k] = nullptr; c->bin_item_cap[k] = 0u; }
            HIP_TRY(c, hipMalloc(&c->bin_items[k], (size_t)want_items * sizeof(uint4)));
            c->bin_item_cap[k] = want_items;
        }
        if (c->bin_bins[k] < bins) {
            if (c->bin_count[k]) { (void)hipFree(c->bin_count[k]); c->bin_count[k] = nullptr; c->bin_bins[k] = 0u; }
            HIP_TRY(c, hipMalloc(&c->bin_count[k], (size_t)bins * 2 * VCT_BIN_CSTRIDE * sizeof(uint32_t)));
            c->bin_bins[k] = bins;
            c->raster_dirty[k] = true;
        }
        if (!c->bin_huge[k]) {
            HIP_TRY(c, hipMalloc(&c->bin_huge[k], (size_t)(VCT_BIN_HUGE_CAP + 16 + 32) * sizeof(uint32_t)));
            c->raster_dirty[k] = true;
        }
    } else {
        if (!c->raster_lists[k]) HIP_TRY(c, hipMalloc(&c->raster_lists[k], (size_t)c->ntri * 4 * sizeof(int32_t)));
        if (!c->raster_recs[k]) HIP_TRY(c, hipMalloc(&c->raster_recs[k], (size_t)c->ntri * 2 * 96));
        if (!c->raster_counts[k]) { HIP_TRY(c, hipMalloc(&c->raster_counts[k], 8 * sizeof(uint32_t))); c->raster_dirty[k] = true; }
        // tile work items: 16x16-pixel pieces of large triangles; pixels/16 entries is ~16x the typical
        // demand (sum of visible bounding boxes ~ a few frames' worth of pixels); overflow is handled
        const size_t want_items = pixels / 16 + 4096;
        if (c->raster_item_capacity[k] < want_items) {
            if (c->raster_items[k]) { (void)hipFree(c->raster_items[k]); c->raster_items[k] = nullptr; }
            HIP_TRY(c, hipMalloc(&c->raster_items[k], want_items * sizeof(uint2)));
            c->raster_item_capacity[k] = (uint32_t)want_items;
        }
    }
    if (c->raster_dirty[k]) {   // first pass, resized buffers, or a pass that failed half way: clear this kind's state once
        if (depth_only) c->shadow_passes = 0u;      // the shadow words restart their epoch cycle with a memset (below)
        else HIP_TRY(c, hipMemsetAsync(c->vis[vs], 0xff, c->vis_words[vs] * sizeof(unsigned long long), s));
        if (binned) {
            HIP_TRY(c, hipMemsetAsync(c->bin_count[k], 0, (size_t)c->bin_bins[k] * 2 * VCT_BIN_CSTRIDE * sizeof(uint32_t), s));
            HIP_TRY(c, hipMemsetAsync(c->bin_huge[k] + VCT_BIN_HUGE_CAP, 0, (16 + 32) * sizeof(uint32_t), s));
        } else {
            HIP_TRY(c, hipMemsetAsync(c->raster_counts[k], 0, 8 * sizeof(uint32_t), s));
        }
        c->raster_dirty[k] = false;
    }
    memset(&a, 0, sizeof(a));
    a.binned = binned ? 1 : 0;
    if (binned) {
        a.bin_recs = c->bin_recs[k]; a.bin_rec_cap = c->bin_rec_cap[k];
        a.bin_entries = c->bin_entries[k]; a.bin_entry_cap = c->bin_entry_cap[k];
        if (c->bin_test_caps[0] && c->bin_test_caps[0] < a.bin_rec_cap) a.bin_rec_cap = c->bin_test_caps[0];
        if (c->bin_test_caps[1] && c->bin_test_caps[1] < a.bin_entry_cap) a.bin_entry_cap = c->bin_test_caps[1];
        a.bin_count = c->bin_count[k]; a.bin_cursor = c->bin_count[k] + (size_t)c->bin_bins[k] * VCT_BIN_CSTRIDE;
        a.bin_items = c->bin_items[k]; a.bin_item_cap = c->bin_item_cap[k];
        a.bin_huge = c->bin_huge[k]; a.bin_huge_cap = VCT_BIN_HUGE_CAP;
        a.bin_ctr = c->bin_huge[k] + VCT_BIN_HUGE_CAP + 8 * c->bin_set[k];
        a.bin_next_ctr = c->bin_huge[k] + VCT_BIN_HUGE_CAP + 8 * (c->bin_set[k] ^ 1);
        c->bin_set[k] ^= 1;
    }
    a.pos = c->tri_pos;
    a.nrm = c->tri_nrm; a.tan = c->tri_tan; a.bit = c->tri_bit;
    a.material = c->tri_mat;
    a.albedo = c->mat_albedo;
    a.specular = c->mat_specular;
    a.ntri = c->ntri;
    a.model_scale = c->cfg.model_scale;
    a.vis = c->vis[vs];
    a.vis32 = nullptr;          // vct_render_shadow_map points it at the shadow-map words
    a.vis32_ebase = 0u;
    if (!binned) {
        a.items = c->raster_items[k];
        // [wave, group, item, -]: the wave and group counters are one 8-byte-aligned pair, k_raster_vis reserves both
        // lists of a workgroup with a single 64-bit atomic
        uint32_t* cur = c->raster_counts[k] + 4 * c->raster_set[k];
        a.wave_list = c->raster_lists[k];
        a.wave_count = cur;
        a.group_list = c->raster_lists[k] + (size_t)c->ntri * 2;
        a.group_count = cur + 1;
        a.item_count = cur + 2;
        a.next_counts = c->raster_counts[k] + 4 * (c->raster_set[k] ^ 1);
        a.recs = c->raster_recs[k];
        c->raster_set[k] ^= 1;
        a.item_capacity = c->raster_item_capacity[k];
    }
    a.tex = textures_of(c);
    if (!depth_only) {          // the main draw's alpha-test class per triangle: once per mesh / texture set
        if (!c->tri_alpha) { HIP_TRY(c, hipMalloc(&c->tri_alpha, (size_t)c->ntri * sizeof(int32_t))); c->tri_alpha_dirty = true; }
        if (c->tri_alpha_dirty) {
            HIP_TRY(c, vct_launch_tri_alpha(a, c->tri_alpha, s));
            c->tri_alpha_dirty = false;
            c->produced_since_switch = true;      // (shared by both frame slots: the other slot's next pass follows this one)
        }
        a.tri_alpha = c->tri_alpha;
    }
    return VCT_OK;
}

int vct_upload_mesh_attributes(vct_ctx* c, const float* normal, const float* tangent,
                               const float* bitangent, const float* specular) {
    if (!c) return VCT_ERR_INVALID;
    if (!normal || !tangent || !bitangent || !specular)
        return fail(c, VCT_ERR_INVALID, "vct_upload_mesh_attributes: null input");
    if (!c->tri_pos) return fail(c, VCT_ERR_INVALID, "vct_upload_mesh_attributes: call vct_upload_triangles first");
    HIP_TRY(c, hipSetDevice(c->device));
    PIPE_TRY(pipeline_drain(c));
    float** dst[3] = {&c->tri_nrm, &c->tri_tan, &c->tri_bit};
    const float* src[3] = {normal, tangent, bitangent};
    const size_t bytes = (size_t)c->ntri * 9 * sizeof(float);
    for (int k = 0; k < 3; ++k) {
        if (*dst[k]) { (void)hipFree(*dst[k]); *dst[k] = nullptr; }
        HIP_TRY(c, hipMalloc(dst[k], bytes));
        HIP_TRY(c, hipMemcpyAsync(*dst[k], src[k], bytes, hipMemcpyHostToDevice, c->stream));
    }
    if (c->mat_specular) { (void)hipFree(c->mat_specular); c->mat_specular = nullptr; }
    HIP_TRY(c, hipMalloc(&c->mat_specular, (size_t)c->nmat * 3 * sizeof(float)));
    HIP_TRY(c, hipMemcpyAsync(c->mat_specular, specular, (size_t)c->nmat * 3 * sizeof(float),
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCT_OK;
}

int vct_upload_mesh_uvs(vct_ctx* c, const float* uv) {
    if (!c) return VCT_ERR_INVALID;
    if (!uv) return fail(c, VCT_ERR_INVALID, "vct_upload_mesh_uvs: null input");
    if (!c->tri_pos) return fail(c, VCT_ERR_INVALID, "vct_upload_mesh_uvs: call vct_upload_triangles first");
    HIP_TRY(c, hipSetDevice(c->device));
    PIPE_TRY(pipeline_drain(c));
    if (c->tri_uv) { (void)hipFree(c->tri_uv); c->tri_uv = nullptr; }
    const size_t bytes = (size_t)c->ntri * 6 * sizeof(float);
    c->tri_alpha_dirty = true;          // textures take effect once the coordinates are there
    c->frag_alb_dirty = true;
    HIP_TRY(c, hipMalloc(&c->tri_uv, bytes));
    HIP_TRY(c, hipMemcpyAsync(c->tri_uv, uv, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCT_OK;
}

int vct_upload_textures(vct_ctx* c, const uint8_t* const* rgba8, const int32_t* width, const int32_t* height,
                        int32_t ntex, const int32_t* mat_tex) {
    if (!c) return VCT_ERR_INVALID;
    if (!c->tri_pos) return fail(c, VCT_ERR_INVALID, "vct_upload_textures: call vct_upload_triangles first");
    HIP_TRY(c, hipSetDevice(c->device));
    PIPE_TRY(pipeline_drain(c));
    if (c->tex_texels) { (void)hipFree(c->tex_texels); c->tex_texels = nullptr; }
    if (c->tex_desc) { (void)hipFree(c->tex_desc); c->tex_desc = nullptr; }
    if (c->mat_tex) { (void)hipFree(c->mat_tex); c->mat_tex = nullptr; }
    c->ntex = 0;
    c->tri_alpha_dirty = true;
    c->frag_alb_dirty = true;
    c->has_alpha_textures = false;
    c->auto_state = 0; c->auto_choice = -1;
    if (ntex == 0) return VCT_OK;                     // detach: flat colours again
    if (ntex < 0 || !rgba8 || !width || !height || !mat_tex)
        return fail(c, VCT_ERR_INVALID, "vct_upload_textures: null or negative input");
    // Packed buffer: every texture's level 0, followed (config.texture_mipmaps) by its mip chain down to 1 x 1 --
    // glGenerateMipmap (R/Model.h:168), built here on the GPU level by level (k_tex_mip).
    std::vector<VctTexDesc> desc((size_t)ntex);
    size_t total = 0;
    for (int32_t i = 0; i < ntex; ++i) {
        if (!rgba8[i] || width[i] <= 0 || height[i] <= 0 || width[i] > 16384 || height[i] > 16384)
            return fail(c, VCT_ERR_INVALID, "vct_upload_textures: bad texture size");
        VctTexDesc& d = desc[(size_t)i];
        memset(&d, 0, sizeof(d));
        d.off = (uint32_t)total;
        d.w = width[i];
        d.h = height[i];
        const size_t n = (size_t)width[i] * height[i];
        uint32_t flags = 0;
        for (size_t k = 0; k < n; ++k)
            if (rgba8[i][4 * k + 3] != 255) { flags = 1u; break; }
        // bit 1: a square power-of-two map -- its level offsets have a closed form (vct_tex_level_offset), which saves the
        // samplers the dependent load of lvl[k]
        if (width[i] == height[i] && (width[i] & (width[i] - 1)) == 0) flags |= 2u;
        d.flags = flags;
        d.nlev = 1;
        if (c->cfg.texture_mipmaps)
            for (int m = width[i] > height[i] ? width[i] : height[i]; m > 1; m >>= 1) ++d.nlev;
        size_t off = 0;
        for (int k = 0; k < d.nlev; ++k) {
            d.lvl[k] = (uint32_t)off;
            off += (size_t)(width[i] >> k > 1 ? width[i] >> k : 1) * (size_t)(height[i] >> k > 1 ? height[i] >> k : 1);
        }
        total += off;
        if (total > 0xffffffffull) return fail(c, VCT_ERR_INVALID, "vct_upload_textures: more than 2^32 texels");
    }
    for (int32_t m = 0; m < c->nmat * 3; ++m)
        if (mat_tex[m] >= ntex) return fail(c, VCT_ERR_INVALID, "vct_upload_textures: texture index out of range");
    HIP_TRY(c, hipMalloc(&c->tex_texels, total * 4));
    HIP_TRY(c, hipMalloc(&c->tex_desc, (size_t)ntex * sizeof(VctTexDesc)));
    HIP_TRY(c, hipMalloc(&c->mat_tex, (size_t)c->nmat * 3 * sizeof(int32_t)));
    for (int32_t i = 0; i < ntex; ++i) {
        const VctTexDesc& d = desc[(size_t)i];
        HIP_TRY(c, hipMemcpyAsync(c->tex_texels + d.off, rgba8[i], (size_t)width[i] * height[i] * 4,
                                  hipMemcpyHostToDevice, c->stream));
        for (int k = 1; k < d.nlev; ++k) {
            const int pw = d.w >> (k - 1) > 1 ? d.w >> (k - 1) : 1, ph = d.h >> (k - 1) > 1 ? d.h >> (k - 1) : 1;
            const int w = d.w >> k > 1 ? d.w >> k : 1, h = d.h >> k > 1 ? d.h >> k : 1;
            HIP_TRY(c, vct_launch_tex_mip(c->tex_texels + d.off + d.lvl[k - 1], pw, ph, c->tex_texels + d.off + d.lvl[k],
                                          w, h, c->stream));
        }
    }
    HIP_TRY(c, hipMemcpyAsync(c->tex_desc, desc.data(), (size_t)ntex * sizeof(VctTexDesc), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->mat_tex, mat_tex, (size_t)c->nmat * 3 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->ntex = ntex;
    for (int32_t m = 0; m < c->nmat; ++m)         // a diffuse map with non-opaque texels: its fragments are alpha-tested
        if (mat_tex[3 * m] >= 0 && (desc[(size_t)mat_tex[3 * m]].flags & 1u)) c->has_alpha_textures = true;
    return VCT_OK;
}

int vct_render_shadow_map(vct_ctx* c, const float light_vp[16]) {
    if (!c) return VCT_ERR_INVALID;
    if (!light_vp) return fail(c, VCT_ERR_INVALID, "vct_render_shadow_map: null matrix");
    const int S = c->cfg.shadow_map_size;
    if (S <= 0) return fail(c, VCT_ERR_INVALID, "vct_render_shadow_map: config.shadow_map_size <= 0");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->shadow && c->shadow_size == S) PIPE_TRY(pipeline_join(c)); else PIPE_TRY(pipeline_drain(c));   // (re)allocation: host wait
    if (c->shadow && c->shadow_size != S) {
        (void)hipFree(c->shadow); c->shadow = nullptr; c->shadow_size = 0;
        if (c->shadow_tiles) { (void)hipFree(c->shadow_tiles); c->shadow_tiles = nullptr; }
    }
    if (!c->shadow) {
        HIP_TRY(c, hipMalloc(&c->shadow, (size_t)S * S * sizeof(uint32_t)));
        c->shadow_passes = 0u;
    }
    // Tile bounds of the map (vct_launch_shadow_minmax) cost one more pass over it (~0.03 ms at 4096^2) and save the PCF
    // consumers their window fetches away from shadow boundaries: 14-28 % of the voxelize pass (street: 0.536 -> 0.461 ms at
    // 1024^3, 0.176 -> 0.126 at 256^3; atrium 0.033 -> 0.031).  Built where that repays the pass: meshes of >= 4 M
    // voxel fragments (VCT_SHADOW_TILES=0 / 1 in the environment: never / always).  Results are identical either way.
    bool want_tiles = c->n_frags >= 4000000u;
    if (const char* st = getenv("VCT_SHADOW_TILES")) want_tiles = st[0] == '1';
    if (want_tiles && !c->shadow_tiles) HIP_TRY(c, hipMalloc(&c->shadow_tiles, shadow_tile_count(S) * sizeof(uint2)));
    if (!want_tiles && c->shadow_tiles) { (void)hipFree(c->shadow_tiles); c->shadow_tiles = nullptr; }
    c->shadow_size = S;
    // A pass that fails from here on leaves a map (the old one, or a partly written one) but NO tile bounds: a stale or
    // uninitialised table would decide PCF windows wrongly, without one the consumers just fetch every window (advisor, round 5).
    auto drop_tiles = [&]() { if (c->shadow_tiles) { (void)hipFree(c->shadow_tiles); c->shadow_tiles = nullptr; } };
    VctRasterArgs a;
    int rc = raster_args(c, S, S, true, c->raster_mode == 2, c->stream, a);
    if (rc) { drop_tiles(); return rc; }
    // The pass's atomicMin words ARE the map (vct_internal.h "shadow map words"): epoch 3, 2, 1, 0, then one memset
    // and 3 again -- a new pass overwrites older epochs by itself, readers see them as depth 1.0.
    const uint32_t epoch = 3u - (c->shadow_passes & 3u);
    if (epoch == 3u) {
        const hipError_t em = hipMemsetAsync(c->shadow, 0xff, (size_t)S * S * sizeof(uint32_t), c->stream);
        if (em != hipSuccess) { drop_tiles(); HIP_TRY(c, em); }
    }
    a.vis32 = c->shadow;
    a.vis32_ebase = VCT_SHADOW_EPOCH(epoch);
    memcpy(c->light_vp, light_vp, 64);
    const hipError_t e = vct_launch_shadow_raster(a, light_vp, S, c->stream);
    if (e != hipSuccess) { c->raster_dirty[0] = true; drop_tiles(); HIP_TRY(c, e); }
    c->shadow_ebase = a.vis32_ebase;
    ++c->shadow_passes;
    // depth bounds per (dilated) 8 x 8 tile of the new map: the PCF consumers (voxelizer, G-buffer shade) decide most windows on them
    if (c->shadow_tiles) {
        const hipError_t et = vct_launch_shadow_minmax(c->shadow, c->shadow_ebase, S, c->shadow_tiles, c->stream);
        if (et != hipSuccess) { drop_tiles(); HIP_TRY(c, et); }
    }
    return VCT_OK;
}

int vct_download_shadow_map(vct_ctx* c, float* depth) {
    if (!c || !depth) return VCT_ERR_INVALID;
    if (!c->shadow) return fail(c, VCT_ERR_INVALID, "no shadow map");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->shadow_size * c->shadow_size;
    float* tmp = nullptr;
    HIP_TRY(c, hipMalloc(&tmp, n * sizeof(float)));
    hipError_t e = vct_launch_shadow_decode(c->shadow, tmp, n, c->shadow_ebase, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(depth, tmp, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    HIP_TRY(c, e);
    return VCT_OK;
}

// `shadow_ready`: when not null the visibility raster is issued at once and only the shading kernel (which reads the
// shadow map) waits for that event -- vct_gi_pass rasterises the main draw's visibility beside the shadow pass.
static int render_gbuffer_rows_on(vct_ctx* c, const float view_proj[16], int32_t row0, int32_t row1, hipStream_t s,
                                  hipEvent_t shadow_ready = nullptr) {
    if (!view_proj) return fail(c, VCT_ERR_INVALID, "vct_render_gbuffer: null matrix");
    if (!c->tri_nrm) return fail(c, VCT_ERR_INVALID, "vct_render_gbuffer: call vct_upload_mesh_attributes first");
    if (row0 < 0 || row1 > tiles_y(c) || row0 > row1)
        return fail(c, VCT_ERR_INVALID, "vct_render_gbuffer_rows: tile-row range outside the frame");
    HIP_TRY(c, hipSetDevice(c->device));
    // (two frames in flight: each frame slot has raster scratch of its own -- raster_set_of -- so this pass waits for
    // nothing of the other slot's frame)
    // the form of the visibility stage (vct_ctx.h raster_mode)
    bool binned = c->raster_mode == 2;
    // Automatic choice: six passes -- direct (warm-up: the first pass after an upload pays for cold caches), direct timed,
    // binned (warm-up: it also allocates its scratch), binned timed, direct timed, binned timed -- then the form with the
    // smaller minimum is kept until the mesh or the textures change.  (Rounds 3-4 compared ONE cold direct pass with one
    // warm binned pass: biased towards the binned form -- advisor, round 4.)  Results are identical either way.
    static const struct { int form, slot; } kAutoSeq[6] = {{0, -1}, {0, 0}, {1, -1}, {1, 1}, {0, 2}, {1, 3}};
    int slot = -1;              // >= 0: this pass is timed sample `slot`
    if (c->raster_mode == 0 && c->has_alpha_textures) {
        if (c->auto_state == 6 && c->auto_choice < 0 && hipEventQuery(c->ev_auto[7]) == hipSuccess) {
            float t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            bool ok = true;
            for (int k = 0; k < 4; ++k) ok = ok && hipEventElapsedTime(&t[k], c->ev_auto[2 * k], c->ev_auto[2 * k + 1]) == hipSuccess;
            if (ok) c->auto_choice = fminf(t[1], t[3]) < fminf(t[0], t[2]) ? 1 : 0;
        }
        // the samples must cover the same rows (a rank of a multi-GPU frame only ever rasterises its slab, and the
        // slab may move while the load-aware boundaries settle: a sequence over different rows starts again)
        if (c->auto_choice < 0 && c->auto_state > 0 && c->auto_state < 6 && (row0 != c->auto_rows[0] || row1 != c->auto_rows[1])) c->auto_state = 0;
        if (c->auto_choice >= 0) binned = c->auto_choice == 1;
        else if (row1 > row0 && c->auto_state < 6) {
            binned = kAutoSeq[c->auto_state].form == 1;
            slot = kAutoSeq[c->auto_state].slot;
            c->auto_rows[0] = row0; c->auto_rows[1] = row1;
        } else if (c->auto_state >= 6) binned = true;        // all sampled, the last event still pending: stay on the last form
    }
    const bool sampling = c->raster_mode == 0 && c->has_alpha_textures && c->auto_choice < 0 && row1 > row0 && c->auto_state < 6;
    VctRasterArgs a;
    int rc = raster_args(c, c->cfg.width, c->cfg.height, false, binned, s, a);
    if (rc) return rc;
    if (slot >= 0) HIP_TRY(c, hipEventRecord(c->ev_auto[2 * slot], s));
    hipError_t e = vct_launch_gbuffer_visibility(a, view_proj, c->cfg.width, c->cfg.height, row0, row1, s);
    if (slot >= 0 && e == hipSuccess) e = hipEventRecord(c->ev_auto[2 * slot + 1], s);
    if (sampling && e == hipSuccess) ++c->auto_state;
    if (e == hipSuccess && shadow_ready) e = hipStreamWaitEvent(s, shadow_ready, 0);
    if (e == hipSuccess)
        e = vct_launch_gbuffer_shade(a, view_proj, c->cfg.width, c->cfg.height, row0, row1, c->shadow, c->shadow_ebase,
                                     c->shadow_size, c->shadow_tiles, c->light_vp, c->gb_tiled, s);
    if (e != hipSuccess) { c->raster_dirty[raster_set_of(c)] = true; HIP_TRY(c, e); }
    c->gb_current = c->gb_tiled;
    c->last_raster_form = binned ? 2 : 1;
    c->last_row0 = row0;
    c->last_row1 = row1;
    c->last_row_stride = 1;
    c->have_gbuffer = true;
#if defined(VCT_BIN_STATS) && VCT_BIN_STATS
    if (binned && getenv("VCT_BIN_STATS_DUMP")) {       // instrumented builds only (tools/r04_binstats.sh)
        uint32_t st[48];
        HIP_TRY(c, hipStreamSynchronize(s));
        HIP_TRY(c, hipMemcpy(st, c->bin_huge[1] + VCT_BIN_HUGE_CAP, sizeof(st), hipMemcpyDeviceToHost));
        const uint32_t* cur = st + 8 * (c->bin_set[1] ^ 1);
        fprintf(stderr, "binstats: entries %u records %u items %u huge %u | ", cur[0], cur[1], cur[2], cur[4]);
        // (the adopted form of k_bin_raster only counts its alpha-queue flushes and the fragments they fetched; the
        // per-step counters of the earlier forms are in profiles/experiments/README.md)
        fprintf(stderr, "alpha_queue_flushes %u fragments_fetched %u\n", st[16 + 7], st[16 + 8]);
        HIP_TRY(c, hipMemset(c->bin_huge[1] + VCT_BIN_HUGE_CAP + 16, 0, 32 * sizeof(uint32_t)));
    }
#endif
    return VCT_OK;
}

int vct_render_gbuffer_rows(vct_ctx* c, const float view_proj[16], int32_t row0, int32_t row1) {
    if (!c) return VCT_ERR_INVALID;
    return render_gbuffer_rows_on(c, view_proj, row0, row1, c->stream);
}

int vct_render_gbuffer(vct_ctx* c, const float view_proj[16]) {
    if (!c) return VCT_ERR_INVALID;
    return vct_render_gbuffer_rows(c, view_proj, 0, tiles_y(c));
}

int vct_download_gbuffer(vct_ctx* c, float* planes) {
    if (!c || !planes) return VCT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->cfg.width * c->cfg.height * VCT_GB_NPLANES;
    if (!c->gb_linear) HIP_TRY(c, hipMalloc(&c->gb_linear, n * sizeof(float)));
    HIP_TRY(c, vct_launch_untile_gbuffer(c->gb_current, c->gb_linear, c->cfg.width, c->cfg.height, c->stream));
    HIP_TRY(c, hipMemcpyAsync(planes, c->gb_linear, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCT_OK;
}

int vct_voxelize(vct_ctx* c, int32_t mode) {
    if (!c) return VCT_ERR_INVALID;
    if (mode != VCT_VOX_CONSERVATIVE_AVG && mode != VCT_VOX_REFERENCE)
        return fail(c, VCT_ERR_INVALID, "vct_voxelize: unknown mode");
    if (!c->tri_pos) return fail(c, VCT_ERR_INVALID, "vct_voxelize: no triangles uploaded");
    HIP_TRY(c, hipSetDevice(c->device));
    PIPE_TRY(pipeline_join(c));       // the staging pool is shared: the other slot's resolve may still read the previous pass
    if (!c->brick_slot || !c->stage || (mode == VCT_VOX_REFERENCE && !c->ref_big))
        return fail(c, VCT_ERR_NOMEM, "vct_voxelize: the voxelization plan of this mesh could not be allocated "
                                      "(vct_upload_triangles reported it)");
    const size_t pool_vox = (size_t)(c->nslots ? c->nslots : 1u) * 512;
    const size_t nbricks = (size_t)c->cfg.voxel_dim * c->cfg.voxel_dim * c->cfg.voxel_dim / 512;
    if (mode == VCT_VOX_REFERENCE && !c->acc) {       // reference mode's accumulators: allocated on first use, zeroed once
        HIP_TRY(c, hipMalloc(&c->acc, pool_vox * 16));
        HIP_TRY(c, hipMemsetAsync(c->acc, 0, pool_vox * 16, c->stream));
    }
    if (c->acc_pending) {   // a pass that was never resolved: discard it
        if (c->acc_mode == VCT_VOX_REFERENCE && c->acc) HIP_TRY(c, hipMemsetAsync(c->acc, 0, pool_vox * 16, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->brick_flags, 0, nbricks * sizeof(uint32_t), c->stream));
    }
    VctVoxParams p = vox_params(c);
    p.mode = mode;
    if (mode == VCT_VOX_REFERENCE) {
        glm_voxel_projections(c, p.proj);
        HIP_TRY(c, vct_launch_voxelize_reference(p, c->ref_big + 1, c->ref_big, c->stream));
    } else {
        if (p.tex.texels && c->n_frags) {
            // every fragment's albedo (texture fetch or material colour): independent of the light, so evaluated once per
            // change of the textures / texture coordinates, not once per pass
            if (!c->frag_alb) { HIP_TRY(c, hipMalloc(&c->frag_alb, (size_t)c->n_frags * 3 * sizeof(float))); c->frag_alb_dirty = true; }
            if (c->frag_alb_dirty) {
                HIP_TRY(c, vct_launch_frag_geom(p, nullptr, c->frag_alb, c->stream));
                c->frag_alb_dirty = false;
            }
            p.frag_alb = c->frag_alb;
        }
        HIP_TRY(c, vct_launch_voxelize(p, c->stream));      // one workgroup per brick: LDS accumulation + resolve into the staging pool
    }
    c->acc_pending = true;
    c->acc_mode = mode;
    return VCT_OK;
}

int vct_inject_light(vct_ctx* c) {
    if (!c) return VCT_ERR_INVALID;
    if (!c->acc_pending) return fail(c, VCT_ERR_INVALID, "vct_inject_light: call vct_voxelize first");
    HIP_TRY(c, hipSetDevice(c->device));
    PIPE_TRY(pipeline_join(c));       // level 0 is rewritten: the other slot's trace may still read the chain
    HIP_TRY(c, vct_launch_resolve(c->acc, c->brick_slot, c->chain, c->brick_flags, c->brick_prev, c->cfg.voxel_dim,
                                  c->level0_dirty, nullptr, c->attr_albedo, c->attr_normal,
                                  c->acc_mode == VCT_VOX_REFERENCE, c->stage, c->stage_albedo, c->stage_normal, c->stream));
    c->acc_pending = false;
    c->level0_dirty = false;
    c->use_chain_b = false;
    c->mips_valid = false;
    c->attrs_valid = c->attr_normal != nullptr && c->acc_mode == VCT_VOX_CONSERVATIVE_AVG;
    return VCT_OK;
}

// Footprint records of the levels >= 1 (vct_set_footprint_records), rebuilt after every change of those levels.
// Dense: 8 x the bytes of those levels = 1.14 x level 0 written per build (0.04 ms at 256^3, 2.0 ms at 1024^3).
static int build_cells(vct_ctx* c) {
    c->cells_valid = false;
    if (!c->want_cells || c->nlev < 2) return VCT_OK;
    const size_t V3 = (size_t)c->cfg.voxel_dim * c->cfg.voxel_dim * c->cfg.voxel_dim;
    if (!c->cells) {
        const hipError_t e = hipMalloc(&c->cells, (c->chain_texels - V3) * 32);
        if (e != hipSuccess) { c->cells = nullptr; return fail(c, VCT_ERR_NOMEM, std::string("footprint records: ") + hipGetErrorString(e)); }
    }
    HIP_TRY(c, vct_launch_build_cells(c->chain, c->cells, c->cfg.voxel_dim, c->stream));
    c->cells_valid = true;
    return VCT_OK;
}

int vct_set_footprint_records(vct_ctx* c, int32_t on) {
    if (!c) return VCT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    c->want_cells = on != 0;
    PIPE_TRY(pipeline_drain(c));
    if (!c->want_cells) {
        c->cells_valid = false;
        if (c->cells) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));      // a trace in flight may still read them
            (void)hipFree(c->cells);
            c->cells = nullptr;
        }
        return VCT_OK;
    }
    return c->mips_valid ? build_cells(c) : VCT_OK;           // a valid chain gets its records now, otherwise at the next build
}

int vct_build_mips(vct_ctx* c) {
    if (!c) return VCT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    PIPE_TRY(pipeline_join(c));
    // Sparse form: only bricks that hold something now (brick_prev) or held something when the mips were
    // last built (mip_seen) are reduced.  Valid while level 0 is the output of a resolve (not an upload)
    // and every other brick has all-zero ancestors -- which an upload destroys until ONE dense build has
    // run over a resolved level 0.
    const bool tracked = c->brick_prev && c->mip_seen && !c->level0_dirty;
    const bool sparse = tracked && c->chain_sparse_ready;
    if (!sparse && c->mip_seen && c->brick_prev)      // dense build: afterwards every brick is "seen" as it is now
        HIP_TRY(c, hipMemcpyAsync(c->mip_seen, c->brick_prev,
                                  ((size_t)c->cfg.voxel_dim * c->cfg.voxel_dim * c->cfg.voxel_dim / 512) * sizeof(uint32_t),
                                  hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, vct_launch_build_mips(c->chain, c->cfg.voxel_dim, sparse ? c->brick_prev : nullptr,
                                     sparse ? c->mip_seen : nullptr, c->stream));
    if (!sparse) c->chain_sparse_ready = tracked;
    if (c->aniso) HIP_TRY(c, vct_launch_build_mips_aniso(c->chain, c->aniso, c->cfg.voxel_dim, c->stream));
    c->mips_valid = true;
    c->use_chain_b = false;
    return build_cells(c);
}

int vct_bounce(vct_ctx* c) {
    if (!c) return VCT_ERR_INVALID;
    if (!c->cfg.voxel_attributes || !c->attr_normal)
        return fail(c, VCT_ERR_INVALID, "vct_bounce: needs config.voxel_attributes = 1 and a voxelize + inject pass");
    if (c->acc_pending || !c->mips_valid)
        return fail(c, VCT_ERR_INVALID, "vct_bounce: call vct_inject_light and vct_build_mips first");
    if (!c->attrs_valid)      // a new mesh was uploaded since: level 0 / brick_prev describe the OLD mesh, the slots the new one
        return fail(c, VCT_ERR_INVALID, "vct_bounce: the voxel attributes belong to a mesh uploaded after the last "
                                        "vct_inject_light (voxelize + inject + mips again first)");
    if (c->level0_dirty || c->acc_mode != VCT_VOX_CONSERVATIVE_AVG)
        return fail(c, VCT_ERR_INVALID, "vct_bounce: level 0 must come from a VCT_VOX_CONSERVATIVE_AVG pass (voxel attributes)");
    HIP_TRY(c, hipSetDevice(c->device));
    PIPE_TRY(pipeline_drain(c));      // (allocates the second chain on first use)
    int rc = refresh_steps(c);
    if (rc) return rc;
    const size_t nvox = (size_t)c->cfg.voxel_dim * c->cfg.voxel_dim * c->cfg.voxel_dim;
    const size_t nbricks = nvox / 512;
    bool b_sparse = c->chain_b != nullptr;
    if (!c->chain_b) {
        HIP_TRY(c, hipMalloc(&c->chain_b, c->chain_texels * 4));
        HIP_TRY(c, hipMemsetAsync(c->chain_b, 0, c->chain_texels * 4, c->stream));
        HIP_TRY(c, hipMalloc(&c->mip_seen_b, nbricks * sizeof(uint32_t)));
        HIP_TRY(c, hipMemsetAsync(c->mip_seen_b, 0, nbricks * sizeof(uint32_t), c->stream));
        b_sparse = true;     // zero-filled chain + empty "seen" set: the sparse form is valid from the start
        // occupied-voxel list: surfaces occupy ~1 % of a grid; V^3/8 entries is a generous bound and
        // bricks that do not fit are handled by the per-brick kernel
        c->bounce_list_cap = (uint32_t)(nvox / 8);
        HIP_TRY(c, hipMalloc(&c->bounce_list, ((size_t)c->bounce_list_cap + 1) * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->brick_over, nbricks * sizeof(uint32_t)));
        HIP_TRY(c, hipMemsetAsync(c->brick_over, 0, nbricks * sizeof(uint32_t), c->stream));   // k_bounce_bricks resets what it serves
    }
    // only the counter: the list itself needs no clear (the one brick that can straddle its end marks its tail empty)
    HIP_TRY(c, hipMemsetAsync(c->bounce_list, 0, sizeof(uint32_t), c->stream));
    VctTraceParams p;
    fill_march_params(c, p, c->chain);
    p.attr_albedo = c->attr_albedo;
    p.attr_normal = c->attr_normal;
    p.brick_slot = c->brick_slot;
    p.brick_prev = c->brick_prev;
    p.bounce_seen = c->mip_seen_b;
    p.bounce_out = c->chain_b;
    p.nbricks = (uint32_t)(nvox / 512);
    p.slot_brick = c->slot_brick;
    p.nslots = c->nslots;
    p.bounce_list_count = c->bounce_list;
    p.bounce_list = c->bounce_list + 1;
    p.bounce_list_cap = c->bounce_list_cap;
    p.brick_over = c->brick_over;
    HIP_TRY(c, hipMemsetAsync(c->step_counter, 0, VCT_STEP_COUNTERS * sizeof(unsigned long long), c->stream));
    if (c->time_traces) HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    HIP_TRY(c, vct_launch_bounce(p, c->stream));
    if (c->time_traces) HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    c->last_trace_timed = c->time_traces;
    HIP_TRY(c, vct_launch_build_mips(c->chain_b, c->cfg.voxel_dim, b_sparse ? c->brick_prev : nullptr,
                                     b_sparse ? c->mip_seen_b : nullptr, c->stream));
    // the directional chains always describe the chain the trace reads (the bounce itself gathers
    // from the isotropic bounce-0 chain, like the oracle's vcto_bounce)
    if (c->aniso) HIP_TRY(c, vct_launch_build_mips_aniso(c->chain_b, c->aniso, c->cfg.voxel_dim, c->stream));
    c->use_chain_b = true;
    c->have_trace = true;      // step counter / event pair now describe the bounce launch
    c->last_was_screen_trace = false;
    return VCT_OK;
}

int vct_download_voxel_attributes(vct_ctx* c, uint8_t* albedo, uint8_t* normal) {
    if (!c || !albedo || !normal) return VCT_ERR_INVALID;
    if (!c->attr_albedo) return fail(c, VCT_ERR_INVALID, "no voxel attributes (config.voxel_attributes, vct_voxelize + vct_inject_light)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->staging) {
        const size_t n = (size_t)c->cfg.voxel_dim * c->cfg.voxel_dim * c->cfg.voxel_dim;
        HIP_TRY(c, hipMalloc(&c->staging, n * 4));
    }
    const int V = c->cfg.voxel_dim;
    const size_t n = (size_t)V * V * V;
    const uint32_t* src[2] = {c->attr_albedo, c->attr_normal};
    uint8_t* dst[2] = {albedo, normal};
    uint32_t* dense = nullptr;          // pooled [slot][512] -> dense Morton volume -> linear staging
    HIP_TRY(c, hipMalloc(&dense, n * 4));
    for (int k = 0; k < 2; ++k) {
        hipError_t e = vct_launch_unpool(src[k], c->brick_slot, dense, (uint32_t)(n / 512), c->stream);
        if (e == hipSuccess) e = vct_launch_morton_to_linear(dense, c->staging, V, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dst[k], c->staging, n * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { (void)hipFree(dense); HIP_TRY(c, e); }
    }
    (void)hipFree(dense);
    return VCT_OK;
}

// ---- volume up/download -----------------------------------------------------------------

static int ensure_staging(vct_ctx* c) {
    if (c->staging) return VCT_OK;
    const size_t n = (size_t)c->cfg.voxel_dim * c->cfg.voxel_dim * c->cfg.voxel_dim;
    HIP_TRY(c, hipMalloc(&c->staging, n * 4));
    return VCT_OK;
}

static int upload_levels(vct_ctx* c, const uint8_t* lin, int nlevels) {
    HIP_TRY(c, hipSetDevice(c->device));
    PIPE_TRY(pipeline_drain(c));
    c->use_chain_b = false;
    c->mips_valid = nlevels > 1;
    c->cells_valid = false;
    c->level0_dirty = true;
    c->chain_sparse_ready = false;        // level 0 no longer mirrors brick_prev: next resolve and mip build are dense
    int rc = ensure_staging(c);
    if (rc) return rc;
    const int V = c->cfg.voxel_dim;
    for (int l = 0; l < nlevels; ++l) {
        const int N = V >> l;
        const size_t off = (size_t)vct_level_offset(V, l), n = (size_t)N * N * N;
        HIP_TRY(c, hipMemcpyAsync(c->staging, lin + off * 4, n * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, vct_launch_linear_to_morton(c->staging, c->chain + off, N, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return VCT_OK;
}

int vct_upload_volume_rgba8(vct_ctx* c, const uint8_t* l0) {
    if (!c) return VCT_ERR_INVALID;
    if (!l0) return fail(c, VCT_ERR_INVALID, "vct_upload_volume_rgba8: null volume");
    return upload_levels(c, l0, 1);
}

int vct_upload_chain_rgba8(vct_ctx* c, const uint8_t* chain) {
    if (!c) return VCT_ERR_INVALID;
    if (!chain) return fail(c, VCT_ERR_INVALID, "vct_upload_chain_rgba8: null chain");
    int rc = upload_levels(c, chain, c->nlev);
    if (rc) return rc;
    if (c->aniso) HIP_TRY(c, vct_launch_build_mips_aniso(c->chain, c->aniso, c->cfg.voxel_dim, c->stream));
    return build_cells(c);
}

int vct_download_aniso_rgba8(vct_ctx* c, uint8_t* out) {
    if (!c || !out) return VCT_ERR_INVALID;
    if (!c->aniso) return fail(c, VCT_ERR_INVALID, "context created without anisotropic_mips");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_staging(c);
    if (rc) return rc;
    const int V = c->cfg.voxel_dim;
    const size_t V3 = (size_t)V * V * V, stride = c->chain_texels - V3;
    for (int d = 0; d < 6; ++d)
        for (int l = 1; l < c->nlev; ++l) {
            const int N = V >> l;
            const size_t off = (size_t)vct_level_offset(V, l) - V3, n = (size_t)N * N * N;
            HIP_TRY(c, vct_launch_morton_to_linear(c->aniso + d * stride + off, c->staging, N, c->stream));
            HIP_TRY(c, hipMemcpyAsync(out + (d * stride + off) * 4, c->staging, n * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    return VCT_OK;
}

int vct_download_chain_rgba8(vct_ctx* c, uint8_t* chain) {
    if (!c) return VCT_ERR_INVALID;
    if (!chain) return fail(c, VCT_ERR_INVALID, "vct_download_chain_rgba8: null destination");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_staging(c);
    if (rc) return rc;
    const int V = c->cfg.voxel_dim;
    for (int l = 0; l < c->nlev; ++l) {
        const int N = V >> l;
        const size_t off = (size_t)vct_level_offset(V, l), n = (size_t)N * N * N;
        const uint32_t* active = c->use_chain_b ? c->chain_b : c->chain;
        HIP_TRY(c, vct_launch_morton_to_linear(active + off, c->staging, N, c->stream));
        HIP_TRY(c, hipMemcpyAsync(chain + off * 4, c->staging, n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return VCT_OK;
}

// ---- trace -------------------------------------------------------------------------------

static int bind_gbuffer(vct_ctx* c, const vct_gbuffer* gb) {
    if (!gb || !gb->planes) return fail(c, VCT_ERR_INVALID, "vct_trace: null G-buffer");
    if (gb->width != c->cfg.width || gb->height != c->cfg.height)
        return fail(c, VCT_ERR_INVALID, "vct_trace: G-buffer size differs from the context's frame");
    const size_t npix = (size_t)c->cfg.width * c->cfg.height;
    if (gb->layout == VCT_GB_TILED) {
        if (gb->location == VCT_MEM_DEVICE) {
            c->gb_current = gb->planes;     // zero-copy: trace reads the caller's HBM buffer
        } else {
            HIP_TRY(c, hipMemcpyAsync(c->gb_tiled, gb->planes, gb_tiled_floats(c) * sizeof(float),
                                      hipMemcpyHostToDevice, c->stream));
            c->gb_current = c->gb_tiled;
        }
        return VCT_OK;
    }
    if (gb->layout != VCT_GB_LINEAR) return fail(c, VCT_ERR_INVALID, "vct_trace: unknown G-buffer layout");
    const float* src = gb->planes;
    if (gb->location == VCT_MEM_HOST) {
        if (!c->gb_linear) HIP_TRY(c, hipMalloc(&c->gb_linear, npix * VCT_GB_NPLANES * sizeof(float)));
        HIP_TRY(c, hipMemcpyAsync(c->gb_linear, gb->planes, npix * VCT_GB_NPLANES * sizeof(float),
                                  hipMemcpyHostToDevice, c->stream));
        src = c->gb_linear;
    }
    HIP_TRY(c, vct_launch_tile_gbuffer(src, c->gb_tiled, c->cfg.width, c->cfg.height, c->stream));
    c->gb_current = c->gb_tiled;
    return VCT_OK;
}

int vct_trace_slab(vct_ctx* c, const vct_gbuffer* gb, int32_t row0, int32_t row1, void* out,
                   int32_t out_location) {
    if (!c) return VCT_ERR_INVALID;
    if (row0 < 0 || row1 > tiles_y(c) || row0 > row1)
        return fail(c, VCT_ERR_INVALID, "vct_trace_slab: tile-row range outside the frame");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = bind_gbuffer(c, gb);
    if (rc) return rc;
    c->have_gbuffer = true;
    rc = launch_trace(c, row0, row1);
    if (rc) return rc;
    if (out) {
        const int y0 = row0 * VCT_TILE;
        const int y1 = row1 * VCT_TILE < c->cfg.height ? row1 * VCT_TILE : c->cfg.height;
        if (y1 > y0) {
            const size_t off = (size_t)y0 * c->cfg.width * 8, bytes = (size_t)(y1 - y0) * c->cfg.width * 8;
            const char* src = (const char*)(c->frame_target ? c->frame_target : c->frame);
            HIP_TRY(c, hipMemcpyAsync((char*)out + off, src + off, bytes,
                                      out_location == VCT_MEM_DEVICE ? hipMemcpyDeviceToDevice
                                                                     : hipMemcpyDeviceToHost,
                                      c->stream));
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCT_OK;
}

int vct_trace(vct_ctx* c, const vct_gbuffer* gb, void* out, int32_t out_location) {
    if (!c) return VCT_ERR_INVALID;
    return vct_trace_slab(c, gb, 0, tiles_y(c), out, out_location);
}

int vct_trace_current(vct_ctx* c, void* out, int32_t out_location) {
    if (!c) return VCT_ERR_INVALID;
    if (!c->have_gbuffer) return fail(c, VCT_ERR_INVALID, "vct_trace_current: no G-buffer resident yet");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = launch_trace(c, 0, tiles_y(c));
    if (rc) return rc;
    if (out) {
        const char* src = (const char*)(c->frame_target ? c->frame_target : c->frame);
        HIP_TRY(c, hipMemcpyAsync(out, src, (size_t)c->cfg.width * c->cfg.height * 8,
                                  out_location == VCT_MEM_DEVICE ? hipMemcpyDeviceToDevice
                                                                 : hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCT_OK;
}

int vct_trace_resident_rows(vct_ctx* c, int32_t row0, int32_t row1) {
    if (!c) return VCT_ERR_INVALID;
    if (!c->have_gbuffer) return fail(c, VCT_ERR_INVALID, "vct_trace_resident_rows: no G-buffer resident yet");
    if (row0 < 0 || row1 > tiles_y(c) || row0 > row1)
        return fail(c, VCT_ERR_INVALID, "vct_trace_resident_rows: tile-row range outside the frame");
    HIP_TRY(c, hipSetDevice(c->device));
    return launch_trace(c, row0, row1);
}

int vct_trace_resident_strided(vct_ctx* c, int32_t row0, int32_t row1, int32_t stride) {
    if (!c) return VCT_ERR_INVALID;
    if (!c->have_gbuffer) return fail(c, VCT_ERR_INVALID, "vct_trace_resident_strided: no G-buffer resident yet");
    if (row0 < 0 || row1 > tiles_y(c) || row0 > row1 || stride < 1)
        return fail(c, VCT_ERR_INVALID, "vct_trace_resident_strided: tile-row range outside the frame or stride < 1");
    HIP_TRY(c, hipSetDevice(c->device));
    return launch_trace(c, row0, row1, nullptr, stride, false);
}

int vct_trace_resident(vct_ctx* c) {
    if (!c) return VCT_ERR_INVALID;
    if (!c->have_gbuffer) return fail(c, VCT_ERR_INVALID, "vct_trace_resident: no G-buffer resident yet");
    HIP_TRY(c, hipSetDevice(c->device));
    return launch_trace(c, c->last_row0, c->last_row1, nullptr, c->last_row_stride, false);
}

int vct_gi_pass(vct_ctx* c, const float light_vp[16], const float view_proj[16], int32_t mode) {
    if (!c) return VCT_ERR_INVALID;
    if (!light_vp || !view_proj) return fail(c, VCT_ERR_INVALID, "vct_gi_pass: null matrix");
    if (c->cfg.shadow_map_size <= 0) return fail(c, VCT_ERR_INVALID, "vct_gi_pass: config.shadow_map_size <= 0");
    // A rank of a multi-GPU frame (vct_comm_init) runs the same pass on its slab: the G-buffer stream is scissored to
    // the rank's tile rows and the pass ends with vct_frame_step (slab trace + the frame's one gather) at the join.
    int row0 = 0, row1 = tiles_y(c);
    const bool rank_ctx = vct_comm_rows(c, &row0, &row1);
    HIP_TRY(c, hipSetDevice(c->device));
    PIPE_TRY(pipeline_join(c));
    {
        // VCT_GI_ONE_STREAM=1 (A/B): the six stages in sequence on the context's stream, no fork / join events
        static const bool one_stream = [] { const char* e = getenv("VCT_GI_ONE_STREAM"); return e && e[0] == '1'; }();
        if (one_stream) {
            int rc1 = vct_render_shadow_map(c, light_vp);
            if (rc1 == VCT_OK) rc1 = vct_voxelize(c, mode);
            if (rc1 == VCT_OK) rc1 = vct_inject_light(c);
            if (rc1 == VCT_OK) rc1 = vct_build_mips(c);
            if (rc1 == VCT_OK) rc1 = render_gbuffer_rows_on(c, view_proj, row0, row1, c->stream);
            if (rc1) return rc1;
            if (rank_ctx) return vct_frame_step(c);
            return launch_trace(c, c->last_row0, c->last_row1, nullptr, c->last_row_stride, false);
        }
    }
    if (!c->aux_stream) {
        if (c->reserved_cus > 0) {
            hipDeviceProp_t prop;
            HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
            HIP_TRY(c, vct_create_masked_stream(&c->aux_stream, c->device, 0, prop.multiProcessorCount - c->reserved_cus));
        } else {
            bool ov = false;      // a stream that shares the context stream's hardware queue would run the two halves in sequence
            PIPE_TRY(create_overlapping_stream(c, c->stream, &c->aux_stream, &ov));
        }
    }
    if (!c->ev_fork) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    if (!c->ev_shadow) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_shadow, hipEventDisableTiming));
    if (!c->ev_join) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    // fork at once: the main draw's VISIBILITY raster needs nothing of this pass (it has its own lists and words);
    // only its shading kernel reads the shadow map (PCF term), so that alone waits for the shadow pass
    HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));                 // everything issued before this call is done
    HIP_TRY(c, hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
    int rc = vct_render_shadow_map(c, light_vp);                       // allocates / sizes the shadow map first
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->ev_shadow, c->stream));
    rc = render_gbuffer_rows_on(c, view_proj, row0, row1, c->aux_stream, c->ev_shadow);
    // join before anything else can fail: later work on the context's stream must see the G-buffer
    const hipError_t ej = hipEventRecord(c->ev_join, c->aux_stream);
    if (rc == VCT_OK) rc = vct_voxelize(c, mode);
    if (rc == VCT_OK) rc = vct_inject_light(c);
    if (rc == VCT_OK) rc = vct_build_mips(c);
    if (ej == hipSuccess) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    else HIP_TRY(c, ej);
    if (rc) return rc;
    if (rank_ctx) return vct_frame_step(c);
    return launch_trace(c, c->last_row0, c->last_row1, nullptr, c->last_row_stride, false);
}

int vct_download_frame(vct_ctx* c, void* out) {
    if (!c || !out) return VCT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const char* src = (const char*)(c->frame_target ? c->frame_target : c->frame);
    HIP_TRY(c, hipMemcpyAsync(out, src, (size_t)c->cfg.width * c->cfg.height * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCT_OK;
}

int vct_set_frame_target(vct_ctx* c, void* dev) {
    if (!c) return VCT_ERR_INVALID;
    c->frame_target = (uint16_t*)dev;
    return VCT_OK;
}

int vct_synchronize(vct_ctx* c) {
    if (!c) return VCT_ERR_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->frames_in_flight > 1) HIP_TRY(c, hipStreamSynchronize(c->slots[1 - c->cur_slot].stream));      // every frame in flight
    return VCT_OK;
}

// ---- two frames in flight (vct_ctx.h VctFrameSlot) ---------------------------------------------------------------------
int vct_set_frames_in_flight(vct_ctx* c, int32_t n) {
    if (!c) return VCT_ERR_INVALID;
    if (n != 1 && n != 2) return fail(c, VCT_ERR_INVALID, "vct_set_frames_in_flight: 1 or 2");
    if (n == c->frames_in_flight) return VCT_OK;
    if (n == 2) {
        if (c->cfg.debug_outputs || c->cfg.trace_variant == 4)
            return fail(c, VCT_ERR_INVALID, "vct_set_frames_in_flight: debug_outputs and trace_variant 4 keep per-context scratch: one frame at a time");
#if defined(VCT_STATS) && VCT_STATS
        return fail(c, VCT_ERR_INVALID, "vct_set_frames_in_flight: instrumented build (VCT_STATS): one frame at a time");
#endif
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->comm) PIPE_TRY(vct_comm_sync(c));      // (deadline-bounded: a frame step in flight may wait for a peer)
    PIPE_TRY(vct_synchronize(c));
    if (n == 1) {
        // back to one frame: slot 0's set into the context, slot 1's released
        if (c->cur_slot != 0) { slot_save(c, c->slots[1]); slot_load(c, c->slots[0]); c->cur_slot = 0; }
        VctFrameSlot& o = c->slots[1];
        void* ob[] = {o.gb_tiled, o.frame, o.tile_steps};
        for (void* b : ob) if (b) (void)hipFree(b);
        if (o.ev0) (void)hipEventDestroy(o.ev0);
        if (o.ev1) (void)hipEventDestroy(o.ev1);
        if (o.stream) (void)hipStreamDestroy(o.stream);
        o = VctFrameSlot();
        // ... and its raster scratch set (vct_ctx.h: set [2], visibility words [1])
        void** sc[] = {(void**)&c->raster_lists[2], &c->raster_recs[2], (void**)&c->raster_counts[2], (void**)&c->raster_items[2],
                       &c->bin_recs[2], (void**)&c->bin_entries[2], (void**)&c->bin_count[2], (void**)&c->bin_items[2],
                       (void**)&c->bin_huge[2], (void**)&c->vis[1]};
        for (void** q : sc) if (*q) { (void)hipFree(*q); *q = nullptr; }
        c->raster_item_capacity[2] = c->bin_rec_cap[2] = c->bin_entry_cap[2] = c->bin_bins[2] = c->bin_item_cap[2] = 0u;
        c->vis_words[1] = 0;
        c->raster_dirty[2] = true;
        c->raster_set[2] = c->bin_set[2] = 0;
        c->frames_in_flight = 1;
        c->produced_since_switch = c->joined_since_switch = c->drained_since_switch = false;
        return VCT_OK;
    }
    // a second slot: its own stream, timing events, G-buffer, frame and per-tile step counts (190 MB + 17 MB at 1080p)
    VctFrameSlot o;
    const size_t npix = (size_t)c->cfg.width * c->cfg.height, nt = (size_t)tiles_x(c) * tiles_y(c);
    // the slot's stream: one that demonstrably runs beside the context's stream (create_overlapping_stream)
    PIPE_TRY(create_overlapping_stream(c, c->stream, &o.stream, &c->slot_streams_overlap));
    hipError_t e = hipEventCreate(&o.ev0);
    if (e == hipSuccess) e = hipEventCreate(&o.ev1);
    if (e == hipSuccess) e = hipMalloc(&o.gb_tiled, gb_tiled_floats(c) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&o.frame, npix * 8);
    if (e == hipSuccess) e = hipMalloc(&o.tile_steps, nt * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(o.gb_tiled, 0, gb_tiled_floats(c) * sizeof(float), o.stream);
    if (e == hipSuccess) e = hipMemsetAsync(o.frame, 0, npix * 8, o.stream);
    if (e == hipSuccess) e = hipMemsetAsync(o.tile_steps, 0, nt * sizeof(uint32_t), o.stream);
    if (e == hipSuccess && !c->ev_xslot) e = hipEventCreateWithFlags(&c->ev_xslot, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamSynchronize(o.stream);
    if (e != hipSuccess) {
        void* ob[] = {o.gb_tiled, o.frame, o.tile_steps};
        for (void* b : ob) if (b) (void)hipFree(b);
        if (o.ev0) (void)hipEventDestroy(o.ev0);
        if (o.ev1) (void)hipEventDestroy(o.ev1);
        if (o.stream) (void)hipStreamDestroy(o.stream);
        return fail(c, e == hipErrorOutOfMemory ? VCT_ERR_NOMEM : VCT_ERR_DEVICE, std::string("vct_set_frames_in_flight: ") + hipGetErrorString(e));
    }
    o.gb_current = o.gb_tiled;
    c->slots[1] = o;
    slot_save(c, c->slots[0]);
    c->cur_slot = 0;
    c->frames_in_flight = 2;
    c->produced_since_switch = c->joined_since_switch = c->drained_since_switch = false;
    return VCT_OK;
}

int vct_get_frames_in_flight(const vct_ctx* c, int32_t* n, int32_t* selected, int32_t* streams_overlap_out) {
    if (!c) return VCT_ERR_INVALID;
    if (n) *n = c->frames_in_flight;
    if (selected) *selected = c->cur_slot;
    if (streams_overlap_out) *streams_overlap_out = (c->frames_in_flight > 1 && c->slot_streams_overlap) ? 1 : 0;
    return VCT_OK;
}

int vct_select_frame_slot(vct_ctx* c, int32_t slot) {
    if (!c) return VCT_ERR_INVALID;
    if (slot < 0 || slot >= c->frames_in_flight)
        return fail(c, VCT_ERR_INVALID, "vct_select_frame_slot: slot outside [0, frames in flight)");
    if (slot == c->cur_slot) return VCT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    bool waited = false;
    if (c->produced_since_switch) {      // shared state was written on this slot's stream: the other stream's next work follows it
        HIP_TRY(c, hipEventRecord(c->ev_xslot, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->slots[slot].stream, c->ev_xslot, 0));
        c->produced_since_switch = false;
        waited = true;
    }
    slot_save(c, c->slots[c->cur_slot]);
    slot_load(c, c->slots[slot]);
    c->cur_slot = slot;
    // the stream left behind may hold work the next producer must follow -- unless the wait above already put the selected
    // stream behind all of it (the stream left behind receives nothing more until it is selected again)
    c->joined_since_switch = waited;
    c->drained_since_switch = false;
    return VCT_OK;
}

int vct_download_steps(vct_ctx* c, uint8_t* steps) {
    if (!c || !steps) return VCT_ERR_INVALID;
    if (!c->dbg_steps) return fail(c, VCT_ERR_INVALID, "context created without debug_outputs");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(steps, c->dbg_steps, (size_t)c->cfg.width * c->cfg.height * 7,
                         hipMemcpyDeviceToHost));
    return VCT_OK;
}

int vct_download_cones(vct_ctx* c, float* cones) {
    if (!c || !cones) return VCT_ERR_INVALID;
    if (!c->dbg_cones) return fail(c, VCT_ERR_INVALID, "context created without debug_outputs");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(cones, c->dbg_cones, (size_t)c->cfg.width * c->cfg.height * 28 * sizeof(float),
                         hipMemcpyDeviceToHost));
    return VCT_OK;
}

// executed steps per tile row of the last screen trace: sums of the waves' slots over the launched rows
static int row_steps(vct_ctx* c, std::vector<uint64_t>& rows) {
    const int tx = tiles_x(c), ty = tiles_y(c);
    rows.assign((size_t)ty, 0);
    const int r0 = c->last_row0, r1 = c->last_row1;
    if (r1 <= r0) return VCT_OK;
    const size_t per_row = (size_t)tx;
    std::vector<uint32_t> v(per_row * (size_t)(r1 - r0));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(v.data(), c->tile_steps + per_row * (size_t)r0, v.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int r = r0; r < r1; ++r) {
        if ((r - r0) % c->last_row_stride) continue;          // an interleaved launch: the other rows belong to other ranks
        uint64_t sum = 0;
        const uint32_t* q = v.data() + per_row * (size_t)(r - r0);
        for (size_t i = 0; i < per_row; ++i) sum += q[i];
        rows[(size_t)r] = sum;
    }
    return VCT_OK;
}

int vct_last_step_count(vct_ctx* c, uint64_t* steps) {
    if (!c || !steps) return VCT_ERR_INVALID;
    if (!c->have_trace) return fail(c, VCT_ERR_INVALID, "no trace has run");
    HIP_TRY(c, hipSetDevice(c->device));
    uint64_t sum = 0;
    if (c->last_was_screen_trace) {
        std::vector<uint64_t> rows;
        const int rc = row_steps(c, rows);
        if (rc) return rc;
        for (uint64_t r : rows) sum += r;
    } else {        // a bounce: its kernels add into the atomic bank
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        unsigned long long v[VCT_STEP_COUNTERS];
        HIP_TRY(c, hipMemcpy(v, c->step_counter, sizeof(v), hipMemcpyDeviceToHost));
        for (int i = 0; i < VCT_STEP_COUNTERS; ++i) sum += v[i];
    }
    *steps = sum;
    return VCT_OK;
}

int vct_get_stage_counts(vct_ctx* c, uint64_t out[8]) {
    if (!c || !out) return VCT_ERR_INVALID;
    memset(out, 0, 8 * sizeof(uint64_t));
    out[0] = (uint64_t)c->ntri;
    out[1] = c->n_frags;
    out[2] = 0;
    out[3] = c->nslots;
    out[5] = (uint64_t)c->reserved_cus;          // compute units kept for the communication stream (VCT_COMM_RESERVED_CUS)
    out[6] = (uint64_t)c->last_raster_form;      // visibility form of the last main-draw pass: 1 direct, 2 tile-binned
    out[7] = (uint64_t)c->n_vox_items;           // work items of the voxelize pass (slots, heavy ones cut into chunks)
    if (c->brick_prev) {
        HIP_TRY(c, hipSetDevice(c->device));
        const size_t nbricks = (size_t)c->cfg.voxel_dim * c->cfg.voxel_dim * c->cfg.voxel_dim / 512;
        std::vector<uint32_t> flags(nbricks);
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipMemcpy(flags.data(), c->brick_prev, nbricks * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint64_t n = 0;
        for (uint32_t f : flags) n += f != 0u;
        out[4] = n;
    }
    return VCT_OK;
}

int vct_last_row_steps(vct_ctx* c, uint64_t* rows, int32_t nrows) {
    if (!c || !rows) return VCT_ERR_INVALID;
    if (!c->have_trace || !c->last_was_screen_trace)
        return fail(c, VCT_ERR_INVALID, "vct_last_row_steps: the last march was not a screen trace");
    if (nrows != tiles_y(c)) return fail(c, VCT_ERR_INVALID, "vct_last_row_steps: nrows must be the frame's tile rows, ceil(height / 8)");
    // trace_variant 4 stores its step counts per VIRTUAL tile of the compaction list: only their total means anything
    if (c->last_trace_compacted)
        return fail(c, VCT_ERR_INVALID, "vct_last_row_steps: the last trace was compacted (config.trace_variant 4): no per-row histogram");
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<uint64_t> v;
    const int rc = row_steps(c, v);
    if (rc) return rc;
    memcpy(rows, v.data(), (size_t)nrows * sizeof(uint64_t));
    return VCT_OK;
}

int vct_last_trace_stats(vct_ctx* c, uint64_t out[16]) {
    if (!c || !out) return VCT_ERR_INVALID;
#if defined(VCT_STATS) && VCT_STATS
    if (!c->have_trace) return fail(c, VCT_ERR_INVALID, "no trace has run");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, c->stats, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return VCT_OK;
#else
    return fail(c, VCT_ERR_INVALID, "vct_last_trace_stats: this library was built without -DVCT_STATS=1 "
                                    "(tools/build_ab.sh stats \"-DVCT_STATS=1\")");
#endif
}

int vct_set_trace_timing(vct_ctx* c, int32_t on) {
    if (!c) return VCT_ERR_INVALID;
    c->time_traces = on != 0;
    return VCT_OK;
}

int vct_last_trace_ms(vct_ctx* c, float* ms) {
    if (!c || !ms) return VCT_ERR_INVALID;
    if (!c->have_trace) return fail(c, VCT_ERR_INVALID, "no trace has run");
    if (!c->last_trace_timed)
        return fail(c, VCT_ERR_INVALID, "vct_last_trace_ms: the last trace was issued with timing off (vct_set_trace_timing)");
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return VCT_OK;
}

int vct_selftest_const_divide(vct_ctx* c, float d, uint64_t* mismatches) {
    if (!c || !mismatches) return VCT_ERR_INVALID;
    if (!divisor_ok(d)) return fail(c, VCT_ERR_INVALID, "divisor outside the set the FMA division is proven for");
    HIP_TRY(c, hipSetDevice(c->device));
    // scratch = the statistics words (never the step-counter bank: vct_last_step_count sums that)
    HIP_TRY(c, hipMemsetAsync(c->stats, 0, 2 * sizeof(unsigned long long), c->stream));
    HIP_TRY(c, vct_launch_divide_selftest(d, c->stats, c->stream));
    unsigned long long v[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(v, c->stats, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *mismatches = v[0];
    if (v[0]) {      // not a failure of the call: leave one offending x readable for diagnosis
        char msg[96];
        snprintf(msg, sizeof(msg), "const divide by %.9g: %llu mismatches, e.g. x bits 0x%08llx", d, v[0], v[1]);
        c->err = msg;
    }
    return VCT_OK;
}

// The trace kernels fetch texels through typed-buffer loads and rely on the texture path converting a UNORM8 channel to
// exactly (float)c / 255.0f.  Every byte value in every channel position (1,024 texels) through that path against the
// library's exact decode; *mismatches = channels that differ (0 on gfx950: tools/unorm_probe.hip).
int vct_selftest_texel_buffer(vct_ctx* c, uint64_t* mismatches) {
    if (!c || !mismatches) return VCT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<uint32_t> h(1024);
    for (uint32_t i = 0; i < 1024u; ++i) {
        const uint32_t b = i & 255u, k = i >> 8;       // byte b in channel k, the other channels vary with it
        const uint32_t o0 = (b * 7u + 3u) & 255u, o1 = 255u - b, o2 = (b * 13u + 5u) & 255u;
        const uint32_t ch[4] = {o0, o1, o2, b};
        h[i] = ch[(0 + 3 - k) & 3] | (ch[(1 + 3 - k) & 3] << 8) | (ch[(2 + 3 - k) & 3] << 16) | (ch[(3 + 3 - k) & 3] << 24);
    }
    uint32_t* d = nullptr;
    HIP_TRY(c, hipMalloc(&d, h.size() * sizeof(uint32_t)));
    hipError_t e = hipMemcpyAsync(d, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->stats, 0, 2 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = vct_launch_texel_buffer_selftest(d, (uint32_t)h.size(), c->stats, c->stream);
    unsigned long long v[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(v, c->stats, sizeof(v), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIP_TRY(c, e);
    *mismatches = v[0];
    if (v[0]) {
        char msg[128];
        snprintf(msg, sizeof(msg), "texel buffer: %llu channel values differ from (float)c / 255.0f, e.g. texel 0x%08llx", v[0], v[1]);
        c->err = msg;
    }
    return VCT_OK;
}

int vct_selftest_area_divide(vct_ctx* c, uint64_t seed, uint64_t count, uint64_t* mismatches) {
    if (!c || !mismatches) return VCT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemsetAsync(c->stats, 0, 2 * sizeof(unsigned long long), c->stream));
    HIP_TRY(c, vct_launch_area_divide_selftest(seed, count, c->stats, c->stream));
    unsigned long long v[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(v, c->stats, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *mismatches = v[0];
    if (v[0]) {
        char msg[96];
        snprintf(msg, sizeof(msg), "area divide: %llu mismatches, e.g. sample %llu of seed %llu", v[0], v[1],
                 (unsigned long long)seed);
        c->err = msg;
    }
    return VCT_OK;
}

int vct_get_stream(vct_ctx* c, void** s) {
    if (!c || !s) return VCT_ERR_INVALID;
    *s = (void*)c->stream;
    return VCT_OK;
}

int vct_get_frame_device(vct_ctx* c, void** p, size_t* bytes) {
    if (!c || !p) return VCT_ERR_INVALID;
    *p = c->frame_target ? c->frame_target : c->frame;
    if (bytes) *bytes = (size_t)c->cfg.width * c->cfg.height * 8;
    return VCT_OK;
}

}  // extern "C"
