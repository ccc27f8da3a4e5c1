// vct_multi.hip -- one frame across the GPUs of a node: screen-tile slabs + ONE RCCL gather per frame
// (BASELINE.json config 4).
//
// The reference is single-GPU (R/main.cpp:77-94 is one GL context), so nothing here has a counterpart in
// it.  Pixels are independent given the read-only chain (S/VoxelConeTracing.fs:165-228 reads only its own
// varyings and the shared textures), so the frame's 8-pixel tile rows are cut into `world` equal
// contiguous slabs; every rank holds its own replica of scene + chain, rasterises and traces only its
// slab, and the root receives the other ranks' slabs with a single ncclGather of padded equal slabs
// (/opt/rocm/include/rccl/rccl.h: ncclGather, in place on the root) -- root fan-in over its direct xGMI
// links, one hop per link, no ring.  No collective touches the data path of the trace itself.
//
// One process per GPU.  Per frame the host issues, natively: wait(buffer k free) -> trace kernel writes
// the slab straight into gather buffer k (vct_set_frame_target-style full-frame addressing, no copy) ->
// event -> ncclGather on a communication stream -> event.  Two buffers: the gather of frame k overlaps
// the trace of frame k+1.
//
// RCCL is loaded lazily (dlopen) so single-GPU users of libvct_amd.so never pay for it and the library
// carries no link-time dependency on a particular librccl build (a Python process that already loaded
// torch's bundled RCCL reuses that one).
#include <dlfcn.h>
#include <string.h>

#include <string>

#include "vct_ctx.h"

namespace {

// the five RCCL entry points used, with the types of rccl.h restated (opaque comm, 128-byte id)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclFloat16 = 6 };
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl* rccl() {
    static Rccl r;
    if (r.lib || !r.err.empty()) return &r;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) { r.err = std::string("cannot load RCCL: ") + dlerror(); return &r; }
    auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p) r.err = std::string("RCCL lacks ") + n; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.Gather = (decltype(r.Gather))sym("ncclGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    return &r;
}

}  // namespace

struct vct_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    int row0 = 0, row1 = 0;            // tile rows of this rank's slab
    int rows_per_rank = 0;             // padded slab height in tile rows (equal on every rank)
    size_t slab_halves = 0;            // halves per padded slab = the gather's sendcount
    hipStream_t comm_stream = nullptr;
    uint16_t* buf[2] = {nullptr, nullptr};   // root: world padded slabs (the frame); others: one padded slab
    hipEvent_t traced[2] = {nullptr, nullptr}, gathered[2] = {nullptr, nullptr};
    unsigned long long frames = 0;     // steps issued
    int last = -1;                     // buffer of the last issued step
};

#define NCCL_TRY(c, expr)                                                                        \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess)                                                                   \
            return vct_fail((c), VCT_ERR_DEVICE, std::string(#expr) + ": " + rccl()->GetErrorString(r_)); \
    } while (0)

void vct_comm_release(vct_ctx* c) {
    if (!c || !c->comm) return;
    vct_comm* m = c->comm;
    if (m->comm_stream) (void)hipStreamSynchronize(m->comm_stream);
    if (m->comm && rccl()->CommDestroy) (void)rccl()->CommDestroy(m->comm);
    for (int k = 0; k < 2; ++k) {
        if (m->buf[k]) (void)hipFree(m->buf[k]);
        if (m->traced[k]) (void)hipEventDestroy(m->traced[k]);
        if (m->gathered[k]) (void)hipEventDestroy(m->gathered[k]);
    }
    if (m->comm_stream) (void)hipStreamDestroy(m->comm_stream);
    delete m;
    c->comm = nullptr;
    c->frame_target = nullptr;
}

extern "C" {

int vct_slab_partition(int32_t height, int32_t world, int32_t rank, int32_t* row0, int32_t* row1,
                       int32_t* rows_per_rank) {
    if (height <= 0 || world <= 0 || rank < 0 || rank >= world) return VCT_ERR_INVALID;
    const int ty = (height + VCT_TILE - 1) / VCT_TILE;
    const int per = (ty + world - 1) / world;
    const int r0 = rank * per < ty ? rank * per : ty;
    if (row0) *row0 = r0;
    if (row1) *row1 = r0 + per < ty ? r0 + per : ty;
    if (rows_per_rank) *rows_per_rank = per;
    return VCT_OK;
}

int vct_comm_get_unique_id(void* id128) {
    if (!id128) return VCT_ERR_INVALID;
    Rccl* r = rccl();
    if (!r->err.empty()) return vct_fail(nullptr, VCT_ERR_DEVICE, r->err);
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess) return vct_fail(nullptr, VCT_ERR_DEVICE, "ncclGetUniqueId failed");
    memcpy(id128, id.internal, 128);
    return VCT_OK;
}

int vct_comm_init(vct_ctx* c, const void* id128, int32_t rank, int32_t world) {
    if (!c) return VCT_ERR_INVALID;
    if (!id128 || world <= 0 || rank < 0 || rank >= world) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_init: bad rank / world / id");
    if (c->comm) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_init: already initialised (vct_comm_destroy first)");
    Rccl* r = rccl();
    if (!r->err.empty()) return vct_fail(c, VCT_ERR_DEVICE, r->err);
    HIP_TRY(c, hipSetDevice(c->device));
    vct_comm* m = new vct_comm();
    c->comm = m;
    m->rank = rank;
    m->world = world;
    vct_slab_partition(c->cfg.height, world, rank, &m->row0, &m->row1, &m->rows_per_rank);
    m->slab_halves = (size_t)m->rows_per_rank * VCT_TILE * c->cfg.width * 4;
    const size_t slab_bytes = m->slab_halves * 2;
    HIP_TRY(c, hipStreamCreateWithFlags(&m->comm_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        const size_t bytes = rank == 0 ? slab_bytes * world : slab_bytes;
        HIP_TRY(c, hipMalloc(&m->buf[k], bytes));
        HIP_TRY(c, hipMemsetAsync(m->buf[k], 0, bytes, c->stream));
        HIP_TRY(c, hipEventCreateWithFlags(&m->traced[k], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&m->gathered[k], hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(m->gathered[k], c->stream));      // "previous gather" of the first use
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    NCCL_TRY(c, r->CommInitRank(&m->comm, world, id, rank));
    return VCT_OK;
}

int vct_comm_destroy(vct_ctx* c) {
    if (!c) return VCT_ERR_INVALID;
    vct_comm_release(c);
    return VCT_OK;
}

int vct_comm_slab(vct_ctx* c, int32_t* row0, int32_t* row1) {
    if (!c) return VCT_ERR_INVALID;
    if (!c->comm) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_slab: call vct_comm_init first");
    if (row0) *row0 = c->comm->row0;
    if (row1) *row1 = c->comm->row1;
    return VCT_OK;
}

// One frame: trace this rank's slab into gather buffer k and start its gather.  Asynchronous.
int vct_frame_step(vct_ctx* c) {
    if (!c) return VCT_ERR_INVALID;
    vct_comm* m = c->comm;
    if (!m) return vct_fail(c, VCT_ERR_INVALID, "vct_frame_step: call vct_comm_init first");
    if (!c->have_gbuffer) return vct_fail(c, VCT_ERR_INVALID, "vct_frame_step: no G-buffer resident yet");
    HIP_TRY(c, hipSetDevice(c->device));
    const int k = (int)(m->frames & 1ull);
    // this rank's padded slab inside buffer k; full-frame addressing = that address minus the slab's first row
    uint16_t* slab = m->buf[k];      // root: slab 0 of its frame buffer (in-place gather); others: their slab buffer
    const size_t first_row_halves = (size_t)m->row0 * VCT_TILE * c->cfg.width * 4;
    HIP_TRY(c, hipStreamWaitEvent(c->stream, m->gathered[k], 0));    // buffer k is free once its last gather is done
    c->frame_target = slab - first_row_halves;
    int rc = VCT_OK;
    if (m->row1 > m->row0) rc = vct_launch_trace_rows(c, m->row0, m->row1);
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(m->traced[k], c->stream));
    HIP_TRY(c, hipStreamWaitEvent(m->comm_stream, m->traced[k], 0));
    // ONE collective per frame.  Root: in place (its slab already sits at offset rank * sendcount = 0).
    NCCL_TRY(c, rccl()->Gather(slab, m->rank == 0 ? m->buf[k] : nullptr, m->slab_halves, ncclFloat16, 0, m->comm,
                               m->comm_stream));
    HIP_TRY(c, hipEventRecord(m->gathered[k], m->comm_stream));
    m->last = k;
    ++m->frames;
    return VCT_OK;
}

int vct_comm_sync(vct_ctx* c) {
    if (!c) return VCT_ERR_INVALID;
    if (!c->comm) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_sync: call vct_comm_init first");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->comm->comm_stream));
    return VCT_OK;
}

int vct_comm_frame(vct_ctx* c, void** dev, size_t* bytes) {
    if (!c || !dev) return VCT_ERR_INVALID;
    vct_comm* m = c->comm;
    if (!m || m->last < 0) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_frame: no frame gathered yet");
    if (m->rank != 0) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_frame: only the root (rank 0) holds the frame");
    *dev = m->buf[m->last];
    if (bytes) *bytes = (size_t)c->cfg.width * c->cfg.height * 8;
    return VCT_OK;
}

int vct_comm_download_frame(vct_ctx* c, void* out) {
    if (!c || !out) return VCT_ERR_INVALID;
    void* dev = nullptr;
    size_t bytes = 0;
    int rc = vct_comm_frame(c, &dev, &bytes);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->comm->comm_stream));
    HIP_TRY(c, hipMemcpy(out, dev, bytes, hipMemcpyDeviceToHost));
    return VCT_OK;
}

}  // extern "C"
