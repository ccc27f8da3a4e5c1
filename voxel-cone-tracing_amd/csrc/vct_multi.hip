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
// event -> ncclGather on a communication stream -> event.  Two buffers, so that frame k+1 may be traced while frame
// k is still being gathered -- on one GPU that overlap was measured NOT to happen (0 of 103 gather copies ran beside a
// trace, profiles/experiments/r04_gather_timeline.txt: the gather's kernels wait for the wave slots the trace holds); a
// frame costs slab trace + dependent dispatch (~23 us) + wire time.  vct_comm_last_gather_ms reports the gather alone.
//
// RCCL is loaded lazily (dlopen) so single-GPU users of libvct_amd.so never pay for it and the library
// carries no link-time dependency on a particular librccl build (a Python process that already loaded
// torch's bundled RCCL reuses that one).
#include <dlfcn.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "vct_ctx.h"

namespace {

// the RCCL entry points used, with the types of rccl.h restated (opaque comm, 128-byte id)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclInProgress = 7, ncclFloat16 = 6 };
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(ncclComm_t, int*) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
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
    r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
    r.CommGetAsyncError = (decltype(r.CommGetAsyncError))sym("ncclCommGetAsyncError");
    r.Gather = (decltype(r.Gather))sym("ncclGather");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    // optional (diagnostics only: vct_comm_info)
    r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.lib, "ncclCommUserRank");
    r.CommCuDevice = (decltype(r.CommCuDevice))dlsym(r.lib, "ncclCommCuDevice");
    r.GetVersion = (decltype(r.GetVersion))dlsym(r.lib, "ncclGetVersion");
    return &r;
}

// gathered[rank][local row j][8 pixel rows][w] -> frame row (j * world + rank) * 8 + ...; one 8-byte pixel per thread step
__global__ void __launch_bounds__(256)
k_deinterleave(const uint2* __restrict__ gathered, uint2* __restrict__ frame, int w, int h, int world, size_t slab_pixels) {
    const size_t n = (size_t)w * h;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / (size_t)w), x = (int)(i - (size_t)y * w);
        const int tr = y >> 3, r = tr % world, j = tr / world;
        frame[i] = gathered[(size_t)r * slab_pixels + (size_t)(j * 8 + (y & 7)) * w + x];
    }
}

// ---- direct slabs (VCT_COMM_MODE=direct; experimental, default off) ------------------------------------------------
// Every rank's trace kernel stores its slab straight into the ROOT's frame buffers (mapped through hipIpc handles: 8 B
// per pixel over xGMI), and the frame's exchange step is two flags instead of a collective: no RCCL at all.
//   rendezvous   a POSIX shared-memory block named after the communicator id: the root publishes the IPC handles of
//                its two frame buffers there, the ranks open them; the same block -- page-locked and mapped into every
//                rank's GPU (coherent host memory) -- carries the flags
//   step f, buffer k = f & 1, generation g = f / 2 + 1
//     rank r > 0:  wait release[k] >= g - 1 (the root is past frame f - 2)  ->  trace into the mapped frame  ->  done[k][r] = g
//     root:        release[k] = g - 1 (ordered behind everything it issued on its stream: the consumer of frame f - 2)
//                  ->  trace its own slab  ->  wait done[k][r] >= g for every r  ->  (de-interleave)  ->  frame f is there
//   Every wait is one lane spinning on a system-scope load with a deadline (the communicator's timeout): a dead peer
//   costs a wrong frame and an error from vct_comm_sync, never a hang.
// NOT validated across GPUs (one GPU per box here): what is tested is two processes on ONE GPU (IPC mapping, flags,
// ordering, the failure path).  Whether peer stores are visible to the root's next kernel without more than the
// end-of-kernel release is what the first run on a multi-GPU node has to show; hence off by default.
#define VCT_DIRECT_MAX_RANKS 64
struct DirectShm {
    uint32_t magic, world;
    volatile uint32_t handles_ready;                    // root: the two handles below are valid
    volatile uint32_t attached[VCT_DIRECT_MAX_RANKS];   // rank r: opened them
    hipIpcMemHandle_t frame[2];
    // flags (GPU-written, GPU-read; host memory mapped into every rank's device)
    uint32_t done[2][VCT_DIRECT_MAX_RANKS];
    uint32_t release[2];
    uint32_t timed_out;                                 // a wait gave up
};

// lanes 0 .. n-1 each wait for flags[lane * stride] >= want; deadline in wall-clock ticks (the device's
// hipDeviceAttributeWallClockRate, vct_comm::wall_khz).  `abort_word` is this process' own page-locked word: its host sets
// it before tearing the communicator down, so a wait still queued behind long compute gives up at once instead of
// polling memory that is about to be unmapped (advisor, round 5).
__global__ void k_flag_wait(const uint32_t* flags, int n, int stride, uint32_t want, uint32_t* timed_out, long long ticks,
                            const uint32_t* abort_word) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const long long t0 = (long long)wall_clock64();
    while (__hip_atomic_load(flags + (size_t)i * stride, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;   // local teardown: not a peer's fault
        if ((long long)wall_clock64() - t0 > ticks) {
            __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        __builtin_amdgcn_s_sleep(32);
    }
}
__global__ void k_flag_set(uint32_t* flag, uint32_t v) {
    __threadfence_system();
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

struct vct_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    int row0 = 0, row1 = 0;            // tile rows of this rank's slab
    int rows_per_rank = 0;             // padded slab height in tile rows of the EQUAL partition (the gather's sendcount)
    size_t slab_halves = 0;            // halves per padded equal slab
    // load-aware partition (vct_comm_set_slab_rows): tile-row boundaries of every rank, [world + 1]; empty = the equal
    // partition above.  Uneven slabs travel by grouped ncclSend / ncclRecv straight to their rows of the root's frame.
    std::vector<int> starts;
    // Interleaved slabs (vct_comm_set_interleaved): tile row r belongs to rank r % world -- every rank samples the whole
    // frame, so the slabs cost the same BY CONSTRUCTION (no histogram, no feedback rounds).  A rank traces its rows back
    // to back into its padded slab, the slabs travel with the one equal-count ncclGather, and the root de-interleaves
    // them into `il_frame` (one 8-byte copy per pixel on the communication stream, behind the gather).
    // VCT_COMM_STREAM=same: the gather is issued on the context's stream right behind the trace instead of on the
    // communication stream behind an event (profiles/experiments/README.md "gather beside the next trace")
    bool same_stream = false;
    bool comm_stream_overlaps = false;     // the communication stream was seen to run beside the context's stream
    bool interleaved = false;
    uint16_t* il_frame[2] = {nullptr, nullptr};      // root only
    size_t buf_halves = 0;             // allocation of each gather buffer
    hipStream_t comm_stream = nullptr;
    uint16_t* buf[2] = {nullptr, nullptr};   // root: the frame (world padded slabs); others: this rank's slab
    hipEvent_t traced[2] = {nullptr, nullptr}, gathered[2] = {nullptr, nullptr};
    hipEvent_t g0[2] = {nullptr, nullptr}, g1[2] = {nullptr, nullptr};     // timing: around the frame's exchange step
    unsigned long long frames = 0;     // steps issued
    int last = -1;                     // buffer of the last issued step
    // direct slabs (VCT_COMM_MODE=direct): no RCCL communicator; see DirectShm above
    bool direct = false;
    char shm_name[48] = {0};
    DirectShm* shm = nullptr;          // host mapping of the rendezvous block
    DirectShm* shm_dev = nullptr;      // the same block as this rank's GPU sees it
    uint16_t* peer_frame[2] = {nullptr, nullptr};      // ranks > 0: the root's frame buffers, mapped
    hipStream_t ctx_stream = nullptr;  // the context's stream (direct mode queues flag kernels and peer stores on it: drained at teardown)
    hipStream_t ctx_stream2 = nullptr; // ... and the other frame slot's, when the context runs two frames in flight
    uint32_t* abort_host = nullptr;    // this process' abort word (page-locked, mapped): k_flag_wait gives up when it is set
    uint32_t* abort_dev = nullptr;
    long long wall_khz = 100000;       // wall_clock64 rate of this device (hipDeviceAttributeWallClockRate)
    int timeout_ms = 60000;            // vct_comm_sync gives up after this long and aborts the communicator
    bool broken = false;               // the communicator was aborted (a peer died or hung): only vct_comm_destroy is left
};

// An RCCL call failed: the communicator may be half way through a collective the peers will never finish, so it
// is aborted (ncclCommAbort frees its resources without waiting for them) and the context keeps only the error.
static int comm_fail(vct_ctx* c, vct_comm* m, const std::string& what, ncclResult_t r) {
    std::string msg = what + ": " + (rccl()->GetErrorString ? rccl()->GetErrorString(r) : "RCCL error");
    if (m && m->comm && rccl()->CommAbort) { (void)rccl()->CommAbort(m->comm); m->comm = nullptr; msg += " (communicator aborted)"; }
    if (m) m->broken = true;
    return vct_fail(c, VCT_ERR_DEVICE, msg);
}

#define NCCL_TRY(c, m, expr)                                              \
    do {                                                                  \
        ncclResult_t r_ = (expr);                                         \
        if (r_ != ncclSuccess) return comm_fail((c), (m), #expr, r_);     \
    } while (0)

static void comm_free(vct_comm* m) {
    if (!m) return;
    if (m->direct) {
        // Direct slabs: flag waits and the trace kernels that store into the root's mapped frame may still be QUEUED
        // (a host-side deadline fires before a wait that sits behind long compute has even started).  They must have
        // left the GPU before the mappings they use go away -- so: raise the abort word (every wait of this process
        // returns at its next poll; each is deadline-bounded anyway), drain BOTH streams, and only then close the IPC
        // handles and unregister the block.  Cannot hang: nothing left in the queues waits on a peer.
        if (m->abort_host) __atomic_store_n(m->abort_host, 1u, __ATOMIC_RELEASE);
        if (m->ctx_stream) (void)hipStreamSynchronize(m->ctx_stream);
        if (m->ctx_stream2) (void)hipStreamSynchronize(m->ctx_stream2);
        if (m->comm_stream) (void)hipStreamSynchronize(m->comm_stream);
        for (int k = 0; k < 2; ++k) if (m->peer_frame[k]) (void)hipIpcCloseMemHandle(m->peer_frame[k]);
        if (m->shm) { (void)hipHostUnregister(m->shm); munmap(m->shm, sizeof(DirectShm)); }
        if (m->rank == 0 && m->shm_name[0]) shm_unlink(m->shm_name);
        if (m->abort_host) (void)hipHostFree(m->abort_host);
    } else if (m->comm_stream && !m->broken) {
        (void)hipStreamSynchronize(m->comm_stream);
    }
    if (m->comm) {
        if (m->broken && rccl()->CommAbort) (void)rccl()->CommAbort(m->comm);
        else if (rccl()->CommDestroy) (void)rccl()->CommDestroy(m->comm);
    }
    for (int k = 0; k < 2; ++k) {
        if (m->il_frame[k]) (void)hipFree(m->il_frame[k]);
        if (m->buf[k]) (void)hipFree(m->buf[k]);
        if (m->traced[k]) (void)hipEventDestroy(m->traced[k]);
        if (m->gathered[k]) (void)hipEventDestroy(m->gathered[k]);
        if (m->g0[k]) (void)hipEventDestroy(m->g0[k]);
        if (m->g1[k]) (void)hipEventDestroy(m->g1[k]);
    }
    if (m->comm_stream) (void)hipStreamDestroy(m->comm_stream);
    delete m;
}

void vct_comm_release(vct_ctx* c) {
    if (!c || !c->comm) return;
    // (frame slots may have been added or switched since vct_comm_init: the streams that carry this communicator's work NOW)
    c->comm->ctx_stream = c->stream;
    c->comm->ctx_stream2 = c->frames_in_flight > 1 ? c->slots[1 - c->cur_slot].stream : nullptr;
    comm_free(c->comm);
    c->comm = nullptr;
}

// slab of the attached communicator (vct_capi.hip: vct_gi_pass on a rank context); false when there is none
bool vct_comm_rows(const vct_ctx* c, int* row0, int* row1) {
    if (!c || !c->comm) return false;
    *row0 = c->comm->row0;
    *row1 = c->comm->row1;
    if (c->comm->interleaved) { *row0 = 0; *row1 = vct_tiles_y(c); }      // its rows are spread over the whole frame
    return true;
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

// Equal-WORK boundaries from a per-tile-row cost histogram (e.g. the executed cone steps per tile row of the previous
// frame, vct_last_row_steps): starts[r] = first tile row of rank r, starts[world] = tile_rows, chosen so that every
// slab's cost is as close to total / world as whole rows allow.  Pure host arithmetic; every rank computes the same
// boundaries from the same (all-reduced or root-broadcast) histogram.
int vct_slab_partition_weighted(const uint64_t* row_cost, int32_t tile_rows, int32_t world, int32_t* starts) {
    if (!row_cost || !starts || tile_rows <= 0 || world <= 0) return VCT_ERR_INVALID;
    long double total = 0.0L;
    for (int i = 0; i < tile_rows; ++i) total += (long double)row_cost[i] + 1.0L;     // +1: empty rows still cost a launch slot
    starts[0] = 0;
    long double acc = 0.0L;
    int row = 0;
    for (int r = 1; r < world; ++r) {
        const long double want = total * (long double)r / (long double)world;
        // advance while taking the next row leaves the prefix closer to the target than stopping here
        while (row < tile_rows) {
            const long double with = acc + (long double)row_cost[row] + 1.0L;
            if (with <= want || (with - want) < (want - acc)) { acc = with; ++row; } else break;
        }
        // every remaining rank must still be able to get a (possibly empty) slab inside the frame
        if (row < starts[r - 1]) row = starts[r - 1];
        starts[r] = row;
    }
    starts[world] = tile_rows;
    return VCT_OK;
}

int vct_comm_get_unique_id(void* id128) {
    if (!id128) return VCT_ERR_INVALID;
    const char* mode = getenv("VCT_COMM_MODE");
    if (mode && mode[0] == 'd') {              // direct slabs never touch RCCL: any 128 unpredictable bytes name the rendezvous
        FILE* fp = fopen("/dev/urandom", "rb");
        const bool ok = fp && fread(id128, 1, 128, fp) == 128;
        if (fp) fclose(fp);
        return ok ? VCT_OK : vct_fail(nullptr, VCT_ERR_DEVICE, "vct_comm_get_unique_id: /dev/urandom unreadable");
    }
    Rccl* r = rccl();
    if (!r->err.empty()) return vct_fail(nullptr, VCT_ERR_DEVICE, r->err);
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess) return vct_fail(nullptr, VCT_ERR_DEVICE, "ncclGetUniqueId failed");
    memcpy(id128, id.internal, 128);
    return VCT_OK;
}

// Rendezvous of the direct mode: shared block, the root's IPC handles, every rank's mapping.  Host-side waits are
// bounded by the communicator's timeout.
static int direct_attach(vct_ctx* c, vct_comm* m, const void* id128) {
    if (m->world > VCT_DIRECT_MAX_RANKS) return vct_fail(c, VCT_ERR_INVALID, "direct slabs: at most 64 ranks");
    const unsigned char* id = (const unsigned char*)id128;
    snprintf(m->shm_name, sizeof(m->shm_name), "/vct_%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x", id[0], id[1], id[2], id[3],
             id[4], id[5], id[6], id[7], id[8], id[9], id[10], id[11]);
    const auto t0 = std::chrono::steady_clock::now();
    auto expired = [&]() { return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > m->timeout_ms; };
    int fd = -1;
    if (m->rank == 0) {
        shm_unlink(m->shm_name);
        fd = shm_open(m->shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(DirectShm)) != 0) { if (fd >= 0) close(fd); return vct_fail(c, VCT_ERR_DEVICE, "direct slabs: cannot create the rendezvous block"); }
    } else {
        while ((fd = shm_open(m->shm_name, O_RDWR, 0600)) < 0) {
            if (expired()) return vct_fail(c, VCT_ERR_DEVICE, "direct slabs: the root's rendezvous block never appeared");
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        struct stat st;
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < sizeof(DirectShm)) {      // created, not sized yet
            if (expired()) { close(fd); return vct_fail(c, VCT_ERR_DEVICE, "direct slabs: rendezvous block not sized"); }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    void* p = mmap(nullptr, sizeof(DirectShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return vct_fail(c, VCT_ERR_DEVICE, "direct slabs: mmap failed");
    m->shm = (DirectShm*)p;
    hipError_t e = hipHostRegister(p, sizeof(DirectShm), hipHostRegisterMapped | hipHostRegisterPortable);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&m->shm_dev, p, 0);
    if (e != hipSuccess) { munmap(p, sizeof(DirectShm)); m->shm = nullptr; return vct_fail(c, VCT_ERR_DEVICE, std::string("direct slabs: hipHostRegister: ") + hipGetErrorString(e)); }
    if (m->rank == 0) {
        m->shm->world = (uint32_t)m->world;
        for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipIpcGetMemHandle(&m->shm->frame[k], m->buf[k]);
        if (e != hipSuccess) return vct_fail(c, VCT_ERR_DEVICE, std::string("direct slabs: hipIpcGetMemHandle: ") + hipGetErrorString(e));
        m->shm->magic = 0x56435444u;
        __atomic_store_n(&m->shm->handles_ready, 1u, __ATOMIC_RELEASE);
        for (int r = 1; r < m->world; ++r)
            while (!__atomic_load_n(&m->shm->attached[r], __ATOMIC_ACQUIRE)) {
                if (expired()) return vct_fail(c, VCT_ERR_DEVICE, "direct slabs: rank " + std::to_string(r) + " never attached");
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
    } else {
        while (!__atomic_load_n(&m->shm->handles_ready, __ATOMIC_ACQUIRE)) {
            if (expired()) return vct_fail(c, VCT_ERR_DEVICE, "direct slabs: the root never published its frame buffers");
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        for (int k = 0; k < 2 && e == hipSuccess; ++k)
            e = hipIpcOpenMemHandle((void**)&m->peer_frame[k], m->shm->frame[k], hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return vct_fail(c, VCT_ERR_DEVICE, std::string("direct slabs: hipIpcOpenMemHandle: ") + hipGetErrorString(e));
        __atomic_store_n(&m->shm->attached[m->rank], 1u, __ATOMIC_RELEASE);
    }
    return VCT_OK;
}

// All or nothing: the communicator is built into a local object and attached to the context only when every
// allocation and ncclCommInitRank succeeded; on any failure everything is released and the context is as before
// (a retry is possible, vct_frame_step keeps refusing).
int vct_comm_init(vct_ctx* c, const void* id128, int32_t rank, int32_t world) {
    if (!c) return VCT_ERR_INVALID;
    if (!id128 || world <= 0 || rank < 0 || rank >= world) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_init: bad rank / world / id");
    if (c->comm) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_init: already initialised (vct_comm_destroy first)");
    const char* mode = getenv("VCT_COMM_MODE");
    const bool direct = mode && mode[0] == 'd';       // direct slabs: no RCCL (see DirectShm)
    Rccl* r = direct ? nullptr : rccl();
    if (r && !r->err.empty()) return vct_fail(c, VCT_ERR_DEVICE, r->err);
    HIP_TRY(c, hipSetDevice(c->device));
    vct_comm* m = new vct_comm();
    m->rank = rank;
    m->world = world;
    m->direct = direct;
    vct_slab_partition(c->cfg.height, world, rank, &m->row0, &m->row1, &m->rows_per_rank);
    m->slab_halves = (size_t)m->rows_per_rank * VCT_TILE * c->cfg.width * 4;
    m->buf_halves = rank == 0 ? m->slab_halves * world : m->slab_halves;
    if (const char* t = getenv("VCT_COMM_TIMEOUT_MS")) { const int v = atoi(t); if (v > 0) m->timeout_ms = v; }
    if (const char* t = getenv("VCT_COMM_STREAM")) m->same_stream = t[0] == 's';
    hipError_t e;
    if (c->reserved_cus > 0) {       // the CUs the context's streams leave alone (VCT_COMM_RESERVED_CUS)
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, c->device);
        if (e == hipSuccess) e = vct_create_masked_stream(&m->comm_stream, c->device, prop.multiProcessorCount - c->reserved_cus, prop.multiProcessorCount);
    } else {
        // the exchange step is meant to run beside the next frame's trace: a stream that shares the context stream's hardware
        // queue would run behind it instead (round 6: HIP hands out four queues; RCCL and torch hold streams of their own)
        bool ov = false;
        e = vct_create_overlapping_stream(c, c->stream, &m->comm_stream, &ov) == VCT_OK ? hipSuccess : hipErrorUnknown;
        m->comm_stream_overlaps = ov;
    }
    m->ctx_stream = c->stream;
    if (direct && e == hipSuccess) {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && khz > 0) m->wall_khz = khz;
        e = hipHostMalloc((void**)&m->abort_host, sizeof(uint32_t), hipHostMallocMapped);
        if (e == hipSuccess) { *m->abort_host = 0u; e = hipHostGetDevicePointer((void**)&m->abort_dev, m->abort_host, 0); }
    }
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        if (!(direct && rank != 0)) {        // (a rank of the direct mode owns no slab buffer: it writes the root's)
            // Direct mode: peers store into the root's frame over IPC / xGMI and the root's next kernel reads it with only
            // a kernel-boundary acquire, which need not invalidate lines of LOCAL coarse-grained memory that the root's L2
            // kept from an earlier frame.  Fine-grained memory is not cached that way, so the frame buffers peers write
            // are allocated fine-grained (advisor, round 5); no fallback -- a coarse-grained frame could be silently stale.
            e = direct ? hipExtMallocWithFlags((void**)&m->buf[k], m->buf_halves * 2, hipDeviceMallocFinegrained)
                       : hipMalloc(&m->buf[k], m->buf_halves * 2);
            if (e == hipSuccess) e = hipMemsetAsync(m->buf[k], 0, m->buf_halves * 2, c->stream);
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->traced[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->gathered[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreate(&m->g0[k]);
        if (e == hipSuccess) e = hipEventCreate(&m->g1[k]);
        if (e == hipSuccess) e = hipEventRecord(m->gathered[k], c->stream);      // "previous gather" of the first use
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        comm_free(m);
        return vct_fail(c, e == hipErrorOutOfMemory ? VCT_ERR_NOMEM : VCT_ERR_DEVICE,
                        std::string("vct_comm_init: ") + hipGetErrorString(e));
    }
    if (direct) {
        const int rc = direct_attach(c, m, id128);
        if (rc) { comm_free(m); return rc; }
        c->comm = m;
        return VCT_OK;
    }
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    const ncclResult_t nr = r->CommInitRank(&m->comm, world, id, rank);
    if (nr != ncclSuccess) {
        m->comm = nullptr;          // nothing to destroy: ncclCommInitRank failed
        comm_free(m);
        return vct_fail(c, VCT_ERR_DEVICE, std::string("ncclCommInitRank: ") + r->GetErrorString(nr));
    }
    c->comm = m;
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

int vct_comm_set_timeout_ms(vct_ctx* c, int32_t ms) {
    if (!c) return VCT_ERR_INVALID;
    if (!c->comm) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_set_timeout_ms: call vct_comm_init first");
    if (ms <= 0) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_set_timeout_ms: timeout must be > 0");
    c->comm->timeout_ms = ms;
    return VCT_OK;
}

// Collective (every rank passes the same boundaries): slab r = tile rows [starts[r], starts[r+1]).  starts == NULL
// returns to the equal partition.  Frames in flight are drained first; a rank whose new slab outgrows its gather
// buffers gets larger ones.  Contiguous slabs and interleaved tile rows are alternatives: this call switches the
// interleaved assignment off (the last of vct_comm_set_slab_rows / vct_comm_set_interleaved wins).
int vct_comm_set_slab_rows(vct_ctx* c, const int32_t* starts) {
    if (!c) return VCT_ERR_INVALID;
    vct_comm* m = c->comm;
    if (!m || m->broken) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_set_slab_rows: no usable communicator (vct_comm_init)");
    const int ty = vct_tiles_y(c);
    if (starts) {
        if (starts[0] != 0 || starts[m->world] != ty) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_set_slab_rows: boundaries must run from 0 to the frame's tile rows");
        for (int r = 0; r < m->world; ++r)
            if (starts[r + 1] < starts[r]) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_set_slab_rows: boundaries must not decrease");
    }
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = vct_comm_sync(c);
    if (rc) return rc;
    m->interleaved = false;
    if (!starts) {
        m->starts.clear();
        vct_slab_partition(c->cfg.height, m->world, m->rank, &m->row0, &m->row1, nullptr);
    } else {
        m->starts.assign(starts, starts + m->world + 1);
        m->row0 = starts[m->rank];
        m->row1 = starts[m->rank + 1];
    }
    const size_t need = m->rank == 0 ? m->buf_halves : (size_t)(m->row1 - m->row0) * VCT_TILE * c->cfg.width * 4;
    if (need > m->buf_halves && !m->direct) {        // (direct slabs: a rank owns no buffer, it writes the root's frame)
        uint16_t* nb[2] = {nullptr, nullptr};
        for (int k = 0; k < 2; ++k) {
            const hipError_t e = hipMalloc(&nb[k], need * 2);
            if (e != hipSuccess) {
                if (nb[0]) (void)hipFree(nb[0]);
                return vct_fail(c, VCT_ERR_NOMEM, std::string("vct_comm_set_slab_rows: ") + hipGetErrorString(e));
            }
        }
        for (int k = 0; k < 2; ++k) { (void)hipFree(m->buf[k]); m->buf[k] = nb[k]; }
        m->buf_halves = need;
    }
    m->last = -1;
    return VCT_OK;
}

// Collective (every rank passes the same value): interleaved tile rows instead of contiguous slabs.  Frames in flight
// are drained first.  The load-aware boundaries are dropped (the two are alternatives).
int vct_comm_set_interleaved(vct_ctx* c, int32_t on) {
    if (!c) return VCT_ERR_INVALID;
    vct_comm* m = c->comm;
    if (!m || m->broken) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_set_interleaved: no usable communicator (vct_comm_init)");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = vct_comm_sync(c);
    if (rc) return rc;
    // the root's de-interleave target first: a failed allocation leaves this rank's state untouched.  (The other
    // ranks cannot see the root's failure: on VCT_ERR_NOMEM the caller must destroy the communicator on every rank.)
    if (on && m->rank == 0)
        for (int k = 0; k < 2; ++k)
            if (!m->il_frame[k]) {
                const hipError_t e = hipMalloc(&m->il_frame[k], (size_t)c->cfg.width * c->cfg.height * 8);
                if (e != hipSuccess) return vct_fail(c, VCT_ERR_NOMEM, std::string("vct_comm_set_interleaved: ") + hipGetErrorString(e));
            }
    m->starts.clear();
    vct_slab_partition(c->cfg.height, m->world, m->rank, &m->row0, &m->row1, nullptr);
    m->interleaved = on != 0;
    if (m->interleaved) {
        const int ty = vct_tiles_y(c);
        m->row0 = m->rank < ty ? m->rank : ty;
        m->row1 = ty;
    }
    m->last = -1;
    return VCT_OK;
}

// One frame: trace this rank's slab into gather buffer k and start its gather.  Asynchronous.
int vct_frame_step(vct_ctx* c) {
    if (!c) return VCT_ERR_INVALID;
    vct_comm* m = c->comm;
    if (!m) return vct_fail(c, VCT_ERR_INVALID, "vct_frame_step: call vct_comm_init first");
    if (m->broken || (!m->comm && !m->direct)) return vct_fail(c, VCT_ERR_DEVICE, "vct_frame_step: the communicator was aborted (vct_comm_destroy, then vct_comm_init again)");
    if (!c->have_gbuffer) return vct_fail(c, VCT_ERR_INVALID, "vct_frame_step: no G-buffer resident yet");
    HIP_TRY(c, hipSetDevice(c->device));
    const int k = (int)(m->frames & 1ull);
    // This rank's slab inside buffer k.  The kernel addresses the full frame, so it gets the slab's address minus the
    // slab's first row as its output base -- passed to the launch, never stored in the context (on a non-root rank
    // that address lies before the allocation).  Root: the frame buffer itself (slab 0 at offset 0, in-place gather).
    uint16_t* slab = m->buf[k];
    const size_t first_row_halves = (size_t)m->row0 * VCT_TILE * c->cfg.width * 4;
    HIP_TRY(c, hipStreamWaitEvent(c->stream, m->gathered[k], 0));    // buffer k is free once its last gather is done
    const bool uneven = !m->starts.empty();
    if (m->direct) {
        // direct slabs (see DirectShm): flags instead of a collective, the slab stored straight into the root's frame
        const uint32_t gen = (uint32_t)(m->frames >> 1) + 1u;
        const long long ticks = (long long)m->timeout_ms * m->wall_khz;        // wall_clock64 ticks per ms = its rate in kHz
        const size_t row_halves = (size_t)VCT_TILE * c->cfg.width * 4;
        if (m->rank == 0) {
            hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(1), 0, c->stream, &m->shm_dev->release[k], gen - 1u);
            HIP_TRY(c, hipGetLastError());
            uint16_t* base = uneven ? m->buf[k] : slab - first_row_halves;
            const int rc = m->interleaved ? vct_launch_trace_rows(c, m->row0, m->row1, slab, m->world, true)
                                          : vct_launch_trace_rows(c, m->row0, m->row1, base);
            if (rc) return rc;
            const hipStream_t cs = m->same_stream ? c->stream : m->comm_stream;
            if (!m->same_stream) {
                HIP_TRY(c, hipEventRecord(m->traced[k], c->stream));
                HIP_TRY(c, hipStreamWaitEvent(m->comm_stream, m->traced[k], 0));
            }
            HIP_TRY(c, hipEventRecord(m->g0[k], cs));
            if (m->world > 1) {
                hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, cs, &m->shm_dev->done[k][1], m->world - 1, 1, gen,
                                   &m->shm_dev->timed_out, ticks, m->abort_dev);
                HIP_TRY(c, hipGetLastError());
            }
            if (m->interleaved) {
                hipLaunchKernelGGL(k_deinterleave, dim3(2048), dim3(256), 0, cs, (const uint2*)m->buf[k],
                                   (uint2*)m->il_frame[k], c->cfg.width, c->cfg.height, m->world, m->slab_halves / 4);
                HIP_TRY(c, hipGetLastError());
            }
            HIP_TRY(c, hipEventRecord(m->g1[k], cs));
            HIP_TRY(c, hipEventRecord(m->gathered[k], cs));
        } else {
            hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, c->stream, &m->shm_dev->release[k], 1, 1, gen - 1u,
                               &m->shm_dev->timed_out, ticks, m->abort_dev);
            HIP_TRY(c, hipGetLastError());
            // where this rank's slab lives in the root's buffer k: packed slab `rank` (equal / interleaved) or its own rows
            uint16_t* peer_slab = m->peer_frame[k] + (uneven ? (size_t)m->starts[m->rank] * row_halves : (size_t)m->rank * m->slab_halves);
            const int rc = m->interleaved ? vct_launch_trace_rows(c, m->row0, m->row1, peer_slab, m->world, true)
                                          : vct_launch_trace_rows(c, m->row0, m->row1, peer_slab - first_row_halves);
            if (rc) return rc;
            HIP_TRY(c, hipEventRecord(m->g0[k], c->stream));
            hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(1), 0, c->stream, &m->shm_dev->done[k][m->rank], gen);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipEventRecord(m->g1[k], c->stream));
            HIP_TRY(c, hipEventRecord(m->gathered[k], c->stream));
        }
        m->last = k;
        ++m->frames;
        return VCT_OK;
    }
    uint16_t* out_base = (m->rank == 0 && uneven) ? m->buf[k] : slab - first_row_halves;
    // interleaved: every world-th tile row from `rank` on, back to back at the start of the slab
    const int rc = m->interleaved ? vct_launch_trace_rows(c, m->row0, m->row1, slab, m->world, true)
                                  : vct_launch_trace_rows(c, m->row0, m->row1, out_base);     // an empty slab launches nothing
    if (rc) return rc;
    const hipStream_t cs = m->same_stream ? c->stream : m->comm_stream;
    if (!m->same_stream) {
        HIP_TRY(c, hipEventRecord(m->traced[k], c->stream));
        HIP_TRY(c, hipStreamWaitEvent(m->comm_stream, m->traced[k], 0));
    }
    Rccl* r = rccl();
    HIP_TRY(c, hipEventRecord(m->g0[k], cs));
    if (m->interleaved) {
        NCCL_TRY(c, m, r->Gather(slab, m->rank == 0 ? m->buf[k] : nullptr, m->slab_halves, ncclFloat16, 0, m->comm,
                                 cs));
        if (m->rank == 0) {
            hipLaunchKernelGGL(k_deinterleave, dim3(2048), dim3(256), 0, cs, (const uint2*)m->buf[k],
                               (uint2*)m->il_frame[k], c->cfg.width, c->cfg.height, m->world, m->slab_halves / 4);
            HIP_TRY(c, hipGetLastError());
        }
    } else if (!uneven) {
        // ONE collective per frame.  Root: in place (its slab already sits at offset rank * sendcount = 0).
        NCCL_TRY(c, m, r->Gather(slab, m->rank == 0 ? m->buf[k] : nullptr, m->slab_halves, ncclFloat16, 0, m->comm, cs));
    } else {
        // load-aware slabs differ in size: one fused group of point-to-point transfers, each slab straight to its
        // rows of the root's frame (what ncclGather is made of inside RCCL, with per-rank counts)
        const size_t row_halves = (size_t)VCT_TILE * c->cfg.width * 4;
        NCCL_TRY(c, m, r->GroupStart());
        ncclResult_t gr = ncclSuccess;
        if (m->rank == 0) {
            for (int peer = 1; peer < m->world && gr == ncclSuccess; ++peer) {
                const size_t n = (size_t)(m->starts[peer + 1] - m->starts[peer]) * row_halves;
                if (n) gr = r->Recv(m->buf[k] + (size_t)m->starts[peer] * row_halves, n, ncclFloat16, peer, m->comm, cs);
            }
        } else {
            const size_t n = (size_t)(m->row1 - m->row0) * row_halves;
            if (n) gr = r->Send(slab, n, ncclFloat16, 0, m->comm, cs);
        }
        const ncclResult_t ge = r->GroupEnd();
        if (gr != ncclSuccess) return comm_fail(c, m, "ncclSend/ncclRecv", gr);
        if (ge != ncclSuccess) return comm_fail(c, m, "ncclGroupEnd", ge);
    }
    HIP_TRY(c, hipEventRecord(m->g1[k], cs));
    HIP_TRY(c, hipEventRecord(m->gathered[k], cs));
    m->last = k;
    ++m->frames;
    return VCT_OK;
}

// Waits for the trace and the gather streams -- with a deadline: a rank that died or hangs leaves its peers inside
// the collective forever, so after timeout_ms (vct_comm_set_timeout_ms / VCT_COMM_TIMEOUT_MS, default 60 s) or on an
// asynchronous RCCL error the communicator is aborted and the call fails; the context itself stays usable for
// single-GPU calls after vct_comm_destroy.
int vct_comm_sync(vct_ctx* c) {
    if (!c) return VCT_ERR_INVALID;
    vct_comm* m = c->comm;
    if (!m) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_sync: call vct_comm_init first");
    if (m->broken) return vct_fail(c, VCT_ERR_DEVICE, "vct_comm_sync: the communicator was aborted");
    HIP_TRY(c, hipSetDevice(c->device));
    // BOTH streams are polled under the deadline: with frames issued back to back the context stream itself waits on
    // gathered[k] (vct_frame_step), i.e. behind a collective a dead peer never finishes -- a blocking
    // hipStreamSynchronize(c->stream) here would hang before the deadline loop was ever reached.
    // After the abort below the stuck collective's kernel ends, its event completes and both streams drain.
    // Compute queued on the context's stream in front of the exchange (several 4K / 1024^3 GI passes, a first-use divisor
    // self-test) is not communication: it gets a deadline of its own -- the last step's "slab traced" event -- and the
    // communication deadline starts when that has fired (advisor, round 4).  Both are bounded: a trace that waits behind
    // the gather of a dead peer two frames back never fires the event, and the second loop then aborts as before.
    if (m->last >= 0 && !m->same_stream && !m->direct && m->traced[m->last]) {
        const auto tc = std::chrono::steady_clock::now();
        int sp = 0;
        while (hipEventQuery(m->traced[m->last]) == hipErrorNotReady) {
            if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - tc).count() > m->timeout_ms) break;
            if (++sp < 2000) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (true) {
        hipError_t qc = hipStreamQuery(c->stream);
        if (qc != hipSuccess && qc != hipErrorNotReady) HIP_TRY(c, qc);
        if (qc == hipSuccess && c->frames_in_flight > 1) {      // two frames in flight: the other slot's stream carries steps too
            qc = hipStreamQuery(c->slots[1 - c->cur_slot].stream);
            if (qc != hipSuccess && qc != hipErrorNotReady) HIP_TRY(c, qc);
        }
        const hipError_t q = hipStreamQuery(m->comm_stream);
        if (q == hipSuccess && qc == hipSuccess) {
            if (m->direct && __atomic_load_n(&m->shm->timed_out, __ATOMIC_ACQUIRE)) {
                m->broken = true;
                return vct_fail(c, VCT_ERR_DEVICE, "vct_comm_sync: a rank's flag wait gave up after " + std::to_string(m->timeout_ms) +
                                                   " ms (a peer died or hangs); the frame is incomplete, communicator unusable");
            }
            return VCT_OK;
        }
        if (q != hipSuccess && q != hipErrorNotReady) HIP_TRY(c, q);
        ncclResult_t async = ncclSuccess;
        if (m->comm && rccl()->CommGetAsyncError && rccl()->CommGetAsyncError(m->comm, &async) == ncclSuccess &&
            async != ncclSuccess && async != ncclInProgress)
            return comm_fail(c, m, "vct_comm_sync: asynchronous RCCL error", async);
        const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (ms > m->timeout_ms) {
            if (m->comm && rccl()->CommAbort) { (void)rccl()->CommAbort(m->comm); m->comm = nullptr; }
            m->broken = true;
            return vct_fail(c, VCT_ERR_DEVICE, "vct_comm_sync: the gather did not complete within " + std::to_string(m->timeout_ms) +
                                               " ms (a peer rank died or hangs); communicator aborted");
        }
        if (++spins < 2000) std::this_thread::yield();                         // the common case: done within microseconds
        else std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

// Diagnostics for multi-GPU bench lines: what RCCL itself says about the communicator.
// out[0] = ncclCommCount, out[1] = ncclCommUserRank, out[2] = ncclCommCuDevice, out[3] = ncclGetVersion (-1: not exported)
int vct_comm_info(vct_ctx* c, int32_t out[4]) {
    if (!c || !out) return VCT_ERR_INVALID;
    vct_comm* m = c->comm;
    if (m && m->direct) {        // direct slabs: no RCCL; what the rendezvous block says, version 0
        out[0] = (int32_t)m->shm->world; out[1] = m->rank; out[2] = c->device; out[3] = 0;
        return VCT_OK;
    }
    if (!m || !m->comm) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_info: no usable communicator (vct_comm_init)");
    Rccl* r = rccl();
    int v[4] = {-1, -1, -1, -1};
    if (r->CommCount) (void)r->CommCount(m->comm, &v[0]);
    if (r->CommUserRank) (void)r->CommUserRank(m->comm, &v[1]);
    if (r->CommCuDevice) (void)r->CommCuDevice(m->comm, &v[2]);
    if (r->GetVersion) (void)r->GetVersion(&v[3]);
    for (int i = 0; i < 4; ++i) out[i] = v[i];
    return VCT_OK;
}

// Device time of the LAST frame's exchange step on this rank (the ncclGather / the send-recv group, plus the root's
// de-interleave in interleaved mode), between two events on the stream it ran on.  Waits for that step (vct_comm_sync).
int vct_comm_last_gather_ms(vct_ctx* c, float* ms) {
    if (!c || !ms) return VCT_ERR_INVALID;
    vct_comm* m = c->comm;
    if (!m || m->last < 0) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_last_gather_ms: no frame stepped yet");
    const int rc = vct_comm_sync(c);
    if (rc) return rc;
    HIP_TRY(c, hipEventElapsedTime(ms, m->g0[m->last], m->g1[m->last]));
    return VCT_OK;
}

int vct_comm_frame(vct_ctx* c, void** dev, size_t* bytes) {
    if (!c || !dev) return VCT_ERR_INVALID;
    vct_comm* m = c->comm;
    if (!m || m->last < 0) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_frame: no frame gathered yet");
    if (m->rank != 0) return vct_fail(c, VCT_ERR_INVALID, "vct_comm_frame: only the root (rank 0) holds the frame");
    *dev = m->interleaved ? m->il_frame[m->last] : m->buf[m->last];
    if (bytes) *bytes = (size_t)c->cfg.width * c->cfg.height * 8;
    return VCT_OK;
}

int vct_comm_download_frame(vct_ctx* c, void* out) {
    if (!c || !out) return VCT_ERR_INVALID;
    void* dev = nullptr;
    size_t bytes = 0;
    int rc = vct_comm_frame(c, &dev, &bytes);
    if (rc) return rc;
    rc = vct_comm_sync(c);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpy(out, dev, bytes, hipMemcpyDeviceToHost));
    return VCT_OK;
}

// The interleaved step's data path on ONE GPU, for any world size (no multi-GPU box is needed to check it): the resident
// G-buffer is traced once per emulated rank -- its rows, strided and packed, into slab `rank` of a stand-in gather
// buffer, exactly as vct_frame_step does -- the root's de-interleave kernel runs over that buffer, and the result is
// compared with the frame one launch traces.  *mismatches = pixels that differ (0 expected).
int vct_selftest_interleaved(vct_ctx* c, int32_t world, uint64_t* mismatches) {
    if (!c || !mismatches || world < 1) return VCT_ERR_INVALID;
    if (!c->have_gbuffer) return vct_fail(c, VCT_ERR_INVALID, "vct_selftest_interleaved: no G-buffer resident yet");
    HIP_TRY(c, hipSetDevice(c->device));
    const int w = c->cfg.width, h = c->cfg.height, ty = vct_tiles_y(c);
    const int per = (ty + world - 1) / world;
    const size_t slab_pixels = (size_t)per * VCT_TILE * w, npix = (size_t)w * h;
    uint2 *gathered = nullptr, *il = nullptr, *plain = nullptr;
    auto done = [&](int rc) { if (gathered) (void)hipFree(gathered); if (il) (void)hipFree(il); if (plain) (void)hipFree(plain); return rc; };
    if (hipMalloc(&gathered, slab_pixels * world * 8) != hipSuccess || hipMalloc(&il, npix * 8) != hipSuccess ||
        hipMalloc(&plain, npix * 8) != hipSuccess)
        return done(vct_fail(c, VCT_ERR_NOMEM, "vct_selftest_interleaved: out of memory"));
    hipError_t e = hipMemsetAsync(gathered, 0, slab_pixels * world * 8, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(plain, 0, npix * 8, c->stream);
    if (e != hipSuccess) return done(vct_fail(c, VCT_ERR_DEVICE, hipGetErrorString(e)));
    for (int r = 0; r < world; ++r) {
        const int rc = vct_launch_trace_rows(c, r < ty ? r : ty, ty, (uint16_t*)(gathered + slab_pixels * r), world, true);
        if (rc) return done(rc);
    }
    hipLaunchKernelGGL(k_deinterleave, dim3(2048), dim3(256), 0, c->stream, gathered, il, w, h, world, slab_pixels);
    int rc = vct_launch_trace_rows(c, 0, ty, (uint16_t*)plain);
    if (rc) return done(rc);
    std::vector<uint2> a(npix), b(npix);
    e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(a.data(), il, npix * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(b.data(), plain, npix * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return done(vct_fail(c, VCT_ERR_DEVICE, hipGetErrorString(e)));
    uint64_t bad = 0;
    for (size_t i = 0; i < npix; ++i) bad += (a[i].x != b[i].x || a[i].y != b[i].y) ? 1u : 0u;
    *mismatches = bad;
    return done(VCT_OK);
}

}  // extern "C"
