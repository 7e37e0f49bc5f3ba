// hpt_multi.hip — multi-GPU behind the C ABI (SURVEY.md §8b "gpus", §8e; include/hpt.h).
//
// The path shards with NO data-path collective: the scene is replicated in every GPU's HBM, shard r renders the 32x32 super-tiles t
// with t % n == r (hpt_render_desc.shard_rank / shard_count — the enumeration the kernel's work queue walks), and the only
// communication is ONE exchange of film data per frame:
//   * default box filter: a GATHER of the tiles each shard owns to shard 0 — each shard packs its tiles (16 KiB apiece) into a
//     contiguous buffer (hpt_pack_tiles_kernel), the root posts one ncclRecv per peer / every peer one ncclSend inside one
//     ncclGroupStart / ncclGroupEnd (point-to-point over xGMI; the film crosses the links exactly once: 33 MB at 1080p), the root scatters
//     the tiles into the frame (hpt_unpack_tiles_kernel);
//   * a reconstruction filter wider than the box (hpt_scene_set_filter): samples reach pixels of neighbouring tiles, every shard's film
//     holds partial sums over the whole frame, and the exchange is one ncclReduce(sum) of the full-frame films to shard 0.
// Two forms share this code:
//   hpt_comm  : one PROCESS per GPU (bench.py under torch.distributed.run, which only carries the 128-byte ncclUniqueId and the timing
//               barrier): ncclCommInitRank;
//   hpt_multi : one process, one host THREAD per GPU (the pbrt plugin, `Renderer "hip" "integer gpus" [N]`): ncclCommInitAll.
// RCCL is reached through dlopen("librccl.so.1") — the copy already in the process (PyTorch ships one) or ROCm's — so that libhpt.so has no
// link-time dependency on a second RCCL.  RCCL refuses two ranks on one device ("Duplicate GPU detected"); hpt_multi_create therefore
// accepts a device list with repeats (several shards on one GPU, e.g. {0, 0}) and moves their tiles with hipMemcpyAsync instead — the
// single-GPU test path of the sharding, packing and threading; RCCL itself then runs with a communicator of one rank.
#include <hip/hip_runtime.h>
#include <cerrno>

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "hpt_internal.h"
#include "hpt_kernels.h"
#include "hpt_rccl_abi.h"

namespace {

// ---- the few RCCL entry points this file needs (hpt_rccl_abi.h: hand-written declarations resolved at run time, checked at build time against
// <rccl/rccl.h> by hpt_rccl_check.cpp) --------------------------------------------------------------------------------------------------------------
typedef hpt_nccl_comm_t ncclComm_t;
typedef hpt_nccl_unique_id ncclUniqueId;
enum { ncclSuccess = HPT_NCCL_SUCCESS };
enum { ncclFloat32 = HPT_NCCL_FLOAT32 };   // ncclDataType_t
enum { ncclSum = HPT_NCCL_SUM };           // ncclRedOp_t
struct Rccl {
    void *lib = nullptr;
    hpt_nccl_get_unique_id_fn GetUniqueId = nullptr;
    hpt_nccl_comm_init_rank_fn CommInitRank = nullptr;
    hpt_nccl_comm_init_all_fn CommInitAll = nullptr;
    hpt_nccl_comm_destroy_fn CommDestroy = nullptr;
    hpt_nccl_comm_count_fn CommCount = nullptr;
    hpt_nccl_group_fn GroupStart = nullptr, GroupEnd = nullptr;
    hpt_nccl_send_fn Send = nullptr;
    hpt_nccl_recv_fn Recv = nullptr;
    hpt_nccl_reduce_fn Reduce = nullptr;
    hpt_nccl_get_error_string_fn GetErrorString = nullptr;
};
Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (r.lib) {
#define SYM(field, name) *(void **)(&r.field) = dlsym(r.lib, name)
            SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommInitAll, "ncclCommInitAll");
            SYM(CommDestroy, "ncclCommDestroy"); SYM(CommCount, "ncclCommCount"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
            SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(Reduce, "ncclReduce"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv || !r.Reduce) r.lib = nullptr;
        }
    }
    return r.lib ? &r : nullptr;
}
#define NCCL_OK(expr) do { int rc_ = (expr); if (rc_ != ncclSuccess) { hpt_set_error("%s failed: %s", #expr, rccl()->GetErrorString ? rccl()->GetErrorString(rc_) : "RCCL error"); return HPT_E_HIP; } } while (0)
#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { hpt_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); return HPT_E_HIP; } } while (0)

// ---- tiles <-> packed buffer -------------------------------------------------------------------------------------------------------
// Shard `rank` of `count` owns the 32x32 super-tiles t = rank, rank + count, ... of the n_stx x n_sty grid over the film; its k-th tile
// sits at float4s [k * HPT_TILE_PX, (k + 1) * HPT_TILE_PX) of the packed buffer.
// Round 4: a tile record is 34 x 34 — the tile AND a one-pixel apron.  A camera sample whose image coordinate is an exact integer lands in
// two pixels (ImageFilm::AddSample under the box filter, film/image.cpp:82-89; 1.2e-4 of the samples); when the second pixel belongs to a
// tile of ANOTHER shard the contribution sits in the rendering shard's film, outside its own tiles, and a gather of 32 x 32 tiles lost it
// (found by the two-process host-transport test: film weights of a 3-shard frame one short in single pixels).  The apron carries it: an
// apron pixel q (in tile B, not this shard's) is included by exactly ONE of this shard's tiles around B — the first in the order left,
// right, up, down, up-left, up-right, down-left, down-right of B whose apron contains q and which this shard owns — and the root ADDS the
// records into its film (its own film holds its own spills there, zeros otherwise).  +13 % bytes, an exact frame.
#define HPT_TILE_W 34
#define HPT_TILE_PX (HPT_TILE_W * HPT_TILE_W)
__device__ __forceinline__ bool apron_carrier_is(int bx, int by, int u, int v, int n_stx, int n_sty, int rank, int count, int want_dx, int want_dy) {
    // neighbours of tile B = (bx, by) whose apron contains B's pixel (u, v), in priority order; true iff the first one owned by `rank` is (want_dx, want_dy)
    const int dxs[8] = {-1, 1, 0, 0, -1, 1, -1, 1}, dys[8] = {0, 0, -1, 1, -1, -1, 1, 1};
    for (int i = 0; i < 8; ++i) {
        const int dx = dxs[i], dy = dys[i];
        if ((dx < 0 && u != 0) || (dx > 0 && u != 31) || (dy < 0 && v != 0) || (dy > 0 && v != 31)) continue;
        const int tx = bx + dx, ty = by + dy;
        if (tx < 0 || ty < 0 || tx >= n_stx || ty >= n_sty) continue;
        if ((ty * n_stx + tx) % count != rank) continue;
        return dx == want_dx && dy == want_dy;
    }
    return false;
}
__global__ void hpt_pack_tiles_kernel(const float4 *film, float4 *packed, int x_count, int y_count, int n_stx, int n_tiles, int rank, int count) {
    const int k = blockIdx.x;                         // local tile
    const int t = k * count + rank;
    if (t >= n_tiles) return;
    const int ax = t % n_stx, ay = t / n_stx, x0 = ax * 32, y0 = ay * 32, n_sty = n_tiles / n_stx;
    for (int p = threadIdx.x; p < HPT_TILE_PX; p += blockDim.x) {
        const int lx = p % HPT_TILE_W - 1, ly = p / HPT_TILE_W - 1;
        const int x = x0 + lx, y = y0 + ly;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x >= 0 && y >= 0 && x < x_count && y < y_count) {
            if (lx >= 0 && lx < 32 && ly >= 0 && ly < 32) val = film[(size_t)y * x_count + x];
            else {
                const int bx = x >> 5, by = y >> 5;
                if ((by * n_stx + bx) % count != rank && apron_carrier_is(bx, by, x & 31, y & 31, n_stx, n_sty, rank, count, ax - bx, ay - by))
                    val = film[(size_t)y * x_count + x];
            }
        }
        packed[(size_t)k * HPT_TILE_PX + p] = val;
    }
}
// the root ADDS a peer's records (atomically: aprons of one launch overlap each other and other tiles' interiors)
__global__ void hpt_unpack_tiles_kernel(float4 *film, const float4 *packed, int x_count, int y_count, int n_stx, int n_tiles, int rank, int count) {
    const int k = blockIdx.x;
    const int t = k * count + rank;
    if (t >= n_tiles) return;
    const int x0 = (t % n_stx) * 32, y0 = (t / n_stx) * 32;
    for (int p = threadIdx.x; p < HPT_TILE_PX; p += blockDim.x) {
        const int x = x0 + p % HPT_TILE_W - 1, y = y0 + p / HPT_TILE_W - 1;
        if (x < 0 || y < 0 || x >= x_count || y >= y_count) continue;
        const float4 v = packed[(size_t)k * HPT_TILE_PX + p];
        if (v.x == 0.f && v.y == 0.f && v.z == 0.f && v.w == 0.f) continue;
        float *f = (float *)&film[(size_t)y * x_count + x];
        unsafeAtomicAdd(f + 0, v.x); unsafeAtomicAdd(f + 1, v.y); unsafeAtomicAdd(f + 2, v.z); unsafeAtomicAdd(f + 3, v.w);
    }
}
inline int local_tiles(int n_tiles, int rank, int count) { return (n_tiles - rank + count - 1) / count; }

} // namespace

// film += other (float4 per pixel): the wide-filter sum of partial films on the peer-copy path, shard after shard in rank order
__global__ __launch_bounds__(256) void hpt_film_add_kernel(float4 *__restrict__ film, const float4 *__restrict__ other, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 a = film[i], b = other[i];
        film[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

// ---- host-staged transport (HPT_COMM_TRANSPORT=host) ------------------------------------------------------------------------------------
// The same exchange — same pack / unpack kernels, same tile bookkeeping, same wide-filter sum — with the hop between the processes done through
// POSIX shared memory instead of ncclSend / ncclRecv / ncclReduce: every non-root rank owns one mailbox segment "/hpt<key>.<rank>" (header +
// payload, created by its owner, grown on demand), copies its packed tiles (or, under a wide filter, its whole film) into it with hipMemcpy
// and publishes the frame's sequence number; the root maps the mailboxes, waits for the sequence number, uploads the payload and
// acknowledges, after which the owner may overwrite the mailbox.  What it is for: RCCL refuses two ranks on one device, so on a one-GPU box
// hpt_comm's multi-rank logic could never run; with this transport two PROCESSES sharing the one device exercise it
// (tests/test_gpu_multi.py).  It also serves hosts without RCCL.  Blocking (the call returns when this rank's part is done), unlike RCCL.
struct HostBox {                       // one mailbox, mapped by its owner and by the root
    struct Hdr { std::atomic<uint64_t> seq, ack, bytes; uint64_t pad[5]; };
    int fd = -1; void *map = nullptr; size_t map_bytes = 0; std::string name; bool owner = false;
    Hdr *hdr() const { return (Hdr *)map; }
    char *payload() const { return (char *)map + sizeof(Hdr); }
    void close_box() {
        if (map) munmap(map, map_bytes);
        if (fd >= 0) close(fd);
        if (owner && !name.empty()) shm_unlink(name.c_str());
        map = nullptr; fd = -1; map_bytes = 0;
    }
    // owner: create (or grow) to hold `payload_bytes`; root: (re)map to the segment's current size
    bool open_box(const std::string &nm, bool own, size_t payload_bytes) {
        name = nm; owner = own;
        const size_t want = sizeof(Hdr) + payload_bytes;
        if (fd < 0) {
            // (the owner creates its mailbox exclusively: a name left over from a crashed run — pid and clock make that unlikely — is removed, not adopted)
            if (own) { fd = shm_open(nm.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600); if (fd < 0 && errno == EEXIST) { shm_unlink(nm.c_str()); fd = shm_open(nm.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600); } }
            else fd = shm_open(nm.c_str(), O_RDWR, 0600);
            if (fd < 0) return false;
        }
        struct stat sb;
        if (fstat(fd, &sb) != 0) return false;
        size_t have = (size_t)sb.st_size;
        if (own && have < want) { if (ftruncate(fd, (off_t)want) != 0) return false; have = want; }   // (growing keeps the header's contents)
        if (have < sizeof(Hdr)) return false;
        if (map && map_bytes == have) return true;
        if (map) munmap(map, map_bytes);
        map = mmap(nullptr, have, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (map == MAP_FAILED) { map = nullptr; return false; }
        map_bytes = have;
        return true;
    }
};
static bool host_wait(const std::atomic<uint64_t> &v, uint64_t want, double timeout_s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0; v.load(std::memory_order_acquire) < want; ++spin) {
        if ((spin & 63) == 63) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
            usleep(50);
        } else sched_yield();
    }
    return true;
}

// ---- one process per GPU -----------------------------------------------------------------------------------------------------------------
struct hpt_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    float4 *packed = nullptr; size_t packed_tiles = 0;      // send buffer (peers) / receive buffer for all peers' tiles (root)
    // host-staged transport
    bool host = false; std::string key; uint64_t frame = 0;
    uint64_t published = 0;                                  // host transport, non-root: the last frame whose bytes were actually published (hpt_comm_destroy waits for ITS acknowledgement)
    // what the last exchange was (hpt_comm_info): its duration on the stream between two events, and the peers rank 0 took records from
    hipEvent_t ev0 = nullptr, ev1 = nullptr; bool timed = false; int peers_last = 0;
    std::vector<HostBox> boxes;                              // non-root: [0] = its own mailbox; root: one per peer (index peer - 1)
    void *d_stage = nullptr; size_t d_stage_bytes = 0;       // root, wide filter: a peer's film on the device while it is added
};
static bool comm_transport_is_host() { const char *e = getenv("HPT_COMM_TRANSPORT"); return e && !strcmp(e, "host"); }
static double comm_timeout_s() { const char *e = getenv("HPT_COMM_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 120.0; }

extern "C" int hpt_comm_unique_id(void *out128) {
    if (!out128) { hpt_set_error("null argument"); return HPT_E_INVALID; }
    if (comm_transport_is_host()) {     // the id names the mailboxes: "HPTHOST" + a key unique to this communicator
        memset(out128, 0, 128);
        snprintf((char *)out128, 128, "HPTHOST%08x%08x", (unsigned)getpid(), (unsigned)std::chrono::steady_clock::now().time_since_epoch().count());
        return HPT_OK;
    }
    Rccl *r = rccl();
    if (!r) { hpt_set_error("RCCL (librccl.so.1) is not available in this process"); return HPT_E_NODEVICE; }
    ncclUniqueId id;
    NCCL_OK(r->GetUniqueId(&id));
    memcpy(out128, &id, sizeof(id));
    return HPT_OK;
}

extern "C" hpt_comm *hpt_comm_create(const void *id128, int rank, int world, int device) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) { hpt_set_error("bad communicator arguments"); return nullptr; }
    if (!memcmp(id128, "HPTHOST", 7)) {                      // an id made by hpt_comm_unique_id under HPT_COMM_TRANSPORT=host (every rank follows the id, not its own environment)
        if (hipSetDevice(device) != hipSuccess) { hpt_set_error("hipSetDevice(%d) failed", device); return nullptr; }
        hpt_comm *c = new hpt_comm();
        c->rank = rank; c->world = world; c->device = device; c->host = true;
        char k[32]; memset(k, 0, sizeof(k)); memcpy(k, (const char *)id128 + 7, 16);
        c->key = std::string("/hpt") + k;
        if (rank != 0) {                                     // the owner creates its mailbox now, so that the root finds it at the first exchange
            c->boxes.resize(1);
            if (!c->boxes[0].open_box(c->key + "." + std::to_string(rank), true, 4096)) { hpt_set_error("hpt_comm (host transport): shm_open(%s.%d) failed", c->key.c_str(), rank); delete c; return nullptr; }
        } else c->boxes.resize((size_t)(world > 1 ? world - 1 : 0));
        return c;
    }
    Rccl *r = rccl();
    if (!r) { hpt_set_error("RCCL (librccl.so.1) is not available in this process"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { hpt_set_error("hipSetDevice(%d) failed", device); return nullptr; }
    hpt_comm *c = new hpt_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId id; memcpy(&id, id128, sizeof(id));
    int rc = r->CommInitRank(&c->comm, world, id, rank);
    if (rc != ncclSuccess) { hpt_set_error("ncclCommInitRank failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "?"); delete c; return nullptr; }
    return c;
}

extern "C" void hpt_comm_destroy(hpt_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->packed) (void)hipFree(c->packed);
    if (c->d_stage) (void)hipFree(c->d_stage);
    // host transport, non-root: the mailbox is unlinked only when rank 0 has taken the last frame published in it — a rank that exchanged once and
    // left at once used to remove the name before rank 0 had opened it, and rank 0 then span until "no mailbox" 
    // (the wait is for the last frame PUBLISHED — a host_send that failed before publishing used to cost a second full timeout here)
    if (c->host && c->rank != 0 && c->published > 0 && !c->boxes.empty() && c->boxes[0].map)
        (void)host_wait(c->boxes[0].hdr()->ack, c->published, comm_timeout_s());
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (HostBox &b : c->boxes) b.close_box();
    if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
    delete c;
}

// The host-staged hop of one exchange.  Non-root: `src` (device, `bytes`) -> own mailbox, publish.  Root: for each peer in rank order wait,
// upload into dst(peer) and hand the device pointer to `consume` (unpack / add), acknowledge.
static int host_send(hpt_comm *c, const void *d_src, size_t bytes, hipStream_t stream) {
    HostBox &b = c->boxes[0];
    const uint64_t f = ++c->frame;
    if (f > 1 && !host_wait(b.hdr()->ack, f - 1, comm_timeout_s())) { hpt_set_error("hpt_comm (host transport): rank 0 did not take frame %llu of rank %d", (unsigned long long)(f - 1), c->rank); return HPT_E_HIP; }
    if (!b.open_box(b.name, true, bytes)) { hpt_set_error("hpt_comm (host transport): cannot grow the mailbox to %zu bytes", bytes); return HPT_E_HIP; }
    HIP_OK(hipStreamSynchronize(stream));
    if (bytes) HIP_OK(hipMemcpy(b.payload(), d_src, bytes, hipMemcpyDeviceToHost));
    b.hdr()->bytes.store(bytes, std::memory_order_relaxed);
    b.hdr()->seq.store(f, std::memory_order_release);
    c->published = f;
    return HPT_OK;
}
template <class F> static int host_recv_all(hpt_comm *c, F &&per_peer) {
    const uint64_t f = ++c->frame;
    for (int p = 1; p < c->world; ++p) {
        HostBox &b = c->boxes[(size_t)p - 1];
        const std::string nm = c->key + "." + std::to_string(p);
        const auto t0 = std::chrono::steady_clock::now();
        while (!b.open_box(nm, false, 0)) {                  // (the peer may not have created it yet)
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > comm_timeout_s()) { hpt_set_error("hpt_comm (host transport): no mailbox %s", nm.c_str()); return HPT_E_HIP; }
            usleep(200);
        }
        if (!host_wait(b.hdr()->seq, f, comm_timeout_s())) { hpt_set_error("hpt_comm (host transport): rank %d did not deliver frame %llu", p, (unsigned long long)f); return HPT_E_HIP; }
        const size_t bytes = (size_t)b.hdr()->bytes.load(std::memory_order_relaxed);
        if (!b.open_box(nm, false, 0) || b.map_bytes < sizeof(HostBox::Hdr) + bytes) { hpt_set_error("hpt_comm (host transport): mailbox %s is smaller than its payload", nm.c_str()); return HPT_E_HIP; }
        const int rc = per_peer(p, b.payload(), bytes);
        b.hdr()->ack.store(f, std::memory_order_release);
        if (rc != HPT_OK) {                                   // the frame is lost, but no peer is left waiting for its acknowledgement until its own timeout
            for (int q = p + 1; q < c->world; ++q) {
                HostBox &bq = c->boxes[(size_t)q - 1];
                if (bq.open_box(c->key + "." + std::to_string(q), false, 0) && host_wait(bq.hdr()->seq, f, comm_timeout_s())) bq.hdr()->ack.store(f, std::memory_order_release);
            }
            return rc;
        }
    }
    return HPT_OK;
}

// The end-of-frame film exchange on `stream` (the stream the shard was rendered on; asynchronous, like an RCCL call).
// d_film: this rank's x_count * y_count * 4 floats; on rank 0 it holds the whole frame afterwards.
static int comm_exchange_film(hpt_comm *c, const hpt_render_desc *rd, void *d_film, void *stream_v, int wide_filter);
extern "C" int hpt_comm_exchange_film(hpt_comm *c, const hpt_render_desc *rd, void *d_film, void *stream_v, int wide_filter) {
    if (!c || !rd || !d_film) { hpt_set_error("null argument"); return HPT_E_INVALID; }
    if (c->world == 1) return HPT_OK;
    // the exchange between two events on its stream: hpt_comm_info reports the last one's duration (pack + transfer + unpack as the device saw them)
    HIP_OK(hipSetDevice(c->device));
    if (!c->ev0) { HIP_OK(hipEventCreate(&c->ev0)); HIP_OK(hipEventCreate(&c->ev1)); }
    c->timed = false; c->peers_last = 0;
    HIP_OK(hipEventRecord(c->ev0, (hipStream_t)stream_v));
    const int rc = comm_exchange_film(c, rd, d_film, stream_v, wide_filter);
    if (rc != HPT_OK) return rc;
    HIP_OK(hipEventRecord(c->ev1, (hipStream_t)stream_v));
    c->timed = true;
    return HPT_OK;
}
// What the communicator is and what its last exchange did (bench.py's `rccl_ranks` / `exchange_ms`: a driver must be able to tell an 8-rank RCCL gather
// from anything else).  *ranks: ncclCommCount of the RCCL communicator (host transport: the world size the id was created for); *transport: 0 RCCL, 1 the
// host-staged one-device dry run; *peers: ranks whose records rank 0 received in the last exchange (world - 1 when every shard owns a tile; 0 on other
// ranks); *exchange_ms: duration of the last exchange on its stream (waits for it; < 0: none yet).  Any pointer may be NULL.
extern "C" int hpt_comm_info(hpt_comm *c, int *ranks, int *transport, int *peers, float *exchange_ms) {
    if (!c) { hpt_set_error("null argument"); return HPT_E_INVALID; }
    if (ranks) {
        *ranks = c->world;
        if (!c->host && c->comm) { Rccl *r = rccl(); int n = 0; if (!r || !r->CommCount) { hpt_set_error("ncclCommCount is not available"); return HPT_E_NODEVICE; } NCCL_OK(r->CommCount(c->comm, &n)); *ranks = n; }
    }
    if (transport) *transport = c->host ? 1 : 0;
    if (peers) *peers = c->peers_last;
    if (exchange_ms) {
        *exchange_ms = -1.f;
        if (c->timed) { HIP_OK(hipSetDevice(c->device)); HIP_OK(hipEventSynchronize(c->ev1)); HIP_OK(hipEventElapsedTime(exchange_ms, c->ev0, c->ev1)); }
    }
    return HPT_OK;
}
static int comm_exchange_film(hpt_comm *c, const hpt_render_desc *rd, void *d_film, void *stream_v, int wide_filter) {
    Rccl *r = c->host ? nullptr : rccl();
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_OK(hipSetDevice(c->device));
    const size_t n_floats = (size_t)rd->x_count * rd->y_count * 4;
    // Sampler "halton" on a pixel extent that does not start on the global 32x32 grid: a shard's windows are not its film tiles — sum as well
    if (HPT_SAMPLER_KIND(rd->sampler_mode) == HPT_SAMPLER_HALTON_HASH && ((rd->x_start | rd->y_start) & 31)) wide_filter = 1;
    if (HPT_SAMPLER_KIND(rd->sampler_mode) == HPT_SAMPLER_BESTCANDIDATE_HASH) wide_filter = 1;   // Sampler "bestcandidate": the shards are table tiles, not film tiles
    if (wide_filter && c->host) {                        // host-staged sum: the peers' films added on the root's device, in rank order
        if (c->rank != 0) return host_send(c, d_film, n_floats * sizeof(float), stream);
        if (c->d_stage_bytes < n_floats * sizeof(float)) {
            if (c->d_stage) (void)hipFree(c->d_stage);
            c->d_stage = nullptr; c->d_stage_bytes = 0;
            HIP_OK(hipMalloc(&c->d_stage, n_floats * sizeof(float)));
            c->d_stage_bytes = n_floats * sizeof(float);
        }
        return host_recv_all(c, [&](int, const char *payload, size_t bytes) -> int {
            ++c->peers_last;
            if (bytes != n_floats * sizeof(float)) { hpt_set_error("hpt_comm (host transport): a peer's film has %zu bytes, expected %zu", bytes, n_floats * sizeof(float)); return HPT_E_INVALID; }
            HIP_OK(hipMemcpyAsync(c->d_stage, payload, bytes, hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(hpt_film_add_kernel, dim3(1024), dim3(256), 0, stream, (float4 *)d_film, (const float4 *)c->d_stage, n_floats / 4);
            HIP_OK(hipStreamSynchronize(stream));            // the mailbox is released (acknowledged) when its bytes are on the device
            return HPT_OK;
        });
    }
    if (wide_filter) {                                   // partial sums over the whole frame: one sum-reduce to rank 0
        NCCL_OK(r->Reduce(d_film, d_film, n_floats, ncclFloat32, ncclSum, 0, c->comm, stream));
        if (c->rank == 0) c->peers_last = c->world - 1;
        return HPT_OK;
    }
    const int n_stx = (rd->x_count + 31) / 32, n_sty = (rd->y_count + 31) / 32, n_tiles = n_stx * n_sty;
    const size_t need = c->rank == 0 ? (size_t)n_tiles : (size_t)local_tiles(n_tiles, c->rank, c->world);
    if (c->packed_tiles < need) {
        if (c->packed) (void)hipFree(c->packed);
        c->packed = nullptr; c->packed_tiles = 0;
        HIP_OK(hipMalloc((void **)&c->packed, need * HPT_TILE_PX * sizeof(float4)));
        c->packed_tiles = need;
    }
    if (c->rank != 0) {
        const int mine = local_tiles(n_tiles, c->rank, c->world);
        if (mine > 0) hipLaunchKernelGGL(hpt_pack_tiles_kernel, dim3(mine), dim3(256), 0, stream, (const float4 *)d_film, c->packed, rd->x_count, rd->y_count, n_stx, n_tiles, c->rank, c->world);
        if (c->host) return host_send(c, c->packed, (size_t)mine * HPT_TILE_PX * sizeof(float4), stream);
        NCCL_OK(r->GroupStart());
        if (mine > 0) NCCL_OK(r->Send(c->packed, (size_t)mine * HPT_TILE_PX * 4, ncclFloat32, 0, c->comm, stream));
        NCCL_OK(r->GroupEnd());
        return HPT_OK;
    }
    size_t off = 0;
    if (c->host) {
        HIP_OK(hipStreamSynchronize(stream));               // (the receive buffer may still be read by the previous frame's unpack kernels)
        const int rc = host_recv_all(c, [&](int p, const char *payload, size_t bytes) -> int {
            const int n = local_tiles(n_tiles, p, c->world);
            if (bytes != (size_t)n * HPT_TILE_PX * sizeof(float4)) { hpt_set_error("hpt_comm (host transport): rank %d sent %zu bytes for %d tiles", p, bytes, n); return HPT_E_INVALID; }
            if (n > 0) HIP_OK(hipMemcpy(c->packed + off * HPT_TILE_PX, payload, bytes, hipMemcpyHostToDevice));
            if (n > 0) ++c->peers_last;
            off += (size_t)n;
            return HPT_OK;
        });
        if (rc != HPT_OK) return rc;
    } else {
        NCCL_OK(r->GroupStart());
        for (int p = 1; p < c->world; ++p) {
            const int n = local_tiles(n_tiles, p, c->world);
            if (n > 0) { NCCL_OK(r->Recv(c->packed + off * HPT_TILE_PX, (size_t)n * HPT_TILE_PX * 4, ncclFloat32, p, c->comm, stream)); ++c->peers_last; }
            off += (size_t)n;
        }
        NCCL_OK(r->GroupEnd());
    }
    off = 0;
    for (int p = 1; p < c->world; ++p) {
        const int n = local_tiles(n_tiles, p, c->world);
        if (n > 0) hipLaunchKernelGGL(hpt_unpack_tiles_kernel, dim3(n), dim3(256), 0, stream, (float4 *)d_film, c->packed + off * HPT_TILE_PX, rd->x_count, rd->y_count, n_stx, n_tiles, p, c->world);
        off += (size_t)n;
    }
    HIP_OK(hipGetLastError());
    return HPT_OK;
}


// ---- one process, one host thread per GPU -----------------------------------------------------------------------------------------------------
struct hpt_multi {
    int n = 0;
    std::vector<int> devices;
    std::vector<hpt_scene *> scenes;
    std::vector<void *> films;          // per shard: full-frame device film
    std::vector<hipStream_t> streams;
    std::vector<float4 *> packed;       // per shard: its packed tiles ON THE ROOT'S DEVICE side of the transfer (peer path) / send buffer (RCCL path)
    std::vector<float4 *> recv;         // RCCL path: root's receive buffers
    std::vector<ncclComm_t> comms;      // RCCL path (distinct devices); empty: peer-copy path
    size_t film_bytes = 0, tiles_cap = 0;
    void *stage = nullptr; size_t stage_bytes = 0;   // peer-copy path under a wide filter: one shard's film on the root's device while it is added
    bool wide = false;
    std::vector<int> chunks_taken;      // last frame: sub-shards each device rendered (dynamic hand-out, HPT_MULTI_CHUNKS; 1 each otherwise)
};

extern "C" void hpt_multi_destroy(hpt_multi *m) {
    if (!m) return;
    if (m->stage && m->n > 0) { (void)hipSetDevice(m->devices[0]); (void)hipFree(m->stage); }
    for (int i = 0; i < m->n; ++i) {
        (void)hipSetDevice(m->devices[(size_t)i]);
        if (i < (int)m->films.size() && m->films[(size_t)i]) (void)hipFree(m->films[(size_t)i]);
        if (i < (int)m->packed.size() && m->packed[(size_t)i]) (void)hipFree(m->packed[(size_t)i]);
        if (i < (int)m->recv.size() && m->recv[(size_t)i]) (void)hipFree(m->recv[(size_t)i]);
        if (i < (int)m->streams.size() && m->streams[(size_t)i]) (void)hipStreamDestroy(m->streams[(size_t)i]);
        if (i < (int)m->comms.size() && m->comms[(size_t)i] && rccl()) (void)rccl()->CommDestroy(m->comms[(size_t)i]);
        if (i < (int)m->scenes.size() && m->scenes[(size_t)i]) hpt_scene_destroy(m->scenes[(size_t)i]);
    }
    delete m;
}

extern "C" hpt_multi *hpt_multi_create(const hpt_scene_desc *desc, const int *devices, int n_devices) {
    if (!desc || !devices || n_devices < 1 || n_devices > 64) { hpt_set_error("bad device list"); return nullptr; }
    hpt_multi *m = new hpt_multi();
    m->n = n_devices;
    m->devices.assign(devices, devices + n_devices);
    m->scenes.assign((size_t)n_devices, nullptr); m->films.assign((size_t)n_devices, nullptr); m->streams.assign((size_t)n_devices, nullptr);
    m->packed.assign((size_t)n_devices, nullptr); m->recv.assign((size_t)n_devices, nullptr);
    // the scene is replicated: one hpt_scene per shard, built concurrently (BVH build on the host is per scene; uploads overlap)
    std::vector<std::thread> th;
    std::vector<std::string> errs((size_t)n_devices);
    for (int i = 0; i < n_devices; ++i)
        th.emplace_back([&, i]() {
            m->scenes[(size_t)i] = hpt_scene_create(desc, devices[i]);
            if (!m->scenes[(size_t)i]) errs[(size_t)i] = hpt_last_error();                 // (the error text is thread-local)
            else if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreate(&m->streams[(size_t)i]) != hipSuccess) errs[(size_t)i] = "hipStreamCreate failed";
        });
    for (auto &t : th) t.join();
    for (int i = 0; i < n_devices; ++i)
        if (!errs[(size_t)i].empty()) { hpt_set_error("shard %d (device %d): %s", i, devices[i], errs[(size_t)i].c_str()); hpt_multi_destroy(m); return nullptr; }
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i) for (int j = 0; j < i; ++j) if (devices[i] == devices[j]) distinct = false;
    const char *force = getenv("HPT_GATHER");                // "peer": hipMemcpyAsync between devices instead of RCCL
    if (n_devices > 1 && distinct && !(force && !strcmp(force, "peer"))) {
        Rccl *r = rccl();
        if (!r) { hpt_set_error("RCCL (librccl.so.1) is not available (HPT_GATHER=peer gathers with hipMemcpyPeerAsync)"); hpt_multi_destroy(m); return nullptr; }
        m->comms.assign((size_t)n_devices, nullptr);
        int rc = r->CommInitAll(m->comms.data(), n_devices, devices);
        if (rc != ncclSuccess) { hpt_set_error("ncclCommInitAll failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "?"); m->comms.clear(); hpt_multi_destroy(m); return nullptr; }
    } else if (n_devices > 1) {
        for (int i = 1; i < n_devices; ++i) {                // peer path: let the root's device read / write the others' memory
            if (devices[i] == devices[0]) continue;
            (void)hipSetDevice(devices[0]); (void)hipDeviceEnablePeerAccess(devices[i], 0);
            (void)hipSetDevice(devices[i]); (void)hipDeviceEnablePeerAccess(devices[0], 0);
        }
        (void)hipGetLastError();
    }
    return m;
}

extern "C" int hpt_multi_set_filter(hpt_multi *m, const hpt_filter *f) {
    if (!m) { hpt_set_error("null handle"); return HPT_E_INVALID; }
    for (int i = 0; i < m->n; ++i) { int rc = hpt_scene_set_filter(m->scenes[(size_t)i], f); if (rc != HPT_OK) return rc; }
    m->wide = f != nullptr;
    return HPT_OK;
}

extern "C" int hpt_multi_set_camera_motion(hpt_multi *m, const hpt_instance *c2w) {
    if (!m) { hpt_set_error("null handle"); return HPT_E_INVALID; }
    for (int i = 0; i < m->n; ++i) { int rc = hpt_scene_set_camera_motion(m->scenes[(size_t)i], c2w); if (rc != HPT_OK) return rc; }
    return HPT_OK;
}

extern "C" int hpt_multi_set_sample_table(hpt_multi *m, const float *table, int n_entries) {
    if (!m) { hpt_set_error("null handle"); return HPT_E_INVALID; }
    for (int i = 0; i < m->n; ++i) { int rc = hpt_scene_set_sample_table(m->scenes[(size_t)i], table, n_entries); if (rc != HPT_OK) return rc; }
    return HPT_OK;
}

extern "C" int hpt_multi_chunks_taken(hpt_multi *m, int *out_n_devices) {
    if (!m || !out_n_devices) { hpt_set_error("null argument"); return HPT_E_INVALID; }
    for (int i = 0; i < m->n; ++i) out_n_devices[i] = i < (int)m->chunks_taken.size() ? m->chunks_taken[(size_t)i] : 0;
    return HPT_OK;
}

extern "C" int hpt_multi_scene(hpt_multi *m, int shard, hpt_scene **out) {
    if (!m || !out || shard < 0 || shard >= m->n) { hpt_set_error("bad shard"); return HPT_E_INVALID; }
    *out = m->scenes[(size_t)shard];
    return HPT_OK;
}

// One frame: every shard renders its tiles on its own device (one host thread each), then the film exchange; the frame lands in
// film_xyzw_host (rank 0's film).  stats: n_devices records (may be null).  rd->shard_rank / shard_count are ignored.
extern "C" int hpt_multi_render(hpt_multi *m, const hpt_camera *cam, const hpt_render_desc *rd, float *film_host, hpt_stats *stats) {
    if (!m || !cam || !rd || !film_host) { hpt_set_error("null argument"); return HPT_E_INVALID; }
    const int n = m->n;
    const size_t bytes = sizeof(float) * 4 * (size_t)rd->x_count * rd->y_count;
    const int n_stx = (rd->x_count + 31) / 32, n_sty = (rd->y_count + 31) / 32, n_tiles = n_stx * n_sty;
    const bool use_rccl = !m->comms.empty();
    // (Sampler "halton" on a pixel extent off the global 32x32 grid: a shard's windows are not its film tiles — partial films are summed)
    // HPT_MULTI_CHUNKS=<k> (2 .. 16): dynamic hand-out of n x k sub-shards (below); box filter only (the two-pass film of a table filter writes whole pixels)
    int chunks = 1;
    if (const char *e = getenv("HPT_MULTI_CHUNKS")) { chunks = atoi(e); if (chunks < 1) chunks = 1; if (chunks > 16) chunks = 16; }
    const bool dynamic = chunks > 1 && n > 1 && !m->wide;
    m->chunks_taken.assign((size_t)n, dynamic ? 0 : 1);
    const bool wide = m->wide || dynamic || (HPT_SAMPLER_KIND(rd->sampler_mode) == HPT_SAMPLER_HALTON_HASH && ((rd->x_start | rd->y_start) & 31))
                      || HPT_SAMPLER_KIND(rd->sampler_mode) == HPT_SAMPLER_BESTCANDIDATE_HASH;       // (its shards are table tiles, not film tiles)
    // buffers (grown on demand, kept with the handle)
    if (m->film_bytes < bytes || m->tiles_cap < (size_t)n_tiles) {
        for (int i = 0; i < n; ++i) {
            HIP_OK(hipSetDevice(m->devices[(size_t)i]));
            if (m->films[(size_t)i]) (void)hipFree(m->films[(size_t)i]);
            if (m->packed[(size_t)i]) (void)hipFree(m->packed[(size_t)i]);
            if (m->recv[(size_t)i]) (void)hipFree(m->recv[(size_t)i]);
            m->films[(size_t)i] = nullptr; m->packed[(size_t)i] = nullptr; m->recv[(size_t)i] = nullptr;
            HIP_OK(hipMalloc(&m->films[(size_t)i], bytes));
            const size_t mine = (size_t)local_tiles(n_tiles, i, n);
            if (i > 0 && mine > 0) HIP_OK(hipMalloc((void **)&m->packed[(size_t)i], mine * HPT_TILE_PX * sizeof(float4)));
            if (i > 0 && mine > 0) { HIP_OK(hipSetDevice(m->devices[0])); HIP_OK(hipMalloc((void **)&m->recv[(size_t)i], mine * HPT_TILE_PX * sizeof(float4))); }
        }
        m->film_bytes = bytes; m->tiles_cap = (size_t)n_tiles;
    }
    std::vector<int> rcs((size_t)n, HPT_OK);
    std::vector<std::string> errs((size_t)n);
    std::vector<hpt_stats> st((size_t)n);
    std::vector<std::thread> th;
    std::atomic<int> next_chunk(0);
    for (int i = 0; i < n; ++i)
        th.emplace_back([&, i]() {
            hpt_render_desc r = *rd;
            if (dynamic) {
                // Dynamic hand-out (SURVEY.md §8e "optional dynamic balancing: host-side atomic tile counter per node"): the frame is cut into
                // n x chunks round-robin sub-shards and every device's thread pulls the next one when its kernel has drained, so a device that
                // got cheap tiles — or is faster — takes more of them and the frame ends with the slowest SUB-shard's tail, not the slowest shard's.
                // A device's sub-shards are disjoint tile sets of one film (cleared once, then added to); the exchange is the sum.
                int rc = hipSetDevice(m->devices[(size_t)i]) == hipSuccess && hipMemsetAsync(m->films[(size_t)i], 0, bytes, m->streams[(size_t)i]) == hipSuccess ? HPT_OK : HPT_E_HIP;
                hpt_stats acc; memset(&acc, 0, sizeof(acc));
                int taken = 0;
                for (;;) {
                    const int c = next_chunk.fetch_add(1);
                    if (c >= n * chunks || rc != HPT_OK) break;
                    r.shard_rank = c; r.shard_count = n * chunks;
                    hpt_stats one; memset(&one, 0, sizeof(one));
                    rc = hpt_render_device_into(m->scenes[(size_t)i], cam, &r, m->films[(size_t)i], m->streams[(size_t)i], &one, false);
                    const double ms = acc.kernel_ms + one.kernel_ms;
                    const uint64_t cs = acc.camera_samples + one.camera_samples, bs = acc.bad_samples + one.bad_samples, cr = acc.closest_rays + one.closest_rays,
                                   sr = acc.shadow_rays + one.shadow_rays, nv = acc.nodes_visited + one.nodes_visited, tt = acc.tris_tested + one.tris_tested;
                    acc = one; acc.kernel_ms = ms; acc.camera_samples = cs; acc.bad_samples = bs; acc.closest_rays = cr; acc.shadow_rays = sr; acc.nodes_visited = nv; acc.tris_tested = tt;
                    ++taken;
                }
                if (rc != HPT_OK) errs[(size_t)i] = hpt_last_error();
                st[(size_t)i] = acc; rcs[(size_t)i] = rc; m->chunks_taken[(size_t)i] = taken;
                return;
            }
            r.shard_rank = i; r.shard_count = n;
            int rc = hpt_render_device(m->scenes[(size_t)i], cam, &r, m->films[(size_t)i], m->streams[(size_t)i], &st[(size_t)i]);
            const int mine = local_tiles(n_tiles, i, n);   // (0 for a frame smaller than an earlier one of this handle: nothing to pack)
            if (rc == HPT_OK && i > 0 && !wide && m->packed[(size_t)i] && mine > 0) {
                hipLaunchKernelGGL(hpt_pack_tiles_kernel, dim3(mine), dim3(256), 0, m->streams[(size_t)i], (const float4 *)m->films[(size_t)i], m->packed[(size_t)i],
                                   rd->x_count, rd->y_count, n_stx, n_tiles, i, n);
                if (hipStreamSynchronize(m->streams[(size_t)i]) != hipSuccess) rc = HPT_E_HIP;
            }
            rcs[(size_t)i] = rc;
            if (rc != HPT_OK) errs[(size_t)i] = hpt_last_error();
        });
    for (auto &t : th) t.join();
    for (int i = 0; i < n; ++i) if (rcs[(size_t)i] != HPT_OK) { hpt_set_error("shard %d: %s", i, errs[(size_t)i].c_str()); return rcs[(size_t)i]; }
    if (stats) for (int i = 0; i < n; ++i) stats[i] = st[(size_t)i];
    // ---- film exchange to shard 0 ---------------------------------------------------------------------------------------------------
    if (n > 1) {
        if (wide) {                    // sum of partial films
            if (use_rccl) {
                Rccl *r = rccl();
                NCCL_OK(r->GroupStart());
                for (int i = 0; i < n; ++i) NCCL_OK(r->Reduce(m->films[(size_t)i], m->films[(size_t)i], bytes / 4, ncclFloat32, ncclSum, 0, m->comms[(size_t)i], m->streams[(size_t)i]));
                NCCL_OK(r->GroupEnd());
            } else {
                // peer-copy path (shards that share a device, or HPT_GATHER=peer): the root adds the shards' films one after the other in rank
                // order — a fixed order, so the film is bit-reproducible; a film of another device is staged on the root's device first
                HIP_OK(hipSetDevice(m->devices[0]));
                const size_t npx = bytes / sizeof(float4);
                for (int i = 1; i < n; ++i) {
                    const void *src = m->films[(size_t)i];
                    if (m->devices[(size_t)i] != m->devices[0]) {
                        if (m->stage_bytes < bytes) {
                            if (m->stage) (void)hipFree(m->stage);
                            m->stage = nullptr; m->stage_bytes = 0;
                            HIP_OK(hipMalloc(&m->stage, bytes));
                            m->stage_bytes = bytes;
                        }
                        HIP_OK(hipMemcpyPeerAsync(m->stage, m->devices[0], src, m->devices[(size_t)i], bytes, m->streams[0]));
                        src = m->stage;
                    }
                    hipLaunchKernelGGL(hpt_film_add_kernel, dim3(2048), dim3(256), 0, m->streams[0], (float4 *)m->films[0], (const float4 *)src, npx);
                }
                HIP_OK(hipGetLastError());
            }
        } else {
            if (use_rccl) {
                Rccl *r = rccl();
                NCCL_OK(r->GroupStart());
                for (int i = 1; i < n; ++i) {
                    const int mine = local_tiles(n_tiles, i, n);
                    if (mine <= 0) continue;
                    NCCL_OK(r->Send(m->packed[(size_t)i], (size_t)mine * HPT_TILE_PX * 4, ncclFloat32, 0, m->comms[(size_t)i], m->streams[(size_t)i]));
                    NCCL_OK(r->Recv(m->recv[(size_t)i], (size_t)mine * HPT_TILE_PX * 4, ncclFloat32, i, m->comms[0], m->streams[0]));
                }
                NCCL_OK(r->GroupEnd());
            } else {
                HIP_OK(hipSetDevice(m->devices[0]));
                for (int i = 1; i < n; ++i) {
                    const int mine = local_tiles(n_tiles, i, n);
                    if (mine <= 0) continue;
                    if (m->devices[(size_t)i] == m->devices[0]) HIP_OK(hipMemcpyAsync(m->recv[(size_t)i], m->packed[(size_t)i], (size_t)mine * HPT_TILE_PX * sizeof(float4), hipMemcpyDeviceToDevice, m->streams[0]));
                    else HIP_OK(hipMemcpyPeerAsync(m->recv[(size_t)i], m->devices[0], m->packed[(size_t)i], m->devices[(size_t)i], (size_t)mine * HPT_TILE_PX * sizeof(float4), m->streams[0]));
                }
            }
            HIP_OK(hipSetDevice(m->devices[0]));
            for (int i = 1; i < n; ++i) {
                const int mine = local_tiles(n_tiles, i, n);
                if (mine > 0) hipLaunchKernelGGL(hpt_unpack_tiles_kernel, dim3(mine), dim3(256), 0, m->streams[0], (float4 *)m->films[0], m->recv[(size_t)i], rd->x_count, rd->y_count, n_stx, n_tiles, i, n);
            }
        }
        for (int i = 0; i < n; ++i) { HIP_OK(hipSetDevice(m->devices[(size_t)i])); HIP_OK(hipStreamSynchronize(m->streams[(size_t)i])); }
    }
    HIP_OK(hipSetDevice(m->devices[0]));
    HIP_OK(hipMemcpy(film_host, m->films[0], bytes, hipMemcpyDeviceToHost));
    return HPT_OK;
}
