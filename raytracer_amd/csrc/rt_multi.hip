// rt_multi.hip -- one frame over several MI355X: the frame is cut into squares that are shared out over the GPUs, each GPU
// renders its squares from its own copy of the scene into a compact buffer, and the buffers are gathered on one GPU
// with RCCL point-to-point transfers over xGMI and written into FrameBuffer order there.
//
// The reference's counterpart is a set of separate processes that render disjoint tiles (--xstart/--xend/--ystart/--yend,
// exe/Raytracer.cpp:65-68), write TILE files (io/TileFile.cpp:8-27) and are merged by --mergetiles (exe/Raytracer.cpp:260-277,
// io/TileFile.cpp:29-70).
//
//   squares : T x T pixels (32 unless RTGPU_TILE_SIZE says otherwise) over the settings' tile, numbered row by row; square k
//             belongs to rank k % G.  Interleaving balances an unevenly expensive image; a pixel's random numbers are keyed by
//             its frame position, so the assembled frame is bit-identical for every G.
//   buffer  : rank r holds float3[squares of r][T * T] (squares that stick out of the tile keep their full size).
//   gather  : ranks 1..G-1 ncclSend their buffer, rank 0 posts one ncclRecv per peer (one group: the transfers run
//             concurrently, each over the direct xGMI link between its two GPUs), then a kernel on rank 0 copies every pixel
//             of the tile from its square's slot into the frame buffer.  3840 x 2160: 99.5 MB in total, 12.4 MB per peer at G = 8.
//
// Two ways in: one process per GPU (rtgpu_comm_* + rtgpu_render_tiles_device + rtgpu_gather_tiles_device; the host
// program moves the 128-byte communicator id between its processes), or one process for all GPUs (rtgpu_render with
// settings->n_gpus > 1: the library copies the scene to the other devices, keeps one host thread per device while rendering
// and issues the gather for all devices from one thread).
//
// RCCL is loaded when the first communicator is made, not at link time: a single-GPU user never maps it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rt_scene.h"

namespace rt {
namespace {

struct Rccl {
    void* lib = nullptr;
    std::string error;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return lib != nullptr && error.empty(); }
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy the process has already mapped (a host program that brought its own) comes first
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!r.lib)
            for (const char* n : names)
                if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!r.lib) { r.error = std::string("cannot load librccl.so: ") + dlerror(); return; }
        auto sym = [&](const char* name) -> void* {
            void* p = dlsym(r.lib, name);
            if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + name;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}

int hip_fail(hipError_t e, const char* what)
{
    return set_error(e == hipErrorOutOfMemory ? RTGPU_ERR_OUT_OF_MEMORY : RTGPU_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define MG_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(e_, #call); } while (0)
#define MG_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return set_error(RTGPU_ERR_COMM, std::string(#call) + ": " + rccl().GetErrorString(r_)); } while (0)

struct TileGrid {
    uint32_t T, tilesX, nTiles;
};
TileGrid tile_grid(const rtgpu_render_settings& st, uint32_t T)
{
    const uint32_t tw = (uint32_t)(st.x_end - st.x_start), th = (uint32_t)(st.y_end - st.y_start);
    TileGrid g;
    g.T = T;
    g.tilesX = (tw + T - 1) / T;
    g.nTiles = g.tilesX * ((th + T - 1) / T);
    return g;
}
uint64_t rank_floats(const TileGrid& g, uint32_t rank, uint32_t world)
{
    const uint64_t mine = g.nTiles > rank ? (g.nTiles - rank + world - 1) / world : 0;
    return mine * g.T * g.T * 3;
}

// One thread per pixel of the settings' tile: the pixel's square k, its owner k % world and its slot in the owner's buffer.
__global__ __launch_bounds__(256) void scatter_tiles_kernel(rtgpu_render_settings st, uint32_t T, uint32_t tilesX, uint32_t world,
                                                            const float* __restrict__ rank0, const float* __restrict__ others, uint64_t stride,
                                                            float* __restrict__ frame)
{
    const uint32_t tw = (uint32_t)(st.x_end - st.x_start), th = (uint32_t)(st.y_end - st.y_start);
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint64_t)tw * th) return;
    const uint32_t lx = (uint32_t)(i % tw), ly = (uint32_t)(i / tw);
    const uint32_t k = (ly / T) * tilesX + lx / T;
    const uint32_t r = k % world, j = k / world;
    const uint32_t w = (ly % T) * T + lx % T;
    const float* src = (r == 0 ? rank0 : others + (uint64_t)(r - 1) * stride) + ((uint64_t)j * T * T + w) * 3;
    float* dst = frame + ((uint64_t)(st.y_start + (int)ly) * (uint32_t)st.width + (uint32_t)(st.x_start + (int)lx)) * 3;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

int check_tiles(const rtgpu_render_settings* st, int tile_size, int rank, int world)
{
    if (!st) return set_error(RTGPU_ERR_INVALID_ARGUMENT, "settings are null");
    if (st->width <= 0 || st->height <= 0 || st->x_start < 0 || st->y_start < 0 || st->x_end > st->width || st->y_end > st->height ||
        st->x_start > st->x_end || st->y_start > st->y_end)
        return set_error(RTGPU_ERR_INVALID_ARGUMENT, "Invalid tile bounds");   // film/Tile.cpp:9-12
    if (tile_size < 1 || tile_size > 1024) return set_error(RTGPU_ERR_INVALID_ARGUMENT, "tile size must be 1..1024");
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return set_error(RTGPU_ERR_INVALID_ARGUMENT, "rank / number of ranks out of range");
    return RTGPU_OK;
}

int launch_scatter(const rtgpu_render_settings& st, uint32_t T, uint32_t world, const float* rank0, const float* others, float* frame, hipStream_t stream)
{
    const TileGrid g = tile_grid(st, T);
    const uint64_t n = (uint64_t)(st.x_end - st.x_start) * (uint64_t)(st.y_end - st.y_start);
    if (n == 0) return RTGPU_OK;
    const uint64_t stride = rank_floats(g, 0, world);
    hipLaunchKernelGGL(scatter_tiles_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, st, T, g.tilesX, world, rank0, others, stride, frame);
    MG_HIP(hipGetLastError());
    return RTGPU_OK;
}

int default_tile_size()
{
    if (const char* e = std::getenv("RTGPU_TILE_SIZE")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 1024) return v;
    }
    return 32;
}

} // namespace
} // namespace rt

struct rtgpu_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    float* stage = nullptr;       // rank 0: the peers' buffers, [world - 1][stride]
    uint64_t stageFloats = 0;
    int timeoutMs = 0;            // how long a gather may take before the communicator is aborted; 0: the call only enqueues
    hipEvent_t done = nullptr;
    bool broken = false;          // aborted after a deadline: every later call fails
};

// What a scene keeps for rtgpu_render with n_gpus > 1: its copies on the other devices, one communicator and one stream
// per device, the compact buffers and the device frame buffer.
struct rtgpu_multi {
    std::vector<int> devices;               // devices[0] = the scene's own
    std::vector<rtgpu_scene*> replicas;     // replicas[0] = null (the scene itself)
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> done;           // per rank: its part of the gather has completed
    std::vector<float*> tiles;
    std::vector<uint64_t> tileFloats;
    float* stage = nullptr;
    uint64_t stageFloats = 0;
    float* frame = nullptr;
    uint64_t frameFloats = 0;
    int loopback = 0;                       // kLoopbackCopy / kLoopbackRccl: every rank on the scene's own device
    uint32_t* stallFlag = nullptr;          // pinned; test hook of the loopback (a transfer that does not arrive)
};

namespace rt {

void destroy_multi(rtgpu_multi* m)
{
    if (!m) return;
    for (size_t i = 0; i < m->devices.size(); ++i) {
        (void)hipSetDevice(m->devices[i]);
        if (i < m->comms.size() && m->comms[i]) (void)rccl().CommDestroy(m->comms[i]);
        if (i < m->streams.size() && m->streams[i]) (void)hipStreamDestroy(m->streams[i]);
        if (i < m->done.size() && m->done[i]) (void)hipEventDestroy(m->done[i]);
        if (i < m->tiles.size() && m->tiles[i]) (void)hipFree(m->tiles[i]);
        if (i < m->replicas.size() && m->replicas[i]) rtgpu_scene_destroy(m->replicas[i]);
    }
    if (!m->devices.empty()) (void)hipSetDevice(m->devices[0]);
    if (m->stage) (void)hipFree(m->stage);
    if (m->frame) (void)hipFree(m->frame);
    if (m->stallFlag) (void)hipHostFree(m->stallFlag);
    delete m;
}

namespace {

int ensure_floats(float*& p, uint64_t& have, uint64_t need)
{
    if (need <= have && p) return RTGPU_OK;
    if (p) { (void)hipFree(p); p = nullptr; have = 0; }
    MG_HIP(hipMalloc((void**)&p, std::max<uint64_t>(need, 1) * sizeof(float)));
    have = need;
    return RTGPU_OK;
}

// How long a gather may take before it is given up (RTGPU_COMM_TIMEOUT_MS; 0 = no limit).  A transfer between two GPUs of
// one node takes well under a millisecond; what the limit is for is a peer that never posts its side.
int default_timeout_ms()
{
    if (const char* e = std::getenv("RTGPU_COMM_TIMEOUT_MS")) {
        const long v = std::atol(e);
        if (v >= 0 && v <= 86400000) return (int)v;
    }
    return 60000;
}

// Waits until every event has completed or `timeoutMs` have gone by (0: no limit).  Returns the index of an event that
// has not completed at the deadline, -1 when all have, -2 on a HIP error (message set).
int wait_events(const std::vector<hipEvent_t>& events, const std::vector<int>& devices, int timeoutMs)
{
    const auto t0 = std::chrono::steady_clock::now();
    size_t next = 0;
    unsigned spins = 0;
    while (next < events.size()) {
        if (hipSetDevice(devices[next]) != hipSuccess) { set_error(RTGPU_ERR_HIP, "hipSetDevice failed"); return -2; }
        const hipError_t q = hipEventQuery(events[next]);
        if (q == hipSuccess) { ++next; continue; }
        if (q != hipErrorNotReady) { hip_fail(q, "hipEventQuery"); return -2; }
        if (timeoutMs > 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeoutMs)) return (int)next;
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));   // the first 2000 polls spin: a frame's gather ends within them
    }
    return -1;
}

// Test hook of the loopback transport: a transfer that does not arrive.  One thread waits until the host releases it
// (the deadline's abort does) or until `ticks` of the 100 MHz wall clock have passed, so the grid always drains.
__global__ void stall_kernel(volatile uint32_t* release, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (*release == 0u && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

enum { kLoopbackOff = 0, kLoopbackCopy = 1, kLoopbackRccl = 2 };
// RTGPU_MULTI_LOOPBACK: every rank of rtgpu_render(n_gpus = G) on the scene's own device -- G copies of the scene, G host
// threads, G compact buffers, the stage layout and the scatter kernel as on G devices.  "1" / "copy": device-to-device copies
// stand in for the send / receive pairs; "rccl": the pairs go through a one-rank communicator (sends to itself).  For
// boxes with fewer GPUs than ranks.
int loopback_mode()
{
    const char* e = std::getenv("RTGPU_MULTI_LOOPBACK");
    if (!e || !*e || std::strcmp(e, "0") == 0) return kLoopbackOff;
    return std::strcmp(e, "rccl") == 0 ? kLoopbackRccl : kLoopbackCopy;
}
// RTGPU_MULTI_TEST_FAIL = replicate | stream | comm: the set-up of the next multi-GPU render fails at that step;
// recv: rank 0 posts no receives (test hooks)
bool test_fail_at(const char* step)
{
    const char* e = std::getenv("RTGPU_MULTI_TEST_FAIL");
    return e && std::strcmp(e, step) == 0;
}

// Copies of the scene, streams, events and communicators for `G` ranks.  Built apart from the scene and attached only
// when complete: a failure half-way leaves the scene as it was, and the next call starts over.
int make_multi(const rtgpu_scene& scene, int G, int nDev, int loopback, rtgpu_multi** out)
{
    Rccl& R = rccl();
    struct Guard {
        rtgpu_multi* m;
        ~Guard() { destroy_multi(m); }
    } guard{new rtgpu_multi()};
    rtgpu_multi* m = guard.m;
    m->loopback = loopback;
    m->devices.push_back(scene.device);
    for (int d = 0; (int)m->devices.size() < G; ++d) {
        if (loopback) m->devices.push_back(scene.device);
        else if (d >= nDev) break;
        else if (d != scene.device) m->devices.push_back(d);
    }
    m->replicas.assign((size_t)G, nullptr);
    m->comms.assign((size_t)G, nullptr);
    m->streams.assign((size_t)G, nullptr);
    m->done.assign((size_t)G, nullptr);
    m->tiles.assign((size_t)G, nullptr);
    m->tileFloats.assign((size_t)G, 0);
    for (int r = 1; r < G; ++r) {
        if (test_fail_at("replicate") && r == G - 1) return set_error(RTGPU_ERR_OUT_OF_MEMORY, "copying the scene to another device: injected failure");
        if (int rc = rtgpu_scene_replicate(&scene, m->devices[(size_t)r], &m->replicas[(size_t)r])) return rc;
    }
    for (int r = 0; r < G; ++r) {
        MG_HIP(hipSetDevice(m->devices[(size_t)r]));
        if (test_fail_at("stream") && r == G - 1) return set_error(RTGPU_ERR_HIP, "hipStreamCreateWithFlags: injected failure");
        MG_HIP(hipStreamCreateWithFlags(&m->streams[(size_t)r], hipStreamNonBlocking));
        MG_HIP(hipEventCreateWithFlags(&m->done[(size_t)r], hipEventDisableTiming));
    }
    MG_HIP(hipSetDevice(scene.device));
    if (test_fail_at("comm")) return set_error(RTGPU_ERR_COMM, "ncclCommInitAll: injected failure");
    if (loopback == kLoopbackOff) MG_NCCL(R.CommInitAll(m->comms.data(), G, m->devices.data()));
    else if (loopback == kLoopbackRccl) MG_NCCL(R.CommInitAll(m->comms.data(), 1, m->devices.data()));   // one rank; every pair is a send to itself
    else MG_HIP(hipHostMalloc((void**)&m->stallFlag, sizeof(uint32_t)));
    guard.m = nullptr;
    *out = m;
    return RTGPU_OK;
}

// Renderer::render over `G` devices of this process.  One call at a time per scene (scene.multiMutex).
int render_multi(const rtgpu_scene& scene, const rtgpu_camera& cam, const rtgpu_render_settings& st, float* rgbOut, rtgpu_render_stats* stats)
{
    const int G = st.n_gpus;
    const int loopback = loopback_mode();
    int nDev = 0;
    MG_HIP(hipGetDeviceCount(&nDev));
    if (!loopback && G > nDev) return set_error(RTGPU_ERR_NO_DEVICE, "settings ask for " + std::to_string(G) + " GPUs, " + std::to_string(nDev) + " present");
    Rccl& R = rccl();
    if (loopback != kLoopbackCopy && !R.ok()) return set_error(RTGPU_ERR_COMM, R.error);
    std::lock_guard<std::mutex> multiLock(scene.multiMutex);
    MG_HIP(hipSetDevice(scene.device));
    if (scene.multi && ((int)scene.multi->devices.size() != G || scene.multi->loopback != loopback)) { destroy_multi(scene.multi); scene.multi = nullptr; }
    if (!scene.multi)
        if (int rc = make_multi(scene, G, nDev, loopback, &scene.multi)) return rc;
    rtgpu_multi& m = *scene.multi;
    const uint32_t T = (uint32_t)default_tile_size();
    const TileGrid grid = tile_grid(st, T);
    for (int r = 0; r < G; ++r) {
        MG_HIP(hipSetDevice(m.devices[(size_t)r]));
        if (int rc = ensure_floats(m.tiles[(size_t)r], m.tileFloats[(size_t)r], rank_floats(grid, (uint32_t)r, (uint32_t)G))) return rc;
    }
    MG_HIP(hipSetDevice(scene.device));
    const uint64_t stride = rank_floats(grid, 0, (uint32_t)G);
    if (int rc = ensure_floats(m.stage, m.stageFloats, stride * (uint64_t)(G - 1))) return rc;
    const uint64_t frameFloats = (uint64_t)st.width * st.height * 3;
    if (int rc = ensure_floats(m.frame, m.frameFloats, frameFloats)) return rc;
    // pixels outside the tile keep the caller's values (Renderer::render only writes inside the tile)
    MG_HIP(hipMemcpyAsync(m.frame, rgbOut, frameFloats * sizeof(float), hipMemcpyHostToDevice, m.streams[0]));

    // ---- render: one host thread per device ------------------------------------------------------------
    std::vector<int> rcs((size_t)G, RTGPU_OK);
    std::vector<std::string> msgs((size_t)G);
    std::vector<rtgpu_render_stats> rstats((size_t)G);
    auto work = [&](int r) {
        const rtgpu_scene& sc = r == 0 ? scene : *m.replicas[(size_t)r];
        if (hipSetDevice(sc.device) != hipSuccess) { rcs[(size_t)r] = RTGPU_ERR_HIP; msgs[(size_t)r] = "hipSetDevice failed"; return; }
        const TileOrder order{T, (uint32_t)G, (uint32_t)r};
        std::string err;
        const hipError_t e = render_wavefront(sc, cam, st, &order, m.tiles[(size_t)r], &rstats[(size_t)r], nullptr, m.streams[(size_t)r], err);
        if (e != hipSuccess) { rcs[(size_t)r] = e == hipErrorOutOfMemory ? RTGPU_ERR_OUT_OF_MEMORY : RTGPU_ERR_HIP; msgs[(size_t)r] = err.empty() ? hipGetErrorString(e) : err; }
    };
    {
        std::vector<std::thread> threads;
        for (int r = 1; r < G; ++r) threads.emplace_back(work, r);
        work(0);
        for (std::thread& t : threads) t.join();
    }
    // the gather is only entered when every device has rendered: a device that failed leaves no peer waiting
    for (int r = 0; r < G; ++r)
        if (rcs[(size_t)r] != RTGPU_OK) return set_error(rcs[(size_t)r], "GPU " + std::to_string(m.devices[(size_t)r]) + ": " + msgs[(size_t)r]);

    // ---- gather: all devices' transfers in one group, issued from this thread ------------------------------
    if (loopback == kLoopbackCopy) {
        // render_wavefront has waited for its stream, so the ranks' buffers are complete: the copies only need rank 0's stream
        if (const char* e = std::getenv("RTGPU_MULTI_TEST_STALL_MS")) {
            *m.stallFlag = 0u;
            hipLaunchKernelGGL(stall_kernel, dim3(1), dim3(1), 0, m.streams[0], (volatile uint32_t*)m.stallFlag, (unsigned long long)std::max(0L, std::atol(e)) * 100000ull);
            MG_HIP(hipGetLastError());
        }
        for (int r = 1; r < G; ++r) {
            const uint64_t n = rank_floats(grid, (uint32_t)r, (uint32_t)G);
            if (n) MG_HIP(hipMemcpyAsync(m.stage + (uint64_t)(r - 1) * stride, m.tiles[(size_t)r], n * sizeof(float), hipMemcpyDeviceToDevice, m.streams[0]));
        }
    } else {
        // the group is closed on every path: an error inside it is reported after ncclGroupEnd
        ncclResult_t first = R.GroupStart();
        const char* where = "ncclGroupStart";
        if (first == ncclSuccess) {
            for (int r = 1; r < G && first == ncclSuccess; ++r) {
                const uint64_t n = rank_floats(grid, (uint32_t)r, (uint32_t)G);
                if (n == 0) continue;
                const bool self = loopback == kLoopbackRccl;
                first = R.Send(m.tiles[(size_t)r], n, ncclFloat, 0, m.comms[self ? 0 : (size_t)r], m.streams[self ? 0 : (size_t)r]);
                where = "ncclSend";
                if (first != ncclSuccess) break;
                if (test_fail_at("recv")) continue;   // test hook: a send whose receive is never posted
                first = R.Recv(m.stage + (uint64_t)(r - 1) * stride, n, ncclFloat, self ? 0 : r, m.comms[0], m.streams[0]);
                where = "ncclRecv";
            }
            const ncclResult_t end = R.GroupEnd();
            if (first == ncclSuccess && end != ncclSuccess) { first = end; where = "ncclGroupEnd"; }
        }
        if (first != ncclSuccess) {
            const int rc = set_error(RTGPU_ERR_COMM, std::string(where) + ": " + R.GetErrorString(first));
            for (ncclComm_t& c : m.comms) if (c) { (void)R.CommAbort(c); c = nullptr; }
            destroy_multi(scene.multi);   // what a failed group leaves behind is not reused
            scene.multi = nullptr;
            return rc;
        }
    }
    MG_HIP(hipSetDevice(scene.device));
    if (int rc = launch_scatter(st, T, (uint32_t)G, m.tiles[0], m.stage, m.frame, m.streams[0])) return rc;
    for (int r = 0; r < G; ++r) {
        MG_HIP(hipSetDevice(m.devices[(size_t)r]));
        MG_HIP(hipEventRecord(m.done[(size_t)r], m.streams[(size_t)r]));
    }
    // (the frame is copied out after the wait: a copy into pageable host memory waits for the stream inside the runtime,
    // where no deadline applies)
    // ---- deadline: a transfer whose other side never arrives must not hold the caller for ever -----------------------------
    const int timeoutMs = default_timeout_ms();
    const int late = wait_events(m.done, m.devices, timeoutMs);
    if (late == -2) return RTGPU_ERR_HIP;
    if (late >= 0) {
        const int rc = set_error(RTGPU_ERR_COMM, "the gather did not complete within " + std::to_string(timeoutMs) + " ms (rank " + std::to_string(late) +
                                                     " of " + std::to_string(G) + " still waiting); communicators aborted");
        if (m.stallFlag) *(volatile uint32_t*)m.stallFlag = 1u;
        for (ncclComm_t& c : m.comms) if (c) { (void)R.CommAbort(c); c = nullptr; }   // ends the transfer kernels that wait for their peers
        for (int r = 0; r < G; ++r) {
            (void)hipSetDevice(m.devices[(size_t)r]);
            (void)hipStreamSynchronize(m.streams[(size_t)r]);
        }
        destroy_multi(scene.multi);
        scene.multi = nullptr;
        (void)hipSetDevice(scene.device);
        return rc;
    }
    MG_HIP(hipSetDevice(scene.device));
    MG_HIP(hipMemcpyAsync(rgbOut, m.frame, frameFloats * sizeof(float), hipMemcpyDeviceToHost, m.streams[0]));
    MG_HIP(hipStreamSynchronize(m.streams[0]));
    if (stats) {
        *stats = rtgpu_render_stats{};
        for (const rtgpu_render_stats& s : rstats) {
            stats->primary_rays += s.primary_rays;
            stats->extension_rays += s.extension_rays;
            stats->shadow_rays += s.shadow_rays;
            stats->device_ms = std::max(stats->device_ms, s.device_ms);
        }
    }
    return RTGPU_OK;
}

} // namespace

int render_multi_entry(const rtgpu_scene& scene, const rtgpu_camera& cam, const rtgpu_render_settings& st, float* rgbOut, rtgpu_render_stats* stats)
{
    return guarded([&]() -> int { return render_multi(scene, cam, st, rgbOut, stats); });
}

} // namespace rt

extern "C" {

int rtgpu_comm_unique_id(uint8_t id[RTGPU_COMM_ID_BYTES])
{
    using namespace rt;
    if (!id) return set_error(RTGPU_ERR_INVALID_ARGUMENT, "null id");
    static_assert(sizeof(ncclUniqueId) == RTGPU_COMM_ID_BYTES, "communicator id size");
    Rccl& R = rccl();
    if (!R.ok()) return set_error(RTGPU_ERR_COMM, R.error);
    ncclUniqueId u;
    MG_NCCL(R.GetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return RTGPU_OK;
}

int rtgpu_comm_create(const uint8_t id[RTGPU_COMM_ID_BYTES], int rank, int world, rtgpu_comm** out)
{
    using namespace rt;
    if (!id || !out) return set_error(RTGPU_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return set_error(RTGPU_ERR_INVALID_ARGUMENT, "rank / number of ranks out of range");
    Rccl& R = rccl();
    if (!R.ok()) return set_error(RTGPU_ERR_COMM, R.error);
    int device = 0;
    MG_HIP(hipGetDevice(&device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    MG_NCCL(R.CommInitRank(&c, world, u, rank));
    rtgpu_comm* cm = new (std::nothrow) rtgpu_comm();
    if (!cm) { (void)R.CommDestroy(c); return set_error(RTGPU_ERR_OUT_OF_MEMORY, "out of host memory"); }
    cm->comm = c;
    cm->rank = rank;
    cm->world = world;
    cm->device = device;
    cm->timeoutMs = default_timeout_ms();
    if (hipEventCreateWithFlags(&cm->done, hipEventDisableTiming) != hipSuccess) cm->done = nullptr;   // without it the gather only enqueues
    *out = cm;
    return RTGPU_OK;
}

int rtgpu_comm_set_timeout_ms(rtgpu_comm* c, int timeout_ms)
{
    using namespace rt;
    if (!c || timeout_ms < 0) return set_error(RTGPU_ERR_INVALID_ARGUMENT, "null communicator / negative time limit");
    c->timeoutMs = timeout_ms;
    return RTGPU_OK;
}

void rtgpu_comm_destroy(rtgpu_comm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stage) (void)hipFree(c->stage);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->comm) (void)rt::rccl().CommDestroy(c->comm);
    delete c;
}

uint64_t rtgpu_tile_buffer_floats(const rtgpu_render_settings* settings, int tile_size, int rank, int world)
{
    if (rt::check_tiles(settings, tile_size, rank, world)) return 0;
    return rt::rank_floats(rt::tile_grid(*settings, (uint32_t)tile_size), (uint32_t)rank, (uint32_t)world);
}

int rtgpu_render_tiles_device(const rtgpu_scene* scene, const rtgpu_camera* camera, const rtgpu_render_settings* settings,
                              int tile_size, int rank, int world, float* d_tiles, rtgpu_render_stats* stats, void* hip_stream)
{
    using namespace rt;
    if (!scene) return set_error(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    if (!camera) return set_error(RTGPU_ERR_NO_CAMERA, "No camera in scene.");
    if (int rc = check_tiles(settings, tile_size, rank, world)) return rc;
    if (settings->geometry_aa < 1 || settings->material_aa < 1 || settings->max_path_length < 1 || settings->max_path_length > 64)
        return set_error(RTGPU_ERR_INVALID_ARGUMENT, "sample counts and path length out of range");
    if ((int64_t)settings->geometry_aa * settings->material_aa > (int64_t)RTGPU_MAX_PATHS_PER_PIXEL)
        return set_error(RTGPU_ERR_LIMIT, "geometry_aa * material_aa exceeds 2^25 paths per pixel");
    if (camera->aperture < 0.0 || (camera->aperture > 0.0 && settings->geometry_aa < 2))
        return set_error(RTGPU_ERR_INVALID_ARGUMENT, "a thin-lens camera (aperture > 0) needs geometry_aa >= 2");
    if (!d_tiles && rtgpu_tile_buffer_floats(settings, tile_size, rank, world)) return set_error(RTGPU_ERR_INVALID_ARGUMENT, "tile buffer is null");
    MG_HIP(hipSetDevice(scene->device));
    return guarded([&]() -> int {
        std::string err;
        const TileOrder order{(uint32_t)tile_size, (uint32_t)world, (uint32_t)rank};
        const hipError_t e = render_wavefront(*scene, *camera, *settings, &order, d_tiles, stats, nullptr, (hipStream_t)hip_stream, err);
        if (e != hipSuccess) return set_error(e == hipErrorOutOfMemory ? RTGPU_ERR_OUT_OF_MEMORY : RTGPU_ERR_HIP, err.empty() ? hipGetErrorString(e) : err);
        return RTGPU_OK;
    });
}

int rtgpu_scatter_tiles_device(const rtgpu_render_settings* settings, int tile_size, int world, const float* d_all_tiles, float* d_frame, void* hip_stream)
{
    using namespace rt;
    if (int rc = check_tiles(settings, tile_size, 0, world)) return rc;
    if (!d_all_tiles || !d_frame) return set_error(RTGPU_ERR_INVALID_ARGUMENT, "null buffer");
    const uint64_t stride = rank_floats(tile_grid(*settings, (uint32_t)tile_size), 0, (uint32_t)world);
    return launch_scatter(*settings, (uint32_t)tile_size, (uint32_t)world, d_all_tiles, d_all_tiles + stride, d_frame, (hipStream_t)hip_stream);
}

int rtgpu_gather_tiles_device(rtgpu_comm* comm, const rtgpu_render_settings* settings, int tile_size, const float* d_tiles, float* d_frame, void* hip_stream)
{
    using namespace rt;
    if (!comm) return set_error(RTGPU_ERR_INVALID_ARGUMENT, "communicator is null");
    if (int rc = check_tiles(settings, tile_size, comm->rank, comm->world)) return rc;
    const TileGrid grid = tile_grid(*settings, (uint32_t)tile_size);
    const uint64_t mine = rank_floats(grid, (uint32_t)comm->rank, (uint32_t)comm->world);
    if (mine && !d_tiles) return set_error(RTGPU_ERR_INVALID_ARGUMENT, "tile buffer is null");
    if (comm->rank == 0 && !d_frame) return set_error(RTGPU_ERR_INVALID_ARGUMENT, "frame buffer is null on rank 0");
    if (comm->broken) return set_error(RTGPU_ERR_COMM, "the communicator was aborted after a gather ran out of time");
    Rccl& R = rccl();
    const hipStream_t stream = (hipStream_t)hip_stream;
    MG_HIP(hipSetDevice(comm->device));
    // Aborts the communicator: what has been enqueued stops waiting for its peers, and the handle refuses further use.
    auto give_up = [&](const std::string& why) -> int {
        const int rc = set_error(RTGPU_ERR_COMM, why);
        if (comm->comm) { (void)R.CommAbort(comm->comm); comm->comm = nullptr; }
        comm->broken = true;
        (void)hipStreamSynchronize(stream);
        return rc;
    };
    const uint64_t stride = rank_floats(grid, 0, (uint32_t)comm->world);
    if (comm->rank != 0) {
        if (mine) {
            const ncclResult_t r = R.Send(d_tiles, mine, ncclFloat, 0, comm->comm, stream);
            if (r != ncclSuccess) return give_up(std::string("ncclSend: ") + R.GetErrorString(r));
        }
    } else {
        if (comm->world > 1) {
            if (int rc = ensure_floats(comm->stage, comm->stageFloats, stride * (uint64_t)(comm->world - 1))) return rc;
            // the group is closed on every path: an error inside it is reported after ncclGroupEnd
            ncclResult_t first = R.GroupStart();
            const char* where = "ncclGroupStart";
            if (first == ncclSuccess) {
                for (int r = 1; r < comm->world && first == ncclSuccess; ++r) {
                    const uint64_t n = rank_floats(grid, (uint32_t)r, (uint32_t)comm->world);
                    if (n) first = R.Recv(comm->stage + (uint64_t)(r - 1) * stride, n, ncclFloat, r, comm->comm, stream);
                    where = "ncclRecv";
                }
                const ncclResult_t end = R.GroupEnd();
                if (first == ncclSuccess && end != ncclSuccess) { first = end; where = "ncclGroupEnd"; }
            }
            if (first != ncclSuccess) return give_up(std::string(where) + ": " + R.GetErrorString(first));
        }
        if (int rc = launch_scatter(*settings, (uint32_t)tile_size, (uint32_t)comm->world, d_tiles, comm->stage, d_frame, stream)) return rc;
    }
    if (comm->timeoutMs > 0 && comm->done && comm->world > 1) {
        // the deadline: a peer that died after the communicator was made never posts its side, and the transfer kernel on
        // this device would wait for it for ever
        MG_HIP(hipEventRecord(comm->done, stream));
        const int late = wait_events({comm->done}, {comm->device}, comm->timeoutMs);
        if (late == -2) return RTGPU_ERR_HIP;
        if (late >= 0) return give_up("the gather did not complete within " + std::to_string(comm->timeoutMs) + " ms on rank " + std::to_string(comm->rank) +
                                      " of " + std::to_string(comm->world) + "; communicator aborted");
    }
    return RTGPU_OK;
}

int rtgpu_scene_replicate(const rtgpu_scene* src, int device, rtgpu_scene** out)
{
    using namespace rt;
    if (!src || !out) return set_error(RTGPU_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    int nDev = 0;
    MG_HIP(hipGetDeviceCount(&nDev));
    if (device < 0 || device >= nDev) return set_error(RTGPU_ERR_NO_DEVICE, "no such device");
    return guarded([&]() -> int {
        int before = 0;
        MG_HIP(hipGetDevice(&before));
        rtgpu_scene* sc = new rtgpu_scene();
        sc->device = device;
        sc->view = src->view;
        sc->info = src->info;
        sc->photons = src->photons;
        // every array pointer is cleared first, so that a failure half-way frees only what the copy owns
        for (const SceneArray& a : src->arrays) *(const void**)((char*)&sc->view + a.viewOffset) = nullptr;
        hipError_t e = hipSetDevice(device);
        for (const SceneArray& a : src->arrays) {
            if (e != hipSuccess) break;
            const void* from = *(const void* const*)((const char*)&src->view + a.viewOffset);
            if (!from || a.bytes == 0) continue;
            void* p = nullptr;
            e = hipMalloc(&p, a.bytes);
            if (e != hipSuccess) break;
            *(const void**)((char*)&sc->view + a.viewOffset) = p;
            sc->arrays.push_back(a);
            e = device == src->device ? hipMemcpy(p, from, a.bytes, hipMemcpyDeviceToDevice) : hipMemcpyPeer(p, device, from, src->device, a.bytes);
        }
        (void)hipSetDevice(before);
        if (e != hipSuccess) {
            sc->release();
            delete sc;
            return hip_fail(e, "copying the scene to another device");
        }
        static_workspace_prepare(device);
        *out = sc;
        return RTGPU_OK;
    });
}

} // extern "C"
