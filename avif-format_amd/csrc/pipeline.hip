// pipeline.hip -- the bound device contexts and the in-process multi-GPU row-tile scheduler.
//
// What the reference does on this path: ONE thread, one row per advanceState(), convert, next row
// (WriteHeifImage.cpp:1017-1029, ReadHeifImage.cpp:141-160; buffer sized at Write.cpp:279-299).  What an 8-GPU MI355X node
// wants instead: the image cut into contiguous even-row tiles, one per GPU, no exchange step between them,
// every tile streamed through its GPU in sub-tiles so that the H2D of sub-tile k+1, the kernel of k and the D2H of k-1 overlap
// -- and all of it driven from the one calling thread the host application gives the plug-in.
//
// Structure:
//   * avifgpu_init_devices binds N contexts.  A context = a HIP device ordinal + ONE worker thread whose current device is
//     that ordinal for its whole life (the caller's current device is never touched) + kSlots staging slots.  A slot owns a
//     stream, a completion event, device in/out buffers and (lazily) two pinned host buffers.
//   * The calling thread only queues tiles (write_tile_enqueue / read_tile_enqueue) and waits for slots; the worker issues the
//     copies and the launch, and finishes tiles in order.  A host buffer that is already page-locked (the shim's tile buffers,
//     planes from avifgpu_image_alloc, memory the caller registered) is the DMA source / target itself; a pageable buffer
//     (libheif's planes, numpy arrays) is bounced through the slot's pinned buffer with the WORKER's memcpy, so N contexts
//     also give N memcpy streams and the caller's memory is never registered / unregistered behind its back.
//   * write_rows_host / read_rows_host cut a row range into one tile per context (row_cut: the rule of sharding.row_cut) and
//     deal sub-tiles round-robin, so every device's DMA engines and its x16 link are busy at the same time.  No collective,
//     no peer traffic: a tile's output bytes depend on its own rows only, so the planes are byte-identical for any N.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

#include "staging.h"

namespace avifgpu {
namespace {

constexpr int kMaxSlots = 8;
constexpr int kDefaultLanes = 2;        // workers per bound device (AVIFGPU_LANES): see contexts_init

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// Device staging pitch of a row of `bytes`: tight when that keeps every row 16-byte aligned (then a contiguous host tile is ONE
// linear DMA transfer and the kernels take their aligned paths), padded otherwise.
size_t staging_pitch(size_t bytes) { return (bytes % 16 == 0) ? bytes : align256(bytes); }

struct Job {
    bool is_write = true;
    int slot = 0;
    avifgpu_write_desc wd{};
    avifgpu_read_desc rd{};
    int row0 = 0, nrows = 0;
    const void* rows_in = nullptr;   int64_t rows_in_stride = 0;      // write: interleaved host rows
    void* planes_out[4] = {};        int64_t planes_out_stride[4] = {};
    const void* planes_in[4] = {};   int64_t planes_in_stride[4] = {}; // read
    void* rows_out = nullptr;        int64_t rows_out_stride = 0;
    IccArgs icc;
    // filled by start(): what finish() still has to copy out of the pinned bounce buffer
    struct Bounce { uint8_t* dst; int64_t dst_stride; size_t off, pitch, bytes; int rows; } bounce[4];
    int nbounce = 0;
    char label[kLabelBytes] = "";
    double t_queued = 0, t_start = 0, t_issued = 0;        // AVIFGPU_TRACE
    uint64_t upload_seq = 0;                               // position in its device's upload order (enqueue())
};

struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t up_done = nullptr;                          // this tile's host-to-device copies (UploadOrder)
    void* d_in = nullptr;  size_t d_in_cap = 0;
    void* d_out = nullptr; size_t d_out_cap = 0;
    void* h_rows = nullptr;   size_t h_rows_cap = 0;      // pinned: the shim's tile buffer / bounce of pageable rows
    void* h_planes = nullptr; size_t h_planes_cap = 0;    // pinned: bounce of pageable planes
    bool busy = false;                                    // queued or in flight (guarded by Ctx::mu)
};

struct Ctx {
    int device = -1;
    int nslots = 4;
    Slot slot[kMaxSlots];
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Job> queue;
    bool stop = false, started = false;
    int init_err = 0; char init_msg[256] = "";
    int err = 0; char err_msg[512] = "";                   // first failure of any tile since the last wait_all
    char last_label[kLabelBytes] = "";
    // a pinned tile buffer the calling thread asked for: allocated by the WORKER, whose affinity is the device's NUMA node, so the
    // pages are first touched (and page-locked) on the socket the GPU's x16 link hangs off
    bool alloc_pending = false; int alloc_slot = 0; size_t alloc_bytes = 0; int alloc_rc = 0;
    bool pinned_to_node = false;
    // what this context moved over its device's link since the binding (or the last avifgpu_device_traffic_reset): payload bytes of the
    // tiles it issued, per direction, and how many of them went through a bounce buffer (pageable caller memory)
    std::atomic<uint64_t> tiles{0}, bytes_h2d{0}, bytes_d2h{0}, bytes_bounced{0};
};

// Where a bound device sits in the host: PCI bus id, NUMA node, the node's CPUs ("report which GPUs hang off which
// root complex").  Read once per binding from sysfs; AVIFGPU_SYSFS_ROOT redirects the reads (tests use a fake tree).
struct DeviceTopo { int device = -1; int numa_node = -1; std::string bdf, cpulist; int workers = 0; bool pinned = false; };
std::vector<DeviceTopo> g_topo;

bool read_small_file(const std::string& path, std::string& out)
{
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[512];
    const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
    fclose(f);
    buf[n] = 0;
    out = buf;
    while (!out.empty() && (out.back() == '\n' || out.back() == ' ' || out.back() == '\r')) out.pop_back();
    return true;
}

// "0-63,128-191" -> cpu_set_t; false for an empty or malformed list
bool parse_cpulist(const std::string& list, cpu_set_t& set)
{
    CPU_ZERO(&set);
    int count = 0;
    const char* p = list.c_str();
    while (*p) {
        char* e = nullptr;
        const long a = strtol(p, &e, 10);
        if (e == p || a < 0) return false;
        long b = a;
        p = e;
        if (*p == '-') { b = strtol(p + 1, &e, 10); if (e == p + 1 || b < a) return false; p = e; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, &set); ++count; }
        if (*p == ',') ++p; else if (*p) return false;
    }
    return count > 0;
}

int topology_probe(const char* sysfs_root, const char* bdf_in, int& numa_node, std::string& cpulist)
{
    numa_node = -1; cpulist.clear();
    if (!bdf_in || !*bdf_in) return AVIFGPU_formatBadParameters;
    std::string bdf(bdf_in);
    for (char& ch : bdf) if (ch >= 'A' && ch <= 'F') ch = (char)(ch - 'A' + 'a');          // sysfs names are lower case
    const std::string root = (sysfs_root && *sysfs_root) ? sysfs_root : "/sys";
    const std::string dev = root + "/bus/pci/devices/" + bdf;
    std::string txt;
    if (!read_small_file(dev + "/numa_node", txt)) return AVIFGPU_readErr;                 // no such device in this tree
    numa_node = atoi(txt.c_str());
    // the node's CPUs; a single-node host (or a VM) reports node -1 and the device's own local_cpulist, if any, is used
    if (numa_node >= 0 && read_small_file(root + "/devices/system/node/node" + std::to_string(numa_node) + "/cpulist", txt)) cpulist = txt;
    else if (read_small_file(dev + "/local_cpulist", txt)) cpulist = txt;
    return 0;
}

std::mutex g_ctx_mu;                                       // init / shutdown
std::vector<Ctx*>* g_ctxs = nullptr;                       // heap-held on purpose: a process that never calls avifgpu_shutdown
                                                           // must not run thread destructors at exit
std::vector<int> g_bound;                                  // the ordinals of the current binding (as passed by the caller)

// AVIFGPU_TRACE=1: one stderr line per tile with the worker-side timeline (microseconds since the context was created)
bool g_trace = false;
double now_us()
{
    using namespace std::chrono;
    static const steady_clock::time_point t0 = steady_clock::now();
    return duration<double, std::micro>(steady_clock::now() - t0).count();
}

int env_int(const char* name, int dflt, int lo, int hi)
{
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    const long x = strtol(v, nullptr, 0);
    return (int)std::min<long>(std::max<long>(x, lo), hi);
}

// Is [p, p + bytes) page-locked memory HIP can DMA to directly?  The whole span has to lie inside ONE page-locked range: the
// runtime reports the range its first byte belongs to (start + size) and the last byte must fall inside it -- two ends that are
// each page-locked say nothing about what lies between them.  A runtime that does not report ranges for host memory degrades to
// the probe of both ends.
bool is_pinned(const void* p, size_t bytes)
{
    auto one = [](const void* q) {
        hipPointerAttribute_t a;
        std::memset(&a, 0, sizeof(a));
        if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    };
    if (!p || !one(p)) return false;
    if (bytes <= 1) return true;
    void* start = nullptr; size_t size = 0;
    if (hipPointerGetAttribute(&start, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, const_cast<void*>(p)) == hipSuccess &&
        hipPointerGetAttribute(&size, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, const_cast<void*>(p)) == hipSuccess && start && size) {
        const uintptr_t lo = reinterpret_cast<uintptr_t>(start), a = reinterpret_cast<uintptr_t>(p);
        return a >= lo && a + bytes <= lo + size;
    }
    (void)hipGetLastError();
    return one(static_cast<const uint8_t*>(p) + bytes - 1);
}

int grow_device(void** p, size_t* cap, size_t need)
{
    if (need <= *cap) return 0;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    const hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess) { *p = nullptr; return hip_fail(e, "hipMalloc(staging)", AVIFGPU_memFullErr); }
    *cap = need;
    return 0;
}

int grow_pinned(void** p, size_t* cap, size_t need)
{
    if (need <= *cap) return 0;
    if (*p) { (void)hipHostFree(*p); *p = nullptr; *cap = 0; }
    const hipError_t e = hipHostMalloc(p, need, hipHostMallocPortable);
    if (e != hipSuccess) { *p = nullptr; (void)hipGetLastError(); return fail(AVIFGPU_memFullErr, "hipHostMalloc(%zu bytes of pinned staging) failed", need); }
    *cap = need;
    return 0;
}

// rows x bytes, host -> device (or back), ONE linear transfer when both sides are contiguous
hipError_t copy_rows(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t bytes, size_t rows, hipMemcpyKind kind, hipStream_t st)
{
    if (rows == 0 || bytes == 0) return hipSuccess;
    if (dst_pitch == bytes && src_pitch == bytes) return hipMemcpyAsync(dst, src, bytes * rows, kind, st);
    return hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, bytes, rows, kind, st);
}

void copy_rows_here(uint8_t* dst, size_t dst_pitch, const uint8_t* src, size_t src_pitch, size_t bytes, size_t rows)
{
    if (dst_pitch == bytes && src_pitch == bytes) { std::memcpy(dst, src, bytes * rows); return; }
    for (size_t r = 0; r < rows; ++r) std::memcpy(dst + r * dst_pitch, src + r * src_pitch, bytes);
}

// ---- CPU copies of pageable caller memory ------------------------------------------------------------------------------
// A tile of pageable rows (or planes) goes through a pinned bounce buffer: a memcpy of 16-32 MiB that one thread does at 10-20 GB/s,
// a third of what the link takes.  The workers share a few helper threads that split such a copy by rows (AVIFGPU_COPY_THREADS,
// default 3 helpers beside the worker itself; 0 = the worker alone, round 2's behaviour).  Helpers are created on first use by the
// worker that needs them and inherit its CPU affinity -- the NUMA node of that worker's GPU.  Page-locked caller memory never gets here.
class CopyHelpers {
public:
    struct Piece { uint8_t* dst; const uint8_t* src; size_t dst_pitch, src_pitch, bytes, rows; std::atomic<int>* left; };
    void copy(uint8_t* dst, size_t dst_pitch, const uint8_t* src, size_t src_pitch, size_t bytes, size_t rows)
    {
        const int helpers = want();
        if (helpers == 0 || bytes * rows < ((size_t)2 << 20) || rows < 2) { copy_rows_here(dst, dst_pitch, src, src_pitch, bytes, rows); return; }
        ensure(helpers);
        const size_t parts = std::min<size_t>((size_t)helpers + 1, rows);
        std::atomic<int> left((int)parts - 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t k = 1; k < parts; ++k) {
                const size_t r0 = rows * k / parts, r1 = rows * (k + 1) / parts;
                queue_.push_back({ dst + r0 * dst_pitch, src + r0 * src_pitch, dst_pitch, src_pitch, bytes, r1 - r0, &left });
            }
        }
        cv_.notify_all();
        copy_rows_here(dst, dst_pitch, src, src_pitch, bytes, rows / parts);                 // the caller's own share: rows [0, rows/parts)
        // help with whatever is still queued (this copy's pieces or another worker's), then wait for the pieces others took
        for (;;) {
            Piece p;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (queue_.empty()) break;
                p = queue_.front(); queue_.pop_front();
            }
            run(p);
        }
        while (left.load(std::memory_order_acquire) > 0) std::this_thread::yield();
    }
    void shutdown()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        std::vector<std::thread> mine;
        { std::lock_guard<std::mutex> lk(mu_); mine.swap(threads_); }
        for (std::thread& t : mine) if (t.joinable()) t.join();
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = false;
    }
    int thread_count() { std::lock_guard<std::mutex> lk(mu_); return (int)threads_.size(); }
private:
    static int want() { return env_int("AVIFGPU_COPY_THREADS", 3, 0, 15); }
    static void run(const Piece& p)
    {
        copy_rows_here(p.dst, p.dst_pitch, p.src, p.src_pitch, p.bytes, p.rows);
        p.left->fetch_sub(1, std::memory_order_release);
    }
    void ensure(int n)
    {
        std::lock_guard<std::mutex> lk(mu_);
        while ((int)threads_.size() < n) threads_.emplace_back([this] { loop(); });
    }
    void loop()
    {
        for (;;) {
            Piece p;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
                if (queue_.empty()) return;                     // stop
                p = queue_.front(); queue_.pop_front();
            }
            run(p);
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Piece> queue_;
    std::vector<std::thread> threads_;
    bool stop_ = false;
};
// One pool per NUMA node (round 4; round 3 had one per process): helper threads are created by the first worker that needs them and
// inherit THAT worker's CPU affinity, so a single pool on a two-socket, eight-GPU node would have done the bounce copies of the other
// socket's GPUs across the socket link -- the traffic the worker pinning exists to avoid.  A worker registers its node once
// (t_copy_node: the NUMA node of its device, or 0 when the host does not say); a pool is only ever used by workers of its node.
// Never destroyed (like the contexts: their threads may outlive static destruction at process exit); contexts_shutdown() joins the helpers.
thread_local int t_copy_node = 0;
std::mutex g_helpers_mu;
std::map<int, CopyHelpers*>& helper_pools() { static std::map<int, CopyHelpers*>* m = new std::map<int, CopyHelpers*>; return *m; }
CopyHelpers& copy_helpers()
{
    std::lock_guard<std::mutex> lk(g_helpers_mu);
    CopyHelpers*& p = helper_pools()[t_copy_node];
    if (!p) p = new CopyHelpers;
    return *p;
}
void copy_helpers_shutdown()
{
    // The pools STAY in the map (round 6; round 5 dropped them from it and leaked the objects): a worker that still holds a
    // reference from copy_helpers() and calls copy() after this point re-spawns helper threads on a pool a later shutdown still finds and
    // joins, and a re-binding reuses its node's pool instead of making another.  shutdown() joins the threads; the thread-less object is a
    // few hundred bytes and is never destroyed (like the contexts: a reference may be on some worker's stack at process exit).
    std::vector<CopyHelpers*> pools;
    { std::lock_guard<std::mutex> lk(g_helpers_mu); for (auto& kv : helper_pools()) pools.push_back(kv.second); }
    for (CopyHelpers* p : pools) p->shutdown();
}
// "pools alive" (avifgpu_device_traffic): the pools of the CURRENT binding that have helper threads, not every pool the process ever made
int copy_helper_pool_count()
{
    std::lock_guard<std::mutex> lk(g_helpers_mu);
    int n = 0;
    for (auto& kv : helper_pools()) if (kv.second && kv.second->thread_count() > 0) ++n;
    return n;
}

void host_copy_rows(uint8_t* dst, size_t dst_pitch, const uint8_t* src, size_t src_pitch, size_t bytes, size_t rows)
{
    copy_helpers().copy(dst, dst_pitch, src, src_pitch, bytes, rows);
}

// ---- upload order ------------------------------------------------------------------------------------------------------
// Every slot has its own stream, so the copies of up to lanes x slots tiles are in flight at once -- and left alone they march in
// step: all uploads share the link and finish together, then all kernels, then all downloads, during which the upward direction of
// the link idles (25 % of a C4 save in the copy trace of round 3, profiles/r03/pcie_copy_trace_summary.txt).  The uploads of one
// DEVICE are therefore chained in issue order, at most AVIFGPU_UPLOAD_DEPTH (default 1) of them running at a time: tile t+DEPTH's
// upload waits for tile t's, and tile t's planes travel down while it runs -- the link is busy both ways, as a classic three-stage
// pipeline has it, with the per-slot streams still giving every tile its own kernel and download order.  0 = no chaining.
// The order is the order in which the caller QUEUED the tiles (a ticket per tile, taken in enqueue()), whichever lane a tile went
// to: the caller waits for staging slots in that same order, so tiles must also complete in it -- two lanes taking the chain in the
// order their threads happened to arrive cost a third of the throughput (27 ms instead of 16 for C4, profiles/r03/upload_order_sweep.jsonl).
constexpr int kMaxUploadDepth = 4;
struct UploadOrder {
    std::mutex mu;
    std::condition_variable turn;
    uint64_t next_ticket = 0;                              // handed out by enqueue()
    uint64_t serving = 0;                                  // the ticket whose uploads may be issued now
    hipEvent_t ring[kMaxUploadDepth] = {};                 // upload-done events of the last DEPTH tiles issued on this device
    int head = 0;
};
std::mutex g_upload_mu;
std::vector<std::unique_ptr<UploadOrder>> g_upload;       // by device index
int g_upload_depth = 1;

UploadOrder& upload_order(int device)
{
    std::lock_guard<std::mutex> lk(g_upload_mu);
    if ((int)g_upload.size() <= device) g_upload.resize((size_t)device + 1);
    if (!g_upload[device]) g_upload[device].reset(new UploadOrder);
    return *g_upload[device];
}

// Taken around the host-to-device copies of one tile: construct -> the stream waits for the upload DEPTH tiles back; finish() ->
// this tile's upload is recorded as the newest.  Holding the device's mutex in between keeps two lanes from interleaving.
// A started tile's place in the order.  EVERY started tile passes its turn on -- from issue() when it gets that far, from the
// destructor when it fails earlier -- or the tiles behind it would wait for ever.
class UploadTurn {
public:
    UploadTurn(int device, uint64_t ticket) : order_(upload_order(device)), ticket_(ticket) {}
    ~UploadTurn() { if (!passed_) { std::unique_lock<std::mutex> lk(order_.mu); wait(lk); pass(); } }
    // issue the host-to-device copies of the tile through `copies` (returns hipError_t) in this tile's turn
    template <typename F> hipError_t issue(Slot& sl, F&& copies)
    {
        std::unique_lock<std::mutex> lk(order_.mu);
        wait(lk);
        hipError_t e = hipSuccess;
        if (g_upload_depth > 0) {
            hipEvent_t prev = order_.ring[order_.head];     // the oldest of the last DEPTH uploads
            if (prev && prev != sl.up_done) e = hipStreamWaitEvent(sl.stream, prev, 0);
        }
        if (e == hipSuccess) e = copies();
        if (e == hipSuccess && g_upload_depth > 0) {
            e = hipEventRecord(sl.up_done, sl.stream);
            if (e == hipSuccess) { order_.ring[order_.head] = sl.up_done; order_.head = (order_.head + 1) % g_upload_depth; }
        }
        pass();
        return e;
    }
private:
    void wait(std::unique_lock<std::mutex>& lk) { order_.turn.wait(lk, [&] { return order_.serving == ticket_; }); }
    void pass() { ++order_.serving; passed_ = true; order_.turn.notify_all(); }
    UploadOrder& order_;
    const uint64_t ticket_;
    bool passed_ = false;
};

// a slot's events are about to be destroyed: nothing may wait on them afterwards
void forget_uploads_of(int device, const Slot& sl)
{
    if (!sl.up_done) return;
    UploadOrder& o = upload_order(device);
    std::lock_guard<std::mutex> lk(o.mu);
    for (hipEvent_t& e : o.ring) if (e == sl.up_done) e = nullptr;
}

// ---- worker side -------------------------------------------------------------------------------------------------------
// Issue the copies and the launch of one tile on its slot's stream.  Returns an OSErr; the message is in this thread's
// last_error().  After the first asynchronous operation any failure drains the stream before returning, so no DMA is left
// running into (or out of) caller memory behind an error return.
int start_write(Ctx& c, Job& j)
{
    UploadTurn turn(c.device, j.upload_seq);
    Slot& sl = c.slot[j.slot];
    const avifgpu_write_desc* d = &j.wd;
    WriteGeom g;
    int err = check_write(d, j.row0, j.nrows, g);
    if (err) return err;
    WriteParams p;
    if ((err = fill_write_params(d, j.row0, j.nrows, g, j.icc, p))) return err;

    const size_t row_bytes = (size_t)d->width * d->planes * (d->depth / 8);
    const size_t in_pitch = staging_pitch(row_bytes);
    if ((err = grow_device(&sl.d_in, &sl.d_in_cap, in_pitch * (size_t)j.nrows))) return err;
    size_t off[4] = {}, pitch[4] = {}, out_total = 0;
    int prow[4] = {}; int64_t pbytes[4] = {};
    for (int pl = 0; pl < 4; ++pl) {
        if (!write_plane_used(d, g, pl)) continue;
        write_plane_extent(d, g, pl, j.nrows, prow[pl], pbytes[pl]);
        pitch[pl] = staging_pitch((size_t)pbytes[pl]);
        off[pl] = out_total;
        out_total += align256(pitch[pl] * (size_t)prow[pl]);
    }
    if ((err = grow_device(&sl.d_out, &sl.d_out_cap, out_total))) return err;

    hipStream_t st = sl.stream;
    hipError_t e;
    const size_t src_span = (size_t)j.rows_in_stride * (size_t)(j.nrows - 1) + row_bytes;
    const bool own_tile = j.rows_in == sl.h_rows;            // the shim's pinned tile buffer
    const bool direct = own_tile || is_pinned(j.rows_in, src_span);
    if (!direct) {                                           // the CPU copy of pageable rows: before the device's upload order is taken
        if ((err = grow_pinned(&sl.h_rows, &sl.h_rows_cap, in_pitch * (size_t)j.nrows))) return err;
        host_copy_rows(static_cast<uint8_t*>(sl.h_rows), in_pitch, static_cast<const uint8_t*>(j.rows_in), (size_t)j.rows_in_stride, row_bytes, (size_t)j.nrows);
    }
    e = turn.issue(sl, [&] {
        if (direct) return copy_rows(sl.d_in, in_pitch, j.rows_in, (size_t)j.rows_in_stride, row_bytes, (size_t)j.nrows, hipMemcpyHostToDevice, st);
        return hipMemcpyAsync(sl.d_in, sl.h_rows, in_pitch * (size_t)j.nrows, hipMemcpyHostToDevice, st);
    });
    if (e != hipSuccess) { (void)hipStreamSynchronize(st); return hip_fail(e, "H2D copy", AVIFGPU_writErr); }
    c.tiles.fetch_add(1, std::memory_order_relaxed);
    c.bytes_h2d.fetch_add((uint64_t)row_bytes * (uint64_t)j.nrows, std::memory_order_relaxed);
    if (!direct) c.bytes_bounced.fetch_add((uint64_t)row_bytes * (uint64_t)j.nrows, std::memory_order_relaxed);

    p.src = static_cast<const uint8_t*>(sl.d_in); p.src_row_bytes = (int64_t)in_pitch;
    for (int pl = 0; pl < 4; ++pl) {
        p.dst[pl] = write_plane_used(d, g, pl) ? static_cast<uint8_t*>(sl.d_out) + off[pl] : nullptr;
        p.dst_stride[pl] = (int64_t)pitch[pl];
    }
    e = launch_write(p, d->depth, d->planes, g.dst16, d->output, g.xs, g.ys, hot_variant(), st, j.label);
    if (e != hipSuccess) { (void)hipStreamSynchronize(st); return hip_fail(e, "kernel launch", AVIFGPU_writErr); }

    // planes back: straight into page-locked destinations, through the pinned bounce buffer otherwise
    j.nbounce = 0;
    bool any_pageable = false;
    bool pinned_dst[4] = {};
    for (int pl = 0; pl < 4; ++pl) {
        if (!write_plane_used(d, g, pl) || prow[pl] == 0) continue;
        const size_t span = (size_t)j.planes_out_stride[pl] * (size_t)(prow[pl] - 1) + (size_t)pbytes[pl];
        pinned_dst[pl] = is_pinned(j.planes_out[pl], span);
        any_pageable = any_pageable || !pinned_dst[pl];
    }
    if (any_pageable && (err = grow_pinned(&sl.h_planes, &sl.h_planes_cap, out_total))) { (void)hipStreamSynchronize(st); return err; }
    for (int pl = 0; pl < 4; ++pl) {
        if (!write_plane_used(d, g, pl) || prow[pl] == 0) continue;
        const uint8_t* dev = static_cast<const uint8_t*>(sl.d_out) + off[pl];
        if (pinned_dst[pl]) {
            e = copy_rows(j.planes_out[pl], (size_t)j.planes_out_stride[pl], dev, pitch[pl], (size_t)pbytes[pl], (size_t)prow[pl], hipMemcpyDeviceToHost, st);
        } else {
            e = hipMemcpyAsync(static_cast<uint8_t*>(sl.h_planes) + off[pl], dev, pitch[pl] * (size_t)prow[pl], hipMemcpyDeviceToHost, st);
            j.bounce[j.nbounce++] = { static_cast<uint8_t*>(j.planes_out[pl]), j.planes_out_stride[pl], off[pl], pitch[pl], (size_t)pbytes[pl], prow[pl] };
        }
        if (e != hipSuccess) { (void)hipStreamSynchronize(st); return hip_fail(e, "D2H copy", AVIFGPU_writErr); }
        c.bytes_d2h.fetch_add((uint64_t)pbytes[pl] * (uint64_t)prow[pl], std::memory_order_relaxed);
        if (!pinned_dst[pl]) c.bytes_bounced.fetch_add((uint64_t)pbytes[pl] * (uint64_t)prow[pl], std::memory_order_relaxed);
    }
    if ((e = hipEventRecord(sl.done, st)) != hipSuccess) { (void)hipStreamSynchronize(st); return hip_fail(e, "event record", AVIFGPU_writErr); }
    return 0;
}

int start_read(Ctx& c, Job& j)
{
    UploadTurn turn(c.device, j.upload_seq);
    Slot& sl = c.slot[j.slot];
    const avifgpu_read_desc* d = &j.rd;
    ReadGeom g;
    int err = check_read(d, j.row0, j.nrows, g);
    if (err) return err;
    ReadParams p;
    if ((err = fill_read_params(d, j.nrows, g, p))) return err;

    size_t off[4] = {}, pitch[4] = {}, in_total = 0;
    int prow[4] = {}; int64_t pbytes[4] = {};
    for (int pl = 0; pl < 4; ++pl) {
        if (!read_plane_used(d, g, pl)) continue;
        read_plane_extent(d, g, pl, j.nrows, prow[pl], pbytes[pl]);
        pitch[pl] = staging_pitch((size_t)pbytes[pl]);
        off[pl] = in_total;
        in_total += align256(pitch[pl] * (size_t)prow[pl]);
    }
    if ((err = grow_device(&sl.d_in, &sl.d_in_cap, in_total))) return err;
    const size_t row_bytes = (size_t)d->width * g.nch * (d->depth / 8);
    const size_t out_pitch = staging_pitch(row_bytes);
    if ((err = grow_device(&sl.d_out, &sl.d_out_cap, out_pitch * (size_t)j.nrows))) return err;

    hipStream_t st = sl.stream;
    hipError_t e = hipSuccess;
    bool pinned_src[4] = {}, any_pageable = false;
    for (int pl = 0; pl < 4; ++pl) {
        if (!read_plane_used(d, g, pl) || prow[pl] == 0) continue;
        const size_t span = (size_t)j.planes_in_stride[pl] * (size_t)(prow[pl] - 1) + (size_t)pbytes[pl];
        pinned_src[pl] = is_pinned(j.planes_in[pl], span);
        any_pageable = any_pageable || !pinned_src[pl];
    }
    if (any_pageable && (err = grow_pinned(&sl.h_planes, &sl.h_planes_cap, in_total))) return err;
    for (int pl = 0; pl < 4; ++pl) {                         // the CPU copies of pageable planes: before the device's upload order is taken
        if (!read_plane_used(d, g, pl) || prow[pl] == 0 || pinned_src[pl]) continue;
        host_copy_rows(static_cast<uint8_t*>(sl.h_planes) + off[pl], pitch[pl], static_cast<const uint8_t*>(j.planes_in[pl]),
                       (size_t)j.planes_in_stride[pl], (size_t)pbytes[pl], (size_t)prow[pl]);
    }
    e = turn.issue(sl, [&] {
        hipError_t ce = hipSuccess;
        for (int pl = 0; pl < 4 && ce == hipSuccess; ++pl) {
            if (!read_plane_used(d, g, pl)) continue;
            uint8_t* dev = static_cast<uint8_t*>(sl.d_in) + off[pl];
            p.src[pl] = dev; p.src_stride[pl] = (int64_t)pitch[pl];
            if (prow[pl] == 0) continue;
            if (pinned_src[pl]) ce = copy_rows(dev, pitch[pl], j.planes_in[pl], (size_t)j.planes_in_stride[pl], (size_t)pbytes[pl], (size_t)prow[pl], hipMemcpyHostToDevice, st);
            else ce = hipMemcpyAsync(dev, static_cast<uint8_t*>(sl.h_planes) + off[pl], pitch[pl] * (size_t)prow[pl], hipMemcpyHostToDevice, st);
        }
        return ce;
    });
    if (e != hipSuccess) { (void)hipStreamSynchronize(st); return hip_fail(e, "H2D copy", AVIFGPU_readErr); }
    c.tiles.fetch_add(1, std::memory_order_relaxed);
    for (int pl = 0; pl < 4; ++pl) {
        if (!read_plane_used(d, g, pl) || prow[pl] == 0) continue;
        c.bytes_h2d.fetch_add((uint64_t)pbytes[pl] * (uint64_t)prow[pl], std::memory_order_relaxed);
        if (!pinned_src[pl]) c.bytes_bounced.fetch_add((uint64_t)pbytes[pl] * (uint64_t)prow[pl], std::memory_order_relaxed);
    }
    p.dst = static_cast<uint8_t*>(sl.d_out); p.dst_row_bytes = (int64_t)out_pitch;
    e = launch_read(p, d->colorspace, d->depth, g.alpha, g.xs, g.ys, st, j.label);
    if (e != hipSuccess) { (void)hipStreamSynchronize(st); return hip_fail(e, "kernel launch", AVIFGPU_readErr); }

    j.nbounce = 0;
    const size_t dst_span = (size_t)j.rows_out_stride * (size_t)(j.nrows - 1) + row_bytes;
    if (j.rows_out == sl.h_rows || is_pinned(j.rows_out, dst_span)) {
        e = copy_rows(j.rows_out, (size_t)j.rows_out_stride, sl.d_out, out_pitch, row_bytes, (size_t)j.nrows, hipMemcpyDeviceToHost, st);
    } else {
        if ((err = grow_pinned(&sl.h_rows, &sl.h_rows_cap, out_pitch * (size_t)j.nrows))) { (void)hipStreamSynchronize(st); return err; }
        e = hipMemcpyAsync(sl.h_rows, sl.d_out, out_pitch * (size_t)j.nrows, hipMemcpyDeviceToHost, st);
        j.bounce[j.nbounce++] = { static_cast<uint8_t*>(j.rows_out), j.rows_out_stride, 0, out_pitch, row_bytes, j.nrows };
    }
    if (e != hipSuccess) { (void)hipStreamSynchronize(st); return hip_fail(e, "D2H copy", AVIFGPU_readErr); }
    c.bytes_d2h.fetch_add((uint64_t)row_bytes * (uint64_t)j.nrows, std::memory_order_relaxed);
    if (j.nbounce) c.bytes_bounced.fetch_add((uint64_t)row_bytes * (uint64_t)j.nrows, std::memory_order_relaxed);
    if ((e = hipEventRecord(sl.done, st)) != hipSuccess) { (void)hipStreamSynchronize(st); return hip_fail(e, "event record", AVIFGPU_readErr); }
    return 0;
}

// Wait for a started tile and land what went through the bounce buffer.
int finish(Ctx& c, Job& j)
{
    Slot& sl = c.slot[j.slot];
    const hipError_t e = hipEventSynchronize(sl.done);
    if (e != hipSuccess) return hip_fail(e, "event synchronize", j.is_write ? AVIFGPU_writErr : AVIFGPU_readErr);
    const uint8_t* hb = static_cast<const uint8_t*>(j.is_write ? sl.h_planes : sl.h_rows);
    for (int i = 0; i < j.nbounce; ++i) {
        const Job::Bounce& b = j.bounce[i];
        host_copy_rows(b.dst, (size_t)b.dst_stride, hb + b.off, b.pitch, b.bytes, (size_t)b.rows);
    }
    return 0;
}

void record_error(Ctx& c, int code)
{
    std::lock_guard<std::mutex> lk(c.mu);
    if (!c.err) { c.err = code; snprintf(c.err_msg, sizeof(c.err_msg), "%s", last_error()); }
}

void release_slot(Ctx& c, const Job& j)
{
    {
        std::lock_guard<std::mutex> lk(c.mu);
        c.slot[j.slot].busy = false;
        if (j.label[0]) snprintf(c.last_label, sizeof(c.last_label), "%s", j.label);
    }
    c.cv_done.notify_all();
}

void worker_main(Ctx* cp)
{
    Ctx& c = *cp;
    // run on the CPUs of the device's NUMA node (when the host says which they are): the bounce memcpy of pageable caller memory,
    // the first touch of every pinned staging buffer and the driver's submission path then stay on the socket the GPU hangs off
    for (const DeviceTopo& t : g_topo) {
        cpu_set_t set;
        if (t.device == c.device && env_int("AVIFGPU_PIN_WORKERS", 1, 0, 1) && parse_cpulist(t.cpulist, set))
            c.pinned_to_node = pthread_setaffinity_np(pthread_self(), sizeof(set), &set) == 0;
        if (t.device == c.device) t_copy_node = t.numa_node >= 0 ? t.numa_node : 0;      // this worker's bounce copies use its own node's helper pool
    }
    hipError_t e = hipSetDevice(c.device);
    for (int s = 0; e == hipSuccess && s < c.nslots; ++s) {
        e = hipStreamCreateWithFlags(&c.slot[s].stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c.slot[s].done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c.slot[s].up_done, hipEventDisableTiming);
    }
    {
        std::lock_guard<std::mutex> lk(c.mu);
        if (e != hipSuccess) { c.init_err = AVIFGPU_memFullErr; snprintf(c.init_msg, sizeof(c.init_msg), "device %d: %s", c.device, hipGetErrorString(e)); }
        c.started = true;
    }
    c.cv_done.notify_all();
    if (e != hipSuccess) return;

    // Issue every queued tile at once (copies and launches are asynchronous: ~10 us per tile) and land finished tiles in order.
    // The worker never sleeps inside the runtime while tiles may arrive: it polls the oldest tile's event and naps on the
    // work condition variable in between, so a newly queued tile is issued within microseconds and the copy queues stay fed
    // (blocking in hipEventSynchronize here left the H2D queue dry for ~1 ms per tile: 38 -> 4x GB/s, profiles/r02/pcie_pipeline.md).
    std::deque<Job> inflight;
    for (;;) {
        Job j;
        bool have = false, alloc = false;
        {
            std::unique_lock<std::mutex> lk(c.mu);
            if (inflight.empty()) c.cv_work.wait(lk, [&] { return c.stop || !c.queue.empty() || c.alloc_pending; });
            if (c.alloc_pending) alloc = true;
            else if (!c.queue.empty()) { j = c.queue.front(); c.queue.pop_front(); have = true; }
            else if (c.stop && inflight.empty()) break;
        }
        if (alloc) {
            // the slot is idle (its producer is the one waiting for this buffer)
            Slot& sl = c.slot[c.alloc_slot];
            const int rc = grow_pinned(&sl.h_rows, &sl.h_rows_cap, c.alloc_bytes);
            { std::lock_guard<std::mutex> lk(c.mu); c.alloc_rc = rc; c.alloc_pending = false; }
            c.cv_done.notify_all();
            continue;
        }
        if (have) {
            // a slot never holds two tiles: the producer waited for it before queueing
            if (g_trace) j.t_start = now_us();
            const int rc = j.is_write ? start_write(c, j) : start_read(c, j);
            if (g_trace) j.t_issued = now_us();
            if (rc) { record_error(c, rc); release_slot(c, j); }
            else inflight.push_back(j);
            continue;
        }
        const hipError_t q = hipEventQuery(c.slot[inflight.front().slot].done);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            // nap until a tile (or a buffer request) arrives.  `stop` is NOT part of the predicate: a shutdown with tiles in flight
            // would turn this into a busy spin on hipEventQuery; it is looked at when the last tile has landed
            std::unique_lock<std::mutex> lk(c.mu);
            c.cv_work.wait_for(lk, std::chrono::microseconds(30), [&] { return !c.queue.empty() || c.alloc_pending; });
            continue;
        }
        Job done = inflight.front();
        inflight.pop_front();
        const double t_wait = g_trace ? now_us() : 0;
        const int rc = finish(c, done);                  // event already complete (or failed): returns at once, then the bounce copy
        if (g_trace)
            fprintf(stderr, "[avifgpu trace] dev %d ctx %p slot %d rows [%d,+%d) queued %.0f start %.0f issued %.0f ready %.0f done %.0f us\n",
                    c.device, (void*)&c, done.slot, done.row0, done.nrows, done.t_queued, done.t_start, done.t_issued, t_wait, now_us());
        if (rc) record_error(c, rc);
        release_slot(c, done);
    }
    for (int s = 0; s < c.nslots; ++s) {
        Slot& sl = c.slot[s];
        if (sl.stream) (void)hipStreamSynchronize(sl.stream);
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.h_rows) (void)hipHostFree(sl.h_rows);
        if (sl.h_planes) (void)hipHostFree(sl.h_planes);
        forget_uploads_of(c.device, sl);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.up_done) (void)hipEventDestroy(sl.up_done);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
        sl = Slot();
    }
}

Ctx* ctx_at(int i)
{
    if (!g_ctxs || i < 0 || i >= (int)g_ctxs->size()) return nullptr;
    return (*g_ctxs)[i];
}

int enqueue(int ctx, Job& j)
{
    Ctx* c = ctx_at(ctx);
    if (!c || j.slot < 0 || j.slot >= c->nslots) return fail(AVIFGPU_formatBadParameters, "bad context / slot (%d, %d)", ctx, j.slot);
    {
        std::lock_guard<std::mutex> lk(c->mu);
        if (c->slot[j.slot].busy) return fail(AVIFGPU_formatBadParameters, "staging slot (%d, %d) is still busy", ctx, j.slot);
        c->slot[j.slot].busy = true;
        if (g_trace) j.t_queued = now_us();
        {
            UploadOrder& o = upload_order(c->device);
            std::lock_guard<std::mutex> l2(o.mu);
            j.upload_seq = o.next_ticket++;
        }
        c->queue.push_back(j);
    }
    c->cv_work.notify_one();
    return 0;
}

} // namespace

// =========================================================================================================================
int row_cut(int height, int world, int k, bool even)
{
    if (k >= world) return height;
    int b = (int)(((long long)height * k) / world);
    if (even) b -= b & 1;
    return b;
}

int context_count() { return g_ctxs ? (int)g_ctxs->size() : 0; }
int bound_device_count() { return g_ctxs ? (int)g_bound.size() : 0; }
int context_device(int ctx) { Ctx* c = ctx_at(ctx); return c ? c->device : -1; }
int slots_per_context() { Ctx* c = ctx_at(0); return c ? c->nslots : 0; }

void contexts_shutdown()
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (!g_ctxs) return;
    for (Ctx* c : *g_ctxs) {
        { std::lock_guard<std::mutex> l2(c->mu); c->stop = true; }
        c->cv_work.notify_all();
        if (c->worker.joinable()) c->worker.join();
        delete c;
    }
    delete g_ctxs;
    g_ctxs = nullptr;
    g_bound.clear();
    g_topo.clear();
    copy_helpers_shutdown();                               // the next binding's workers make their own (affinity of THEIR node)
    {
        // Upload-order tickets start again with the next binding.  Every ticketed job has been through its UploadTurn by now (the
        // workers drain their queues before they exit), so next_ticket == serving; resetting both makes that an invariant of a
        // fresh binding instead of something a dropped job could break for the rest of the process.
        std::lock_guard<std::mutex> l3(g_upload_mu);
        for (auto& o : g_upload)
            if (o) { std::lock_guard<std::mutex> l4(o->mu); o->next_ticket = o->serving = 0; o->head = 0; for (hipEvent_t& ev : o->ring) ev = nullptr; }   // (the ring's events died with their slots)
    }
}

int contexts_init(const int32_t* devices, int count)
{
    if (!devices || count < 1 || count > 64) return fail(AVIFGPU_formatBadParameters, "avifgpu_init_devices: bad device list");
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(AVIFGPU_formatBadParameters, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    for (int i = 0; i < count; ++i)
        if (devices[i] < 0 || devices[i] >= n) return fail(AVIFGPU_formatBadParameters, "device %d out of range [0,%d)", devices[i], n);
    // Lanes: contexts per bound device.  One worker keeps one copy queue per direction busy; a second lane on the same device
    // overlaps its submissions -- and its CPU copies of pageable caller memory -- with the first one's.
    const int lanes = env_int("AVIFGPU_LANES", kDefaultLanes, 1, 4);
    // Slots: tiles in flight per lane, each with a stream of its own.  Default lanes x slots = 4 streams per device: the runtime
    // multiplexes a process's streams onto 4 hardware queues per device, and the barrier that orders a tile's upload behind the
    // previous one's then also holds up whatever OTHER stream shares its queue -- 2 x 3, 1 x 6 or 2 x 4 streams measured 19-27 ms for
    // the C4 job against 15.5-16 ms for 2 x 2 or 1 x 4 (profiles/r03/upload_order_sweep.jsonl).
    const int nslots = env_int("AVIFGPU_SLOTS", std::max(2, 4 / lanes), 2, kMaxSlots);
    const bool trace = env_int("AVIFGPU_TRACE", 0, 0, 1) != 0;
    const int upload_depth = env_int("AVIFGPU_UPLOAD_DEPTH", 1, 0, kMaxUploadDepth);
    {
        // same binding AND same knobs: nothing to do (a changed AVIFGPU_LANES / AVIFGPU_SLOTS / AVIFGPU_TRACE / AVIFGPU_UPLOAD_DEPTH re-binds)
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        if (g_ctxs && g_bound == std::vector<int>(devices, devices + count) && (int)g_ctxs->size() == count * lanes &&
            (*g_ctxs)[0]->nslots == nslots && g_trace == trace && g_upload_depth == upload_depth) return 0;
    }
    contexts_shutdown();                                   // re-binding releases every stream, event and buffer of the old devices
    release_device_caches();                               // ... and their table caches (never reused on another device)
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto* v = new std::vector<Ctx*>();
    g_trace = trace;
    g_upload_depth = upload_depth;                         // no tile is in flight here: the old contexts are gone
    // where each bound device sits: PCI bus id -> NUMA node -> that node's CPUs (sysfs); the workers pin themselves to them
    g_topo.clear();
    for (int i = 0; i < count; ++i) {
        bool seen = false;
        for (const DeviceTopo& t : g_topo) seen = seen || t.device == devices[i];
        if (seen) continue;
        DeviceTopo t;
        t.device = devices[i];
        char bdf[64] = "";
        if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), devices[i]) == hipSuccess) {
            t.bdf = bdf;
            (void)topology_probe(getenv("AVIFGPU_SYSFS_ROOT"), bdf, t.numa_node, t.cpulist);
        } else (void)hipGetLastError();
        g_topo.push_back(t);
    }
    int rc = 0;
    for (int i = 0; i < count * lanes && !rc; ++i) {
        Ctx* c = new Ctx();
        c->device = devices[i / lanes];
        c->nslots = nslots;
        c->worker = std::thread(worker_main, c);
        v->push_back(c);
        std::unique_lock<std::mutex> l2(c->mu);
        c->cv_done.wait(l2, [&] { return c->started; });
        if (c->init_err) rc = fail(c->init_err, "%s", c->init_msg);
    }
    g_ctxs = v;
    g_bound.assign(devices, devices + count);
    for (DeviceTopo& t : g_topo) {
        t.workers = 0; t.pinned = true;
        for (Ctx* c : *v) if (c->device == t.device) { ++t.workers; t.pinned = t.pinned && c->pinned_to_node; }
        t.pinned = t.pinned && t.workers > 0;
    }
    if (rc) {
        // unwind outside the lock-free helpers: same steps as contexts_shutdown
        for (Ctx* c : *g_ctxs) {
            { std::lock_guard<std::mutex> l2(c->mu); c->stop = true; }
            c->cv_work.notify_all();
            if (c->worker.joinable()) c->worker.join();
            delete c;
        }
        delete g_ctxs; g_ctxs = nullptr; g_bound.clear(); g_topo.clear();
    }
    return rc;
}

int device_topology(int index, avifgpu_device_info* out)
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (!out || index < 0 || index >= (int)g_topo.size()) return AVIFGPU_formatBadParameters;
    const DeviceTopo& t = g_topo[index];
    std::memset(out, 0, sizeof(*out));
    out->device = t.device; out->numa_node = t.numa_node; out->workers = t.workers; out->workers_pinned = t.pinned ? 1 : 0;
    snprintf(out->pci_bus_id, sizeof(out->pci_bus_id), "%s", t.bdf.c_str());
    snprintf(out->cpulist, sizeof(out->cpulist), "%s", t.cpulist.c_str());
    return 0;
}

// The placement a binding of these PCI devices would get, without touching a device: NUMA node and CPU list per device (sysfs),
// workers per device (AVIFGPU_LANES), and -- the return value -- how many distinct helper pools / pinned-allocation domains that
// makes (one per NUMA node seen; devices whose node is unknown share pool 0).  contexts_init() does exactly this per bound device.
int topology_plan(const char* sysfs_root, const char* const* bdfs, int count, avifgpu_device_info* out)
{
    if (!bdfs || count < 1 || count > 64 || !out) return AVIFGPU_formatBadParameters;
    const int lanes = env_int("AVIFGPU_LANES", kDefaultLanes, 1, 4);
    const bool pin = env_int("AVIFGPU_PIN_WORKERS", 1, 0, 1) != 0;
    std::vector<int> nodes;
    for (int i = 0; i < count; ++i) {
        if (!bdfs[i]) return AVIFGPU_formatBadParameters;
        int node = -1; std::string list;
        const int rc = topology_probe(sysfs_root, bdfs[i], node, list);
        if (rc < 0) return rc;
        std::memset(&out[i], 0, sizeof(out[i]));
        out[i].device = i; out[i].numa_node = node; out[i].workers = lanes;
        cpu_set_t set;
        out[i].workers_pinned = pin && parse_cpulist(list, set) ? 1 : 0;
        snprintf(out[i].pci_bus_id, sizeof(out[i].pci_bus_id), "%s", bdfs[i]);
        snprintf(out[i].cpulist, sizeof(out[i].cpulist), "%s", list.c_str());
        const int pool = node >= 0 ? node : 0;
        if (std::find(nodes.begin(), nodes.end(), pool) == nodes.end()) nodes.push_back(pool);
    }
    return (int)nodes.size();
}

int device_traffic(int index, avifgpu_device_traffic* out, bool reset)
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (index < 0 || index >= (int)g_topo.size() || !g_ctxs) return AVIFGPU_formatBadParameters;
    const int dev = g_topo[index].device;
    avifgpu_device_traffic t;
    std::memset(&t, 0, sizeof(t));
    t.device = dev;
    for (Ctx* c : *g_ctxs) {
        if (c->device != dev) continue;
        // read-and-zero in ONE atomic step when resetting: a tile issued between a load and a separate store would be lost
        if (reset) { t.tiles += c->tiles.exchange(0); t.bytes_h2d += c->bytes_h2d.exchange(0); t.bytes_d2h += c->bytes_d2h.exchange(0); t.bytes_bounced += c->bytes_bounced.exchange(0); }
        else { t.tiles += c->tiles.load(); t.bytes_h2d += c->bytes_h2d.load(); t.bytes_d2h += c->bytes_d2h.load(); t.bytes_bounced += c->bytes_bounced.load(); }
    }
    t.copy_helper_pools = copy_helper_pool_count();
    if (out) *out = t;
    return 0;
}

int topology_probe_c(const char* sysfs_root, const char* bdf, int32_t* numa_node, char* cpulist, int32_t cpulist_len)
{
    int node = -1; std::string list;
    const int rc = topology_probe(sysfs_root, bdf, node, list);
    if (numa_node) *numa_node = node;
    if (cpulist && cpulist_len > 0) snprintf(cpulist, (size_t)cpulist_len, "%s", list.c_str());
    if (rc) return rc;
    cpu_set_t set;
    return list.empty() ? 0 : (parse_cpulist(list, set) ? CPU_COUNT(&set) : AVIFGPU_readErr);
}

void* tile_buffer(int ctx, int slot, size_t bytes)
{
    Ctx* c = ctx_at(ctx);
    if (!c || slot < 0 || slot >= c->nslots) return nullptr;
    Slot& sl = c->slot[slot];                              // the slot is idle (the producer owns it between wait_slot and enqueue)
    const size_t need = bytes ? bytes : 64;
    if (need <= sl.h_rows_cap) return sl.h_rows;
    // (re)allocated by the context's worker: first touched and page-locked on the NUMA node of the device it feeds
    std::unique_lock<std::mutex> lk(c->mu);
    c->alloc_slot = slot; c->alloc_bytes = need; c->alloc_rc = 0; c->alloc_pending = true;
    c->cv_work.notify_one();
    c->cv_done.wait(lk, [&] { return !c->alloc_pending; });
    if (c->alloc_rc) { set_error("hipHostMalloc of a pinned tile buffer failed"); return nullptr; }
    return sl.h_rows;
}

int write_tile_enqueue(int ctx, int slot, const avifgpu_write_desc* d, int row0, int nrows, const void* src, int64_t src_row_bytes,
                       void* const dst[4], const int64_t dst_stride[4], const IccArgs& icc)
{
    WriteGeom g;
    int err = check_write(d, row0, nrows, g);              // early, on the calling thread: parameter errors come back synchronously
    if (err) return err;
    if ((err = check_write_buffers(d, g, nrows, src, src_row_bytes, dst, dst_stride))) return err;
    if (nrows == 0) return 0;
    Job j;
    j.is_write = true; j.slot = slot; j.wd = *d; j.row0 = row0; j.nrows = nrows;
    j.rows_in = src; j.rows_in_stride = src_row_bytes;
    for (int pl = 0; pl < 4; ++pl) { j.planes_out[pl] = dst[pl]; j.planes_out_stride[pl] = dst_stride[pl]; }
    j.icc = icc;
    return enqueue(ctx, j);
}

int read_tile_enqueue(int ctx, int slot, const avifgpu_read_desc* d, int row0, int nrows, const void* const src[4],
                      const int64_t src_stride[4], void* dst, int64_t dst_row_bytes)
{
    ReadGeom g;
    int err = check_read(d, row0, nrows, g);
    if (err) return err;
    if ((err = check_read_buffers(d, g, nrows, src, src_stride, dst, dst_row_bytes))) return err;
    if (nrows == 0) return 0;
    Job j;
    j.is_write = false; j.slot = slot; j.rd = *d; j.row0 = row0; j.nrows = nrows;
    for (int pl = 0; pl < 4; ++pl) { j.planes_in[pl] = src[pl]; j.planes_in_stride[pl] = src_stride[pl]; }
    j.rows_out = dst; j.rows_out_stride = dst_row_bytes;
    return enqueue(ctx, j);
}

int wait_slot(int ctx, int slot)
{
    Ctx* c = ctx_at(ctx);
    if (!c || slot < 0 || slot >= c->nslots) return fail(AVIFGPU_formatBadParameters, "bad context / slot (%d, %d)", ctx, slot);
    std::unique_lock<std::mutex> lk(c->mu);
    c->cv_done.wait(lk, [&] { return !c->slot[slot].busy; });
    if (c->last_label[0]) set_last_kernel(c->last_label);
    if (c->err) { set_error(c->err_msg); return c->err; }  // sticky until wait_all: the caller stops queueing and drains
    return 0;
}

int wait_all()
{
    int first = 0;
    char msg[512] = "";
    for (int i = 0; i < context_count(); ++i) {
        Ctx* c = ctx_at(i);
        std::unique_lock<std::mutex> lk(c->mu);
        c->cv_done.wait(lk, [&] {
            if (!c->queue.empty()) return false;
            for (int s = 0; s < c->nslots; ++s) if (c->slot[s].busy) return false;
            return true;
        });
        if (c->last_label[0]) set_last_kernel(c->last_label);
        if (c->err && !first) { first = c->err; snprintf(msg, sizeof(msg), "%s", c->err_msg); }
        c->err = 0; c->err_msg[0] = 0;
    }
    if (first) set_error(msg);
    return first;
}

// ---- whole-range host conversions ---------------------------------------------------------------------------------------
namespace {

// rows per sub-tile, even, at least 2.  With the uploads in order a tile only has to be long enough to hide its own launch and the
// ~27 us between two uploads, and short enough that the job still has a dozen tiles per worker to pipeline: a twelfth of the worker's
// share, between 4 and 32 MiB of the rows side (C4 from page-locked memory: 17.0 ms with 8 MiB tiles, 16.0 with 16, 15.5 with 32,
// profiles/r03/upload_order_sweep.jsonl).  AVIFGPU_CHUNK_MB fixes it.
int chunk_rows_for(size_t bytes_per_row, int rows_per_worker)
{
    size_t budget;
    if (getenv("AVIFGPU_CHUNK_MB")) budget = (size_t)env_int("AVIFGPU_CHUNK_MB", 16, 1, 4096) << 20;
    else budget = std::min<size_t>(std::max<size_t>(bytes_per_row * (size_t)std::max(rows_per_worker, 1) / 12, (size_t)4 << 20), (size_t)32 << 20);
    size_t rows = budget / std::max<size_t>(bytes_per_row, 1);
    rows = std::max<size_t>(rows & ~(size_t)1, 2);
    return (int)std::min<size_t>(rows, 1u << 30);
}

} // namespace

// Host-pointer conversions share the contexts' slots and their sticky error state: ONE at a time per process.  Photoshop calls a
// plug-in serially (AvifFormat.cpp:104-199); any other caller's threads are serialised here instead of corrupting each other.
std::recursive_mutex g_host_call_mu;
void host_call_lock() { g_host_call_mu.lock(); }
void host_call_unlock() { g_host_call_mu.unlock(); }

int write_rows_host(const avifgpu_write_desc* d, int row0, int nrows, const void* src, int64_t src_row_bytes,
                    void* const dst[4], const int64_t dst_stride[4], const IccArgs& icc)
{
    std::lock_guard<std::recursive_mutex> serial(g_host_call_mu);
    const int n = context_count();
    if (n == 0) return fail(AVIFGPU_formatBadParameters, "avifgpu_init has not succeeded: no HIP device bound (no CPU fallback)");
    WriteGeom g;
    int err = check_write(d, row0, nrows, g);
    if (err) return err;
    const int nslots = slots_per_context();
    const size_t row_bytes = (size_t)d->width * d->planes * (d->depth / 8);
    const int chunk = chunk_rows_for(row_bytes, (nrows + n - 1) / n);
    const bool ycc = d->output == AVIFGPU_OUT_YCBCR;
    std::vector<int> next(n), end(n);
    for (int c = 0; c < n; ++c) { next[c] = row0 + row_cut(nrows, n, c, true); end[c] = row0 + row_cut(nrows, n, c + 1, true); }
    bool failed = false;
    for (int step = 0; !failed; ++step) {
        bool any = false;
        for (int c = 0; c < n && !failed; ++c) {
            if (next[c] >= end[c]) continue;
            any = true;
            const int r0 = next[c], nr = std::min(chunk, end[c] - r0);
            next[c] = r0 + nr;
            const int slot = step % nslots;
            if ((err = wait_slot(c, slot))) { failed = true; break; }
            void* tdst[4];
            for (int pl = 0; pl < 4; ++pl) {
                const bool chroma = ycc && (pl == 1 || pl == 2);
                const int64_t r = chroma ? ((r0 - row0) >> g.ys) : (r0 - row0);
                tdst[pl] = dst[pl] ? static_cast<uint8_t*>(dst[pl]) + r * dst_stride[pl] : nullptr;
            }
            const uint8_t* tsrc = static_cast<const uint8_t*>(src) + (int64_t)(r0 - row0) * src_row_bytes;
            if ((err = write_tile_enqueue(c, slot, d, r0, nr, tsrc, src_row_bytes, tdst, dst_stride, icc))) failed = true;
        }
        if (!any) break;
    }
    const int werr = wait_all();                          // drains everything, also after a failure
    return err ? err : werr;
}

int read_rows_host(const avifgpu_read_desc* d, int row0, int nrows, const void* const src[4], const int64_t src_stride[4],
                   void* dst, int64_t dst_row_bytes)
{
    std::lock_guard<std::recursive_mutex> serial(g_host_call_mu);
    const int n = context_count();
    if (n == 0) return fail(AVIFGPU_formatBadParameters, "avifgpu_init has not succeeded: no HIP device bound (no CPU fallback)");
    ReadGeom g;
    int err = check_read(d, row0, nrows, g);
    if (err) return err;
    const int nslots = slots_per_context();
    const size_t row_bytes = (size_t)d->width * g.nch * (d->depth / 8);
    const int chunk = chunk_rows_for(row_bytes, (nrows + n - 1) / n);
    const bool ycc = d->colorspace == AVIFGPU_COLORSPACE_YCBCR;
    std::vector<int> next(n), end(n);
    for (int c = 0; c < n; ++c) { next[c] = row0 + row_cut(nrows, n, c, true); end[c] = row0 + row_cut(nrows, n, c + 1, true); }
    bool failed = false;
    for (int step = 0; !failed; ++step) {
        bool any = false;
        for (int c = 0; c < n && !failed; ++c) {
            if (next[c] >= end[c]) continue;
            any = true;
            const int r0 = next[c], nr = std::min(chunk, end[c] - r0);
            next[c] = r0 + nr;
            const int slot = step % nslots;
            if ((err = wait_slot(c, slot))) { failed = true; break; }
            const void* tsrc[4];
            for (int pl = 0; pl < 4; ++pl) {
                const bool chroma = ycc && (pl == 1 || pl == 2);
                const int64_t r = chroma ? ((r0 - row0) >> g.ys) : (r0 - row0);
                tsrc[pl] = src[pl] ? static_cast<const uint8_t*>(src[pl]) + r * src_stride[pl] : nullptr;
            }
            uint8_t* tdst = static_cast<uint8_t*>(dst) + (int64_t)(r0 - row0) * dst_row_bytes;
            if ((err = read_tile_enqueue(c, slot, d, r0, nr, tsrc, src_stride, tdst, dst_row_bytes))) failed = true;
        }
        if (!any) break;
    }
    const int werr = wait_all();
    return err ? err : werr;
}

} // namespace avifgpu
