// group.hip -- hipr_group_*: one frame rendered by several GPUs of a node from ONE process (include/hiprenderer_c.h, "Device groups").
//
// The path shards by independent units: pixel tiles of 8 x 8, dealt round-robin (tile % N == member), scene and tables replicated,
// every member keeps the f64 accumulation of its own tiles for the whole run -- no data-path collective per sample.
// A group holds one HiprContext per member device and drives each from its own host thread (the wavefront loop of a pass waits on
// per-bounce queue sizes, so a thread per device keeps the devices independent): member 0 on the calling thread, the others on worker
// threads that live as long as the group (a render() on 8 GPUs is a few milliseconds: no thread is created on the frame path). The only exchange is once per displayed frame: the
// members' compact half4 tiles are gathered on member 0's device and k_scatter_tiles assembles the frame there.
//   gather: RCCL point-to-point (ncclCommInitAll in this process; member 0 posts one ncclRecv per peer inside a group call, every
//           peer one ncclSend) over xGMI; RCCL is loaded with dlopen so that the library has no link-time dependency on it. When
//           it cannot be loaded, when two members share a device (tests on a one-GPU box) or with HIPR_GROUP_GATHER=copy, the
//           same bytes move by peer-to-peer hipMemcpyAsync (device to device over xGMI as well).
//   watchdog (round 5): no member ever blocks in the exchange. Every thread brackets its RCCL calls with ncclGroupStart / ncclGroupEnd (the discipline RCCL documents for
//           several communicators in one process) and then POLLS its stream (hipStreamQuery) against a deadline (HIPR_GROUP_GATHER_TIMEOUT_MS, 10 s by default) instead
//           of waiting on it. An exchange that does not finish -- a peer that never posted its half -- ends the call with HIPR_ERROR_TIMEOUT naming the member; the
//           communicators are aborted (ncclCommAbort), the members get fresh streams and the group gathers with peer-to-peer copies for the rest of its life. Nothing
//           is re-executed and no process is replaced. tests/test_gpu_coverage.py::test_device_group_watchdog_ends_a_stalled_exchange injects the stall.
// The reference renders on one device (OR/Renderer.cpp:289-291); there is no reference behaviour here beyond "the same image", which
// holds bit for bit: the RNG is a pure function of (pixel, accumulation, bounce) and every member walks the same BVH.
#include "../../include/hiprenderer_c.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" void hipr_internal_set_last_error(const char* message);   // hiprenderer.hip: hipr_last_error() is per thread; a worker's message is handed to the caller's

namespace {

// The slice of the RCCL API the gather needs (rccl.h: ncclResult_t is an int enum with ncclSuccess = 0, ncclUint8 = 1 of ncclDataType_t).
struct Rccl {
    void* library = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*CommAbort)(void* comm) = nullptr;      // optional: frees a communicator whose operations will never complete
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void* sendbuff, size_t count, int datatype, int peer, void* comm, hipStream_t stream) = nullptr;
    int (*Recv)(void* recvbuff, size_t count, int datatype, int peer, void* comm, hipStream_t stream) = nullptr;
    bool load() {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            library = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (library) break;
        }
        if (!library) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(library, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(library, "ncclCommDestroy"));
        CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(library, "ncclCommAbort"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(library, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(library, "ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(dlsym(library, "ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(dlsym(library, "ncclRecv"));
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv;
    }
};
constexpr int NCCL_UINT8 = 1;

struct Member {
    int device = 0;
    HiprContext* context = nullptr;
    void* compact = nullptr;        // half4 per owned pixel, on this member's device
    hipStream_t stream = nullptr;   // gather traffic of this member
    void* comm = nullptr;           // ncclComm_t
    int status = HIPR_OK;
    std::string message;            // hipr_last_error() of the thread that ran this member's share of the last call
};

// One worker thread per member beyond the first, parked on a condition variable between calls.
struct Workers {
    std::vector<std::thread> threads;
    std::mutex mutex;
    std::condition_variable start, done;
    unsigned long long generation = 0;
    uint32_t pending = 0;
    bool stop = false;
    std::function<int(Member&, uint32_t)> work;
};

} // namespace

struct HiprGroup {
    std::vector<Member> members;
    Rccl rccl;
    bool use_rccl = false;
    uint32_t width = 0, height = 0;
    uint64_t compact_pixels = 0;    // per member, padded to member 0's count (it owns the most tiles)
    void* gathered = nullptr;       // members x compact_pixels half4 on member 0's device
    bool frame_ready = false;
    std::string gather_description;
    Workers workers;
    // What a stalled exchange leaves behind (fall_back_to_copies): streams that may still hold queued work, and the buffers that work reads and writes. They are
    // parked here, never reused, and released in hipr_group_destroy once their stream has drained (leaked with a message if it never does).
    struct Retired { int device; hipStream_t stream; std::vector<void*> buffers; };
    std::vector<Retired> retired;
    int gather_timeout_ms = 10000;  // HIPR_GROUP_GATHER_TIMEOUT_MS
    int stall_member_once = -1;     // HIPR_GROUP_TEST_STALL_MEMBER: test hook, the member whose half of the NEXT exchange is held back past the deadline
};

namespace {

void worker_loop(HiprGroup* g, uint32_t index) {
    Workers& w = g->workers;
    unsigned long long seen = 0;
    for (;;) {
        std::function<int(Member&, uint32_t)> work;
        {
            std::unique_lock<std::mutex> lock(w.mutex);
            w.start.wait(lock, [&] { return w.stop || w.generation != seen; });
            if (w.stop) return;
            seen = w.generation;
            work = w.work;
        }
        Member& m = g->members[index];
        // an exception on this thread (std::bad_alloc in a host-side vector of an entry point) would end the process: it becomes the member's status
        try {
            m.status = work(m, index);
            m.message = m.status != HIPR_OK ? hipr_last_error() : "";
        } catch (const std::exception& e) {
            m.status = HIPR_ERROR_OUT_OF_MEMORY;
            m.message = std::string("exception on the member's worker thread: ") + e.what();
        }
        {
            std::lock_guard<std::mutex> lock(w.mutex);
            if (--w.pending == 0) w.done.notify_one();
        }
    }
}

// Runs work(member, index) for every member, member 0 on the calling thread, and returns the first failure; the failing member's message becomes
// the calling thread's hipr_last_error().
int for_each_member(HiprGroup* g, std::function<int(Member&, uint32_t)> work) {
    std::vector<Member>& members = g->members;
    Workers& w = g->workers;
    const uint32_t others = uint32_t(w.threads.size());
    if (others) {
        std::lock_guard<std::mutex> lock(w.mutex);
        w.work = work;
        w.pending = others;
        ++w.generation;
        w.start.notify_all();
    }
    members[0].status = work(members[0], 0u);
    members[0].message = members[0].status != HIPR_OK ? hipr_last_error() : "";
    for (size_t i = 1 + others; i < members.size(); ++i) {   // no worker (the group is being built or torn down): in line
        members[i].status = work(members[i], uint32_t(i));
        members[i].message = members[i].status != HIPR_OK ? hipr_last_error() : "";
    }
    if (others) {
        std::unique_lock<std::mutex> lock(w.mutex);
        w.done.wait(lock, [&] { return w.pending == 0; });
    }
    for (size_t i = 0; i < members.size(); ++i)
        if (members[i].status != HIPR_OK) {
            const std::string text = "device group member " + std::to_string(i) + " (device " + std::to_string(members[i].device) + "): " + members[i].message;
            hipr_internal_set_last_error(text.c_str());
            return members[i].status;
        }
    return HIPR_OK;
}

// Waits for `stream` by polling, up to `timeout_ms`: a member's thread must come back from the exchange whatever its peers do. A HIP failure on the stream (a fault,
// a lost device) is reported as what it is, not as a timeout.
enum class Waited { Done, TimedOut, Failed };
Waited wait_for_stream(hipStream_t stream, int timeout_ms, std::string& failure) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    for (;;) {
        const hipError_t status = hipStreamQuery(stream);
        if (status == hipSuccess) return Waited::Done;
        if (status != hipErrorNotReady) {
            failure = std::string("hipStreamQuery on the gather stream: ") + hipGetErrorString(status);
            (void)hipGetLastError();
            return Waited::Failed;
        }
        if (std::chrono::steady_clock::now() > deadline) return Waited::TimedOut;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// After an exchange that stalled: the communicators are given up (their pending operations would hold the streams for ever) and the group gathers with peer-to-peer
// copies from now on -- on FRESH streams and into FRESH buffers. What is still queued on the old streams (the held-back copy, an ncclSend kernel) may run at any later
// time: it must find the memory it was given, and must not touch anything a later call uses. So the old streams are not destroyed (hipStreamDestroy may wait for
// them; with a copy that never ends, for ever) and the old compact / gathered buffers are not freed or reused: both are parked in g->retired until the group is
// destroyed. Returns false when the fresh streams or buffers cannot be had; the group then refuses further frames (frame_ready = false) instead of racing.
bool fall_back_to_copies(HiprGroup* g, const char* why) {
    bool ok = true;
    const size_t bytes = size_t(g->compact_pixels) * 8;
    for (size_t i = 0; i < g->members.size(); ++i) {
        Member& m = g->members[i];
        (void)hipSetDevice(m.device);
        if (m.comm) {
            if (g->rccl.CommAbort) g->rccl.CommAbort(m.comm);      // without ncclCommAbort the communicator is left alone: destroying it would wait for the stalled operations
            m.comm = nullptr;
        }
        HiprGroup::Retired old = {m.device, m.stream, {m.compact}};
        if (i == 0) old.buffers.push_back(g->gathered);
        m.stream = nullptr; m.compact = nullptr;
        if (i == 0) g->gathered = nullptr;
        ok = ok && hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking) == hipSuccess && hipMalloc(&m.compact, bytes) == hipSuccess;
        if (i == 0) ok = ok && hipMalloc(&g->gathered, g->members.size() * bytes) == hipSuccess;
        g->retired.push_back(std::move(old));
    }
    (void)hipGetLastError();
    g->use_rccl = false;
    if (!ok) g->frame_ready = false;      // hipr_group_set_frame allocates again
    g->gather_description = std::string("peer-to-peer hipMemcpyAsync (fallen back: ") + why + ")";
    fprintf(stderr, "hiprenderer: device group: %s; the group gathers with peer-to-peer copies from now on\n", why);
    return ok;
}

void stop_workers(HiprGroup* g) {
    Workers& w = g->workers;
    {
        std::lock_guard<std::mutex> lock(w.mutex);
        w.stop = true;
        w.start.notify_all();
    }
    for (std::thread& t : w.threads) t.join();
    w.threads.clear();
}

} // namespace

extern "C" {

int hipr_group_destroy(HiprGroup* g) {
    if (!g) return HIPR_OK;
    stop_workers(g);
    for (Member& m : g->members) {
        (void)hipSetDevice(m.device);
        if (m.comm && g->rccl.CommDestroy) g->rccl.CommDestroy(m.comm);
        if (m.compact) (void)hipFree(m.compact);
        if (m.stream) (void)hipStreamDestroy(m.stream);
    }
    for (HiprGroup::Retired& r : g->retired) {      // what a stalled exchange left behind: released once drained, given a last second to drain, leaked otherwise
        (void)hipSetDevice(r.device);
        std::string ignored;
        if (r.stream && wait_for_stream(r.stream, 1000, ignored) != Waited::Done) {
            fprintf(stderr, "hiprenderer: device group: a stalled gather stream of device %d never drained; its stream and %zu buffers are left to the process\n", r.device, r.buffers.size());
            continue;
        }
        for (void* b : r.buffers) if (b) (void)hipFree(b);
        if (r.stream) (void)hipStreamDestroy(r.stream);
    }
    for (Member& m : g->members) if (m.context) hipr_destroy(m.context);
    if (g->gathered && !g->members.empty()) { (void)hipSetDevice(g->members[0].device); (void)hipFree(g->gathered); }
    delete g;
    return HIPR_OK;
}

int hipr_group_create(const int* device_ids, uint32_t count, HiprGroup** out_group) {
    if (!device_ids || count == 0 || count > 64 || !out_group) return HIPR_ERROR_INVALID_ARGUMENT;
    *out_group = nullptr;
    HiprGroup* g = new HiprGroup();
    g->members.resize(count);
    bool distinct = true;
    for (uint32_t i = 0; i < count; ++i) {
        Member& m = g->members[i];
        m.device = device_ids[i];
        for (uint32_t j = 0; j < i; ++j) distinct = distinct && device_ids[j] != device_ids[i];
        if (int s = hipr_create(m.device, &m.context)) { hipr_group_destroy(g); return s; }
        if (hipSetDevice(m.device) != hipSuccess || hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking) != hipSuccess) { hipr_group_destroy(g); return HIPR_ERROR_HIP; }
    }
    // peer access for the copy gather and for RCCL's direct transport; a failure only means the copies are staged by the runtime
    for (uint32_t i = 1; i < count && distinct; ++i) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, g->members[0].device, g->members[i].device) == hipSuccess && can) {
            (void)hipSetDevice(g->members[0].device);
            (void)hipDeviceEnablePeerAccess(g->members[i].device, 0);
            (void)hipSetDevice(g->members[i].device);
            (void)hipDeviceEnablePeerAccess(g->members[0].device, 0);
        }
    }
    (void)hipGetLastError();
    if (const char* v = getenv("HIPR_GROUP_GATHER_TIMEOUT_MS")) g->gather_timeout_ms = std::max(1, atoi(v));
    if (const char* v = getenv("HIPR_GROUP_TEST_STALL_MEMBER")) g->stall_member_once = atoi(v);
    const char* mode = getenv("HIPR_GROUP_GATHER");
    g->gather_description = "peer-to-peer hipMemcpyAsync";
    if (count > 1 && distinct && !(mode && !strcmp(mode, "copy")) && g->rccl.load()) {
        std::vector<void*> comms(count, nullptr);
        std::vector<int> devices(device_ids, device_ids + count);
        if (g->rccl.CommInitAll(comms.data(), int(count), devices.data()) == 0) {
            for (uint32_t i = 0; i < count; ++i) g->members[i].comm = comms[i];
            g->use_rccl = true;
            g->gather_description = "RCCL ncclSend / ncclRecv (ncclCommInitAll)";
        } else
            fprintf(stderr, "hiprenderer: ncclCommInitAll failed; the group gathers with peer-to-peer copies\n");
    }
    for (uint32_t i = 1; i < count; ++i) g->workers.threads.emplace_back(worker_loop, g, i);
    *out_group = g;
    return HIPR_OK;
}

uint32_t hipr_group_size(HiprGroup* g) { return g ? uint32_t(g->members.size()) : 0u; }
HiprContext* hipr_group_context(HiprGroup* g, uint32_t member) { return g && member < g->members.size() ? g->members[member].context : nullptr; }
const char* hipr_group_gather_description(HiprGroup* g) { return g ? g->gather_description.c_str() : ""; }

int hipr_group_upload_tables(HiprGroup* g, const HiprTables* tables) {
    if (!g) return HIPR_ERROR_INVALID_ARGUMENT;
    return for_each_member(g, [&](Member& m, uint32_t) { return hipr_upload_tables(m.context, tables); });
}

int hipr_group_upload_scene(HiprGroup* g, const HiprSceneDesc* scene) {
    if (!g) return HIPR_ERROR_INVALID_ARGUMENT;
    return for_each_member(g, [&](Member& m, uint32_t) { return hipr_upload_scene(m.context, scene); });   // replicated: every device walks the same BVH
}

int hipr_group_update_scene_geometry(HiprGroup* g, const HiprSceneDesc* scene) {
    if (!g) return HIPR_ERROR_INVALID_ARGUMENT;
    return for_each_member(g, [&](Member& m, uint32_t) { return hipr_update_scene_geometry(m.context, scene); });
}

int hipr_group_set_scene_state(HiprGroup* g, const HiprSceneState* state) {
    if (!g) return HIPR_ERROR_INVALID_ARGUMENT;
    for (Member& m : g->members)
        if (int s = hipr_set_scene_state(m.context, state)) return s;
    return HIPR_OK;
}

int hipr_group_set_entry_point(HiprGroup* g, int entry) {
    if (!g) return HIPR_ERROR_INVALID_ARGUMENT;
    for (Member& m : g->members)
        if (int s = hipr_set_entry_point(m.context, entry)) return s;
    return HIPR_OK;
}

int hipr_group_use_scratch_accumulation(HiprGroup* g, int enable) {
    if (!g) return HIPR_ERROR_INVALID_ARGUMENT;
    for (Member& m : g->members)
        if (int s = hipr_use_scratch_accumulation(m.context, enable)) return s;
    return HIPR_OK;
}

int hipr_group_set_frame(HiprGroup* g, uint32_t width, uint32_t height, uint32_t samples_per_pass) {
    if (!g || width == 0 || height == 0) return HIPR_ERROR_INVALID_ARGUMENT;
    const uint32_t n = uint32_t(g->members.size());
    g->frame_ready = false;
    int status = for_each_member(g, [&](Member& m, uint32_t i) {
        HiprFrameDesc frame = {width, height, i, n, samples_per_pass};
        return hipr_set_frame(m.context, &frame);
    });
    if (status) return status;
    g->width = width; g->height = height;
    if (n > 1) {
        uint32_t owned = 0;
        if (int s = hipr_owned_pixel_count(g->members[0].context, &owned)) return s;
        g->compact_pixels = owned;
        for (Member& m : g->members) {
            if (hipSetDevice(m.device) != hipSuccess) return HIPR_ERROR_HIP;
            if (m.compact) (void)hipFree(m.compact);
            m.compact = nullptr;
            if (hipMalloc(&m.compact, g->compact_pixels * 8) != hipSuccess) return HIPR_ERROR_OUT_OF_MEMORY;
        }
        if (hipSetDevice(g->members[0].device) != hipSuccess) return HIPR_ERROR_HIP;
        if (g->gathered) (void)hipFree(g->gathered);
        g->gathered = nullptr;
        if (hipMalloc(&g->gathered, uint64_t(n) * g->compact_pixels * 8) != hipSuccess) return HIPR_ERROR_OUT_OF_MEMORY;
    }
    g->frame_ready = true;
    return HIPR_OK;
}

int hipr_group_set_samples_per_pass(HiprGroup* g, uint32_t samples_per_pass) {
    if (!g) return HIPR_ERROR_INVALID_ARGUMENT;
    for (Member& m : g->members)
        if (int s = hipr_set_samples_per_pass(m.context, samples_per_pass)) return s;
    return HIPR_OK;
}

int hipr_group_trace_pass(HiprGroup* g, const HiprCameraState* camera) {
    if (!g || !camera) return HIPR_ERROR_INVALID_ARGUMENT;
    return for_each_member(g, [&](Member& m, uint32_t) { return hipr_trace_pass(m.context, camera); });
}

int hipr_group_accumulate_samples(HiprGroup* g, uint32_t first_sample, uint32_t sample_count, uint32_t first_accumulation, void* out_half4_device, uint32_t out_pitch_pixels,
                                  int synchronize) {
    if (!g || !g->frame_ready) return HIPR_ERROR_NOT_READY;
    const uint32_t n = uint32_t(g->members.size());
    if (n == 1) return hipr_accumulate_samples(g->members[0].context, first_sample, sample_count, first_accumulation, out_half4_device, out_pitch_pixels, synchronize);
    if (out_half4_device && out_pitch_pixels < g->width) return HIPR_ERROR_INVALID_ARGUMENT;
    const size_t bytes = size_t(g->compact_pixels) * 8;
    char* gathered = static_cast<char*>(g->gathered);
    // Phase 1: every member folds its samples and writes its compact tiles. Phase 2, only when every member got there: the tiles travel to
    // member 0's device. (A member that failed before posting its send or receive would leave its peer waiting in the exchange for ever.)
    int status = for_each_member(g, [&](Member& m, uint32_t) -> int {
        return hipr_accumulate_samples(m.context, first_sample, sample_count, first_accumulation, out_half4_device ? m.compact : nullptr, 0, 1);
    });
    if (status || !out_half4_device) return status;
    const int stalled = g->stall_member_once;
    g->stall_member_once = -1;
    std::atomic<int> timed_out{-1};
    status = for_each_member(g, [&](Member& m, uint32_t i) -> int {
        auto hip_failed = [&](const char* what) { hipr_internal_set_last_error(what); return HIPR_ERROR_HIP; };
        if (hipSetDevice(m.device) != hipSuccess) return hip_failed("hipSetDevice failed in the tile gather");
        // test hook: this member's stream is held past the deadline by a host function that sleeps on it -- queued BEFORE a copy (the copy itself is what stalls: it
        // runs later, into the retired buffers) and AFTER an RCCL call (an ncclSend kernel held back behind the sleep would launch after ncclCommAbort had freed the
        // communicator's device state; queued after, the exchange itself completes and only the stream's tail is late)
        auto hold_stream = [&] {
            static int sleep_ms;
            sleep_ms = g->gather_timeout_ms + 500;
            (void)hipLaunchHostFunc(m.stream, [](void* ms) { std::this_thread::sleep_for(std::chrono::milliseconds(*static_cast<int*>(ms))); }, &sleep_ms);
        };
        if (int(i) == stalled && !g->use_rccl) hold_stream();
        if (g->use_rccl) {
            // one communicator per thread, every thread's calls inside its own group call (RCCL: several communicators driven from one process)
            int r = 0;
            if (i == 0) {
                if (hipMemcpyAsync(gathered, m.compact, bytes, hipMemcpyDeviceToDevice, m.stream) != hipSuccess) return hip_failed("copy of member 0's own tiles failed");
                r = g->rccl.GroupStart();
                for (uint32_t peer = 1; peer < n && r == 0; ++peer) r = g->rccl.Recv(gathered + size_t(peer) * bytes, bytes, NCCL_UINT8, int(peer), m.comm, m.stream);
                r = g->rccl.GroupEnd() | r;
                if (r != 0) return hip_failed("ncclRecv of the members' tiles failed");
            } else {
                r = g->rccl.GroupStart();
                if (r == 0) r = g->rccl.Send(m.compact, bytes, NCCL_UINT8, 0, m.comm, m.stream);
                r = g->rccl.GroupEnd() | r;
                if (r != 0) return hip_failed("ncclSend of a member's tiles failed");
            }
            if (int(i) == stalled) hold_stream();
        } else if (hipMemcpyAsync(gathered + size_t(i) * bytes, m.compact, bytes, hipMemcpyDeviceToDevice, m.stream) != hipSuccess)
            return hip_failed("peer-to-peer copy of a member's tiles failed");
        std::string failure;
        const Waited waited = wait_for_stream(m.stream, g->gather_timeout_ms, failure);
        if (waited == Waited::Done) return HIPR_OK;
        if (waited == Waited::Failed) return hip_failed(failure.c_str());
        int none = -1;
        timed_out.compare_exchange_strong(none, int(i));
        const std::string text = "the tile exchange (" + g->gather_description + ") did not finish within " + std::to_string(g->gather_timeout_ms) + " ms on this member's stream";
        hipr_internal_set_last_error(text.c_str());
        return HIPR_ERROR_TIMEOUT;
    });
    if (timed_out.load() >= 0 || (status && g->use_rccl)) {
        // the frame of this call is not delivered; the next call finds a group that copies
        const std::string message = hipr_last_error();
        if (!fall_back_to_copies(g, timed_out.load() >= 0 ? "an exchange did not finish within its deadline" : "an RCCL call failed"))
            fprintf(stderr, "hiprenderer: device group: no fresh streams or buffers after the stalled exchange; hipr_group_set_frame must be called again\n");
        hipr_internal_set_last_error(message.c_str());
    }
    if (status) return status;
    if (int s = hipr_scatter_tiles(g->members[0].context, g->gathered, g->compact_pixels, n, g->width, g->height, out_half4_device, out_pitch_pixels)) return s;
    return synchronize ? hipr_synchronize(g->members[0].context) : HIPR_OK;
}

int hipr_group_read_accumulation(HiprGroup* g, double* out_rgba, uint64_t capacity_pixels) {
    if (!g || !g->frame_ready || !out_rgba) return HIPR_ERROR_INVALID_ARGUMENT;
    const uint32_t n = uint32_t(g->members.size());
    if (n == 1) return hipr_read_accumulation(g->members[0].context, out_rgba, capacity_pixels);
    if (capacity_pixels < uint64_t(g->width) * g->height) return HIPR_ERROR_INVALID_ARGUMENT;
    const uint32_t tiles_x = (g->width + 7) / 8, tiles_total = tiles_x * ((g->height + 7) / 8);
    std::vector<double> compact;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t owned = 0;
        if (int s = hipr_owned_pixel_count(g->members[i].context, &owned)) return s;
        compact.resize(size_t(owned) * 4);
        if (int s = hipr_read_accumulation(g->members[i].context, compact.data(), owned)) return s;
        for (uint32_t k = 0; k < owned; ++k) {
            const uint32_t tile = (k >> 6) * n + i, lane = k & 63u;
            const uint32_t x = (tile % tiles_x) * 8 + (lane & 7u), y = (tile / tiles_x) * 8 + (lane >> 3);
            if (tile < tiles_total && x < g->width && y < g->height) std::memcpy(out_rgba + 4 * (size_t(y) * g->width + x), compact.data() + 4 * size_t(k), 32);
        }
    }
    return HIPR_OK;
}

int hipr_group_get_counters(HiprGroup* g, HiprCounters* out) {
    if (!g || !out) return HIPR_ERROR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof(*out));
    for (Member& m : g->members) {
        HiprCounters c;
        if (int s = hipr_get_counters(m.context, &c)) return s;
        out->camera_rays += c.camera_rays; out->closest_rays += c.closest_rays; out->shadow_rays += c.shadow_rays; out->shaded_hits += c.shaded_hits;
        out->closest_nodes += c.closest_nodes; out->closest_triangles += c.closest_triangles; out->shadow_nodes += c.shadow_nodes; out->shadow_triangles += c.shadow_triangles;
        out->iterations = std::max(out->iterations, c.iterations);
    }
    return HIPR_OK;
}

} // extern "C"
