// stub_rccl.cpp — memcpy-backed stand-ins for the RCCL / HIP runtime entry points mipgen_amd/host/gather.cpp binds at run time.
// TEST INFRASTRUCTURE ONLY (part of the stub libmipgen_accel.so of tests/stub_accel; never shipped, never linked by the product).
//
// gather.cpp looks its entry points up in the process first (dlsym(RTLD_DEFAULT, ...)), so on a machine without a GPU the `-gpu_gather rccl`
// route of the front end runs against these with 2 and 4 "ranks" under ThreadSanitizer / AddressSanitizer.  What is modelled is the part of the
// semantics the gather's correctness rests on:
//   * a stream is a thread of its own that executes what was enqueued in order (copies, event records) - work is ASYNCHRONOUS to the caller, so
//     a buffer the front end reuses or frees before its transfer has been waited for is a data race / use after free the sanitizers see;
//   * ncclSend / ncclRecv only take effect at ncclGroupEnd, where every send must meet a receive of the same size on the peer's communicator
//     (an unmatched operation - a hang on real hardware - is ncclInvalidUsage here); the copy runs on the receiver's stream after everything
//     enqueued on the sender's stream before the group;
//   * an event is complete when the stream reaches its record; hipEventSynchronize / hipStreamSynchronize block the caller until then.
// "Device memory" is host memory.  Knob (environment, read here only):  STUB_RCCL_FAIL=<call>:<n>  the n-th call (1-based) of
// ncclCommInitAll | ncclGroupEnd | hipMemcpyAsync | hipEventSynchronize | hipMalloc fails.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Stream {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false;
    std::thread th;
    Stream() : th([this] { run(); }) {}
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
            }
            f();
        }
    }
    void push(std::function<void()> f) { { std::lock_guard<std::mutex> lk(m); q.push_back(std::move(f)); } cv.notify_all(); }
    void finish() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv.notify_all(); th.join(); }
};

struct Event {
    std::mutex m;
    std::condition_variable cv;
    uint64_t recorded = 0, done = 0;
    void complete(uint64_t v) { std::lock_guard<std::mutex> lk(m); if (v > done) done = v; cv.notify_all(); }   // (notified under the lock: a waiter may destroy the event as soon as it returns)
    void wait() { std::unique_lock<std::mutex> lk(m); const uint64_t want = recorded; cv.wait(lk, [&] { return done >= want; }); }
};

struct Comm { int rank = 0, n = 0, device = 0; };

struct Op { bool send; void* ptr; size_t bytes; int peer; Comm* comm; Stream* stream; };
thread_local int t_group_depth = 0;
thread_local std::vector<Op> t_ops;
thread_local int t_device = 0;

std::mutex g_count_m;
std::map<std::string, int> g_calls;
bool injected(const char* call)
{
    const char* e = getenv("STUB_RCCL_FAIL");
    if (!e) return false;
    int n;
    { std::lock_guard<std::mutex> lk(g_count_m); n = ++g_calls[call]; }
    const char* colon = strchr(e, ':');
    if (!colon) return false;
    return std::string(e, (size_t)(colon - e)) == call && atoi(colon + 1) == n;
}

int visible_devices()
{
    const char* e = getenv("STUB_ACCEL_DEVICES");
    const int n = e ? atoi(e) : 1;
    return n > 0 ? n : 1;
}

}  // namespace

extern "C" {

const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorInvalidDevice ? "invalid device ordinal (stub)" : e == hipErrorOutOfMemory ? "out of memory (stub, injected)" : "error (stub, injected)"; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidUsage ? "invalid usage (stub: an unmatched send / receive)" : "unhandled system error (stub, injected)"; }

hipError_t hipSetDevice(int d) { if (d < 0 || d >= visible_devices()) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t) new Stream(); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s)
{
    Event e;
    e.recorded = 1;
    ((Stream*)s)->push([&e] { e.complete(1); });
    e.wait();
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) { Stream* st = (Stream*)s; st->finish(); delete st; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t) new Event(); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    Event* ev = (Event*)e;
    uint64_t v;
    { std::lock_guard<std::mutex> lk(ev->m); v = ++ev->recorded; }
    ((Stream*)s)->push([ev, v] { ev->complete(v); });
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
    ((Event*)e)->wait();                                              // (the work is waited for either way: a failure must not leave copies running)
    return injected("hipEventSynchronize") ? hipErrorUnknown : hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) { delete (Event*)e; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { if (injected("hipMalloc")) return hipErrorOutOfMemory; *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t s)
{
    if (injected("hipMemcpyAsync")) return hipErrorUnknown;
    ((Stream*)s)->push([dst, src, n] { memcpy(dst, src, n); });
    return hipSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devs)
{
    if (injected("ncclCommInitAll")) return ncclSystemError;
    for (int i = 0; i < n; i++) {
        if (devs[i] < 0 || devs[i] >= visible_devices()) return ncclInvalidArgument;
        Comm* c = new Comm();
        c->rank = i; c->n = n; c->device = devs[i];
        comms[i] = (ncclComm_t)c;
    }
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete (Comm*)c; return ncclSuccess; }
ncclResult_t ncclGroupStart() { t_group_depth++; return ncclSuccess; }

static ncclResult_t run_group()
{
    std::vector<Op> ops;
    ops.swap(t_ops);
    if (injected("ncclGroupEnd")) return ncclSystemError;
    std::vector<char> used(ops.size(), 0);
    for (size_t i = 0; i < ops.size(); i++) {
        if (!ops[i].send) continue;
        size_t j = 0;
        for (; j < ops.size(); j++)
            if (!used[j] && !ops[j].send && ops[j].comm->rank == ops[i].peer && ops[j].peer == ops[i].comm->rank && ops[j].bytes == ops[i].bytes) break;
        if (j == ops.size()) return ncclInvalidUsage;
        used[i] = used[j] = 1;
        // the copy runs on the receiver's stream once the sender's stream has reached the group
        std::shared_ptr<Event> reached(new Event());
        reached->recorded = 1;
        if (ops[i].stream != ops[j].stream) ops[i].stream->push([reached] { reached->complete(1); });
        else reached->done = 1;
        void* dst = ops[j].ptr; const void* src = ops[i].ptr; const size_t n = ops[i].bytes;
        ops[j].stream->push([reached, dst, src, n] { reached->wait(); memcpy(dst, src, n); });
    }
    for (size_t j = 0; j < ops.size(); j++) if (!used[j]) return ncclInvalidUsage;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (t_group_depth <= 0) return ncclInvalidUsage;
    if (--t_group_depth > 0) return ncclSuccess;
    return run_group();
}
static ncclResult_t add_op(bool send, void* p, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
    Comm* c = (Comm*)comm;
    if (!c || peer < 0 || peer >= c->n || t != ncclUint8) return ncclInvalidArgument;
    t_ops.push_back(Op{send, p, count, peer, c, (Stream*)s});
    if (t_group_depth == 0) return run_group();                         // (outside a group: a lone operation can never be matched)
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* p, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) { return add_op(true, (void*)p, count, t, peer, comm, s); }
ncclResult_t ncclRecv(void* p, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) { return add_op(false, p, count, t, peer, comm, s); }

}  // extern "C"
