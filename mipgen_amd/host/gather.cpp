// gather.cpp — the exchange step of the multi-GPU front end (`-gpu_gather rccl`): per result window ONE grouped RCCL send / receive moves the
// window's emitted counts, condensed survivors and collapse results (+ the SVR re-scores of a mixed design, + the all_mips text of a non-silent
// one) from the HBM of the device that scored it into one packed buffer on GPU 0 - xGMI is point to point, every peer has its own link to the
// root, nothing rings -, and one D2H copy hands the buffer to the sequential selection stage (the consumer of the reference's per-region tables:
// /root/reference/mipgen.cpp:503-515 condense -> collapse -> pick in region order; rand() at :1863 and the used-arm sets of :1925-1938 couple
// the regions, so the pick runs in ONE place).  Single process, one communicator per device (ncclCommInitAll), every RCCL call from the
// consumer thread; rank 0's own windows take the same route (a self send / receive inside the group).
//
// RCCL and the HIP runtime are bound at the first RcclGather::init, not at link time: only this opt-in route needs them, and every other run of
// libmipgen_host.so (the default PCIe gather, designs that never reach the accelerator, usage errors, ranks of a multi-process harness) neither
// pays RCCL's load time nor needs librccl on the machine.  Symbols already in the process are taken first (a program that links RCCL itself; the
// memcpy-backed stand-ins of tests/stub_accel, which run this file with several ranks under ThreadSanitizer on a machine without a GPU).
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "gather.hpp"

namespace mipgen {

namespace {

// the entry points this file calls, by their own prototypes (the headers above are used for the types only)
#define GATHER_HIP_API(X) \
    X(hipSetDevice) X(hipStreamCreateWithFlags) X(hipStreamSynchronize) X(hipStreamDestroy) X(hipEventCreateWithFlags) X(hipEventRecord) \
    X(hipEventSynchronize) X(hipEventDestroy) X(hipFree) X(hipHostRegister) X(hipHostUnregister) X(hipMemcpyAsync) X(hipGetErrorString)
#define GATHER_RCCL_API(X) X(ncclCommInitAll) X(ncclCommDestroy) X(ncclGroupStart) X(ncclGroupEnd) X(ncclSend) X(ncclRecv) X(ncclGetErrorString)

struct Api {
#define X(name) decltype(&::name) name = nullptr;
    GATHER_HIP_API(X)
    GATHER_RCCL_API(X)
#undef X
    // hipMalloc is an overload set in the C++ header: the plain (void**, size_t) form is the exported symbol
    hipError_t (*hipMallocRaw)(void**, size_t) = nullptr;
};
Api g_api;
std::once_flag g_api_once;
std::string g_api_err;

void* open_first(const char* const* names)
{
    for (; *names; names++) if (void* h = dlopen(*names, RTLD_NOW | RTLD_LOCAL)) return h;
    return nullptr;
}

void bind_api()
{
    static const char* const rccl_names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", nullptr};
    static const char* const hip_names[] = {"libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so.7", "libamdhip64.so", nullptr};
    // (RTLD_DEFAULT is a null pointer: "found in the process" is a flag of its own)
    auto why = [] { const char* e = dlerror(); return std::string(e ? e : "no loader message"); };
    const bool rccl_here = dlsym(RTLD_DEFAULT, "ncclCommInitAll") != nullptr;
    void* rccl = rccl_here ? RTLD_DEFAULT : open_first(rccl_names);
    if (!rccl_here && !rccl) { g_api_err = "rccl gather: librccl not found (" + why() + ")"; return; }
    const bool hip_here = dlsym(RTLD_DEFAULT, "hipStreamCreateWithFlags") != nullptr;
    void* hip = hip_here ? RTLD_DEFAULT : open_first(hip_names);
    if (!hip_here && !hip) { g_api_err = "rccl gather: libamdhip64 not found (" + why() + ")"; return; }
    const char* missing = nullptr;
#define X(name) if (!(g_api.name = (decltype(g_api.name))dlsym(hip, #name)) && !missing) missing = #name;
    X(hipSetDevice) X(hipStreamCreateWithFlags) X(hipStreamSynchronize) X(hipStreamDestroy) X(hipEventCreateWithFlags) X(hipEventRecord)
    X(hipEventSynchronize) X(hipEventDestroy) X(hipFree) X(hipHostRegister) X(hipHostUnregister) X(hipMemcpyAsync) X(hipGetErrorString)
#undef X
    if (!(g_api.hipMallocRaw = (hipError_t (*)(void**, size_t))dlsym(hip, "hipMalloc")) && !missing) missing = "hipMalloc";
#define X(name) if (!(g_api.name = (decltype(g_api.name))dlsym(rccl, #name)) && !missing) missing = #name;
    GATHER_RCCL_API(X)
#undef X
    if (missing) g_api_err = std::string("rccl gather: entry point ") + missing + " not found";
}

// (the names below shadow the headers' declarations inside this namespace: every call goes through the bound table)
#define hipSetDevice g_api.hipSetDevice
#define hipStreamCreateWithFlags g_api.hipStreamCreateWithFlags
#define hipStreamSynchronize g_api.hipStreamSynchronize
#define hipStreamDestroy g_api.hipStreamDestroy
#define hipEventCreateWithFlags g_api.hipEventCreateWithFlags
#define hipEventRecord g_api.hipEventRecord
#define hipEventSynchronize g_api.hipEventSynchronize
#define hipEventDestroy g_api.hipEventDestroy
#define hipMalloc g_api.hipMallocRaw
#define hipFree g_api.hipFree
#define hipHostRegister g_api.hipHostRegister
#define hipHostUnregister g_api.hipHostUnregister
#define hipMemcpyAsync g_api.hipMemcpyAsync
#define hipGetErrorString g_api.hipGetErrorString
#define ncclCommInitAll g_api.ncclCommInitAll
#define ncclCommDestroy g_api.ncclCommDestroy
#define ncclGroupStart g_api.ncclGroupStart
#define ncclGroupEnd g_api.ncclGroupEnd
#define ncclSend g_api.ncclSend
#define ncclRecv g_api.ncclRecv
#define ncclGetErrorString g_api.ncclGetErrorString

std::string hip_msg(const char* what, hipError_t e) { return std::string("rccl gather: ") + what + ": " + hipGetErrorString(e); }
std::string nccl_msg(const char* what, ncclResult_t r) { return std::string("rccl gather: ") + what + ": " + ncclGetErrorString(r); }
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

struct RcclGather::Impl {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> send_streams;       // one per rank, on its device
    bool bound = false;                          // the RCCL / HIP entry points are there (init got that far)
    hipStream_t recv_stream = nullptr;           // on the root device
    struct Slot {
        void* dev = nullptr; size_t dev_cap = 0;     // packed receive buffer on the root device
        void* host = nullptr; size_t host_cap = 0;   // pinned: target of the one D2H copy
        hipEvent_t done = nullptr;
        bool in_flight = false;
    } slot[kSlots];
};

RcclGather::RcclGather() : p_(new Impl()) {}
RcclGather::~RcclGather() { destroy(); delete p_; }

int RcclGather::check_devices(const std::vector<int>& devices, std::string* err)
{
    if (devices.empty()) { *err = "rccl gather: no devices"; return -1; }
    for (size_t i = 0; i < devices.size(); i++)
        for (size_t j = i + 1; j < devices.size(); j++)
            if (devices[i] == devices[j]) { *err = "rccl gather: one communicator rank per device - more device workers than visible GPUs (use -gpu_gather pcie)"; return -1; }
    return 0;
}

int RcclGather::init(const std::vector<int>& devices, std::string* err)
{
    Impl& P = *p_;
    const double t_init0 = now_s();
    if (check_devices(devices, err)) return -1;
    std::call_once(g_api_once, bind_api);
    if (!g_api_err.empty()) { *err = g_api_err; return -1; }
    P.bound = true;
    P.devices = devices;
    P.comms.assign(devices.size(), nullptr);
    ncclResult_t r = ncclCommInitAll(P.comms.data(), (int)devices.size(), devices.data());
    if (r != ncclSuccess) { P.comms.clear(); *err = nccl_msg("ncclCommInitAll", r); return -1; }
    P.send_streams.assign(devices.size(), nullptr);
    for (size_t k = 0; k < devices.size(); k++) {
        hipError_t e = hipSetDevice(devices[k]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&P.send_streams[k], hipStreamNonBlocking);
        if (e != hipSuccess) { *err = hip_msg("send stream", e); return -1; }
    }
    hipError_t e = hipSetDevice(devices[0]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&P.recv_stream, hipStreamNonBlocking);
    for (int s = 0; s < kSlots && e == hipSuccess; s++) e = hipEventCreateWithFlags(&P.slot[s].done, hipEventDisableTiming);
    if (e != hipSuccess) { *err = hip_msg("receive stream / events", e); return -1; }
    seconds_init = now_s() - t_init0;
    return 0;
}

int RcclGather::post(int src, const GatherPiece* pieces, int n, size_t total, int s, std::string* err)
{
    Impl& P = *p_;
    if (src < 0 || src >= (int)P.comms.size() || s < 0 || s >= kSlots || P.slot[s].in_flight) { *err = "rccl gather: bad post"; return -1; }
    const double t0 = now_s();
    Impl::Slot& S = P.slot[s];
    hipError_t e = hipSetDevice(P.devices[0]);
    if (e != hipSuccess) { *err = hip_msg("hipSetDevice", e); return -1; }
    const size_t want = total + 256;
    if (S.dev_cap < want) {
        if (S.dev) (void)hipFree(S.dev);
        S.dev = nullptr; S.dev_cap = 0;
        e = hipMalloc(&S.dev, want + want / 4);
        if (e != hipSuccess) { *err = hip_msg("receive buffer", e); return -1; }
        S.dev_cap = want + want / 4;
    }
    if (S.host_cap < want) {
        if (S.host) { (void)hipHostUnregister(S.host); free(S.host); }
        S.host = nullptr; S.host_cap = 0;
        // ordinary (CPU-cached) pages, pinned in place: the selection stage reads the buffer many times, and memory from hipHostMalloc is mapped
        // uncached on the host side (the pick stage ran 30 % slower from it); visibility is by the event the D2H copy is followed by
        const size_t bytes = (want + want / 4 + 4095) & ~(size_t)4095;
        if (posix_memalign(&S.host, 4096, bytes)) { S.host = nullptr; *err = "rccl gather: out of host memory"; return -1; }
        memset(S.host, 0, bytes);                                  // (touch the pages before they are pinned)
        e = hipHostRegister(S.host, bytes, hipHostRegisterDefault);
        if (e != hipSuccess) { free(S.host); S.host = nullptr; *err = hip_msg("pinning the host buffer", e); return -1; }
        S.host_cap = want + want / 4;
    }
    // one group = one fused transfer: every piece is a send on the source rank's communicator and a receive at its offset of the packed buffer
    ncclResult_t r = ncclGroupStart();
    for (int i = 0; i < n && r == ncclSuccess; i++) {
        if (pieces[i].bytes == 0) continue;
        r = ncclSend(pieces[i].dev, pieces[i].bytes, ncclUint8, 0, P.comms[(size_t)src], P.send_streams[(size_t)src]);
        if (r == ncclSuccess) r = ncclRecv((char*)S.dev + pieces[i].offset, pieces[i].bytes, ncclUint8, src, P.comms[0], P.recv_stream);
    }
    const ncclResult_t r2 = ncclGroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess) { *err = nccl_msg("grouped send / receive", r != ncclSuccess ? r : r2); return -1; }
    (void)hipSetDevice(P.devices[0]);
    if (total) e = hipMemcpyAsync(S.host, S.dev, total, hipMemcpyDeviceToHost, P.recv_stream);
    if (e == hipSuccess) e = hipEventRecord(S.done, P.recv_stream);
    if (e != hipSuccess) { *err = hip_msg("D2H of the packed buffer", e); return -1; }
    S.in_flight = true;
    windows++; bytes_moved += total;
    seconds_posting += now_s() - t0;
    return 0;
}

int RcclGather::wait(int s, std::string* err)
{
    Impl& P = *p_;
    if (s < 0 || s >= kSlots || !P.slot[s].in_flight) { *err = "rccl gather: bad wait"; return -1; }
    const double t0 = now_s();
    const hipError_t e = hipEventSynchronize(P.slot[s].done);
    seconds_waiting += now_s() - t0;
    if (e != hipSuccess) { *err = hip_msg("waiting for a window", e); return -1; }
    P.slot[s].in_flight = false;
    return 0;
}

char* RcclGather::host(int s) { return (char*)p_->slot[s].host; }

void RcclGather::destroy()
{
    Impl& P = *p_;
    if (!P.bound) return;                        // never initialised: nothing was created
    const double t0 = now_s();
    // everything posted must have drained before a buffer goes: a post() that failed AFTER its group (the D2H copy, the event) leaves the receive
    // of that group running into the slot's device buffer with no event to wait on - the streams themselves are waited for, senders first
    for (size_t k = 0; k < P.send_streams.size(); k++) if (P.send_streams[k]) { (void)hipSetDevice(P.devices[k]); (void)hipStreamSynchronize(P.send_streams[k]); }
    if (P.recv_stream) { (void)hipSetDevice(P.devices[0]); (void)hipStreamSynchronize(P.recv_stream); }
    for (int s = 0; s < kSlots; s++) {
        Impl::Slot& S = P.slot[s];
        S.in_flight = false;
        if (S.dev) { (void)hipSetDevice(P.devices.empty() ? 0 : P.devices[0]); (void)hipFree(S.dev); S.dev = nullptr; S.dev_cap = 0; }
        if (S.host) { (void)hipHostUnregister(S.host); free(S.host); S.host = nullptr; S.host_cap = 0; }
        if (S.done) { (void)hipEventDestroy(S.done); S.done = nullptr; }
    }
    for (size_t k = 0; k < P.send_streams.size(); k++) if (P.send_streams[k]) { (void)hipSetDevice(P.devices[k]); (void)hipStreamDestroy(P.send_streams[k]); }
    P.send_streams.clear();
    if (P.recv_stream) { (void)hipSetDevice(P.devices[0]); (void)hipStreamDestroy(P.recv_stream); P.recv_stream = nullptr; }
    for (ncclComm_t c : P.comms) if (c) (void)ncclCommDestroy(c);
    if (!P.comms.empty()) seconds_destroy = now_s() - t0;
    P.comms.clear();
}

}  // namespace mipgen
