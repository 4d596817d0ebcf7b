// gather.hpp — the RCCL exchange step of the multi-GPU front end (gather.cpp); no HIP / RCCL type crosses this header.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace mipgen {

struct GatherPiece {
    const void* dev;     // device pointer on the source rank
    size_t bytes;
    size_t offset;       // where the piece lands in the packed buffer (16-byte aligned)
};

// One communicator rank per device worker (single process: ncclCommInitAll), rank 0 = the root the selection stage reads from.
// post() is asynchronous: grouped ncclSend (source rank) / ncclRecv (root) of the pieces into slot s of the root's packed receive buffers,
// then one D2H copy into the slot's pinned host buffer; wait() returns when that copy has landed.  kSlots windows may be in flight.
class RcclGather {
public:
    static constexpr int kSlots = 2;
    RcclGather();
    ~RcclGather();
    RcclGather(const RcclGather&) = delete;
    RcclGather& operator=(const RcclGather&) = delete;
    static int check_devices(const std::vector<int>& devices, std::string* err);   // cheap: what init() would refuse
    int init(const std::vector<int>& devices, std::string* err);                     // ncclCommInitAll: seconds (RCCL's own set-up) - callers overlap it
    int post(int src_rank, const GatherPiece* pieces, int n, size_t total_bytes, int slot, std::string* err);
    int wait(int slot, std::string* err);
    char* host(int slot);
    void destroy();
    double seconds_posting = 0.0, seconds_waiting = 0.0, seconds_init = 0.0, seconds_destroy = 0.0;
    size_t bytes_moved = 0;
    int64_t windows = 0;

private:
    struct Impl;
    Impl* p_;
};

}  // namespace mipgen
