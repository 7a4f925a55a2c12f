// TEST INFRASTRUCTURE, not product: a stand-in for librccl inside ONE process on ONE GPU, so that everything around the one
// collective of the C ABI (sf_comm_unique_id / sf_comm_init / sf_allgather_status / sf_comm_destroy, include/simfire_hip.h) can be
// exercised with a world of 8 "ranks" = 8 handles: the unique-id hand-off, the ranks' arguments, the size and rank-major order of
// the gathered block, the error paths.  RCCL itself refuses several ranks on one GPU; on an 8-GPU node the only line this leaves
// untested is ncclAllGather.  The library loads it instead of librccl when SIMFIRE_RCCL_LIB names it (tests/test_hip_resident.py
// builds it with hipcc into a temporary directory).
// Semantics: a rank's ncclAllGather copies its block into the slot of its rank in the receive buffer of every rank that has called
// so far, and their blocks into its own - when the last rank has called, every receive buffer holds every block.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2 } ncclDataType_t;
struct FakeComm { uint64_t world_id; int nranks, rank; const void *send = nullptr; void *recv = nullptr; size_t count = 0; bool called = false; };
typedef FakeComm *ncclComm_t;
}

static const char kMagic[8] = {'F', 'A', 'K', 'E', 'R', 'C', 'C', 'L'};
static uint64_t g_next_id = 1;
static std::map<uint64_t, std::vector<FakeComm *>> g_worlds;

extern "C" ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof *id);
    std::memcpy(id->internal, kMagic, 8);
    const uint64_t v = g_next_id++;
    std::memcpy(id->internal + 8, &v, 8);
    return ncclSuccess;
}

extern "C" ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (std::memcmp(id.internal, kMagic, 8) != 0) return ncclInvalidArgument;            // not an id of ncclGetUniqueId
    uint64_t v;
    std::memcpy(&v, id.internal + 8, 8);
    auto &w = g_worlds[v];
    if (w.empty()) w.assign((size_t)nranks, nullptr);
    if ((int)w.size() != nranks || w[(size_t)rank]) return ncclInvalidUsage;              // two sizes of one world / a rank taken twice
    FakeComm *c = new FakeComm();
    c->world_id = v; c->nranks = nranks; c->rank = rank;
    w[(size_t)rank] = c;
    *comm = c;
    return ncclSuccess;
}

extern "C" ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || !sendbuff || !recvbuff || datatype != ncclInt32) return ncclInvalidArgument;
    auto &w = g_worlds[comm->world_id];
    const size_t bytes = sendcount * 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;         // (the block is produced on that stream)
    comm->send = sendbuff; comm->recv = recvbuff; comm->count = sendcount; comm->called = true;
    for (FakeComm *o : w) {
        if (!o || !o->called) continue;
        if (o->count != sendcount) return ncclInvalidArgument;                             // every rank the same number of environments
        if (hipMemcpy(static_cast<char *>(comm->recv) + (size_t)o->rank * bytes, o->send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
        if (hipMemcpy(static_cast<char *>(o->recv) + (size_t)comm->rank * bytes, comm->send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

extern "C" ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclInvalidArgument;
    auto it = g_worlds.find(comm->world_id);
    if (it != g_worlds.end()) {
        it->second[(size_t)comm->rank] = nullptr;
        bool any = false;
        for (FakeComm *o : it->second) any = any || o;
        if (!any) g_worlds.erase(it);
    }
    delete comm;
    return ncclSuccess;
}

extern "C" const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclInvalidArgument: return "invalid argument";
        case ncclInvalidUsage: return "invalid usage";
        default: return "error";
    }
}
