// RCCL called directly ("RCCL broadcast of the WY block", per-column all-reduce of the
// sharded gemv): the collectives of the sharded Hessenberg reduction enqueued on the reduction's own
// stream from C++, instead of one Python callback into torch.distributed per panel column.
// librccl.so is opened at run time (no link-time dependency: the library still loads on a machine
// without RCCL and the callback path keeps working).  One communicator per process; the unique id
// travels through whatever the caller has (starneig_amd/distributed.py: a torch.distributed broadcast).
#include "common.h"
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <rccl/rccl.h>

namespace sn {

namespace {

struct RcclApi {
    void *handle = nullptr;
    bool failed = false;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load()
    {
        if (handle) return true;
        if (failed) return false;
        // the copy already in the process (torch's, under its soname) before any other on the path:
        // two RCCL runtimes in one process is the thing to avoid
        char const *names[] = {"librccl.so.1", "librccl.so"};
        for (char const *name : names)
            if (!handle) handle = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        for (char const *name : names)
            if (!handle) handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (!handle) { failed = true; return false; }
        auto sym = [&](const char *name) { return dlsym(handle, name); };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        Broadcast = (decltype(Broadcast))sym("ncclBroadcast");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        auto version = (ncclResult_t (*)(int *))sym("ncclGetVersion");
        int v = 0;
        bool ok = GetUniqueId && CommInitRank && CommDestroy && AllReduce && Broadcast && version &&
                  version(&v) == ncclSuccess && v / 10000 == NCCL_VERSION_CODE / 10000;
        if (!ok) {
            if (version && v) fprintf(stderr, "[starneig-amd] RCCL %d at run time, built against %d: not used\n", v, NCCL_VERSION_CODE);
            dlclose(handle); handle = nullptr; failed = true;
            return false;
        }
        return true;
    }
};
RcclApi g_api;
std::once_flag g_api_once;
// the communicator of the calling thread: one per process in the one-process-per-GPU mode, one per
// device thread in the in-process multi-GPU mode (node_team.hip)
struct RcclComm { ncclComm_t comm = nullptr; int rank = -1, world = 0; };
thread_local RcclComm g_comm;
struct { RcclApi *operator->() { std::call_once(g_api_once, [] { g_api.load(); }); return &g_api; } } g_rccl;
bool api_ready() { return g_rccl->handle != nullptr; }

bool check(ncclResult_t r, const char *what)
{
    if (r == ncclSuccess) return true;
    fprintf(stderr, "[starneig-amd] RCCL %s failed: %s\n", what, g_rccl->GetErrorString ? g_rccl->GetErrorString(r) : "?");
    return false;
}

} // namespace

int rccl_unique_id(void *id128)
{
    if (!api_ready()) return 1;
    ncclUniqueId id;
    if (!check(g_rccl->GetUniqueId(&id), "ncclGetUniqueId")) return 2;
    std::memcpy(id128, &id, sizeof id);
    return 0;
}

int rccl_init(int rank, int world, void const *id128)
{
    if (!api_ready()) return 1;
    if (g_comm.comm) { g_rccl->CommDestroy(g_comm.comm); g_comm.comm = nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    if (!check(g_rccl->CommInitRank(&g_comm.comm, world, id, rank), "ncclCommInitRank")) { g_comm.comm = nullptr; return 2; }
    g_comm.rank = rank; g_comm.world = world;
    return 0;
}

void rccl_finalize()
{
    if (g_comm.comm) { g_rccl->CommDestroy(g_comm.comm); g_comm.comm = nullptr; }
    g_comm.rank = -1; g_comm.world = 0;
}

// ranks of the calling thread's communicator as RCCL itself reports them (ncclCommCount), 0 without one
int rccl_comm_count()
{
    if (!g_comm.comm) return 0;
    auto count = (ncclResult_t (*)(ncclComm_t, int *))dlsym(g_rccl->handle, "ncclCommCount");
    int c = 0;
    if (!count || count(g_comm.comm, &c) != ncclSuccess) return -1;
    return c;
}

bool rccl_ready(int rank, int world) { return g_comm.comm && g_comm.rank == rank && g_comm.world == world; }

int rccl_allreduce_sum(double *buf, long count, hipStream_t s)
{
    if (!g_comm.comm) return 1;
    return check(g_rccl->AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, g_comm.comm, s), "ncclAllReduce") ? 0 : 2;
}

int rccl_broadcast(double *buf, long count, int root, hipStream_t s)
{
    if (!g_comm.comm) return 1;
    return check(g_rccl->Broadcast(buf, buf, (size_t)count, ncclDouble, root, g_comm.comm, s), "ncclBroadcast") ? 0 : 2;
}

} // namespace sn
