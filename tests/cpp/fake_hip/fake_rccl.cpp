// fake_rccl.cpp -> librccl.so.1 for the sanitizer builds: a communicator of ONE rank in one process.  Sends and receives to oneself
// are matched at ncclGroupEnd; ncclAllGather of one rank is a copy.  FAKE_RCCL_FAIL_AT=<k> makes the k-th call fail.
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <vector>

struct fake_nccl_comm { int world, rank; };
namespace {
long g_calls = 0;
ncclResult_t tick() { const char* e = getenv("FAKE_RCCL_FAIL_AT"); ++g_calls; return (e && atol(e) == g_calls) ? ncclInternalError : ncclSuccess; }
struct Pending { const void* src; void* dst; size_t bytes; };
std::vector<Pending> g_sends, g_recvs;
size_t width(ncclDataType_t t) { return t == ncclInt8 ? 1 : 4; }
}
extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { if (tick() != ncclSuccess) return ncclInternalError; memset(id, 7, sizeof *id); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t* c, int world, ncclUniqueId, int rank)
{
    *c = nullptr;
    if (tick() != ncclSuccess) return ncclInternalError;
    if (world != 1 || rank != 0) return ncclInvalidArgument;
    *c = new fake_nccl_comm{world, rank};
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { if (tick() != ncclSuccess) return ncclInternalError; *n = c->world; return ncclSuccess; }
ncclResult_t ncclAllGather(const void* s, void* d, size_t n, ncclDataType_t t, ncclComm_t, hipStream_t) { if (tick() != ncclSuccess) return ncclInternalError; if (s != d) memmove(d, s, n * width(t)); return ncclSuccess; }
ncclResult_t ncclGroupStart(void) { if (tick() != ncclSuccess) return ncclInternalError; g_sends.clear(); g_recvs.clear(); return ncclSuccess; }
ncclResult_t ncclSend(const void* s, size_t n, ncclDataType_t t, int, ncclComm_t, hipStream_t) { if (tick() != ncclSuccess) return ncclInternalError; g_sends.push_back({s, nullptr, n * width(t)}); return ncclSuccess; }
ncclResult_t ncclRecv(void* d, size_t n, ncclDataType_t t, int, ncclComm_t, hipStream_t) { if (tick() != ncclSuccess) return ncclInternalError; g_recvs.push_back({nullptr, d, n * width(t)}); return ncclSuccess; }
ncclResult_t ncclGroupEnd(void)
{
    const ncclResult_t r = tick();
    for (size_t i = 0; i < g_sends.size() && i < g_recvs.size(); i++) memcpy(g_recvs[i].dst, g_sends[i].src, g_sends[i].bytes < g_recvs[i].bytes ? g_sends[i].bytes : g_recvs[i].bytes);
    g_sends.clear(); g_recvs.clear();
    return r;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "injected failure (fake rccl)"; }
}
extern "C" void fake_rccl_reset(void) { g_calls = 0; g_sends.clear(); g_recvs.clear(); }
