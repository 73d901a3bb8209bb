// tests/host_shim/stub_rccl.cpp -- an in-process stand-in for librccl (built as libstub_rccl.so and named through MKZG_RCCL_LIB) for
// driving the library's exchange step (go-eth-kzg_amd/csrc/api_exchange.hip) on a CPU under ThreadSanitizer: the "ranks" are threads
// of one process, device memory is the stub HIP runtime's host memory, and a collective completes inside the call once every rank of
// the communicator has arrived.  Only the six entry points the library resolves.  Test infrastructure only.
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {
struct Group {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long gen = 0;
    std::vector<const void*> send;
    // runs `copy` on every rank once all have published their send pointer; nobody leaves before everybody has copied
    template <class F>
    void collective(int rank, const void* sendbuf, F copy) {
        std::unique_lock<std::mutex> lk(m);
        send[(size_t)rank] = sendbuf;
        unsigned long g = gen;
        if (++arrived == world) { arrived = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
        copy(send);                                   // under the lock: the sends are stable until the second barrier below
        g = gen;
        if (++arrived == world) { arrived = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};
struct Comm { Group* g; int rank; };
std::mutex g_mu;
std::map<long, Group*> g_groups;
long g_next_id = 1;
}  // namespace

extern "C" {
struct ncclUniqueId { char internal[128]; };
int ncclGetUniqueId(ncclUniqueId* id) {
    std::lock_guard<std::mutex> l(g_mu);
    memset(id, 0, sizeof(*id));
    long v = g_next_id++;
    memcpy(id->internal, &v, sizeof(v));
    return 0;
}
int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
    long v;
    memcpy(&v, id.internal, sizeof(v));
    Group* g;
    {
        std::lock_guard<std::mutex> l(g_mu);
        Group*& slot = g_groups[v];
        if (!slot) { slot = new Group(); slot->world = nranks; slot->send.resize((size_t)nranks); }
        g = slot;
    }
    if (g->world != nranks || rank < 0 || rank >= nranks) return 4;      // ncclInvalidArgument
    *comm = new Comm{g, rank};
    g->collective(rank, nullptr, [](const std::vector<const void*>&) {});   // ncclCommInitRank is itself collective
    return 0;
}
int ncclCommDestroy(void* comm) { delete (Comm*)comm; return 0; }
int ncclAllGather(const void* send, void* recv, size_t count, int /*datatype: bytes only*/, void* comm, void* /*stream*/) {
    Comm* c = (Comm*)comm;
    c->g->collective(c->rank, send, [&](const std::vector<const void*>& all) {
        for (int k = 0; k < c->g->world; k++) memcpy((char*)recv + (size_t)k * count, all[(size_t)k], count);
    });
    return 0;
}
int ncclBroadcast(const void* send, void* recv, size_t count, int, int root, void* comm, void*) {
    Comm* c = (Comm*)comm;
    c->g->collective(c->rank, send, [&](const std::vector<const void*>& all) {
        if (recv != all[(size_t)root]) memcpy(recv, all[(size_t)root], count);
    });
    return 0;
}
const char* ncclGetErrorString(int) { return "stub rccl"; }
}
