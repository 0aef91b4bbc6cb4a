// comm.hip -- RCCL transport of the row-band protocol, inside the library.
//
// A band context that was created with an ncclUniqueId owns one RCCL communicator over all bands (one rank per
// GPU, `ncclCommInitRank`).  The only data-path traffic is the neighbour exchange of edge rows (a few hundred KB per
// message, latency bound: one of the seven xGMI links per neighbour) plus a one-word all-reduce that ends the fill
// loops; both are enqueued on the stream the band's kernels run on, so they are ordered with the kernels without a
// host hop.  Rows are sent straight out of the rasters (an edge row is contiguous) and received into a small staging
// buffer; a compare-and-store kernel moves them into the halo row and reports whether anything changed.
//
// librccl is opened at run time (dlopen): single-GPU users never load it, and a process that has already loaded
// another copy of the library (a launcher that imported torch) shares that copy instead of initialising a second one.
#include "common.hpp"

#include <dlfcn.h>
#include <mutex>

namespace mh {

namespace {

// the few RCCL entry points the band protocol needs (signatures of rccl/rccl.h)
typedef struct { char internal[128]; } UniqueId;   // ncclUniqueId, NCCL_UNIQUE_ID_BYTES == 128
typedef void *Comm;                                  // ncclComm_t
enum { kNcclSuccess = 0 };
enum { kNcclUint8 = 1, kNcclFloat64 = 8 };           // ncclDataType_t: ncclUint8 = 1, ncclFloat64 = ncclDouble = 8
enum { kNcclMax = 2 };                               // ncclRedOp_t: sum 0, prod 1, max 2, min 3

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?");
            return;
        }
        auto sym = [&](const char *s) {
            void *p = dlsym(r.handle, s);
            if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + s;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &r;
}

int need_rccl(Rccl **out)
{
    Rccl *r = rccl();
    if (!r->handle || !r->why.empty()) {
        set_error("RCCL unavailable: %s", r->why.c_str());
        return MHIP_ECOMM;
    }
    *out = r;
    return MHIP_OK;
}

#define MH_NCCL(r, expr)                                                                                 \
    do {                                                                                                 \
        int _e = (expr);                                                                                 \
        if (_e != kNcclSuccess) {                                                                        \
            mh::set_error("%s failed: %s (%s:%d)", #expr, (r)->GetErrorString ? (r)->GetErrorString(_e) : "?", __FILE__, __LINE__); \
            return MHIP_ECOMM;                                                                           \
        }                                                                                                \
    } while (0)

}  // namespace

// cheap probe (dlopen + symbols only): every rank votes on it BEFORE any rank enters ncclCommInitRank, which would wait for
// a rank that cannot join
int comm_available()
{
    Rccl *r;
    return need_rccl(&r) == MHIP_OK ? 1 : 0;
}

int comm_unique_id(void *id128)
{
    Rccl *r;
    MH_TRY(need_rccl(&r));
    UniqueId id;
    MH_NCCL(r, r->GetUniqueId(&id));
    memcpy(id128, id.internal, sizeof(id.internal));
    return MHIP_OK;
}

// collective over all `nranks` bands (every rank calls it with rank 0's id); the device must be current
int comm_create(void **comm, const void *id128, int rank, int nranks)
{
    Rccl *r;
    MH_TRY(need_rccl(&r));
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    Comm c = nullptr;
    MH_NCCL(r, r->CommInitRank(&c, nranks, id, rank));
    *comm = c;
    return MHIP_OK;
}

void comm_destroy(void *comm)
{
    Rccl *r = rccl();
    if (comm && r->CommDestroy) (void)r->CommDestroy(comm);
}

// Neighbour exchange of one raster's edge rows, all on stream `s`: my first owned row goes to rank-1 and my last owned
// row to rank+1; their rows arrive in `stage` (2 * rowbytes: [from above | from below]).  Pointers are device memory.
int comm_exchange_rows(void *comm, int rank, int nranks, const void *first_row, const void *last_row, void *stage, size_t rowbytes,
                       hipStream_t s)
{
    Rccl *r;
    MH_TRY(need_rccl(&r));
    const bool up = rank > 0, down = rank < nranks - 1;
    if (!up && !down) return MHIP_OK;
    MH_NCCL(r, r->GroupStart());
    // every exit closes the group: a communicator left in group mode would swallow the next collective (the vote)
    int first_err = kNcclSuccess;
    const char *what = "";
    auto step = [&](int e, const char *w) {
        if (e != kNcclSuccess && first_err == kNcclSuccess) {
            first_err = e;
            what = w;
        }
    };
    if (up) {
        step(r->Send(first_row, rowbytes, kNcclUint8, rank - 1, comm, s), "ncclSend to the band above");
        if (first_err == kNcclSuccess) step(r->Recv(stage, rowbytes, kNcclUint8, rank - 1, comm, s), "ncclRecv from the band above");
    }
    if (down && first_err == kNcclSuccess) {
        step(r->Send(last_row, rowbytes, kNcclUint8, rank + 1, comm, s), "ncclSend to the band below");
        if (first_err == kNcclSuccess)
            step(r->Recv(static_cast<char *>(stage) + rowbytes, rowbytes, kNcclUint8, rank + 1, comm, s), "ncclRecv from the band below");
    }
    const int end_err = r->GroupEnd();
    if (first_err != kNcclSuccess || end_err != kNcclSuccess) {
        const int e = first_err != kNcclSuccess ? first_err : end_err;
        set_error("%s failed: %s", first_err != kNcclSuccess ? what : "ncclGroupEnd", r->GetErrorString ? r->GetErrorString(e) : "?");
        return MHIP_ECOMM;
    }
    return MHIP_OK;
}

// max over all bands of one double (in place on a device word), on stream `s`
int comm_allreduce_max(void *comm, double *d_value, hipStream_t s)
{
    Rccl *r;
    MH_TRY(need_rccl(&r));
    MH_NCCL(r, r->AllReduce(d_value, d_value, 1, kNcclFloat64, kNcclMax, comm, s));
    return MHIP_OK;
}

}  // namespace mh
