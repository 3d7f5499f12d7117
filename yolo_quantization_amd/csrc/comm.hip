// comm.hip -- the one collective of the path: a one-shot RCCL broadcast of the packed quantized weights over xGMI at
// start-up.  The reference has no inference-time communication at all (its multi-GPU code is host-staged
// weight averaging for training, ref: src/network.c:1100-1194), so there is nothing to translate: images shard by rank, every
// device holds a full replica of the packed blobs, and this is how the replica gets there.
//
// librccl is bound at run time (dlopen): the library has no link-time dependency on it, and in a process that already
// carries an RCCL (PyTorch bundles one) the loaded copy is reused instead of a second one being mapped.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../../include/mi355_yolo_int8.h"

namespace {
typedef struct { char internal[128]; } rcclUniqueId;   // NCCL_UNIQUE_ID_BYTES
typedef void *rcclComm;
typedef int (*fn_get_unique_id)(rcclUniqueId *);
typedef int (*fn_comm_init_rank)(rcclComm *, int, rcclUniqueId, int);
typedef int (*fn_comm_destroy)(rcclComm);
typedef int (*fn_broadcast)(const void *, void *, size_t, int /*ncclDataType_t*/, int, rcclComm, hipStream_t);
typedef const char *(*fn_error_string)(int);

struct Rccl {
    void *h = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_broadcast broadcast = nullptr;
    fn_error_string error_string = nullptr;
};
Rccl g_rccl;
thread_local char g_comm_err[384] = "";

int load_rccl()
{
    if (g_rccl.h) return MI355_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // a copy the process already mapped wins
        if (h) break;
    }
    for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        snprintf(g_comm_err, sizeof(g_comm_err), "librccl not found: %s", dlerror());
        return MI355_ENODEV;
    }
    Rccl r;
    r.h = h;
    r.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    r.broadcast = (fn_broadcast)dlsym(h, "ncclBroadcast");
    r.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.broadcast) {
        snprintf(g_comm_err, sizeof(g_comm_err), "librccl lacks the ncclGetUniqueId / ncclCommInitRank / ncclBroadcast entry points");
        return MI355_ENODEV;
    }
    g_rccl = r;
    return MI355_OK;
}

int rccl_fail(int rc, const char *what)
{
    snprintf(g_comm_err, sizeof(g_comm_err), "%s: RCCL error %d (%s)", what, rc,
             g_rccl.error_string ? g_rccl.error_string(rc) : "?");
    return MI355_EHIP;
}
}  // namespace

extern "C" {

const char *mi355_comm_last_error(void) { return g_comm_err; }

int mi355_comm_unique_id(void *id128)
{
    if (!id128) return MI355_EINVAL;
    int rc = load_rccl();
    if (rc) return rc;
    rcclUniqueId id;
    rc = g_rccl.get_unique_id(&id);
    if (rc) return rccl_fail(rc, "ncclGetUniqueId");
    memcpy(id128, &id, sizeof(id));
    return MI355_OK;
}

int mi355_comm_init(void **comm, int nranks, const void *id128, int rank)
{
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return MI355_EINVAL;
    int rc = load_rccl();
    if (rc) return rc;
    rcclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    rcclComm c = nullptr;
    rc = g_rccl.comm_init_rank(&c, nranks, id, rank);  // binds the calling thread's current device (mi355_init)
    if (rc) return rccl_fail(rc, "ncclCommInitRank");
    *comm = c;
    return MI355_OK;
}

int mi355_bcast_blob(void *comm, void *dev_buf, size_t bytes, int root, void *stream)
{
    if (!comm || !dev_buf || !bytes) return MI355_EINVAL;
    if (!g_rccl.h) return MI355_EINVAL;
    const int rc = g_rccl.broadcast(dev_buf, dev_buf, bytes, 1 /* ncclUint8 */, root, (rcclComm)comm, (hipStream_t)stream);
    if (rc) return rccl_fail(rc, "ncclBroadcast");
    return MI355_OK;
}

int mi355_comm_destroy(void *comm)
{
    if (!comm) return MI355_OK;
    if (!g_rccl.h) return MI355_EINVAL;
    const int rc = g_rccl.comm_destroy((rcclComm)comm);
    if (rc) return rccl_fail(rc, "ncclCommDestroy");
    return MI355_OK;
}

}  // extern "C"
