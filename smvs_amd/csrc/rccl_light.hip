// The lighting all-reduce over RCCL (include/smvs_rccl.h): local device sum of
// the views' normal equations, ncclAllReduce in place, copy back.
#include "common.h"
#include "../../include/smvs_rccl.h"

#include <rccl/rccl.h>

struct smvs_comm {
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, world = 1;
    hipStream_t stream = nullptr;
    double *sum = nullptr;      // 272 doubles
};

namespace smvs_hip {

constexpr int LIGHT_DOUBLES = 272;

struct LightPointers {
    double *p[64];
};

__global__ void
light_sum_kernel(LightPointers bufs, int n, double *out)
{
    int const i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= LIGHT_DOUBLES)
        return;
    double s = 0.0;
    for (int k = 0; k < n; ++k)   // fixed order: deterministic
        s += bufs.p[k][i];
    out[i] = s;
}

__global__ void
light_scatter_kernel(const double *sum, LightPointers bufs, int n)
{
    int const i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= LIGHT_DOUBLES)
        return;
    double const s = sum[i];
    for (int k = 0; k < n; ++k)
        bufs.p[k][i] = s;
}

#define SMVS_NCCL_CHECK(expr)                                                 \
    do {                                                                      \
        ncclResult_t r__ = (expr);                                            \
        if (r__ != ncclSuccess) {                                             \
            smvs_hip::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, \
                ncclGetErrorString(r__));                                     \
            return SMVS_ERR_HIP;                                              \
        }                                                                     \
    } while (0)

} // namespace smvs_hip

using namespace smvs_hip;

extern "C" int
smvs_comm_unique_id(void *id128)
{
    SMVS_REQUIRE(id128 != nullptr, "null argument");
    static_assert(sizeof(ncclUniqueId) == SMVS_COMM_ID_BYTES, "id size");
    ncclUniqueId id;
    SMVS_NCCL_CHECK(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return SMVS_OK;
}

extern "C" int
smvs_comm_create(int device, int rank, int world, const void *id128,
    smvs_comm **out)
{
    SMVS_REQUIRE(out != nullptr && id128 != nullptr, "null argument");
    SMVS_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank / world size");
    int const count = logical_device_count();
    SMVS_REQUIRE(device >= 0 && device < count, "no such HIP device");
    SMVS_HIP_CHECK(set_device(device));
    smvs_comm *c = new smvs_comm();
    c->device = device;
    c->rank = rank;
    c->world = world;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t const r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank: %s", ncclGetErrorString(r));
        delete c;
        return SMVS_ERR_HIP;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess
        || hipMalloc((void **)&c->sum, LIGHT_DOUBLES * sizeof(double)) != hipSuccess) {
        set_error("smvs_comm_create: stream / buffer allocation failed");
        smvs_comm_destroy(c);
        return SMVS_ERR_HIP;
    }
    *out = c;
    return SMVS_OK;
}

extern "C" int
smvs_comm_destroy(smvs_comm *c)
{
    if (c == nullptr)
        return SMVS_OK;
    (void)set_device(c->device);
    if (c->stream != nullptr)
        (void)hipStreamSynchronize(c->stream);
    if (c->comm != nullptr)
        (void)ncclCommDestroy(c->comm);
    if (c->sum != nullptr)
        (void)hipFree(c->sum);
    if (c->stream != nullptr)
        (void)hipStreamDestroy(c->stream);
    delete c;
    return SMVS_OK;
}

extern "C" int
smvs_comm_ranks(smvs_comm *c, int *num_ranks, int *this_rank)
{
    SMVS_REQUIRE(c != nullptr && c->comm != nullptr, "null communicator");
    int count = 0, rank = -1;
    SMVS_NCCL_CHECK(ncclCommCount(c->comm, &count));
    SMVS_NCCL_CHECK(ncclCommUserRank(c->comm, &rank));
    if (num_ranks != nullptr)
        *num_ranks = count;
    if (this_rank != nullptr)
        *this_rank = rank;
    return SMVS_OK;
}

extern "C" int
smvs_light_allreduce(smvs_comm *comm, smvs_ctx *const *ctxs, int n)
{
    SMVS_REQUIRE(ctxs != nullptr && n >= 1 && n <= 64, "1 .. 64 contexts");
    LightPointers bufs;
    for (int k = 0; k < n; ++k) {
        SMVS_REQUIRE(ctxs[k] != nullptr && ctxs[k]->lightAb != nullptr, "null context");
        SMVS_REQUIRE(ctxs[k]->device == ctxs[0]->device
            && (comm == nullptr || ctxs[k]->device == comm->device),
            "the contexts live on the communicator's device");
        bufs.p[k] = ctxs[k]->lightAb;
    }
    SMVS_HIP_CHECK(set_device(ctxs[0]->device));
    // the buffers were left by smvs_light_accumulate_dev, which returns with
    // the contexts' streams idle
    hipStream_t const stream = comm != nullptr ? comm->stream : ctxs[0]->stream;
    double *sum = comm != nullptr ? comm->sum : bufs.p[0];
    if (comm != nullptr || n > 1) {
        if (comm == nullptr) {
            // (no scratch without a communicator: sum into a copy of the first)
            SMVS_HIP_CHECK(hipMalloc((void **)&sum, LIGHT_DOUBLES * sizeof(double)));
        }
        hipLaunchKernelGGL(light_sum_kernel, dim3(2), dim3(256), 0, stream, bufs, n, sum);
        SMVS_HIP_CHECK(hipGetLastError());
        if (comm != nullptr && comm->world > 1)
            SMVS_NCCL_CHECK(ncclAllReduce(sum, sum, LIGHT_DOUBLES, ncclDouble, ncclSum,
                comm->comm, stream));
        hipLaunchKernelGGL(light_scatter_kernel, dim3(2), dim3(256), 0, stream, sum, bufs, n);
        SMVS_HIP_CHECK(hipGetLastError());
        SMVS_HIP_CHECK(hipStreamSynchronize(stream));
        if (comm == nullptr)
            (void)hipFree(sum);
    }
    return SMVS_OK;
}
