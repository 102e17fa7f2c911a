/*
 * smvs_rccl.h -- the one collective of the multi-GPU configuration
 * (BASELINE.json configs[4]): the all-reduce of the GlobalLighting normal
 * equations over the reference views of a lock-step round, natively over RCCL
 * (xGMI), no PyTorch in the data path.  libsmvs_rccl.so links libsmvs_hip.so
 * and librccl.
 *
 * The reference fits the lighting per reference view
 * (lib/depth_optimizer.cc:110-117, lib/light_optimizer.cc:22-55): "shared
 * lighting" is an extension, off by default.  Reference
 * views are otherwise independent: nothing else crosses GPUs.
 *
 * One communicator per process (= per GPU).  Bootstrap like every NCCL
 * program: rank 0 calls smvs_comm_unique_id and hands the 128 bytes to the
 * other ranks by whatever channel the launcher offers (a file, MPI, a
 * torch.distributed store); every rank then calls smvs_comm_create.
 */
#ifndef SMVS_RCCL_H
#define SMVS_RCCL_H

#include "smvs_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smvs_comm smvs_comm;

#define SMVS_COMM_ID_BYTES 128
int smvs_comm_unique_id(void *id128);
int smvs_comm_create(int device, int rank, int world, const void *id128,
    smvs_comm **out);
int smvs_comm_destroy(smvs_comm *comm);
/* What RCCL itself says about the communicator: ncclCommCount and
 * ncclCommUserRank -- the evidence a multi-GPU run records that N ranks really
 * joined (bench.py: "n_ranks_seen_by_rccl"). */
int smvs_comm_ranks(smvs_comm *comm, int *num_ranks, int *this_rank);

/* A (16 x 16) and b (16) of LightOptimizer::fit_lighting_to_image
 * (light_optimizer.cc:32-49) as smvs_light_accumulate_dev left them in the
 * contexts' device buffers: summed over the n local contexts on the device,
 * all-reduced (sum) over the ranks in place -- 272 doubles, latency-bound --
 * and written back to every context's buffer, so that smvs_light_download of
 * each returns the total.  comm may be NULL (single process: local sum only).
 * All contexts live on the communicator's device. */
int smvs_light_allreduce(smvs_comm *comm, smvs_ctx *const *ctxs, int n);

#ifdef __cplusplus
}
#endif
#endif /* SMVS_RCCL_H */
