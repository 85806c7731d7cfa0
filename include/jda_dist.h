/* libjda_dist.so -- the one exchange step of the multi-GPU detect path: the gather of final
 * (bbox, score, landmarks) tuples on rank 0 over RCCL (xGMI inside a node).
 *
 * Frames shard across GPUs with no data-path collective (every frame and window is independent, the
 * model is replicated); the only exchange is this gather.  The reference has no
 * distributed layer at all -- its only parallel form is an OpenMP loop over FDDB folds
 * (reference src/test.cpp:100) -- so nothing here replaces a reference symbol: the entry points
 * are additive, for C callers of include/jda.h that run one process per GPU.
 *
 * Kept in its own library so that libjda.so depends on the HIP runtime only.  Plain C types.
 */
#ifndef JDA_DIST_H_
#define JDA_DIST_H_

#include "jda.h"

#ifdef __cplusplus
extern "C" {
#endif

#define JDA_DIST_ID_BYTES 128

/* Rank 0 creates the rendezvous id (ncclGetUniqueId) and hands its 128 bytes to the other ranks by any
 * means it has (a file, MPI, torch.distributed broadcast...).  Returns 0 on success. */
JDA_API int jdaDistUniqueId(unsigned char id[JDA_DIST_ID_BYTES]);

/* One communicator per process, on HIP device `device` (ncclCommInitRank).  row_floats = floats per
 * detection row (5 + 2*landmark_n, the row format of jdaResultsPack); block_rows = rows per rank a
 * pipelined gather carries without a second exchange.  NULL on failure (jdaDistLastError). */
JDA_API void *jdaDistCreate(int rank, int world, const unsigned char id[JDA_DIST_ID_BYTES], int device,
                            int row_floats, int block_rows);
JDA_API void jdaDistDestroy(void *dist);
JDA_API const char *jdaDistLastError(void);

/* Gathers every rank's rows on rank 0, in rank order: counts by ncclAllGather, then grouped
 * ncclSend / ncclRecv of exactly the rows.  On rank 0 *all_rows is malloc'ed (free with jdaDistFree)
 * and *n_all its row count; on the other ranks *all_rows = NULL, *n_all = 0.  Blocking. */
JDA_API int jdaDistGatherRows(void *dist, const float *rows, int n_rows, float **all_rows, int *n_all);

/* The same for the per-frame results of a jdaDetectBatch* call (packed in jdaResultsPack's row format:
 * [frame_offset + i, x, y, size, score, shape...]). */
JDA_API int jdaGatherResults(void *dist, const jdaResult *results, int n, int frame_offset,
                             float **all_rows, int *n_all);

/* Pipelined form, one collective per step and no host wait at Start: every rank contributes a fixed
 * block of (1 + block_rows) rows whose first row carries its count (one ncclAllGather on the
 * communicator's own stream); Collect finishes the OLDEST started gather (at most two may be in
 * flight) and returns its rows like jdaDistGatherRows.  A rank with more than block_rows rows makes
 * every rank fall back to the exact two-step exchange for that gather (the counts are gathered, so
 * all ranks agree).  That fallback queues further collectives inside Collect, so -- as for any collective --
 * EVERY rank must call Start and Collect (and jdaDistGatherRows / jdaGatherResults) the same number of times in
 * the same order; a rank that skips or reorders a call hangs the others.  Counts travel as fp32 (exact up to
 * 2^24 rows per rank and step). */
JDA_API int jdaDistGatherStart(void *dist, const float *rows, int n_rows);
JDA_API int jdaDistGatherCollect(void *dist, float **all_rows, int *n_all);
JDA_API int jdaDistPending(void *dist);

JDA_API void jdaDistFree(float *rows);

#ifdef __cplusplus
}
#endif
#endif
