/*
 * mrt_gather.h -- C ABI of the multi-GPU exchange step: the interleaved tile-row shares of one frame,
 * rendered by one process per GPU (mrt_render_device_part in mrt.h), are gathered into rank 0's frame
 * with RCCL point-to-point transfers over xGMI.
 *
 * The reference is a single process whose worker threads write one shared Surface
 * (threads.cpp:215-224 stores straight into screen->GetBuffer()), so it has no exchange step; this is
 * what takes the place of that shared framebuffer when the workers are GPUs.  The split itself follows
 * the reference's tile queue (threads.cpp:158-169) at the grain of 8-row tile rows: share `rank` of
 * `n_ranks` holds the tile rows R with (R / block_tile_rows) % n_ranks == rank.
 *
 * Library: libmrt_gather.so (links librccl).  libmrt.so itself does not depend on RCCL.
 * Every function returns 0 or a negative MRT_E_* code of mrt.h; mrt_gather_last_error gives the text.
 */
#ifndef MRT_GATHER_H_
#define MRT_GATHER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mrt_gather mrt_gather;

/* One per process, on the current GPU.  Rank 0 allocates a receive buffer per peer share (device memory,
 * share rows x width pixels); the other ranks allocate nothing. */
int  mrt_gather_create(int width, int height, int rank, int n_ranks, int block_tile_rows, mrt_gather** out);
void mrt_gather_destroy(mrt_gather* g);
const char* mrt_gather_last_error(const mrt_gather* g);   /* g may be NULL: last error of mrt_gather_create */

/* Pixel rows of a rank's share (equals mrt_part_rows of mrt.h). */
int mrt_gather_share_rows(const mrt_gather* g, int rank);

/* Frame row that holds row `packed_row` of rank's packed share; negative on bad arguments.  Host only. */
int mrt_gather_frame_row(int height, int n_ranks, int block_tile_rows, int rank, int packed_row);

/* The exchange, queued on `stream` (hipStream_t) without synchronising.  comm is an ncclComm_t of n_ranks
 * ranks in which this process has the rank given at creation.
 *   rank 0 : buf_dev is the whole frame (height rows of pitch_px pixels) with its own share already
 *            rendered in place (compact = 0); it receives every other share in one ncclGroupStart /
 *            ncclRecv x (n_ranks - 1) / ncclGroupEnd and scatters the rows to their places.
 *   others : buf_dev is the packed share (compact = 1, pitch_px must equal width): one ncclSend.
 * Shares without rows are skipped on both sides. */
int mrt_gather_exchange(mrt_gather* g, void* nccl_comm, void* buf_dev, int pitch_px, void* stream);

/* Whole frames, one per rank, gathered on rank 0 (a frame loop whose frames are too small to cut: every GPU renders
 * a frame of its own).  Every rank passes the frame it rendered (height rows of pitch_px pixels; on ranks other than 0
 * pitch_px must equal width: the frame is one ncclSend); rank 0 also passes frames_dev, n_ranks x height x width
 * pixels, whose slot 0 receives a copy of its own frame and slot r the frame of rank r (one group of ncclRecv).
 * Queued on `stream` without synchronising. */
int mrt_gather_frames(mrt_gather* g, void* nccl_comm, const void* frame_dev, int pitch_px, void* frames_dev, void* stream);

/* The communicator, for hosts that do not link RCCL themselves (a Python host through ctypes) and so that every host
 * creates it the same way: rank 0 makes an id (MRT_GATHER_ID_BYTES bytes, RCCL's ncclUniqueId) and hands it to the
 * other ranks by whatever means the launcher offers (a file, a key-value store); every rank then calls
 * mrt_gather_comm_create with the GPU it renders on current.  The calls block until all n_ranks ranks have joined. */
#define MRT_GATHER_ID_BYTES 128
int  mrt_gather_comm_id(void* id_out);
int  mrt_gather_comm_create(const void* id_in, int rank, int n_ranks, void** comm_out);
void mrt_gather_comm_destroy(void* comm);

/* The two halves of rank 0's side on their own, for transports other than RCCL (a single process
 * rehearsing the ranks, tests): the device buffer that receives `peer`'s packed share, and the kernel that
 * copies its rows into the frame.  mrt_gather_stage_local is the transport of a one-process rehearsal: it copies a
 * packed share that lies in this process's device memory into the peer's receive buffer. */
void* mrt_gather_staging(mrt_gather* g, int peer);
int   mrt_gather_stage_local(mrt_gather* g, int peer, const void* packed_dev, void* stream);   /* device-to-device copy into that buffer */
int   mrt_gather_scatter(mrt_gather* g, int peer, void* frame_dev, int pitch_px, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRT_GATHER_H_ */
