// Multi-GPU exchange step (include/mrt_gather.h): packed tile-row shares -> rank 0's frame.
// RCCL point-to-point only: every peer sends one contiguous buffer, rank 0 posts all receives in one
// group (each transfer rides its own xGMI link into rank 0) and a small kernel puts the rows in place.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <string.h>

#include <string>
#include <vector>

#include "../../include/mrt.h"
#include "../../include/mrt_gather.h"

namespace {
thread_local std::string g_create_error;

// tile rows (8 pixel rows each) of share `part`: those R < rows with (R / block) % n_parts == part
int share_tile_rows(int rows, int part, int n_parts, int block) {
    const int cycle = block * n_parts;
    const int full = rows / cycle, rest = rows - full * cycle;
    const int extra = rest - part * block;
    return full * block + (extra < 0 ? 0 : (extra > block ? block : extra));
}
__host__ __device__ inline int frame_row_of(int packed_row, int part, int n_parts, int block) {
    const int lr = packed_row >> 3, blk = lr / block;
    return (((blk * n_parts + part) * block + (lr - blk * block)) << 3) + (packed_row & 7);
}

// 16 bytes per thread when rows are 16-byte aligned (vec = 4 pixels), else one pixel per thread
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows(const T* packed, T* frame, int row_items, size_t pitch_items,
                                                    int n_rows, int part, int n_parts, int block) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n_rows * row_items) return;
    const int k = (int)(i / row_items), x = (int)(i - (size_t)k * row_items);
    frame[(size_t)frame_row_of(k, part, n_parts, block) * pitch_items + x] = packed[i];
}
}  // namespace

struct mrt_gather {
    int width = 0, height = 0, rank = 0, n_ranks = 1, block = 1;
    std::vector<uint32_t*> staging;     // rank 0: one per peer (null for empty shares and for itself)
    std::vector<int> rows;              // pixel rows of every share
    std::string err;
};

static int fail(mrt_gather* g, int code, const std::string& msg) {
    if (g) g->err = msg; else g_create_error = msg;
    return code;
}

extern "C" {

const char* mrt_gather_last_error(const mrt_gather* g) { return g ? g->err.c_str() : g_create_error.c_str(); }

int mrt_gather_frame_row(int height, int n_ranks, int block, int rank, int packed_row) {
    if (height <= 0 || (height % 8) || n_ranks < 1 || rank < 0 || rank >= n_ranks || block < 1 || packed_row < 0) return MRT_E_ARG;
    if (packed_row >= 8 * share_tile_rows(height / 8, rank, n_ranks, block)) return MRT_E_ARG;
    return frame_row_of(packed_row, rank, n_ranks, block);
}

int mrt_gather_create(int width, int height, int rank, int n_ranks, int block, mrt_gather** out) {
    if (!out) return fail(nullptr, MRT_E_ARG, "out is null");
    *out = nullptr;
    if (width <= 0 || height <= 0 || (width % 8) || (height % 8)) return fail(nullptr, MRT_E_ARG, "width and height must be positive multiples of 8");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks || block < 1) return fail(nullptr, MRT_E_ARG, "need 0 <= rank < n_ranks and block_tile_rows >= 1");
    mrt_gather* g = new mrt_gather();
    g->width = width; g->height = height; g->rank = rank; g->n_ranks = n_ranks; g->block = block;
    g->rows.resize(n_ranks);
    for (int r = 0; r < n_ranks; ++r) g->rows[r] = 8 * share_tile_rows(height / 8, r, n_ranks, block);
    g->staging.assign(n_ranks, nullptr);
    if (rank == 0)
        for (int r = 1; r < n_ranks; ++r) {
            if (!g->rows[r]) continue;
            const hipError_t e = hipMalloc((void**)&g->staging[r], (size_t)g->rows[r] * width * sizeof(uint32_t));
            if (e != hipSuccess) {
                const std::string msg = std::string("receive buffer: ") + hipGetErrorString(e);
                mrt_gather_destroy(g);
                return fail(nullptr, MRT_E_HIP, msg);
            }
        }
    *out = g;
    return MRT_OK;
}

void mrt_gather_destroy(mrt_gather* g) {
    if (!g) return;
    for (uint32_t* p : g->staging) if (p) (void)hipFree(p);
    delete g;
}

int mrt_gather_share_rows(const mrt_gather* g, int rank) {
    if (!g || rank < 0 || rank >= g->n_ranks) return MRT_E_ARG;
    return g->rows[rank];
}

void* mrt_gather_staging(mrt_gather* g, int peer) {
    if (!g || g->rank != 0 || peer < 1 || peer >= g->n_ranks) return nullptr;
    return g->staging[peer];
}

int mrt_gather_stage_local(mrt_gather* g, int peer, const void* packed_dev, void* stream) {
    if (!g) return MRT_E_ARG;
    if (g->rank != 0) return fail(g, MRT_E_STATE, "only rank 0 receives");
    if (peer < 1 || peer >= g->n_ranks || !packed_dev) return fail(g, MRT_E_ARG, "bad peer or null share");
    if (g->rows[peer] == 0) return MRT_OK;
    const hipError_t e = hipMemcpyAsync(g->staging[peer], packed_dev, (size_t)g->rows[peer] * g->width * sizeof(uint32_t),
                                        hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return fail(g, MRT_E_HIP, std::string("share copy: ") + hipGetErrorString(e));
    return MRT_OK;
}

int mrt_gather_scatter(mrt_gather* g, int peer, void* frame_dev, int pitch_px, void* stream) {
    if (!g) return MRT_E_ARG;
    if (g->rank != 0) return fail(g, MRT_E_STATE, "only rank 0 scatters");
    if (peer < 1 || peer >= g->n_ranks || !frame_dev) return fail(g, MRT_E_ARG, "bad peer or null frame");
    if (pitch_px < g->width) return fail(g, MRT_E_ARG, "pitch smaller than width");
    const int n_rows = g->rows[peer];
    if (n_rows == 0) return MRT_OK;
    const bool wide = (pitch_px % 4) == 0 && ((uintptr_t)frame_dev % 16) == 0;
    if (wide) {
        const int items = g->width / 4;
        const size_t n = (size_t)n_rows * items;
        hipLaunchKernelGGL(scatter_rows<uint4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const uint4*)g->staging[peer], (uint4*)frame_dev, items, (size_t)(pitch_px / 4), n_rows, peer, g->n_ranks, g->block);
    } else {
        const size_t n = (size_t)n_rows * g->width;
        hipLaunchKernelGGL(scatter_rows<uint32_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const uint32_t*)g->staging[peer], (uint32_t*)frame_dev, g->width, (size_t)pitch_px, n_rows, peer, g->n_ranks, g->block);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(g, MRT_E_HIP, std::string("scatter launch: ") + hipGetErrorString(e));
    return MRT_OK;
}

int mrt_gather_exchange(mrt_gather* g, void* nccl_comm, void* buf_dev, int pitch_px, void* stream) {
    if (!g) return MRT_E_ARG;
    if (g->n_ranks == 1) return MRT_OK;
    if (!nccl_comm || !buf_dev) return fail(g, MRT_E_ARG, "null communicator or buffer");
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    hipStream_t s = (hipStream_t)stream;
    auto check = [&](ncclResult_t r, const char* what) {
        if (r == ncclSuccess) return MRT_OK;
        return fail(g, MRT_E_HIP, std::string(what) + ": " + ncclGetErrorString(r));
    };
    if (g->rank != 0) {
        if (pitch_px != g->width) return fail(g, MRT_E_ARG, "a packed share is sent as one buffer: pitch must equal width");
        if (g->rows[g->rank] == 0) return MRT_OK;
        return check(ncclSend(buf_dev, (size_t)g->rows[g->rank] * g->width, ncclUint32, 0, comm, s), "ncclSend");
    }
    if (pitch_px < g->width) return fail(g, MRT_E_ARG, "pitch smaller than width");
    if (int rc = check(ncclGroupStart(), "ncclGroupStart")) return rc;
    int rc = MRT_OK;
    for (int r = 1; r < g->n_ranks && rc == MRT_OK; ++r)
        if (g->rows[r]) rc = check(ncclRecv(g->staging[r], (size_t)g->rows[r] * g->width, ncclUint32, r, comm, s), "ncclRecv");
    const int rc_end = check(ncclGroupEnd(), "ncclGroupEnd");
    if (rc != MRT_OK) return rc;
    if (rc_end != MRT_OK) return rc_end;
    for (int r = 1; r < g->n_ranks; ++r)
        if (int rs = mrt_gather_scatter(g, r, buf_dev, pitch_px, stream)) return rs;
    return MRT_OK;
}

int mrt_gather_frames(mrt_gather* g, void* nccl_comm, const void* frame_dev, int pitch_px, void* frames_dev, void* stream) {
    if (!g) return MRT_E_ARG;
    if (!frame_dev) return fail(g, MRT_E_ARG, "null frame");
    hipStream_t s = (hipStream_t)stream;
    auto check = [&](ncclResult_t r, const char* what) {
        if (r == ncclSuccess) return MRT_OK;
        return fail(g, MRT_E_HIP, std::string(what) + ": " + ncclGetErrorString(r));
    };
    const size_t frame_px = (size_t)g->width * g->height;
    if (g->rank != 0) {
        if (!nccl_comm) return fail(g, MRT_E_ARG, "null communicator");
        if (pitch_px != g->width) return fail(g, MRT_E_ARG, "a frame is sent as one buffer: pitch must equal width");
        return check(ncclSend(frame_dev, frame_px, ncclUint32, 0, (ncclComm_t)nccl_comm, s), "ncclSend");
    }
    if (!frames_dev) return fail(g, MRT_E_ARG, "rank 0 needs the array of frames");
    if (pitch_px < g->width) return fail(g, MRT_E_ARG, "pitch smaller than width");
    const hipError_t e = hipMemcpy2DAsync(frames_dev, (size_t)g->width * 4, frame_dev, (size_t)pitch_px * 4, (size_t)g->width * 4, (size_t)g->height,
                                          hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return fail(g, MRT_E_HIP, std::string("own frame copy: ") + hipGetErrorString(e));
    if (g->n_ranks == 1) return MRT_OK;
    if (!nccl_comm) return fail(g, MRT_E_ARG, "null communicator");
    if (int rc = check(ncclGroupStart(), "ncclGroupStart")) return rc;
    int rc = MRT_OK;
    for (int r = 1; r < g->n_ranks && rc == MRT_OK; ++r)
        rc = check(ncclRecv(static_cast<uint32_t*>(frames_dev) + (size_t)r * frame_px, frame_px, ncclUint32, r, (ncclComm_t)nccl_comm, s), "ncclRecv");
    const int rc_end = check(ncclGroupEnd(), "ncclGroupEnd");
    return rc != MRT_OK ? rc : rc_end;
}

static_assert(sizeof(ncclUniqueId) == MRT_GATHER_ID_BYTES, "mrt_gather.h promises the size of RCCL's communicator id");

int mrt_gather_comm_id(void* id_out) {
    if (!id_out) return fail(nullptr, MRT_E_ARG, "id_out is null");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, MRT_E_HIP, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    memcpy(id_out, &id, sizeof id);
    return MRT_OK;
}

int mrt_gather_comm_create(const void* id_in, int rank, int n_ranks, void** comm_out) {
    if (!id_in || !comm_out) return fail(nullptr, MRT_E_ARG, "null id or comm_out");
    *comm_out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(nullptr, MRT_E_ARG, "need 0 <= rank < n_ranks");
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = ncclCommInitRank(&comm, n_ranks, id, rank);
    if (r != ncclSuccess) return fail(nullptr, MRT_E_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    *comm_out = comm;
    return MRT_OK;
}

void mrt_gather_comm_destroy(void* comm) {
    if (comm) (void)ncclCommDestroy((ncclComm_t)comm);
}

}  // extern "C"
