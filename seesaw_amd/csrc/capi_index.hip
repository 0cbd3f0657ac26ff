// capi_index.hip -- C-ABI of the resident vector index (see include/seesaw_hip.h).
#include <algorithm>
#include <cmath>
#include <chrono>
#include <vector>

#include "ssw_common.h"

namespace ssw {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int num_cus(int device) {
    static int cache[16] = {0};
    const int slot = device & 15;
    if (cache[slot] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
            v <= 0)
            v = 256;
        cache[slot] = v;
    }
    return cache[slot];
}

ssw_status PinnedStage::push(void *dev_dst, const void *src, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return SSW_OK;
    if (pending) {  // the previous copy out of this buffer must have been consumed
        SSW_HIP_TRY(hipEventSynchronize(ev));
        pending = false;
    }
    if (bytes > cap) {
        if (host) (void)hipHostFree(host);
        host = nullptr;
        cap = 0;
        size_t c = 4096;
        while (c < bytes) c <<= 1;
        SSW_HIP_TRY(hipHostMalloc(&host, c, hipHostMallocDefault));
        cap = c;
    }
    if (!ev) SSW_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    memcpy(host, src, bytes);
    SSW_HIP_TRY(hipMemcpyAsync(dev_dst, host, bytes, hipMemcpyHostToDevice, stream));
    SSW_HIP_TRY(hipEventRecord(ev, stream));
    pending = true;
    return SSW_OK;
}

void PinnedStage::release() {
    if (pending && ev) (void)hipEventSynchronize(ev);
    if (host) (void)hipHostFree(host);
    if (ev) (void)hipEventDestroy(ev);
    host = nullptr;
    ev = nullptr;
    cap = 0;
    pending = false;
}

}  // namespace ssw

using namespace ssw;

struct ssw_index {
    int device = 0;
    int64_t n = 0;
    int32_t dim = 0;
    int64_t n_images = 0;
    bool has_map = false;
    float *X = nullptr;
    bool owns_X = false;
    float *scores = nullptr;      // [n]
    float *q_dev = nullptr;       // [dim] device copy of a host query
    PinnedStage q_stage;
    int64_t *row_start = nullptr;  // [n_images + 1] when has_map
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    SelectWorkspace ws;
    bool ws_ready = false;
    // gather staging
    int64_t *gather_idx = nullptr;
    float *gather_out = nullptr;
    int64_t gather_cap = 0;
    PinnedStage rows_stage;
    void *res_host = nullptr;  // pinned result mirror
    // small index (one scan launch + one selection launch, no copies, no stream wait): pinned, device-visible block
    // [query dim f32][excluded ids SMALL_EXCL_CAP i64][packed result], and the sequence number the host spins on
    unsigned char *small_host = nullptr;
    unsigned small_seq = 0;
    unsigned res_pending_seq = 0;  // != 0: the selection in flight publishes into res_host under this sequence number
    unsigned small_pending_seq = 0;  // the same for the small form (topk_small_enqueue / _collect)
    float *q2_dev = nullptr;  // second query vector (score_rows)
    PinnedStage q2_stage;
    // tile geometry + staging of the avg_score aggregation (rescore.hip)
    std::vector<int64_t> row_start_host;  // host mirror of row_start
    float *tile_boxes = nullptr;   // [n, 4] x1, y1, x2, y2
    int32_t *tile_zoom = nullptr;  // [n]
    int64_t *rs_pos = nullptr, *rs_off = nullptr, *rs_row = nullptr;  // [rs_cap]
    float *rs_score = nullptr;     // [rs_cap]
    float *rs_minus = nullptr;     // [rs_minus_cap]
    int64_t rs_cap = 0, rs_minus_cap = 0;
    // profiling of the scan kernel
    bool profiling = false;
    std::vector<hipEvent_t> ev;  // pairs
    int ev_used = 0;
};

static ssw_status ensure_ws(ssw_index *idx) {
    if (idx->ws_ready) return SSW_OK;
    SSW_TRY(select_alloc(idx->ws, idx->n, idx->n_images, idx->has_map));
    idx->ws_ready = true;
    return SSW_OK;
}

static ssw_status check_query(const ssw_index *idx, const float *q_host) {
    for (int i = 0; i < idx->dim; ++i) {
        if (!std::isfinite(q_host[i])) {
            // the reference asserts on NaN query vectors (seesaw/loops/loop_base.py:47)
            set_error("query vector has a non-finite component at %d", i);
            return SSW_ERR_NUMERIC;
        }
    }
    return SSW_OK;
}

// a host query reaches q_dev through the kernel-argument segment of a one-wave kernel (dim <= 768): a launch is a
// third of what the 2-KB copy and its event cost on the host
constexpr int Q_ARG_FLOATS = 768;
struct QArg {
    float v[Q_ARG_FLOATS];
};
__global__ void k_stage_query(QArg q, float *__restrict__ dst, int dim) {
    for (int i = threadIdx.x; i < dim; i += 256) dst[i] = q.v[i];
}
static ssw_status stage_query(ssw_index *idx, const float *q_host) {
    if (idx->dim > Q_ARG_FLOATS)
        return idx->q_stage.push(idx->q_dev, q_host, (size_t)idx->dim * sizeof(float), idx->stream);
    QArg q;
    memcpy(q.v, q_host, (size_t)idx->dim * sizeof(float));
    hipLaunchKernelGGL(k_stage_query, dim3(1), dim3(256), 0, idx->stream, q, idx->q_dev, idx->dim);
    SSW_HIP_TRY(hipGetLastError());
    return SSW_OK;
}

static ssw_status do_scan(ssw_index *idx, const float *q_dev) {
    if (idx->profiling && idx->ev_used + 2 <= (int)idx->ev.size()) {
        SSW_HIP_TRY(hipEventRecord(idx->ev[idx->ev_used], idx->stream));
        SSW_TRY(launch_scan(idx->X, q_dev, idx->scores, idx->n, idx->dim, idx->device, idx->stream));
        SSW_HIP_TRY(hipEventRecord(idx->ev[idx->ev_used + 1], idx->stream));
        idx->ev_used += 2;
        return SSW_OK;
    }
    return launch_scan(idx->X, q_dev, idx->scores, idx->n, idx->dim, idx->device, idx->stream);
}

static ssw_status do_select(ssw_index *idx, int32_t k) {
    SSW_TRY(ensure_ws(idx));
    if (idx->has_map) {
        SSW_TRY(launch_image_max(idx->scores, idx->row_start, idx->n_images, idx->ws.img_score,
                                 idx->ws.img_best, idx->stream));
        return launch_select_topk(idx->ws, idx->ws.img_score, idx->n_images, idx->ws.img_best, k,
                                  idx->device, idx->stream);
    }
    return launch_select_topk(idx->ws, idx->scores, idx->n, nullptr, k, idx->device, idx->stream);
}

extern "C" {

int32_t ssw_abi_version(void) { return SSW_ABI_VERSION; }
const char *ssw_last_error(void) { return g_last_error.c_str(); }

ssw_status ssw_device_count(int32_t *out_count) {
    SSW_REQUIRE(out_count != nullptr, "out_count is NULL");
    int c = 0;
    SSW_HIP_TRY(hipGetDeviceCount(&c));
    *out_count = c;
    return SSW_OK;
}

ssw_status ssw_device_info(int32_t device, char *name, int32_t name_cap, int32_t *out_cus,
                           int64_t *out_hbm_bytes) {
    hipDeviceProp_t p;
    SSW_HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name && name_cap > 0) {
        snprintf(name, (size_t)name_cap, "%s (%s)", p.name, p.gcnArchName);
    }
    if (out_cus) *out_cus = p.multiProcessorCount;
    if (out_hbm_bytes) *out_hbm_bytes = (int64_t)p.totalGlobalMem;
    return SSW_OK;
}

ssw_status ssw_index_create(int32_t device, int64_t n_rows, int32_t dim,
                            const float *dev_vectors_or_null, ssw_index **out) {
    SSW_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    SSW_REQUIRE(n_rows >= 0, "n_rows=%lld < 0", (long long)n_rows);
    if (dim <= 0 || dim % 256 != 0 || dim > 1024) {
        set_error("index: dim=%d unsupported (need a multiple of 256, <= 1024)", dim);
        return SSW_ERR_UNSUPPORTED;
    }
    if (n_rows >= (int64_t)0x7fff0000) {
        set_error("index: %lld rows exceed the 2^31 row limit of one shard", (long long)n_rows);
        return SSW_ERR_UNSUPPORTED;
    }
    SSW_REQUIRE(((uintptr_t)dev_vectors_or_null & 15) == 0, "device matrix is not 16-byte aligned");
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", device);
        return SSW_ERR_HIP;
    }
    ssw_index *idx = new (std::nothrow) ssw_index();
    if (!idx) return SSW_ERR_NOMEM;
    idx->device = device;
    idx->n = n_rows;
    idx->dim = dim;
    idx->n_images = n_rows;
    ssw_status st = SSW_OK;
    auto fail = [&](ssw_status s) {
        ssw_index_destroy(idx);
        return s;
    };
    if (hipStreamCreateWithFlags(&idx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("hipStreamCreate failed");
        return fail(SSW_ERR_HIP);
    }
    idx->stream = idx->own_stream;
    const size_t row_bytes = (size_t)dim * sizeof(float);
    if (dev_vectors_or_null) {
        idx->X = const_cast<float *>(dev_vectors_or_null);
    } else {
        hipError_t e = hipMalloc((void **)&idx->X, (size_t)(n_rows > 0 ? n_rows : 1) * row_bytes);
        if (e != hipSuccess) {
            set_error("hipMalloc of %.2f GB for the index failed: %s",
                      (double)n_rows * row_bytes / 1e9, hipGetErrorString(e));
            idx->X = nullptr;
            return fail(SSW_ERR_NOMEM);
        }
        idx->owns_X = true;
    }
    if (hipMalloc((void **)&idx->scores, (size_t)(n_rows + 64) * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&idx->q_dev, row_bytes) != hipSuccess) {
        set_error("hipMalloc of the score buffer failed");
        return fail(SSW_ERR_NOMEM);
    }
    (void)st;
    *out = idx;
    return SSW_OK;
}

ssw_status ssw_index_destroy(ssw_index *idx) {
    if (!idx) return SSW_OK;
    DeviceGuard guard(idx->device);
    if (idx->own_stream) (void)hipStreamSynchronize(idx->own_stream);
    for (hipEvent_t e : idx->ev) (void)hipEventDestroy(e);
    if (idx->ws_ready) select_free(idx->ws);
    if (idx->owns_X) (void)hipFree(idx->X);
    (void)hipFree(idx->scores);
    (void)hipFree(idx->q_dev);
    idx->q_stage.release();
    (void)hipFree(idx->tile_boxes);
    (void)hipFree(idx->tile_zoom);
    (void)hipFree(idx->rs_pos);
    (void)hipFree(idx->rs_off);
    (void)hipFree(idx->rs_row);
    (void)hipFree(idx->rs_score);
    (void)hipFree(idx->rs_minus);
    (void)hipFree(idx->row_start);
    (void)hipFree(idx->gather_idx);
    (void)hipFree(idx->gather_out);
    (void)hipFree(idx->q2_dev);
    idx->rows_stage.release();
    if (idx->res_host) (void)hipHostFree(idx->res_host);
    if (idx->small_host) (void)hipHostFree(idx->small_host);
    idx->q2_stage.release();
    if (idx->own_stream) (void)hipStreamDestroy(idx->own_stream);
    delete idx;
    return SSW_OK;
}

ssw_status ssw_index_set_stream(ssw_index *idx, void *hip_stream) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    DeviceGuard guard(idx->device);
    SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
    idx->stream = hip_stream ? (hipStream_t)hip_stream : idx->own_stream;
    return SSW_OK;
}

ssw_status ssw_index_sync(ssw_index *idx) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    DeviceGuard guard(idx->device);
    SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
    return SSW_OK;
}

ssw_status ssw_index_shape(const ssw_index *idx, int64_t *n_rows, int32_t *dim, int64_t *n_images) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    if (n_rows) *n_rows = idx->n;
    if (dim) *dim = idx->dim;
    if (n_images) *n_images = idx->n_images;
    return SSW_OK;
}

ssw_status ssw_index_device_ptrs(ssw_index *idx, void **dev_vectors, void **dev_scores) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    if (dev_vectors) *dev_vectors = idx->X;
    if (dev_scores) *dev_scores = idx->scores;
    return SSW_OK;
}

ssw_status ssw_index_upload(ssw_index *idx, const float *host_rows, int64_t first_row, int64_t n) {
    SSW_REQUIRE(idx != nullptr && host_rows != nullptr, "NULL argument");
    SSW_REQUIRE(first_row >= 0 && n >= 0 && first_row + n <= idx->n,
                "rows [%lld, %lld) outside the index of %lld rows", (long long)first_row,
                (long long)(first_row + n), (long long)idx->n);
    DeviceGuard guard(idx->device);
    SSW_HIP_TRY(hipMemcpyAsync(idx->X + first_row * idx->dim, host_rows,
                               (size_t)n * idx->dim * sizeof(float), hipMemcpyHostToDevice,
                               idx->stream));
    SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
    return SSW_OK;
}

ssw_status ssw_index_download(ssw_index *idx, float *host_rows, int64_t first_row, int64_t n) {
    SSW_REQUIRE(idx != nullptr && host_rows != nullptr, "NULL argument");
    SSW_REQUIRE(first_row >= 0 && n >= 0 && first_row + n <= idx->n,
                "rows [%lld, %lld) outside the index of %lld rows", (long long)first_row,
                (long long)(first_row + n), (long long)idx->n);
    DeviceGuard guard(idx->device);
    SSW_HIP_TRY(hipMemcpyAsync(host_rows, idx->X + first_row * idx->dim,
                               (size_t)n * idx->dim * sizeof(float), hipMemcpyDeviceToHost,
                               idx->stream));
    SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
    return SSW_OK;
}

ssw_status ssw_index_fill_random(ssw_index *idx, uint64_t seed, int64_t global_first_row) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    DeviceGuard guard(idx->device);
    SSW_TRY(launch_fill_random(idx->X, idx->n, idx->dim, seed, global_first_row, idx->stream));
    SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
    return SSW_OK;
}

ssw_status ssw_index_set_row2image(ssw_index *idx, const int32_t *row2image_host, int64_t n_images) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    DeviceGuard guard(idx->device);
    SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
    if (idx->ws_ready) {
        select_free(idx->ws);
        idx->ws_ready = false;
    }
    (void)hipFree(idx->row_start);
    idx->row_start = nullptr;
    idx->row_start_host.clear();
    if (row2image_host == nullptr) {
        idx->has_map = false;
        idx->n_images = idx->n;
        return SSW_OK;
    }
    SSW_REQUIRE(n_images >= 0 && n_images <= idx->n, "n_images=%lld outside [0, n_rows]",
                (long long)n_images);
    std::vector<int64_t> start((size_t)n_images + 1, 0);
    int32_t prev = 0;
    for (int64_t r = 0; r < idx->n; ++r) {
        const int32_t m = row2image_host[r];
        if (m < prev || m >= n_images) {
            set_error("row2image[%lld]=%d is not non-decreasing within [0, %lld)", (long long)r, m,
                      (long long)n_images);
            return SSW_ERR_INVALID;
        }
        prev = m;
        start[(size_t)m + 1]++;
    }
    for (int64_t m = 0; m < n_images; ++m) {
        if (start[(size_t)m + 1] == 0) {
            set_error("image position %lld has no rows", (long long)m);
            return SSW_ERR_INVALID;
        }
        start[(size_t)m + 1] += start[(size_t)m];
    }
    SSW_HIP_TRY(hipMalloc((void **)&idx->row_start, ((size_t)n_images + 1) * sizeof(int64_t)));
    SSW_HIP_TRY(hipMemcpy(idx->row_start, start.data(), ((size_t)n_images + 1) * sizeof(int64_t),
                          hipMemcpyHostToDevice));
    idx->row_start_host = std::move(start);
    idx->has_map = true;
    idx->n_images = n_images;
    return SSW_OK;
}

ssw_status ssw_index_set_tile_meta(ssw_index *idx, const float *boxes_host, const int32_t *zoom_host) {
    SSW_REQUIRE(idx != nullptr && boxes_host != nullptr && zoom_host != nullptr, "NULL argument");
    DeviceGuard guard(idx->device);
    for (int64_t r = 0; r < idx->n; ++r)
        SSW_REQUIRE(zoom_host[r] >= 0 && zoom_host[r] <= SSW_RESCORE_MAX_ZOOM, "zoom_level[%lld]=%d outside [0, %d]",
                    (long long)r, zoom_host[r], SSW_RESCORE_MAX_ZOOM);
    SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
    if (!idx->tile_boxes) SSW_HIP_TRY(hipMalloc((void **)&idx->tile_boxes, (size_t)std::max<int64_t>(idx->n, 1) * 16));
    if (!idx->tile_zoom) SSW_HIP_TRY(hipMalloc((void **)&idx->tile_zoom, (size_t)std::max<int64_t>(idx->n, 1) * 4));
    SSW_HIP_TRY(hipMemcpy(idx->tile_boxes, boxes_host, (size_t)idx->n * 16, hipMemcpyHostToDevice));
    SSW_HIP_TRY(hipMemcpy(idx->tile_zoom, zoom_host, (size_t)idx->n * 4, hipMemcpyHostToDevice));
    return SSW_OK;
}

ssw_status ssw_index_rescore_avg(ssw_index *idx, const int64_t *image_positions, int32_t m, int32_t aug_larger,
                                 const float *minus_scores_or_null, float *out_scores, int64_t *out_best_rows) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    if (m <= 0) return SSW_OK;
    SSW_REQUIRE(image_positions && out_scores && out_best_rows, "NULL argument");
    SSW_REQUIRE(aug_larger >= 0 && (aug_larger & 3) <= 2 && aug_larger <= 6,
                "aug_larger=%d is not 0 (all), 1 (greater) or 2 (adjacent), optionally + 4 (aug_weight = cont_weighted)", aug_larger);
    SSW_REQUIRE(idx->has_map && idx->tile_boxes && idx->tile_zoom,
                "rescore_avg needs ssw_index_set_row2image and ssw_index_set_tile_meta first");
    DeviceGuard guard(idx->device);
    std::vector<int64_t> off((size_t)m);
    int64_t total = 0, max_tiles = 0;
    for (int32_t c = 0; c < m; ++c) {
        const int64_t p = image_positions[c];
        SSW_REQUIRE(p >= 0 && p < idx->n_images, "image position %lld outside [0, %lld)", (long long)p,
                    (long long)idx->n_images);
        const int64_t t = idx->row_start_host[(size_t)p + 1] - idx->row_start_host[(size_t)p];
        off[(size_t)c] = total;
        total += t;
        max_tiles = std::max(max_tiles, t);
    }
    SSW_REQUIRE(max_tiles <= SSW_RESCORE_MAX_TILES, "an image with %lld tiles exceeds the %d the kernel keeps in LDS",
                (long long)max_tiles, SSW_RESCORE_MAX_TILES);
    hipStream_t s = idx->stream;
    if (m > idx->rs_cap) {
        SSW_HIP_TRY(hipStreamSynchronize(s));
        for (void *q : {(void *)idx->rs_pos, (void *)idx->rs_off, (void *)idx->rs_row, (void *)idx->rs_score}) (void)hipFree(q);
        idx->rs_pos = idx->rs_off = idx->rs_row = nullptr;
        idx->rs_score = nullptr;
        idx->rs_cap = 0;
        int64_t cap = 256;
        while (cap < m) cap <<= 1;
        SSW_HIP_TRY(hipMalloc((void **)&idx->rs_pos, (size_t)cap * 8));
        SSW_HIP_TRY(hipMalloc((void **)&idx->rs_off, (size_t)cap * 8));
        SSW_HIP_TRY(hipMalloc((void **)&idx->rs_row, (size_t)cap * 8));
        SSW_HIP_TRY(hipMalloc((void **)&idx->rs_score, (size_t)cap * 4));
        idx->rs_cap = cap;
    }
    if (minus_scores_or_null && total > idx->rs_minus_cap) {
        SSW_HIP_TRY(hipStreamSynchronize(s));
        (void)hipFree(idx->rs_minus);
        idx->rs_minus = nullptr;
        idx->rs_minus_cap = 0;
        int64_t cap = 4096;
        while (cap < total) cap <<= 1;
        SSW_HIP_TRY(hipMalloc((void **)&idx->rs_minus, (size_t)cap * 4));
        idx->rs_minus_cap = cap;
    }
    // inputs are a few hundred bytes: plain synchronous copies from the caller's buffers (ordered before the launch)
    SSW_HIP_TRY(hipMemcpyAsync(idx->rs_pos, image_positions, (size_t)m * 8, hipMemcpyHostToDevice, s));
    SSW_HIP_TRY(hipMemcpyAsync(idx->rs_off, off.data(), (size_t)m * 8, hipMemcpyHostToDevice, s));
    if (minus_scores_or_null)
        SSW_HIP_TRY(hipMemcpyAsync(idx->rs_minus, minus_scores_or_null, (size_t)total * 4, hipMemcpyHostToDevice, s));
    SSW_HIP_TRY(hipStreamSynchronize(s));  // `off` is a local; pageable sources are staged by now
    SSW_TRY(launch_avg_score(idx->tile_boxes, idx->tile_zoom, idx->scores, minus_scores_or_null ? idx->rs_minus : nullptr,
                             idx->row_start, idx->rs_pos, idx->rs_off, m, (int32_t)max_tiles, aug_larger,
                             idx->rs_score, idx->rs_row, s));
    SSW_HIP_TRY(hipMemcpyAsync(out_scores, idx->rs_score, (size_t)m * 4, hipMemcpyDeviceToHost, s));
    SSW_HIP_TRY(hipMemcpyAsync(out_best_rows, idx->rs_row, (size_t)m * 8, hipMemcpyDeviceToHost, s));
    SSW_HIP_TRY(hipStreamSynchronize(s));
    return SSW_OK;
}

ssw_status ssw_index_rescore_avg_f64(ssw_index *idx, const double *dev_scores, const int64_t *image_positions, int32_t m,
                                     int32_t aug_larger, double *out_scores, int64_t *out_best_rows) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    if (m <= 0) return SSW_OK;
    SSW_REQUIRE(dev_scores && image_positions && out_scores && out_best_rows, "NULL argument");
    SSW_REQUIRE(aug_larger >= 0 && (aug_larger & 3) <= 2 && aug_larger <= 6,
                "aug_larger=%d is not 0 (all), 1 (greater) or 2 (adjacent), optionally + 4 (aug_weight = cont_weighted)", aug_larger);
    SSW_REQUIRE(idx->has_map && idx->tile_boxes && idx->tile_zoom,
                "rescore_avg needs ssw_index_set_row2image and ssw_index_set_tile_meta first");
    DeviceGuard guard(idx->device);
    std::vector<int64_t> off((size_t)m);
    int64_t total = 0, max_tiles = 0;
    for (int32_t c = 0; c < m; ++c) {
        const int64_t p = image_positions[c];
        SSW_REQUIRE(p >= 0 && p < idx->n_images, "image position %lld outside [0, %lld)", (long long)p,
                    (long long)idx->n_images);
        const int64_t t = idx->row_start_host[(size_t)p + 1] - idx->row_start_host[(size_t)p];
        off[(size_t)c] = total;
        total += t;
        max_tiles = std::max(max_tiles, t);
    }
    hipStream_t s = idx->stream;
    int64_t *d_pos = nullptr, *d_off = nullptr, *d_row = nullptr;
    double *d_score = nullptr;
    auto release = [&]() {
        for (void *q : {(void *)d_pos, (void *)d_off, (void *)d_row, (void *)d_score}) (void)hipFree(q);
    };
    // (a few hundred bytes per call and one call per round of a graph loop: plain allocations keep this entry
    //  independent of the f32 path's cached buffers)
    if (hipMalloc((void **)&d_pos, (size_t)m * 8) != hipSuccess || hipMalloc((void **)&d_off, (size_t)m * 8) != hipSuccess ||
        hipMalloc((void **)&d_row, (size_t)m * 8) != hipSuccess || hipMalloc((void **)&d_score, (size_t)m * 8) != hipSuccess) {
        release();
        set_error("rescore_avg_f64: allocation failed");
        return SSW_ERR_NOMEM;
    }
    ssw_status rc = SSW_OK;
    hipError_t e = hipMemcpyAsync(d_pos, image_positions, (size_t)m * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, off.data(), (size_t)m * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess)
        rc = launch_avg_score_f64(idx->tile_boxes, idx->tile_zoom, dev_scores, idx->row_start, d_pos, d_off, m,
                                  (int32_t)max_tiles, aug_larger, d_score, d_row, s);
    if (e == hipSuccess && rc == SSW_OK) e = hipMemcpyAsync(out_scores, d_score, (size_t)m * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && rc == SSW_OK) e = hipMemcpyAsync(out_best_rows, d_row, (size_t)m * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    else (void)hipStreamSynchronize(s);
    release();
    if (rc != SSW_OK) return rc;
    if (e != hipSuccess) {
        set_error("rescore_avg_f64: %s", hipGetErrorString(e));
        return SSW_ERR_HIP;
    }
    return SSW_OK;
}

ssw_status ssw_index_scan_dev(ssw_index *idx, const float *q_dev) {
    SSW_REQUIRE(idx != nullptr && q_dev != nullptr, "NULL argument");
    DeviceGuard guard(idx->device);
    return do_scan(idx, q_dev);
}

ssw_status ssw_index_scan(ssw_index *idx, const float *q_host, float *out_scores_host_or_null) {
    SSW_REQUIRE(idx != nullptr && q_host != nullptr, "NULL argument");
    SSW_TRY(check_query(idx, q_host));
    DeviceGuard guard(idx->device);
    SSW_TRY(idx->q_stage.push(idx->q_dev, q_host, (size_t)idx->dim * sizeof(float), idx->stream));
    SSW_TRY(do_scan(idx, idx->q_dev));
    if (out_scores_host_or_null && idx->n > 0) {
        SSW_HIP_TRY(hipMemcpyAsync(out_scores_host_or_null, idx->scores,
                                   (size_t)idx->n * sizeof(float), hipMemcpyDeviceToHost,
                                   idx->stream));
    }
    SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
    return SSW_OK;
}

ssw_status ssw_index_load_scores(ssw_index *idx, const float *scores_host) {
    SSW_REQUIRE(idx != nullptr && (idx->n == 0 || scores_host != nullptr), "NULL argument");
    DeviceGuard guard(idx->device);
    if (idx->n > 0) {
        SSW_HIP_TRY(hipMemcpyAsync(idx->scores, scores_host, (size_t)idx->n * sizeof(float),
                                   hipMemcpyHostToDevice, idx->stream));
        SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
    }
    return SSW_OK;
}

ssw_status ssw_index_set_excluded(ssw_index *idx, const int64_t *excluded_images, int64_t n_excluded) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    SSW_REQUIRE(n_excluded == 0 || excluded_images != nullptr, "excluded_images is NULL");
    for (int64_t i = 0; i < n_excluded; ++i) {
        SSW_REQUIRE(excluded_images[i] >= 0 && excluded_images[i] < idx->n_images,
                    "excluded image %lld outside [0, %lld)", (long long)excluded_images[i],
                    (long long)idx->n_images);
    }
    DeviceGuard guard(idx->device);
    SSW_TRY(ensure_ws(idx));
    return select_set_excluded(idx->ws, idx->n_images, excluded_images, n_excluded, idx->stream);
}

ssw_status ssw_index_topk_dev(ssw_index *idx, const float *q_dev, int32_t k) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    DeviceGuard guard(idx->device);
    if (q_dev) SSW_TRY(do_scan(idx, q_dev));
    if (idx->n_images == 0) {  // an empty shard still takes part in the exchange: its message says "0 keys"
        if (idx->ws.xchg.msg_out)
            SSW_HIP_TRY(hipMemsetAsync(idx->ws.xchg.msg_out + (idx->ws.xchg.msg_len - 1), 0, sizeof(uint64_t), idx->stream));
        return SSW_OK;
    }
    return do_select(idx, k);
}

// The fast selection keeps at most 8192 candidates; when more images than that share the 24-bit score prefix
// of the k-th score (duplicated vectors, mass ties) it raises the overflow word next to the count
// (ssw_index_result_ptrs: count[1]).  The host-fetching entry points rerun the deep path by themselves;
// callers of the device-resident form read the flag (e.g. after their exchange step) and call this.
ssw_status ssw_index_select_deep_dev(ssw_index *idx, int32_t k) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    SSW_REQUIRE(k >= 1 && k <= SSW_MAX_TOPK, "k=%d outside [1, %d]", k, SSW_MAX_TOPK);
    if (idx->n_images == 0) return SSW_OK;
    DeviceGuard guard(idx->device);
    SSW_TRY(ensure_ws(idx));
    const float *values = idx->has_map ? idx->ws.img_score : idx->scores;
    const uint32_t *best = idx->has_map ? idx->ws.img_best : nullptr;
    return launch_select_topk_deep(idx->ws, values, idx->n_images, best, k, idx->device, idx->stream);
}

ssw_status ssw_index_result_ptrs(ssw_index *idx, void **dev_keys, void **dev_count,
                                 void **dev_best_rows) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    DeviceGuard guard(idx->device);
    SSW_TRY(ensure_ws(idx));
    if (dev_keys) *dev_keys = idx->ws.out_keys;
    if (dev_count) *dev_count = idx->ws.out_count;
    if (dev_best_rows) *dev_best_rows = idx->ws.out_best;
    return SSW_OK;
}

// one pinned block receives the packed result [count, overflow, k, 0][keys k][best k]:
// one async copy, one synchronisation
static ssw_status ensure_res_host(ssw_index *idx) {
    const size_t cap = 16 + (size_t)SSW_MAX_TOPK * 12;
    if (!idx->res_host) {
        SSW_HIP_TRY(hipHostMalloc((void **)&idx->res_host, cap, hipHostMallocMapped | hipHostMallocCoherent));
        memset(idx->res_host, 0, cap);
    }
    return SSW_OK;
}

// the next selection on this index publishes into res_host (see SelectWorkspace::host_packed)
static ssw_status arm_host_result(ssw_index *idx) {
    SSW_TRY(ensure_ws(idx));
    SSW_TRY(ensure_res_host(idx));
    unsigned char *dev_view = nullptr;
    SSW_HIP_TRY(hipHostGetDevicePointer((void **)&dev_view, idx->res_host, 0));
    unsigned seq = ++idx->small_seq;
    if (seq == 0) seq = ++idx->small_seq;
    idx->ws.host_packed = dev_view;
    idx->ws.host_seq = seq;
    idx->res_pending_seq = seq;
    return SSW_OK;
}

static ssw_status wait_host_seq(hipStream_t stream, const unsigned *flag, unsigned seq) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned it = 0;; ++it) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return SSW_OK;
        if ((it & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
    }
    SSW_HIP_TRY(hipStreamSynchronize(stream));  // a long scan ahead of the selection: sleep in the runtime instead
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
        set_error("topk: the selection kernel finished without publishing its result");
        return SSW_ERR_HIP;
    }
    return SSW_OK;
}

// one pinned block receives the packed result [count, overflow, k, seq][keys k][best k]: written by the selection
// itself when armed (arm_host_result), else one async copy + one synchronisation
static ssw_status fetch_results(ssw_index *idx, int32_t k, int32_t *count, bool *overflow) {
    SSW_TRY(ensure_res_host(idx));
    if (idx->res_pending_seq != 0) {
        const unsigned seq = idx->res_pending_seq;
        idx->res_pending_seq = 0;
        SSW_TRY(wait_host_seq(idx->stream, reinterpret_cast<const unsigned *>(idx->res_host) + 3, seq));
    } else {
        SSW_HIP_TRY(hipMemcpyAsync(idx->res_host, idx->ws.packed, 16 + (size_t)k * 12, hipMemcpyDeviceToHost,
                                   idx->stream));
        SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
    }
    const int32_t *hdr = reinterpret_cast<const int32_t *>(idx->res_host);
    *count = hdr[0];
    *overflow = hdr[1] != 0;
    if (hdr[2] != k) {
        set_error("topk_fetch: k=%d does not match the k=%d of the selection that produced the result", k, hdr[2]);
        return SSW_ERR_INVALID;
    }
    return SSW_OK;
}

ssw_status ssw_index_topk_fetch(ssw_index *idx, int32_t k, int64_t *out_images, float *out_scores,
                                int64_t *out_best_rows, int32_t *out_count) {
    SSW_REQUIRE(idx != nullptr && out_count != nullptr, "NULL argument");
    SSW_REQUIRE(k >= 1 && k <= SSW_MAX_TOPK, "k=%d outside [1, %d]", k, SSW_MAX_TOPK);
    *out_count = 0;
    if (idx->n_images == 0) return SSW_OK;
    DeviceGuard guard(idx->device);
    SSW_TRY(ensure_ws(idx));
    int32_t count = 0;
    bool overflow = false;
    SSW_TRY(fetch_results(idx, k, &count, &overflow));
    if (overflow) {  // massive exact ties: rerun the selection on the deep path
        const float *values = idx->has_map ? idx->ws.img_score : idx->scores;
        const uint32_t *best = idx->has_map ? idx->ws.img_best : nullptr;
        SSW_TRY(arm_host_result(idx));
        SSW_TRY(launch_select_topk_deep(idx->ws, values, idx->n_images, best, k, idx->device,
                                        idx->stream));
        SSW_TRY(fetch_results(idx, k, &count, &overflow));
    }
    const char *h = reinterpret_cast<const char *>(idx->res_host);
    const uint64_t *keys = reinterpret_cast<const uint64_t *>(h + 16);
    const uint32_t *best = reinterpret_cast<const uint32_t *>(h + 16 + (size_t)k * sizeof(uint64_t));
    if (count > k) count = k;
    for (int32_t i = 0; i < count; ++i) {
        const uint64_t key = keys[i];
        if (out_images) out_images[i] = (int64_t)(0xffffffffu - (uint32_t)(key & 0xffffffffull));
        if (out_scores) out_scores[i] = ord_to_f32((uint32_t)(key >> 32));
        if (out_best_rows) out_best_rows[i] = (int64_t)best[i];
    }
    *out_count = count;
    return SSW_OK;
}

// An index of a few thousand images (an LVIS-category subset: 1 109 images x 13 tiles) spends its round in fixed
// costs, not in the scan: three copies, five launches and a stream wait were ~95 us around ~10 us of kernels.  This form
// is three launches and no copy: the query goes to q_dev through a kernel argument, the scan runs on every CU, and ONE
// kernel takes the per-image maximum, strikes out the excluded ids (read from pinned memory the device maps), selects
// and writes the packed result into the same pinned block, releasing a sequence word the host spins on.
constexpr int64_t SMALL_EXCL_CAP = 8192;
constexpr int64_t SMALL_ROWS = 65536;  // the small scan kernel's range (scan.hip)

static SSW_TUNABLE bool g_small_path = true;  // ssw_tune_topk

static bool small_path_ok(const ssw_index *idx, int64_t n_excluded) {
    return g_small_path && idx->n_images >= 1 && idx->n_images <= SELECT_SMALL_IMAGES && idx->n <= SMALL_ROWS &&
           n_excluded <= SMALL_EXCL_CAP;
}

// enqueue half: [stage the query, scan,] exclusion list into the pinned block, ONE selection launch that publishes the
// packed result under a fresh sequence number (idx->small_pending_seq)
static ssw_status topk_small_enqueue(ssw_index *idx, const float *q_host, const int64_t *excluded_images, int64_t n_excluded,
                                     int32_t k) {
    const size_t q_bytes = (size_t)idx->dim * sizeof(float), ex_bytes = (size_t)SMALL_EXCL_CAP * sizeof(int64_t);
    const size_t res_bytes = 16 + (size_t)SSW_MAX_TOPK * 12;
    if (!idx->small_host) {
        SSW_HIP_TRY(hipHostMalloc((void **)&idx->small_host, q_bytes + ex_bytes + res_bytes,
                                  hipHostMallocMapped | hipHostMallocCoherent));
        memset(idx->small_host, 0, q_bytes + ex_bytes + res_bytes);
    }
    unsigned char *dev_view = nullptr;
    SSW_HIP_TRY(hipHostGetDevicePointer((void **)&dev_view, idx->small_host, 0));
    for (int64_t i = 0; i < n_excluded; ++i) {
        SSW_REQUIRE(excluded_images[i] >= 0 && excluded_images[i] < idx->n_images,
                    "excluded image %lld outside [0, %lld)", (long long)excluded_images[i], (long long)idx->n_images);
    }
    SSW_TRY(ensure_ws(idx));
    if (idx->ws.excl_dirty)  // a list installed by ssw_index_set_excluded does not apply to this call
        SSW_TRY(select_set_excluded(idx->ws, idx->n_images, nullptr, 0, idx->stream));
    if (q_host) {
        if (idx->dim <= Q_ARG_FLOATS) {  // through a kernel argument into q_dev: 451 workgroups then read it out of L2
            SSW_TRY(stage_query(idx, q_host));
            SSW_TRY(do_scan(idx, idx->q_dev));
        } else {  // (a wider query stays in the mapped block: every workgroup reads it over the host link)
            memcpy(idx->small_host, q_host, q_bytes);
            SSW_TRY(do_scan(idx, reinterpret_cast<const float *>(dev_view)));
        }
    }
    if (n_excluded > 0) memcpy(idx->small_host + q_bytes, excluded_images, (size_t)n_excluded * sizeof(int64_t));
    unsigned seq = ++idx->small_seq;
    if (seq == 0) seq = ++idx->small_seq;
    SSW_TRY(launch_select_small(idx->ws, idx->scores, idx->has_map ? idx->row_start : nullptr, idx->n_images,
                                reinterpret_cast<const int64_t *>(dev_view + q_bytes), n_excluded, k,
                                dev_view + q_bytes + ex_bytes, seq, idx->stream));
    idx->small_pending_seq = seq;
    return SSW_OK;
}

// collect half: spin on the sequence word, decode the packed result
static ssw_status topk_small_collect(ssw_index *idx, int32_t k, int64_t *out_images, float *out_scores, int64_t *out_best_rows,
                                     int32_t *out_count) {
    const size_t q_bytes = (size_t)idx->dim * sizeof(float), ex_bytes = (size_t)SMALL_EXCL_CAP * sizeof(int64_t);
    const unsigned seq = idx->small_pending_seq;
    SSW_REQUIRE(seq != 0 && idx->small_host != nullptr, "topk: no small selection in flight");
    idx->small_pending_seq = 0;
    unsigned char *res = idx->small_host + q_bytes + ex_bytes;
    SSW_TRY(wait_host_seq(idx->stream, reinterpret_cast<const unsigned *>(res) + 3, seq));
    const int32_t *hdr = reinterpret_cast<const int32_t *>(res);
    int32_t count = hdr[0];
    if (count > k) count = k;
    const uint64_t *keys = reinterpret_cast<const uint64_t *>(res + 16);
    const uint32_t *best = reinterpret_cast<const uint32_t *>(res + 16 + (size_t)k * sizeof(uint64_t));
    for (int32_t i = 0; i < count; ++i) {
        const uint64_t key = keys[i];
        if (out_images) out_images[i] = (int64_t)(0xffffffffu - (uint32_t)(key & 0xffffffffull));
        if (out_scores) out_scores[i] = ord_to_f32((uint32_t)(key >> 32));
        if (out_best_rows) out_best_rows[i] = (int64_t)best[i];
    }
    *out_count = count;
    return SSW_OK;
}

static ssw_status topk_small(ssw_index *idx, const float *q_host, const int64_t *excluded_images, int64_t n_excluded,
                             int32_t k, int64_t *out_images, float *out_scores, int64_t *out_best_rows,
                             int32_t *out_count) {
    SSW_TRY(topk_small_enqueue(idx, q_host, excluded_images, n_excluded, k));
    return topk_small_collect(idx, k, out_images, out_scores, out_best_rows, out_count);
}

// ---- the two halves of ssw_index_topk(q = NULL) for callers that put more work on the stream in between or ahead
// (ssw_labelprop_round: propagation -> scores -> this selection, ONE wait).  `on_stream` replaces the handle's stream for
// the duration of the call; the caller has made sure the handle's own stream is idle (ssw_index_sync).
extern "C++" {
namespace ssw {
struct StreamSwap {
    ssw_index *idx;
    hipStream_t keep;
    StreamSwap(ssw_index *i, hipStream_t s) : idx(i), keep(i->stream) { idx->stream = s; }
    ~StreamSwap() { idx->stream = keep; }
};

ssw_status index_enqueue_topk_resident(ssw_index *idx, hipStream_t on_stream, const int64_t *excluded_images, int64_t n_excluded,
                                       int32_t k) {
    SSW_REQUIRE(idx != nullptr, "NULL argument");
    SSW_REQUIRE(k >= 1 && k <= SSW_MAX_TOPK, "k=%d outside [1, %d]", k, SSW_MAX_TOPK);
    SSW_REQUIRE(n_excluded == 0 || excluded_images != nullptr, "excluded_images is NULL");
    StreamSwap sw(idx, on_stream);
    if (small_path_ok(idx, n_excluded)) return topk_small_enqueue(idx, nullptr, excluded_images, n_excluded, k);
    if (idx->n_images == 0) return SSW_OK;
    SSW_TRY(ssw_index_set_excluded(idx, excluded_images, n_excluded));
    SSW_TRY(arm_host_result(idx));
    const ssw_status st = do_select(idx, k);
    if (st != SSW_OK) {
        idx->ws.host_packed = nullptr;
        idx->res_pending_seq = 0;
    }
    return st;
}

ssw_status index_collect_topk(ssw_index *idx, hipStream_t on_stream, int32_t k, int64_t *out_images, float *out_scores,
                              int64_t *out_best_rows, int32_t *out_count) {
    SSW_REQUIRE(idx != nullptr && out_count != nullptr, "NULL argument");
    *out_count = 0;
    StreamSwap sw(idx, on_stream);
    if (idx->small_pending_seq != 0) return topk_small_collect(idx, k, out_images, out_scores, out_best_rows, out_count);
    return ssw_index_topk_fetch(idx, k, out_images, out_scores, out_best_rows, out_count);
}

int index_device(const ssw_index *idx) { return idx ? idx->device : -1; }
}  // namespace ssw
}  // extern "C++"

ssw_status ssw_index_topk(ssw_index *idx, const float *q_host, const int64_t *excluded_images,
                          int64_t n_excluded, int32_t k, int64_t *out_images, float *out_scores,
                          int64_t *out_best_rows, int32_t *out_count) {
    SSW_REQUIRE(idx != nullptr && out_count != nullptr, "NULL argument");
    SSW_REQUIRE(k >= 1 && k <= SSW_MAX_TOPK, "k=%d outside [1, %d]", k, SSW_MAX_TOPK);
    SSW_REQUIRE(n_excluded == 0 || excluded_images != nullptr, "excluded_images is NULL");
    *out_count = 0;
    DeviceGuard guard(idx->device);
    if (small_path_ok(idx, n_excluded)) {
        if (q_host) SSW_TRY(check_query(idx, q_host));
        return topk_small(idx, q_host, excluded_images, n_excluded, k, out_images, out_scores, out_best_rows, out_count);
    }
    if (q_host) {
        SSW_TRY(check_query(idx, q_host));
        SSW_TRY(stage_query(idx, q_host));
        SSW_TRY(do_scan(idx, idx->q_dev));
    }
    if (idx->n_images == 0) return SSW_OK;
    SSW_TRY(ssw_index_set_excluded(idx, excluded_images, n_excluded));
    // the selection's last kernel writes the packed result into the pinned mirror and releases a sequence word: the
    // host spins on it (no device-to-host copy, no stream wait)
    SSW_TRY(arm_host_result(idx));
    const ssw_status st = do_select(idx, k);
    if (st != SSW_OK) {  // nothing was launched that would publish: disarm
        idx->ws.host_packed = nullptr;
        idx->res_pending_seq = 0;
        return st;
    }
    return ssw_index_topk_fetch(idx, k, out_images, out_scores, out_best_rows, out_count);
}

static ssw_status stage_rows(ssw_index *idx, const int64_t *rows_host, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        SSW_REQUIRE(rows_host[i] >= 0 && rows_host[i] < idx->n, "row %lld outside [0, %lld)",
                    (long long)rows_host[i], (long long)idx->n);
    }
    if (n > idx->gather_cap) {
        SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
        (void)hipFree(idx->gather_idx);
        (void)hipFree(idx->gather_out);
        idx->gather_idx = nullptr;
        idx->gather_out = nullptr;
        idx->gather_cap = 0;
        int64_t cap = 4096;
        while (cap < n) cap <<= 1;
        SSW_HIP_TRY(hipMalloc((void **)&idx->gather_idx, (size_t)cap * sizeof(int64_t)));
        SSW_HIP_TRY(hipMalloc((void **)&idx->gather_out, (size_t)cap * sizeof(float)));
        idx->gather_cap = cap;
    }
    return idx->rows_stage.push(idx->gather_idx, rows_host, (size_t)n * sizeof(int64_t), idx->stream);
}

ssw_status ssw_index_score_rows(ssw_index *idx, const float *q_host, const int64_t *rows_host,
                                int64_t n, float *out_scores_host) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    if (n <= 0) return SSW_OK;
    SSW_REQUIRE(q_host && rows_host && out_scores_host, "NULL argument");
    SSW_TRY(check_query(idx, q_host));
    DeviceGuard guard(idx->device);
    SSW_TRY(stage_rows(idx, rows_host, n));
    if (!idx->q2_dev) SSW_HIP_TRY(hipMalloc((void **)&idx->q2_dev, (size_t)idx->dim * sizeof(float)));
    SSW_TRY(idx->q2_stage.push(idx->q2_dev, q_host, (size_t)idx->dim * sizeof(float), idx->stream));
    SSW_TRY(launch_score_rows(idx->X, idx->q2_dev, idx->gather_idx, n, idx->dim, idx->gather_out,
                              idx->stream));
    SSW_HIP_TRY(hipMemcpyAsync(out_scores_host, idx->gather_out, (size_t)n * sizeof(float),
                               hipMemcpyDeviceToHost, idx->stream));
    SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
    return SSW_OK;
}

ssw_status ssw_index_gather_scores(ssw_index *idx, const int64_t *rows_host, int64_t n,
                                   float *out_scores_host) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    if (n <= 0) return SSW_OK;
    SSW_REQUIRE(rows_host != nullptr && out_scores_host != nullptr, "NULL argument");
    DeviceGuard guard(idx->device);
    SSW_TRY(stage_rows(idx, rows_host, n));
    SSW_TRY(launch_gather_f32(idx->scores, idx->gather_idx, n, idx->gather_out, idx->stream));
    SSW_HIP_TRY(hipMemcpyAsync(out_scores_host, idx->gather_out, (size_t)n * sizeof(float),
                               hipMemcpyDeviceToHost, idx->stream));
    SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
    return SSW_OK;
}

namespace {
__global__ void k_gather_rows_f32(const float *__restrict__ X, const int64_t *__restrict__ rows, int64_t n, int dim,
                                  float *__restrict__ out) {
    const int64_t r = blockIdx.x;
    const float4 *src = reinterpret_cast<const float4 *>(X + rows[r] * dim);
    float4 *dst = reinterpret_cast<float4 *>(out + r * dim);
    for (int c = threadIdx.x; c < dim / 4; c += blockDim.x) dst[c] = src[c];
}
}  // namespace

// the vectors of arbitrary rows (`index.vectors[rows]`: what the fitting loops read of the labelled tiles,
// multi_reg.py:204, loops/util.py:6,11) out of the resident matrix -- for callers that do not hold a host copy of it
// (a rank of the row-sharded index serves its own rows this way)
ssw_status ssw_index_gather_rows(ssw_index *idx, const int64_t *rows_host, int64_t n, float *out_host) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    if (n <= 0) return SSW_OK;
    SSW_REQUIRE(rows_host != nullptr && out_host != nullptr, "NULL argument");
    for (int64_t i = 0; i < n; ++i)
        SSW_REQUIRE(rows_host[i] >= 0 && rows_host[i] < idx->n, "row %lld outside [0, %lld)", (long long)rows_host[i],
                    (long long)idx->n);
    DeviceGuard guard(idx->device);
    SSW_TRY(stage_rows(idx, rows_host, n));
    float *buf = nullptr;
    SSW_HIP_TRY(hipMalloc((void **)&buf, (size_t)n * idx->dim * sizeof(float)));
    hipLaunchKernelGGL(k_gather_rows_f32, dim3((unsigned)n), dim3(128), 0, idx->stream, idx->X, idx->gather_idx, n,
                       (int)idx->dim, buf);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(out_host, buf, (size_t)n * idx->dim * sizeof(float), hipMemcpyDeviceToHost, idx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(idx->stream);
    (void)hipFree(buf);
    if (e != hipSuccess) {
        set_error("ssw_index_gather_rows: %s", hipGetErrorString(e));
        return SSW_ERR_HIP;
    }
    return SSW_OK;
}

ssw_status ssw_topk_merge_dev(int32_t device, void *hip_stream, const uint64_t *dev_keys_in,
                              int32_t n_lists, int32_t list_stride, const int32_t *dev_counts,
                              int32_t k, uint64_t *dev_keys_out, int32_t *dev_count_out) {
    SSW_REQUIRE(dev_keys_in && dev_counts && dev_keys_out && dev_count_out, "NULL argument");
    DeviceGuard guard(device);
    return launch_merge_topk(dev_keys_in, n_lists, list_stride, dev_counts, k, dev_keys_out,
                             dev_count_out, (hipStream_t)hip_stream);
}

// the sharded exchange without elementwise kernels around the collective: the selection's last kernel also writes
// this rank's message (globalised keys, optional best rows, count | overflow << 32) into dev_msg
ssw_status ssw_index_set_exchange_target(ssw_index *idx, uint64_t *dev_msg_or_null, int32_t k_max, int32_t with_best,
                                         int64_t image_offset, int64_t row_offset) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    FinalExchange x;
    if (dev_msg_or_null) {
        SSW_REQUIRE(k_max >= 1 && k_max <= SSW_MAX_TOPK && image_offset >= 0, "bad message geometry");
        x.msg_out = dev_msg_or_null;
        x.image_offset = (uint64_t)image_offset;
        x.row_offset = row_offset;
        x.k_max = k_max;
        x.with_best = with_best ? 1 : 0;
        x.msg_len = (with_best ? 2 : 1) * k_max + 1;
    }
    idx->ws.xchg = x;
    return SSW_OK;
}

ssw_status ssw_topk_merge_msgs_dev(int32_t device, void *hip_stream, const uint64_t *dev_msgs, int32_t world,
                                   int32_t k_max, int32_t with_best, int32_t k, uint64_t *dev_keys_out,
                                   int32_t *dev_count_out, int64_t *dev_flags_or_null, int64_t *dev_flags_seen_or_null) {
    SSW_REQUIRE(dev_msgs && dev_keys_out && dev_count_out, "NULL argument");
    DeviceGuard guard(device);
    return launch_merge_msgs(dev_msgs, world, k_max, with_best, k, dev_keys_out, dev_count_out,
                             reinterpret_cast<long long *>(dev_flags_or_null),
                             reinterpret_cast<long long *>(dev_flags_seen_or_null), (hipStream_t)hip_stream);
}

#ifdef SSW_DEBUG_HOOKS
ssw_status ssw_tune_topk(int32_t flags) {
    g_small_path = (flags & 1) != 0;
    tune_select((flags & 2) != 0);
    return SSW_OK;
}

ssw_status ssw_tune_scan(int32_t variant, int32_t blocks_per_cu) {
    tune_scan(variant, blocks_per_cu);
    return SSW_OK;
}
#endif

ssw_status ssw_index_profile(ssw_index *idx, int32_t enable) {
    SSW_REQUIRE(idx != nullptr, "idx is NULL");
    DeviceGuard guard(idx->device);
    SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
    if (enable && idx->ev.empty()) {
        idx->ev.resize(2 * 4096);
        for (auto &e : idx->ev) SSW_HIP_TRY(hipEventCreate(&e));
    }
    idx->profiling = enable != 0;
    idx->ev_used = 0;
    return SSW_OK;
}

ssw_status ssw_index_profile_read(ssw_index *idx, float *out_ms, int32_t cap, int32_t *out_n) {
    SSW_REQUIRE(idx != nullptr && out_n != nullptr, "NULL argument");
    DeviceGuard guard(idx->device);
    SSW_HIP_TRY(hipStreamSynchronize(idx->stream));
    const int pairs = idx->ev_used / 2;
    int n = 0;
    for (int i = 0; i < pairs && n < cap; ++i) {
        float ms = 0.f;
        SSW_HIP_TRY(hipEventElapsedTime(&ms, idx->ev[2 * i], idx->ev[2 * i + 1]));
        out_ms[n++] = ms;
    }
    *out_n = n;
    idx->ev_used = 0;
    return SSW_OK;
}

}  // extern "C"
