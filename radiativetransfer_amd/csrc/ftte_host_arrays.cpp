// ftte_host_arrays.cpp -- host arrays across PCIe: pinned staging blocks for pageable memory, DMA in place for registered arrays.
#include "ftte_context.h"

namespace ftte {

// ---- host arrays across PCIe ------------------------------------------------------------------------------------
constexpr size_t kStageBytes = (size_t)64 << 20;

bool is_registered(const ftte_ctx *c, const void *p, size_t bytes)
{
    const char *b = (const char *)p;
    for (const auto &r : c->registered)
        if (b >= r.base && b + bytes <= r.base + r.bytes) return true;
    for (const auto &r : c->registered_elsewhere)
        if (b >= r.base && b + bytes <= r.base + r.bytes) return true;
    return false;
}

void parallel_copy(void *dst, const void *src, size_t bytes)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nthreads = std::min<size_t>(std::min(8u, hw), std::max<size_t>(1, bytes >> 22));
    if (nthreads <= 1) { std::memcpy(dst, src, bytes); return; }
    const size_t chunk = ((bytes + nthreads - 1) / nthreads + 63) & ~(size_t)63;
    std::vector<std::thread> pool;
    for (size_t t = 0; t < nthreads; ++t) {
        const size_t lo = t * chunk;
        if (lo >= bytes) break;
        const size_t len = std::min(chunk, bytes - lo);
        pool.emplace_back([=] { std::memcpy((char *)dst + lo, (const char *)src + lo, len); });
    }
    for (auto &th : pool) th.join();
}

int ensure_stage(ftte_ctx *c)
{
    for (int q = 0; q < 2; ++q) {
        if (!c->stage[q]) FTTE_HIP(c, hipHostMalloc(&c->stage[q], kStageBytes, hipHostMallocDefault));
        if (!c->stage_ev[q]) FTTE_HIP(c, hipEventCreateWithFlags(&c->stage_ev[q], hipEventDisableTiming));
    }
    return FTTE_OK;
}

// host -> device on c->stream; returns with the copy complete
int upload(ftte_ctx *c, void *dst_dev, const void *src_host, size_t bytes)
{
    if (is_registered(c, src_host, bytes) || bytes < ((size_t)1 << 20)) {
        FTTE_HIP(c, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
        FTTE_HIP(c, hipStreamSynchronize(c->stream));
        return FTTE_OK;
    }
    int rc = ensure_stage(c);
    if (rc) return rc;
    int q = 0;
    bool busy[2] = {false, false};
    for (size_t off = 0; off < bytes; off += kStageBytes, q ^= 1) {
        const size_t len = std::min(kStageBytes, bytes - off);
        if (busy[q]) FTTE_HIP(c, hipEventSynchronize(c->stage_ev[q]));
        parallel_copy(c->stage[q], (const char *)src_host + off, len);
        FTTE_HIP(c, hipMemcpyAsync((char *)dst_dev + off, c->stage[q], len, hipMemcpyHostToDevice, c->stream));
        FTTE_HIP(c, hipEventRecord(c->stage_ev[q], c->stream));
        busy[q] = true;
    }
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    return FTTE_OK;
}

// device -> host on c->stream (after whatever is queued there); returns with the copy complete
int download(ftte_ctx *c, void *dst_host, const void *src_dev, size_t bytes)
{
    if (is_registered(c, dst_host, bytes) || bytes < ((size_t)1 << 20)) {
        FTTE_HIP(c, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        FTTE_HIP(c, hipStreamSynchronize(c->stream));
        return FTTE_OK;
    }
    int rc = ensure_stage(c);
    if (rc) return rc;
    // block q is filled by the DMA engine while the host threads empty block q^1
    size_t off_prev = 0, len_prev = 0;
    bool have_prev = false;
    int q = 0;
    for (size_t off = 0; off < bytes; off += kStageBytes, q ^= 1) {
        const size_t len = std::min(kStageBytes, bytes - off);
        FTTE_HIP(c, hipMemcpyAsync(c->stage[q], (const char *)src_dev + off, len, hipMemcpyDeviceToHost, c->stream));
        FTTE_HIP(c, hipEventRecord(c->stage_ev[q], c->stream));
        if (have_prev) {
            FTTE_HIP(c, hipEventSynchronize(c->stage_ev[q ^ 1]));
            parallel_copy((char *)dst_host + off_prev, c->stage[q ^ 1], len_prev);
        }
        off_prev = off; len_prev = len; have_prev = true;
    }
    if (have_prev) {
        FTTE_HIP(c, hipEventSynchronize(c->stage_ev[q ^ 1]));
        parallel_copy((char *)dst_host + off_prev, c->stage[q ^ 1], len_prev);
    }
    return FTTE_OK;
}

// host -> device on stream q; returns when the last piece has been handed to the DMA engine (not when it has arrived)
int upload_on(ftte_ctx *c, hipStream_t q, void *dst_dev, const void *src_host, size_t bytes)
{
    if (is_registered(c, src_host, bytes)) {
        FTTE_HIP(c, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, q));
        return FTTE_OK;
    }
    int rc = ensure_stage(c);
    if (rc) return rc;
    int b = 0;
    for (size_t off = 0; off < bytes; off += kStageBytes, b ^= 1) {
        const size_t len = std::min(kStageBytes, bytes - off);
        if (c->stage_used[b]) FTTE_HIP(c, hipEventSynchronize(c->stage_ev[b]));
        parallel_copy(c->stage[b], (const char *)src_host + off, len);
        FTTE_HIP(c, hipMemcpyAsync((char *)dst_dev + off, c->stage[b], len, hipMemcpyHostToDevice, q));
        FTTE_HIP(c, hipEventRecord(c->stage_ev[b], q));
        c->stage_used[b] = true;
    }
    return FTTE_OK;
}

// device -> pageable host memory behind whatever is queued on stream q; returns with the copy complete
int download_on(ftte_ctx *c, hipStream_t q, void *dst_host, const void *src_dev, size_t bytes)
{
    int rc = ensure_stage(c);
    if (rc) return rc;
    for (int b = 0; b < 2; ++b)
        if (c->stage_used[b]) { FTTE_HIP(c, hipEventSynchronize(c->stage_ev[b])); c->stage_used[b] = false; }
    size_t off_prev = 0, len_prev = 0;
    bool have_prev = false;
    int b = 0;
    for (size_t off = 0; off < bytes; off += kStageBytes, b ^= 1) {
        const size_t len = std::min(kStageBytes, bytes - off);
        FTTE_HIP(c, hipMemcpyAsync(c->stage[b], (const char *)src_dev + off, len, hipMemcpyDeviceToHost, q));
        FTTE_HIP(c, hipEventRecord(c->stage_ev[b], q));
        if (have_prev) {
            FTTE_HIP(c, hipEventSynchronize(c->stage_ev[b ^ 1]));
            parallel_copy((char *)dst_host + off_prev, c->stage[b ^ 1], len_prev);
        }
        off_prev = off; len_prev = len; have_prev = true;
    }
    if (have_prev) {
        FTTE_HIP(c, hipEventSynchronize(c->stage_ev[b ^ 1]));
        parallel_copy((char *)dst_host + off_prev, c->stage[b ^ 1], len_prev);
    }
    return FTTE_OK;
}


} // namespace ftte
