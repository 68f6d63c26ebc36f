// q3_capture_lock.h — stream captures and legacy-stream operations exclude each other (see q3_engine.h, above HIPC): a frame capture
// holds q3_capture_mu() exclusively, every synchronous hipMemcpy / null-stream zero-fill / device-wide wait of the library holds it
// shared. One mutex per process (inline function, merged across the units of libq3tts.so).
#pragma once
#include <hip/hip_runtime_api.h>
#include <shared_mutex>
inline std::shared_mutex& q3_capture_mu() { static std::shared_mutex mu; return mu; }
inline hipError_t q3_hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind kind) {
    std::shared_lock<std::shared_mutex> lk(q3_capture_mu());
    return hipMemcpy(dst, src, n, kind);
}
inline hipError_t q3_hipDeviceSynchronize() { std::shared_lock<std::shared_mutex> lk(q3_capture_mu()); return hipDeviceSynchronize(); }
inline hipError_t q3_null_stream_memset(void* p, size_t bytes, bool wait) {       // DevPool: zero-fill on the null stream (+ wait for it)
    std::shared_lock<std::shared_mutex> lk(q3_capture_mu());
    const hipError_t e = hipMemsetAsync(p, 0, bytes, nullptr);
    return (e != hipSuccess || !wait) ? e : hipStreamSynchronize(nullptr);
}
inline hipError_t q3_null_stream_sync() { std::shared_lock<std::shared_mutex> lk(q3_capture_mu()); return hipStreamSynchronize(nullptr); }
