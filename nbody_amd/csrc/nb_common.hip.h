// nb_common.hip.h - the small host-side kit shared by libnbpm and libnbgas: last-error text,
// checked HIP calls, a grow-only device buffer.  Error codes are the common numeric values of
// the NB*_ERR_* enums in include/nbpm.h and include/nbgas.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdio>

namespace nbc {

constexpr int OK = 0, ERR_ARG = -1, ERR_HIP = -2, ERR_NOINIT = -3, ERR_NOMEM = -5;

inline char* error_text() {
    thread_local char text[512] = "";
    return text;
}

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_text(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return nbc::fail(nbc::ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define CHK(expr)                       \
    do {                                \
        int rc_ = (expr);               \
        if (rc_ != nbc::OK) return rc_; \
    } while (0)

// Device memory that only grows; contents are not preserved across a growth.
struct Buf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return OK;
        release();
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            return fail(ERR_NOMEM, "hipMalloc of %zu bytes failed", bytes);
        }
        cap = bytes;
        return OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

inline unsigned cdiv(size_t a, size_t b) { return static_cast<unsigned>((a + b - 1) / b); }

}  // namespace nbc
