// pcr_host.hpp — tiny C++ convenience layer over the C ABI (include/pcr.h) used by the drop-in headers.
// One lazily created context per process (device = $PCR_DEVICE, default 0); errors become exceptions
// (the reference's drivers have no error channel: its tree functions return void).
#ifndef PCR_HOST_HPP
#define PCR_HOST_HPP

#include <cstdlib>
#include <stdexcept>
#include <string>

#include "../pcr.h"

namespace pcr {

struct CtxHolder {
    pcr_ctx* ctx = nullptr;
    CtxHolder()
    {
        const char* dev = std::getenv("PCR_DEVICE");
        int rc = pcr_ctx_create(dev ? std::atoi(dev) : 0, &ctx);
        if (rc != PCR_OK)
            throw std::runtime_error("pcr: no MI355X context (pcr_ctx_create failed, rc = " + std::to_string(rc) +
                                     "); there is no CPU fallback");
    }
    ~CtxHolder() { pcr_ctx_destroy(ctx); }
    CtxHolder(const CtxHolder&) = delete;
    CtxHolder& operator=(const CtxHolder&) = delete;
};

inline pcr_ctx* default_ctx()
{
    static CtxHolder holder;
    return holder.ctx;
}

inline void check(int rc, const char* what)
{
    if (rc != PCR_OK)
        throw std::runtime_error(std::string("pcr: ") + what + " failed (rc = " + std::to_string(rc) + "): " +
                                 pcr_ctx_last_error(default_ctx()));
}

}  // namespace pcr

#endif  // PCR_HOST_HPP
