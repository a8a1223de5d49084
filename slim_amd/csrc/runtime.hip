// Library-wide state: last-error string, version, per-device kernel attributes.
#include "common.hpp"
#include "../../include/slim_hip.h"
#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <utility>

static thread_local char g_err[512] = "";

void slim_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* slim_last_error(void) { return g_err; }

extern "C" int slim_abi_version(void) { return SLIM_HIP_ABI_VERSION; }

hipError_t slim_allow_dynamic_lds(const void* func, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_pair(func, dev);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done[key] = bytes;
    return e;
}
