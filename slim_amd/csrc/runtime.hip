// Library-wide state: last-error string, version, per-device kernel attributes.
#include "common.hpp"
#include "../../include/slim_hip.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
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

// ---- per-launch timing ------------------------------------------------------------------------
#include <string>
#include <vector>
bool g_slim_prof_on = false;
namespace {
struct ProfRec {
    std::string name;
    hipEvent_t a, b;
    double bytes, flops;
    int launches;
};
std::vector<ProfRec> g_prof;
std::mutex g_prof_mu;
}  // namespace

int slim_prof_begin(const char* name, hipStream_t st, double bytes, double flops, int launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.name = name;
    r.bytes = bytes;
    r.flops = flops;
    r.launches = launches > 0 ? launches : 1;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
    (void)hipEventRecord(r.a, st);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}

void slim_prof_end(int token, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (token >= 0 && token < (int)g_prof.size()) (void)hipEventRecord(g_prof[token].b, st);
}

extern "C" int slim_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    g_slim_prof_on = on != 0;
    return SLIM_OK;
}

// Writes "name count total_ms bytes flops\n" lines (one per distinct scope name) into buf; synchronises the events.
extern "C" int slim_profile_summary(char* buf, int64_t buflen) {
    SLIM_REQUIRE(buf && buflen > 0, "slim_profile_summary: bad buffer");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    struct Acc {
        int n = 0;
        double ms = 0, bytes = 0, flops = 0;
    };
    std::map<std::string, Acc> acc;
    for (auto& r : g_prof) {
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        auto& e = acc[r.name];
        e.n += r.launches;
        e.ms += ms;
        e.bytes += r.bytes;
        e.flops += r.flops;
    }
    std::string out;
    for (auto& kv : acc) {
        char line[256];
        snprintf(line, sizeof(line), "%s %d %.6f %.0f %.0f\n", kv.first.c_str(), kv.second.n, kv.second.ms,
                 kv.second.bytes, kv.second.flops);
        out += line;
    }
    if ((int64_t)out.size() + 1 > buflen) {
        slim_set_error("slim_profile_summary: buffer too small");
        return SLIM_ERR_INVALID;
    }
    memcpy(buf, out.c_str(), out.size() + 1);
    return SLIM_OK;
}
