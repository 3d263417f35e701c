// Error string + version for libcpe_hip.so.
#include "cpe_internal.h"
#include <stdarg.h>
#include <string.h>
#include <algorithm>

namespace cpe {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace cpe

extern "C" int32_t cpe_version(void) { return CPE_VERSION; }
extern "C" const char *cpe_last_error_string(void) { return cpe::g_err; }

// ---- per-kernel timers -----------------------------------------------------------------------------
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace cpe {
namespace {
struct ProfRec { const char *name; hipEvent_t a, b; };
std::mutex g_pm;
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
}  // namespace

void prof_begin(const char *name, hipStream_t s)
{
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_pm);
    ProfRec r;
    r.name = name;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
}
void prof_end(hipStream_t s)
{
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_pm);
    if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().b, s);
}
}  // namespace cpe

extern "C" void cpe_profile_enable(int32_t on)
{
    std::lock_guard<std::mutex> lk(cpe::g_pm);
    cpe::g_prof_on = on != 0;
}

// synchronises the recorded events, writes "kernel,calls,total_ms\n" lines (sorted by total, descending)
// into csv (NUL terminated, truncated to cap) and clears the records; returns the number of kernels.
extern "C" int32_t cpe_profile_report(char *csv, size_t cap)
{
    std::lock_guard<std::mutex> lk(cpe::g_pm);
    std::map<std::string, std::pair<int, double>> agg;
    for (auto &r : cpe::g_recs) {
        float ms = 0;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto &e = agg[r.name];
            e.first++;
            e.second += ms;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    cpe::g_recs.clear();
    std::vector<std::pair<std::string, std::pair<int, double>>> v(agg.begin(), agg.end());
    std::sort(v.begin(), v.end(), [](auto &x, auto &y) { return x.second.second > y.second.second; });
    std::string out;
    for (auto &e : v) {
        char line[256];
        snprintf(line, sizeof line, "%s,%d,%.6f\n", e.first.c_str(), e.second.first, e.second.second);
        out += line;
    }
    if (csv && cap) {
        size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(csv, out.data(), n);
        csv[n] = 0;
    }
    return (int32_t)v.size();
}
