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

// the reference's inline constants as the kernels of this build use them (include/cpe.h: CpeDetectConstants).  The kernels
// themselves hold them as literals / constexpr (k_preprocess: 5, 3.0, 15, 0.5, 128; masks.hip: 20, 19, 240, 15, 91, 5 / 200, 7;
// region.hip: NTHR thresholds from 50 in steps of 10, areas 10 .. 5000, distance 10, radius + 4); tests/test_boundary_cpu.py
// compares this table with the values the reference's sources state.
extern "C" int32_t cpe_detect_constants(int32_t target, CpeDetectConstants *out)
{
    CPE_CHECK_ARG(out && (target == CPE_TARGET_CYLINDER || target == CPE_TARGET_PLANE), "cpe_detect_constants: bad argument");
    const bool plane = target == CPE_TARGET_PLANE;
    CpeDetectConstants c = {};
    c.blur_ksize = 5; c.hessian_sigma = 3.0; c.sauvola_window = 15; c.sauvola_k = 0.5; c.sauvola_R = 128.0;
    c.open_len = 20;
    c.clahe_clip = plane ? 0.0 : 4.5; c.clahe_tiles = plane ? 0 : 4;
    c.blob_thr_min = plane ? 0 : 50; c.blob_thr_step = plane ? 0 : 10; c.blob_thr_count = plane ? 0 : 17;
    c.blob_min_area = plane ? 0.0 : 10.0; c.blob_max_area = plane ? 0.0 : 5000.0; c.blob_min_dist = plane ? 0.0 : 10.0;
    c.blob_min_repeat = plane ? 0 : 2; c.disc_extra_radius = plane ? 0 : 4;
    c.spot_blur_ksize = 19; c.spot_threshold = 240;
    c.spot_small_radius = plane ? 0 : 30; c.spot_small_add = plane ? 0 : 20; c.spot_large_add = plane ? 0 : 5;
    c.frag_patch = 15; c.frag_min_pixels = plane ? 8 : 5; c.frag_max_pixels = plane ? 700 : 200;
    c.frag_kernel_base = plane ? 201 : 91;
    c.index_blur_ksize = 7; c.poly_degree = plane ? 1 : 2;
    c.plane_threshold = plane ? 127 : 0; c.plane_dilate_ksize = plane ? 11 : 0;
    c.max_points = CPE_MAXP; c.max_lines = CPE_MAXL; c.max_joints = CPE_MAXJ; c.max_groups_per_dir = CPE_MAXL;
    c.max_joints_per_group = CPE_MAXLP;
    *out = c;
    return CPE_OK;
}
