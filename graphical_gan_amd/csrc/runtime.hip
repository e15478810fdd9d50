// libggan runtime: error strings, launch checks, per-kernel hipEvent timing.
#include "common.h"
#include <stdarg.h>
#include <mutex>
#include <vector>

namespace ggan {

static thread_local char t_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* name) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", name, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

void trace_launch(const char* name, dim3 grid, dim3 block, size_t shmem, double flops) {
    static const bool on = getenv("GGAN_TRACE_LAUNCHES") != nullptr;
    if (on)
        fprintf(stderr, "[ggan launch] %-44s grid %5u x %3u x %3u  block %4u  lds %6zu  MFLOP %10.2f\n", name, grid.x, grid.y, grid.z, block.x, shmem,
                flops * 1e-6);
}

bool launch_skipped(const char* name) {
    // GGAN_SKIP_KERNELS = ';'-separated entries (kernel names contain commas) "substr" (every launch whose name contains it) or "substr@k/n" (of the launches
    // matching substr, those whose ordinal is k modulo n: one launch SITE of an iteration that issues n of them)
    static const char* list = getenv("GGAN_SKIP_KERNELS");
    if (!list || !*list) return false;
    static std::mutex mu;
    static std::vector<long> counts;
    std::lock_guard<std::mutex> lk(mu);
    const char* p = list;
    bool skip = false;
    for (size_t idx = 0; *p; ++idx) {
        const char* q = strchr(p, ';');
        const size_t n = q ? (size_t)(q - p) : strlen(p);
        if (counts.size() <= idx) counts.push_back(0);
        if (n && n < 96) {
            char buf[96];
            memcpy(buf, p, n);
            buf[n] = 0;
            long k = -1, m = 1;
            if (char* at = strrchr(buf, '@')) {
                if (sscanf(at + 1, "%ld/%ld", &k, &m) == 2 && m > 0) *at = 0; else k = -1;
            }
            if (strstr(name, buf)) {
                const long ord = counts[idx]++;
                if (k < 0 || ord % m == k) skip = true;
            }
        }
        if (!q) break;
        p = q + 1;
    }
    return skip;
}

// ---- profiling --------------------------------------------------------------------------------
struct ProfEntry {
    const char* name;
    hipEvent_t a, b;
    double flops, bytes;
    long grid;
};
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<ProfEntry> g_prof;

ProfScope::ProfScope(const char* name, hipStream_t s, double flops, double bytes, long grid) : s_(s), idx_(-1) {
    if (!g_prof_on) return;
    ProfEntry e;
    e.name = name;
    e.flops = flops;
    e.bytes = bytes;
    e.grid = grid;
    if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
    (void)hipEventRecord(e.a, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(e);
    idx_ = (int)g_prof.size() - 1;
}

ProfScope::~ProfScope() {
    if (idx_ < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof[idx_].b, s_);
}

}  // namespace ggan

using namespace ggan;

extern "C" {

int ggan_version(void) { return GGAN_ABI_VERSION; }
const char* ggan_last_error(void) { return t_err; }
int ggan_prof_enable(int on) {
    g_prof_on = on != 0;
    return 0;
}

int ggan_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    g_prof.clear();
    return 0;
}

int ggan_prof_report(ggan_prof_rec* out, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    std::vector<double> fpl;          // flop per launch of record j (part of its key)
    for (auto& e : g_prof) {
        if (hipEventSynchronize(e.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) continue;
        int j = 0;
        for (; j < n; ++j)        // one record per problem shape: (kernel, grid, flop per launch)
            if (out[j].grid == e.grid && fpl[j] == e.flops && strncmp(out[j].name, e.name, sizeof(out[j].name) - 1) == 0) break;
        if (j == n) {
            if (n >= cap) continue;
            memset(&out[n], 0, sizeof(out[n]));
            strncpy(out[n].name, e.name, sizeof(out[n].name) - 1);
            out[n].grid = e.grid;
            fpl.push_back(e.flops);
            ++n;
        }
        out[j].total_ms += ms;
        out[j].launches += 1;
        out[j].flops += e.flops;
        out[j].bytes += e.bytes;
    }
    return n;
}

}  // extern "C"
