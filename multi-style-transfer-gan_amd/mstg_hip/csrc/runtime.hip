// Host-side runtime of libmstg_hip.so: version / error text, and the MSTG_* switches read once per load.
#include <stdlib.h>

#include <cxxabi.h>

#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace mstg {

thread_local char g_last_error[256] = "";
thread_local bool t_ws_packed = false;

static const char* const kEnvNames[ENV_COUNT] = {
    "MSTG_ATTN_BLK4", "MSTG_ATTN_BLK64", "MSTG_WGRAD_1X1", "MSTG_WGRAD_TS_MAXCH", "MSTG_WGRAD_PLAIN", "MSTG_WGRAD_OLD", "MSTG_NO_DPACK",
    "MSTG_IGEMM", "MSTG_STREAM", "MSTG_PF", "MSTG_WGLOB", "MSTG_HEAVY_PER_CU", "MSTG_DBG", "MSTG_DBG_LDS_KB", "MSTG_MS_WGRAD_PACKED",
    "MSTG_MS_FWD4", "MSTG_NO_PACK_CACHE", "MSTG_P32", "MSTG_P32_TH", "MSTG_P32_WLDS", "MSTG_P32_DBG", "MSTG_P32_OCC", "MSTG_ATTN_REG", "MSTG_NFW_ADAPT", "MSTG_BSUMS_ALL", "MSTG_ATTN_BIG32", "MSTG_NORM_WGS",
    "MSTG_CONV_IMG"};

struct EnvCache {
    char val[ENV_COUNT][32];
    bool set[ENV_COUNT];
    EnvCache() { refresh(); }
    void refresh() {
        for (int k = 0; k < ENV_COUNT; ++k) {
            const char* e = getenv(kEnvNames[k]);
            set[k] = e != nullptr;
            snprintf(val[k], sizeof(val[k]), "%s", e ? e : "");
        }
    }
};
static EnvCache g_env;  // constructed when the shared object is loaded

const char* env_get(EnvKnob k) { return g_env.set[k] ? g_env.val[k] : nullptr; }

// ---- per-launch profiler ---------------------------------------------------------------------------------------------------
bool g_prof_on = false;
struct ProfRec {
    const void* fn;
    hipStream_t st;
    hipEvent_t e0, e1;
};
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;
static thread_local int t_prof_open = -1;  // record this thread opened and has not closed (autograd launches from its own thread)

void prof_begin(const void* fn, hipStream_t st) {
    ProfRec r{fn, st, nullptr, nullptr};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    t_prof_open = (int)g_prof.size();
    g_prof.push_back(r);
}
void prof_end(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (t_prof_open < 0 || t_prof_open >= (int)g_prof.size()) return;
    (void)hipEventRecord(g_prof[t_prof_open].e1, st);
    t_prof_open = -1;
}
static void prof_clear() {
    for (auto& r : g_prof) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
}

}  // namespace mstg

extern "C" const char* mstg_version(void) { return "mstg-hip 0.2.0 gfx950"; }
extern "C" const char* mstg_arch(void) { return "gfx950"; }
extern "C" const char* mstg_last_error(void) { return mstg::g_last_error; }
extern "C" void mstg_env_refresh(void) { mstg::g_env.refresh(); }

extern "C" int mstg_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(mstg::g_prof_mu);
    if (on) mstg::prof_clear();
    mstg::g_prof_on = on != 0;
    return MSTG_OK;
}
extern "C" int mstg_prof_count(void) {
    std::lock_guard<std::mutex> lk(mstg::g_prof_mu);
    return (int)mstg::g_prof.size();
}
extern "C" int mstg_prof_get(int i, char* name, size_t name_cap, float* ms) {
    mstg::ProfRec r;
    {
        std::lock_guard<std::mutex> lk(mstg::g_prof_mu);
        if (i < 0 || i >= (int)mstg::g_prof.size() || !name || !ms || name_cap < 8) return mstg::fail_arg(MSTG_E_BADARG, "mstg_prof_get: bad index or buffer");
        r = mstg::g_prof[i];
    }
    hipError_t e = hipEventSynchronize(r.e1);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, r.e0, r.e1);
    if (e != hipSuccess) return mstg::fail_launch(e, "mstg_prof_get");
    const char* mangled = hipKernelNameRefByPtr(r.fn, r.st);
    std::string nm = mangled ? mangled : "?";
    int status = 0;
    char* dem = mangled ? abi::__cxa_demangle(mangled, nullptr, nullptr, &status) : nullptr;
    if (dem && status == 0) nm = dem;
    free(dem);
    // "void mstg::kernel<args>(parameter list)" -> "kernel<args>": cut the parameter list (the last top-level parenthesis group),
    // the return type and the namespace, so that the symbol reads as rocprofv3's kernel-trace prints it minus the decoration
    int depth = 0, cut = -1;
    for (int k = (int)nm.size() - 1; k >= 0; --k) {
        if (nm[k] == ')') ++depth;
        else if (nm[k] == '(' && --depth == 0) { cut = k; break; }
    }
    if (cut > 0) nm.resize(cut);
    if (nm.rfind("void ", 0) == 0) nm.erase(0, 5);
    if (nm.rfind("mstg::", 0) == 0) nm.erase(0, 6);
    snprintf(name, name_cap, "%s", nm.c_str());
    return MSTG_OK;
}
