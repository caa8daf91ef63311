// Host-side runtime of libmstg_hip.so: version / error text, and the MSTG_* switches read once per load.
#include <stdlib.h>

#include <mutex>

#include "common.h"

namespace mstg {

thread_local char g_last_error[256] = "";

static const char* const kEnvNames[ENV_COUNT] = {
    "MSTG_ATTN_BLK4", "MSTG_ATTN_BLK64", "MSTG_WGRAD_1X1", "MSTG_WGRAD_TS_MAXCH", "MSTG_WGRAD_PLAIN", "MSTG_WGRAD_OLD", "MSTG_NO_DPACK",
    "MSTG_IGEMM", "MSTG_STREAM", "MSTG_PF", "MSTG_WGLOB", "MSTG_HEAVY_PER_CU", "MSTG_DBG", "MSTG_DBG_LDS_KB", "MSTG_MS_WGRAD_PACKED",
    "MSTG_MS_FWD4", "MSTG_NO_PACK_CACHE", "MSTG_P32", "MSTG_P32_TH", "MSTG_P32_WLDS", "MSTG_P32_DBG", "MSTG_P32_OCC"};

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

}  // namespace mstg

extern "C" const char* mstg_version(void) { return "mstg-hip 0.2.0 gfx950"; }
extern "C" const char* mstg_arch(void) { return "gfx950"; }
extern "C" const char* mstg_last_error(void) { return mstg::g_last_error; }
extern "C" void mstg_env_refresh(void) { mstg::g_env.refresh(); }
