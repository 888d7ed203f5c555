// Error plumbing, version and run-time options of the C ABI.
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <atomic>

namespace ytvln {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// option table: name (environment variable = "YTVLN_" + name), default
struct OptionEntry { const char* name; int dflt; };
static const OptionEntry kOptions[OPT_COUNT] = {
    {"ATTN_W1", 7}, {"ATTN_W1_DKV_ANY", 0}, {"ATTN_DSPLIT", 1}, {"GEMM_TILE", -1}, {"GEMM_SPLITS", -1}, {"GEMM_SPLIT_MAP", 1}, {"GEMM_GENERIC", 0},
    {"GEMM_SW", 0}, {"GEMM_SK", 0}, {"GEMM_SK_TILE", -1}, {"GEMM_SK_GROUPS", 8}, {"GEMM_T224", 1},
    {"GEMM_BF16_FORM", 0}, {"GEMM_STAGGER", 0}, {"ATTN_DKV_SPLIT", 0}, {"GEMM_BF16_WIDE", 0},
};
static std::atomic<int> g_opt_value[OPT_COUNT];
static std::atomic<int> g_opt_set[OPT_COUNT];          // 0: not read yet, 1: holds a value

int opt(int id) {
    if (id < 0 || id >= OPT_COUNT) return 0;
    if (!g_opt_set[id].load(std::memory_order_acquire)) {
        char env[64];
        snprintf(env, sizeof(env), "YTVLN_%s", kOptions[id].name);
        const char* v = getenv(env);
        g_opt_value[id].store(v ? atoi(v) : kOptions[id].dflt, std::memory_order_relaxed);
        g_opt_set[id].store(1, std::memory_order_release);
    }
    return g_opt_value[id].load(std::memory_order_relaxed);
}

static int option_index(const char* name) {
    if (!name) return -1;
    if (strncmp(name, "YTVLN_", 6) == 0) name += 6;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(name, kOptions[i].name) == 0) return i;
    return -1;
}

}  // namespace ytvln

extern "C" int ytvln_option_count(void) { return ytvln::OPT_COUNT; }
extern "C" const char* ytvln_option_name(int index) { return (index >= 0 && index < ytvln::OPT_COUNT) ? ytvln::kOptions[index].name : ""; }
extern "C" int ytvln_set_option(const char* name, int value) {
    const int i = ytvln::option_index(name);
    if (i < 0) return ytvln::fail(-1, "set_option: unknown option '%s'", name ? name : "(null)");
    ytvln::g_opt_value[i].store(value, std::memory_order_relaxed);
    ytvln::g_opt_set[i].store(1, std::memory_order_release);
    return 0;
}
extern "C" int ytvln_get_option(const char* name, int* value) {
    const int i = ytvln::option_index(name);
    if (i < 0 || !value) return ytvln::fail(-1, "get_option: unknown option '%s' or NULL output", name ? name : "(null)");
    *value = ytvln::opt(i);
    return 0;
}

extern "C" int ytvln_version(void) { return YTVLN_ABI_VERSION; }
extern "C" int64_t ytvln_attn_problem_size(void) { return (int64_t)sizeof(ytvln_attn_problem); }

extern "C" int ytvln_set_host_wait(int device, int blocking) {
    using namespace ytvln;
    YT_REQUIRE(device >= 0, "set_host_wait: bad device %d", device);
    int prev = -1;
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess) e = hipSetDevice(device);
    if (e == hipSuccess) e = hipSetDeviceFlags(blocking ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(-3, "set_host_wait: %s", hipGetErrorString(e)); }
    return 0;
}
extern "C" const char* ytvln_last_error(void) { return ytvln::err_buf(); }
