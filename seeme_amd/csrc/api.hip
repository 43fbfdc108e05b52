// api.hip -- version / error plumbing of the C-ABI.
#include "api_util.hpp"
#include "../../include/seeme_hip.h"

static thread_local char g_err[512] = "";

int seeme_fail(const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return 1;
}
int seeme_fail_hip(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return 2;
}
int seeme_check_launch(const char* kernel) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "launch of %s failed: %s", kernel, hipGetErrorString(e));
        return 3;
    }
    return 0;
}
extern "C" int seeme_version(void) { return 100; }
extern "C" const char* seeme_last_error(void) { return g_err; }
