// Error reporting, per-step state, HIP-graph capture helpers.
#include <stdarg.h>
#include <string.h>

#include "cr_common.hpp"

static thread_local char g_err[512] = "";

int cr_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int cr_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cr_set_error(CR_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return CR_OK;
}

extern "C" int cr_version(void) { return 100; }
extern "C" const char* cr_last_error(void) { return g_err; }

__global__ void k_step_begin(float* state) {
    if (threadIdx.x < 4) state[threadIdx.x] = 0.0f;
    if (threadIdx.x == 4) {
        uint32_t* step = reinterpret_cast<uint32_t*>(state + 4);
        *step = *step + 1u;
    }
}

extern "C" int cr_step_begin(float* state, void* stream) {
    CR_REQUIRE(state != nullptr, "cr_step_begin: state is NULL");
    hipLaunchKernelGGL(k_step_begin, dim3(1), dim3(64), 0, cr_stream(stream), state);
    return cr_check_launch("cr_step_begin");
}

// ---- graph capture ---------------------------------------------------------------
extern "C" int cr_graph_begin(void* stream) {
    hipError_t e = hipStreamBeginCapture(cr_stream(stream), hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return cr_set_error(CR_ERR_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
    return CR_OK;
}

extern "C" int cr_graph_end(void* stream, void** graph_exec_out) {
    CR_REQUIRE(graph_exec_out != nullptr, "cr_graph_end: output pointer is NULL");
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(cr_stream(stream), &graph);
    if (e != hipSuccess) return cr_set_error(CR_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) return cr_set_error(CR_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    *graph_exec_out = exec;
    return CR_OK;
}

extern "C" int cr_graph_launch(void* graph_exec, void* stream) {
    CR_REQUIRE(graph_exec != nullptr, "cr_graph_launch: graph is NULL");
    hipError_t e = hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), cr_stream(stream));
    if (e != hipSuccess) return cr_set_error(CR_ERR_HIP, "hipGraphLaunch: %s", hipGetErrorString(e));
    return CR_OK;
}

extern "C" int cr_graph_destroy(void* graph_exec) {
    if (graph_exec) hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec));
    return CR_OK;
}
