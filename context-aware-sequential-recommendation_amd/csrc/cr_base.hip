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

// ---- the next step's id batch out of a resident ring ------------------------------------
// dst[0 .. n4) (16-byte groups) <- slot ((*step) + 1) mod n_slots of the ring: launched beside cr_adam_step (which reads no ids
// unless it is row-sparse) on a forked branch of the step's graph, so the batch of step t + 1 is in the static id buffers when
// step t ends and no copy stands between two steps.  `step` is the head kernel's snapshot of the step number (state[11]): Adam
// advances state[4] while this kernel runs.
__global__ __launch_bounds__(256) void k_ids_ring_next(const int32_t* ring, int n_slots, long long slot_elems, int32_t* dst, const uint32_t* step) {
    const uint32_t t = *step + 1u;
    const int32_t* src1 = ring + (long long)(t % (uint32_t)n_slots) * slot_elems;
    const long long first = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    if ((slot_elems & 3) == 0 && (((uintptr_t)ring | (uintptr_t)dst) & 15) == 0) {
        const int4* src = reinterpret_cast<const int4*>(src1);
        int4* dst4 = reinterpret_cast<int4*>(dst);
        for (long long i = first; i < (slot_elems >> 2); i += stride) dst4[i] = src[i];
    } else {                                             // (a slot that is no multiple of 16 bytes)
        for (long long i = first; i < slot_elems; i += stride) dst[i] = src1[i];
    }
}

extern "C" int cr_ids_ring_next(const int32_t* ring, int n_slots, int64_t slot_elems, int32_t* dst, const uint32_t* step, void* stream) {
    CR_REQUIRE(ring && dst && step, "cr_ids_ring_next: NULL pointer");
    CR_REQUIRE(n_slots > 0 && slot_elems > 0, "cr_ids_ring_next: n_slots > 0 and slot_elems > 0 expected (got %d, %lld)", n_slots, (long long)slot_elems);
    const long long slot4 = (slot_elems + 3) / 4;
    const int grid = (int)((slot4 + 255) / 256 > 1024 ? 1024 : (slot4 + 255) / 256);
    hipLaunchKernelGGL(k_ids_ring_next, dim3(grid), dim3(256), 0, cr_stream(stream), ring, n_slots, (long long)slot_elems, dst, step);
    return cr_check_launch("cr_ids_ring_next");
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
