// Third translation unit of libsimfire_hip.so (see simfire_hip_run2.hip): k_run for two and four bitmap words per thread - more than 1024
// rows per workgroup, or 8-wave workgroups on 1024 rows (the many-environments regime).  Everything it shares with the first unit
// comes from the same headers (all in anonymous namespaces: each unit has its own copy of the device helpers); the launch arguments
// cross the boundary as bytes.
// Replaces (like sf_run_kernels.h): n calls of RothermelFireManager.update per environment, simfire/game/managers/fire.py:616-719.
// (only the k_run instantiations below are compiled here: the kernels every handle launches live in simfire_hip.hip alone)
#define SF_RUN_UNIT 1
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/simfire_hip.h"
#include "sf_common.h"
#include "sf_step_kernels.h"
#include "sf_aux_kernels.h"
#include "sf_run_kernels.h"

namespace {
typedef void (*run_fn)(StepArgs, int, int, int);
}

// the closed loop of sf_loop_start with two bitmap rows per thread: 8-wave workgroups on 1024 rows, two to a CU - the light loop, which leaves half
// of every CU to the harness's own kernels (SF_TUNE_LOOP_LIGHT).  Diagonal spread is looked up at run time.
hipError_t sf_run3_launch_loop2(int att, unsigned grid, unsigned block, size_t lds, bool set_lds, hipStream_t stream,
                                const void *args, size_t args_bytes, int vcap)
{
    static const run_fn table[2] = {k_run<2, 0, -1, -2>, k_run<2, 1, -1, -2>};
    if (args_bytes != sizeof(StepArgs)) return hipErrorInvalidValue;
    StepArgs a;
    memcpy(&a, args, sizeof a);
    const run_fn kern = table[att ? 1 : 0];
    if (set_lds) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, a, 0x7FFFFFFF, vcap, 64);
    return hipSuccess;
}

// which: 1 / 2 = two / four bitmap words per thread
hipError_t sf_run3_launch_plain(int which, int att, int diag, unsigned grid, unsigned block, size_t lds, bool set_lds, hipStream_t stream,
                                const void *args, size_t args_bytes, int n_steps, int vcap, int bsz)
{
    // [words per thread 2 / 4][attenuation off / on][diagonal spread read at run time / known to be on] (four words: read at run time only)
    static const run_fn table[2][2][2] = {
        {{k_run<2, 0, -1, -1>, k_run<2, 0, 1, -1>}, {k_run<2, 1, -1, -1>, k_run<2, 1, 1, -1>}},
        {{k_run<kRunMaxD, 0, -1, -1>, k_run<kRunMaxD, 0, -1, -1>}, {k_run<kRunMaxD, 1, -1, -1>, k_run<kRunMaxD, 1, -1, -1>}}};
    if (args_bytes != sizeof(StepArgs) || which < 1 || which > 2) return hipErrorInvalidValue;
    StepArgs a;
    memcpy(&a, args, sizeof a);
    // two words per thread, diagonal spread known to be on, no control lines inside the launch: the instantiations with the window phase
    static const run_fn table_nomit[2] = {k_run<2, 0, 1, 0>, k_run<2, 1, 1, 0>};
    const run_fn kern = (which == 1 && diag && !a.mit) ? table_nomit[att ? 1 : 0] : table[which - 1][att ? 1 : 0][diag ? 1 : 0];
    if (set_lds) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, a, n_steps, vcap, bsz);
    return hipSuccess;
}
