// Fourth translation unit of libsimfire_hip.so (see simfire_hip_run2.hip): the team kernels for two bitmap words per thread - grids of
// 1025 .. 2048 columns, BASELINE config C4.  Everything it shares with the first unit
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

// Workgroups of this instantiation (block threads, lds bytes of dynamic LDS) one CU holds at once, as the runtime computes it from the
// kernel's registers and LDS (hipOccupancyMaxActiveBlocksPerMultiprocessor): what the host sizes a team launch's grid by - the members of a
// team wait for each other inside the launch, so a grid the chip cannot hold at once would be a team that is never complete.
static hipError_t occupancy_of(run_fn kern, unsigned block, size_t lds, int *per_cu)
{
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, reinterpret_cast<const void *>(kern), (int)block, lds);
}

hipError_t sf_run4_team2_occupancy(int att, int diag, int mit, unsigned block, size_t lds, int *per_cu)
{
    static const run_fn table[2][2] = {{k_run<2, 0, -1, -1, 1>, k_run<2, 0, 1, -1, 1>}, {k_run<2, 1, -1, -1, 1>, k_run<2, 1, 1, -1, 1>}};
    static const run_fn table_c4[2] = {k_run<2, 0, 1, 0, 1>, k_run<2, 1, 1, 0, 1>};
    return occupancy_of((diag && !mit) ? table_c4[att ? 1 : 0] : table[att ? 1 : 0][diag ? 1 : 0], block, lds, per_cu);
}

hipError_t sf_run4_launch_team2(int att, int diag, unsigned grid, unsigned block, size_t lds, bool set_lds, hipStream_t stream,
                                const void *args, size_t args_bytes, int n_steps, int vcap)
{
    // [attenuation][diagonal spread read at run time / known to be on]
    static const run_fn table[2][2] = {{k_run<2, 0, -1, -1, 1>, k_run<2, 0, 1, -1, 1>}, {k_run<2, 1, -1, -1, 1>, k_run<2, 1, 1, -1, 1>}};
    if (args_bytes != sizeof(StepArgs)) return hipErrorInvalidValue;
    StepArgs a;
    memcpy(&a, args, sizeof a);
    // (diagonal spread on, no control lines inside the launch = BASELINE config C4: its own instantiations, without the control-line
    // code and its registers)
    static const run_fn table_c4[2] = {k_run<2, 0, 1, 0, 1>, k_run<2, 1, 1, 0, 1>};
    const run_fn kern = (diag && !a.mit) ? table_c4[att ? 1 : 0] : table[att ? 1 : 0][diag ? 1 : 0];
    if (set_lds) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, a, n_steps, vcap, 64);
    return hipSuccess;
}

#ifdef SF_WIN_PROF
// (profiles/win_prof.sh: this unit's own copy of the window loop's phase clocks - C4's team kernels live here)
extern "C" int sf_debug_win_prof4(unsigned long long *out /* [1024][16][8] */)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_win_prof), sizeof(unsigned long long) * 1024 * 16 * 8) == hipSuccess ? 0 : -1;
}
#endif
