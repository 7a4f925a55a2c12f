// Second translation unit of libsimfire_hip.so: instantiations of k_run that only some handles ever launch - the team launch on one-word
// rows (k_run<1, ..., TEAM = 1>) and the closed loop (k_run<MIT = -2>) - compiled beside simfire_hip.hip so that the library builds in the
// time of the largest unit (python -m simfire_amd.build runs the compiles side by side; simfire_hip_run3.hip: the plain kernels for
// several bitmap words per thread, simfire_hip_run4.hip: the team kernels for two).  Everything it shares with the first unit
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

// One bitmap word per thread.  att: attenuate_line_ros; diag: diagonal_spread.  set_lds: raise the kernel's dynamic-LDS limit first.
// (two words per thread: sf_run4_launch_team2, simfire_hip_run4.hip)
hipError_t sf_run2_launch_team(int att, int diag, unsigned grid, unsigned block, size_t lds, bool set_lds, hipStream_t stream,
                               const void *args, size_t args_bytes, int n_steps, int vcap)
{
    // [attenuation][diagonal spread read at run time / known to be on]
    static const run_fn table[2][2] = {{k_run<1, 0, -1, -1, 1>, k_run<1, 0, 1, -1, 1>}, {k_run<1, 1, -1, -1, 1>, k_run<1, 1, 1, -1, 1>}};
    if (args_bytes != sizeof(StepArgs)) return hipErrorInvalidValue;
    StepArgs a;
    memcpy(&a, args, sizeof a);
    const run_fn kern = table[att ? 1 : 0][diag ? 1 : 0];
    if (set_lds) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, a, n_steps, vcap, 64);
    return hipSuccess;
}

hipError_t sf_run2_team_occupancy(int att, int diag, unsigned block, size_t lds, int *per_cu)
{
    static const run_fn table[2][2] = {{k_run<1, 0, -1, -1, 1>, k_run<1, 0, 1, -1, 1>}, {k_run<1, 1, -1, -1, 1>, k_run<1, 1, 1, -1, 1>}};
    return occupancy_of(table[att ? 1 : 0][diag ? 1 : 0], block, lds, per_cu);
}
hipError_t sf_run2_join_occupancy(int att, unsigned block, size_t lds, int *per_cu)
{
    static const run_fn table[2] = {k_run<1, 0, 1, 0, 2>, k_run<1, 1, 1, 0, 2>};
    return occupancy_of(table[att ? 1 : 0], block, lds, per_cu);
}

// Teams that grow inside the launch (k_run<TEAM = 2>): one-word rows, diagonal spread, no control lines inside the launch.
hipError_t sf_run2_launch_join(int att, unsigned grid, unsigned block, size_t lds, bool set_lds, hipStream_t stream,
                               const void *args, size_t args_bytes, int n_steps, int vcap)
{
    static const run_fn table[2] = {k_run<1, 0, 1, 0, 2>, k_run<1, 1, 1, 0, 2>};
    if (args_bytes != sizeof(StepArgs)) return hipErrorInvalidValue;
    StepArgs a;
    memcpy(&a, args, sizeof a);
    const run_fn kern = table[att ? 1 : 0];
    if (set_lds) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, a, n_steps, vcap, 64);
    return hipSuccess;
}

// the closed loop of sf_loop_start: one workgroup per environment, steps until the host's stop
hipError_t sf_run2_launch_loop(int att, int diag, unsigned grid, unsigned block, size_t lds, bool set_lds, hipStream_t stream,
                               const void *args, size_t args_bytes, int vcap)
{
    // (diagonal spread is looked up at run time: the two instantiations that knew it at compile time were retired in round 6 for k_win's two -
    // a select or two per batch in a call whose time is the signalling between host and device)
    static const run_fn table[2] = {k_run<1, 0, -1, -2>, k_run<1, 1, -1, -2>};
    if (args_bytes != sizeof(StepArgs)) return hipErrorInvalidValue;
    (void)diag;
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
