// Train-mode BatchNorm statistics out of a conv epilogue: commit one workgroup's per-channel partial sums (see ConvDesc::stat_rep).
#pragma once
#include <hip/hip_runtime.h>

// Threads tid < bn carry the workgroup's sums a1 / a2 of channel n0 + tid.  rep > 1: into replica blockIdx.x % rep (replica r sits at
// stat_sum + (1 + r) * 2 * cout; hn_launch_stat_replica_sum adds them into the leading [sum | sq] afterwards).
// (A ticket -- "the last workgroup adds the replicas up" -- was tried first: the agent-scope release fence in front of the ticket
//  writes the workgroup's L2 lines back and made the convs 3.5x slower.)
__device__ __forceinline__ void hn_stat_commit(double* stat_sum, double* stat_sq, int rep, int cout, int n0, int bn, int tid, float a1, float a2)
{
    if (tid < bn) {
        const size_t r = rep > 1 ? (size_t)(1 + (blockIdx.x & (unsigned)(rep - 1))) * 2 * (size_t)cout : 0;
        atomicAdd(stat_sum + r + n0 + tid, (double)a1);
        atomicAdd(stat_sq + r + n0 + tid, (double)a2);
    }
}
