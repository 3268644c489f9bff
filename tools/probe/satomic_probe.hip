// MEASUREMENT PROBE (not part of the library): can a persistent workgroup hand tiles out DYNAMICALLY without LDS and without a returning
// vector-memory operation?  (LDS is full in the ping-pong kernels, and anything that returns through vmcnt waits for the LDS-DMA queue.)
// Candidate: scalar memory atomics, which return through lgkmcnt.  Wave 0 of a workgroup claims the next tile with s_atomic_add on a
// global counter and publishes it into a per-workgroup mailbox ring with s_atomic_swap; the other waves poll the mailbox with
// s_atomic_or(.., 0).  Every spin is BOUNDED (a stuck wave reports instead of hanging).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/satomic_probe.hip -o tools/probe/satomic_probe && tools/probe/satomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned s_atomic_add_u32(unsigned* p, unsigned v)
{
    unsigned r = v;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(r) : "s"(p) : "memory");
    return r;
}
__device__ __forceinline__ unsigned s_atomic_swap_u32(unsigned* p, unsigned v)
{
    unsigned r = v;
    asm volatile("s_atomic_swap %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(r) : "s"(p) : "memory");
    return r;
}
__device__ __forceinline__ unsigned s_atomic_or_u32(unsigned* p, unsigned v)
{
    unsigned r = v;
    asm volatile("s_atomic_or %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(r) : "s"(p) : "memory");
    return r;
}

// counter: next unclaimed tile; mailbox[wg][4]: (sequence + 1) << 20 | tile (0 = nothing yet); hits[tile][wave]++ ; err[0] = stuck waves
__global__ __launch_bounds__(512) void probe(unsigned* counter, unsigned* mailbox, unsigned* hits, unsigned* err, unsigned total, int work)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    unsigned* mb = mailbox + blockIdx.x * 4;
    float x = (float)lane;
    for (unsigned seq = 0;; ++seq) {
        unsigned tile;
        if (wave == 0) {
            tile = s_atomic_add_u32(counter, 1u);
            if (tile > 0xfffffu) tile = 0xfffffu;
            s_atomic_swap_u32(mb + (seq & 3), ((seq + 1) << 20) | tile);
        } else {
            unsigned v = 0;
            int spins = 0;
            for (; spins < 2000000; ++spins) {
                v = s_atomic_or_u32(mb + (seq & 3), 0u);
                if ((v >> 20) == ((seq + 1) & 0xfffu)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (spins == 2000000) { if (lane == 0) atomicAdd(err, 1u); return; }
            tile = v & 0xfffffu;
        }
        if (tile >= total) break;
        if (lane == 0) atomicAdd(hits + (size_t)tile * 8 + wave, 1u);
        for (int i = 0; i < work; ++i) x = x * 1.0001f + 0.5f;        // "the tile"
        __syncthreads();                                            // the kernels have >= 8 barriers per tile: waves stay within one tile of each other
    }
    if (x == 12345.678f) err[1] = 1;
}

int main()
{
    const unsigned total = 20000;
    for (int grid : {64, 256, 512}) {
        for (int work : {200, 5000}) {
            unsigned *counter, *mailbox, *hits, *err;
            CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&mailbox, grid * 16)); CHECK(hipMalloc(&hits, (size_t)total * 32)); CHECK(hipMalloc(&err, 8));
            CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(mailbox, 0, grid * 16)); CHECK(hipMemset(hits, 0, (size_t)total * 32)); CHECK(hipMemset(err, 0, 8));
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 0, 0, counter, mailbox, hits, err, total, work);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned> h((size_t)total * 8);
            unsigned e[2], c;
            CHECK(hipMemcpy(h.data(), hits, h.size() * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(e, err, 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&c, counter, 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (unsigned v : h) bad += v != 1;
            printf("grid %3d work %5d: %.3f ms, %.2f us per tile and workgroup, counter %u, tile x wave entries != 1: %zu, stuck waves %u\n", grid, work, ms,
                   ms * 1e3 * grid / total, c, bad, e[0]);
            CHECK(hipFree(counter)); CHECK(hipFree(mailbox)); CHECK(hipFree(hits)); CHECK(hipFree(err));
        }
    }
    return 0;
}
