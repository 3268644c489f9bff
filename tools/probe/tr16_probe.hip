// Build: hipcc --offload-arch=gfx950 -O2 -o tr16_probe tr16_probe.hip ; run on an MI355X.  Output pinned the lane mapping
// used by csrc/conv_wgrad_bf16.hip: within 16 lanes, lane i receives element i%4 of lanes i/4, i/4+4, i/4+8, i/4+12.
// Probe of ds_read_b64_tr_b16 (gfx950): which LDS elements does lane l receive?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int stride_elems)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    // lane l points at element  (l % 16) * stride + (l / 16) * 4   (16 rows per 16-lane group, groups step 4 columns)
    const int l = threadIdx.x;
    const unsigned short* p = lds + (l % 16) * stride_elems + (l / 16) * 4;
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main()
{
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {64, 16}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
        unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d elements: lane -> (row,col) x4 where element = row*stride + col\n", stride);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" (%2d,%2d)", h[l * 4 + j] / stride, h[l * 4 + j] % stride);
            printf("\n");
        }
    }
    return 0;
}
