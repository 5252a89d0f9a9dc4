// ROCm 7.2 clang: __builtin_bit_cast(float, v[i]) / (.., v.y) on an ELEMENT of a vector (k1) reads element 0 for every i -- here it even narrows the
// dwordx4 load to one dword; bit_cast of the whole vector (k2), or the element passed through a by-value parameter, is correct.  hipcc -O3 --offload-arch=gfx950 -S
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int v4u __attribute__((vector_size(16)));
__global__ void k1(const float *tab, int n, float ra, float *out)
{
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)tab, (short)0, n * 4, 0x00020000);
    const uint32_t tq = 16u * threadIdx.x;
    const v4u cs4 = __builtin_amdgcn_raw_buffer_load_b128(rt, tq, 0, 0);
    out[4 * threadIdx.x + 0] = ra * __builtin_bit_cast(float, cs4[0]);
    out[4 * threadIdx.x + 1] = ra * __builtin_bit_cast(float, cs4[1]) + 1.f;
    out[4 * threadIdx.x + 2] = ra * __builtin_bit_cast(float, cs4[2]) + 2.f;
    out[4 * threadIdx.x + 3] = ra * __builtin_bit_cast(float, cs4[3]) + 3.f;
}
__global__ void k2(const float *tab, int n, float ra, float *out)
{
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)tab, (short)0, n * 4, 0x00020000);
    const uint32_t tq = 16u * threadIdx.x;
    const v4u cs4 = __builtin_amdgcn_raw_buffer_load_b128(rt, tq, 0, 0);
    float4 f = __builtin_bit_cast(float4, cs4);
    out[4 * threadIdx.x + 0] = ra * f.x;
    out[4 * threadIdx.x + 1] = ra * f.y + 1.f;
    out[4 * threadIdx.x + 2] = ra * f.z + 2.f;
    out[4 * threadIdx.x + 3] = ra * f.w + 3.f;
}
