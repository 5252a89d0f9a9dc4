// buffer_load_dwordx4 ... lds on gfx950: lane L of a wave writes its 16 bytes at M0 + 16 * L (1 KiB contiguous per wave instruction);
// out-of-range lanes (hardware bounds check of the buffer resource) write zeros.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const uint8_t *src, uint32_t bytes, uint32_t pitch, u32x4_t *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[12288];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, (short)0, (int)bytes, 0x00020000);
    const int tid = threadIdx.x, wave = tid >> 6;
    const uint32_t voff = (uint32_t)(tid >> 4) * pitch + 16u * (tid & 15) + 48u;
    for (int k = 0; k < 3; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(s_tile + k * 4096 + wave * 1024), 16, voff, 16u * k * pitch, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    __syncthreads();
    for (int k = 0; k < 3; ++k) out[k * 256 + tid] = *(const u32x4_t *)(s_tile + k * 4096 + 16 * tid);
}
int main()
{
    const uint32_t pitch = 1024, rows = 40, bytes = pitch * rows;      // rows 40..47 are out of range
    std::vector<uint8_t> h(bytes);
    for (uint32_t i = 0; i < bytes; ++i) h[i] = (uint8_t)(i * 131u + (i >> 8));
    uint8_t *d; u32x4_t *o;
    hipMalloc(&d, bytes); hipMalloc(&o, 768 * 16);
    hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, bytes, pitch, o);
    std::vector<uint8_t> r(768 * 16);
    hipMemcpy(r.data(), o, r.size(), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int row = 0; row < 48; ++row)
        for (int b = 0; b < 256; ++b) {
            const uint32_t srcoff = row * pitch + 48 + b;
            const uint8_t want = srcoff < bytes ? h[srcoff] : 0;
            if (r[row * 256 + b] != want) { if (bad < 8) printf("row %d byte %d: got %u want %u\n", row, b, r[row * 256 + b], want); ++bad; }
        }
    printf("lds dma: %d mismatches of %d\n", bad, 48 * 256);
    return bad != 0;
}
