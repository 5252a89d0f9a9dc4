// Micro-benchmark: cost of one wave-level vector memory instruction on gfx950 as a function of width, alignment and lane stride.
// Data set is small (L2 / L1 resident) so the number is the texture-addresser / L1 path, not HBM.
//   hipcc -O3 --offload-arch=gfx950 tools/ta_microbench.hip -o tools/ta_microbench && tools/ta_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(1)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32x1 __attribute__((aligned(1)));

template <typename V> __device__ inline uint32_t fold(V v);
template <> __device__ inline uint32_t fold<u32x1>(u32x1 v) { return v; }
template <> __device__ inline uint32_t fold<u32x2>(u32x2 v) { return v.x ^ v.y; }
template <> __device__ inline uint32_t fold<u32x3>(u32x3 v) { return v.x ^ v.y ^ v.z; }
template <> __device__ inline uint32_t fold<u32x4>(u32x4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// every lane reads `iters` x 8 vectors; lane l of wave w starts at base + w*wave_stride + l*lane_stride + misalign, successive loads step by row_stride
template <typename V>
__global__ __launch_bounds__(256) void k_loads(const char *buf, size_t limit, int lane_stride, int row_stride, int misalign, int iters, uint32_t *sink, int floor4 = 0)
{
    // every access is buf[off + k*row_stride .. +16) with off < limit and limit + 8*row_stride + 16 <= buffer size (host checks)
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    size_t off = ((size_t)wave * 4096 + (size_t)lane * lane_stride + misalign) % limit;
    if (floor4) off &= ~(size_t)3;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        V v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *(const V *)(buf + off + (size_t)k * row_stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= fold<V>(v[k]);
        off += 8 * (size_t)row_stride;
        if (off >= limit) off -= limit;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <typename V>
static void run(const char *name, const char *buf, size_t bytes, int lane_stride, int row_stride, int misalign, uint32_t *sink, int floor4 = 0)
{
    const int iters = 64, blocks = 256 * 8;  // 8 blocks (32 waves) per CU
    const size_t margin = 8 * (size_t)row_stride + 64;
    if (bytes <= 2 * margin) { printf("buffer too small for %s\n", name); return; }
    const size_t limit = bytes - margin;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_loads<V>, dim3(blocks), dim3(256), 0, 0, buf, limit, lane_stride, row_stride, misalign, iters, sink, floor4);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_loads<V>, dim3(blocks), dim3(256), 0, 0, buf, limit, lane_stride, row_stride, misalign, iters, sink, floor4);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_cu = 5.0 * blocks * 4 * iters * 8 / 256.0;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-34s lane_stride %3d row_stride %5d misalign %d : %6.1f cycles / wave-instruction / CU  (%.0f GB/s requested)\n", name, lane_stride, row_stride, misalign,
           cycles / insts_per_cu, 5.0 * blocks * 256 * (double)iters * 8 * sizeof(V) / (ms * 1e-3) / 1e9);
}

int main()
{
    const size_t bytes = 8u << 20;  // 8 MiB: L2 resident
    char *buf; uint32_t *sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 256);
    hipMemset(buf, 1, bytes);
    for (int mis = 0; mis < 2; ++mis) {
        run<u32x1>("dword, contiguous", buf, bytes, 4, 256, mis, sink);
        run<u32x2>("dwordx2, contiguous", buf, bytes, 8, 512, mis, sink);
        run<u32x3>("dwordx3, contiguous", buf, bytes, 12, 768, mis, sink);
        run<u32x4>("dwordx4, contiguous", buf, bytes, 16, 1024, mis, sink);
    }
    run<u32x2>("dwordx2, 3-byte lane stride (warp)", buf, bytes, 3, 11520, 0, sink);
    run<u32x3>("dwordx3 floor4, 3-byte lane stride", buf, bytes, 3, 11520, 0, sink, 1);
    run<u32x2>("dwordx2 floor4, 3-byte lane stride", buf, bytes, 3, 11520, 0, sink, 1);
    run<u32x4>("dwordx4 floor4, 3-byte lane stride", buf, bytes, 3, 11520, 0, sink, 1);
    run<u32x3>("dwordx3, 12-byte stride misaligned 1", buf, bytes, 12, 11520, 1, sink);
    run<u32x4>("dwordx4 floor4, 12-byte stride mis 1", buf, bytes, 12, 11520, 1, sink, 1);
    run<u32x1>("dword, 3-byte lane stride", buf, bytes, 3, 11520, 0, sink);
    run<u32x1>("dword floor4, 3-byte lane stride", buf, bytes, 3, 11520, 0, sink, 1);
    run<u32x2>("dwordx2, 12-byte lane stride", buf, bytes, 12, 11520, 0, sink);
    run<u32x4>("dwordx4, 12-byte lane stride", buf, bytes, 12, 11520, 0, sink);
    run<u32x4>("dwordx4, 24-byte lane stride", buf, bytes, 24, 11520, 0, sink);
    run<u32x1>("dword, 4-byte stride, far rows", buf, bytes, 4, 11520, 0, sink);
    run<u32x4>("dwordx4, contiguous, far rows", buf, bytes, 16, 11520, 0, sink);
    run<u32x1>("dword, 64-byte lane stride", buf, bytes, 64, 4096, 0, sink);
    run<u32x1>("dword, 128-byte lane stride", buf, bytes, 128, 8192, 0, sink);
    return 0;
}
