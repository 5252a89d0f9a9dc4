// does v_cvt_pk_u8_f32 round to nearest even and saturate like saturate_cast<uchar>(cvRound(x))?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float *in, unsigned *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned r = 0;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(in[i]));
    out[i] = r;
}
int main()
{
    std::vector<float> h;
    for (int k = -8; k <= 2100; ++k) { h.push_back(k * 0.125f); }
    h.push_back(-1e9f); h.push_back(1e9f); h.push_back(NAN); h.push_back(254.5f); h.push_back(255.5f); h.push_back(255.49999f); h.push_back(-0.5f); h.push_back(-0.50001f);
    for (int i = 0; i < 2000; ++i) { float v = (rand() % 26000) / 100.0f; h.push_back(v); h.push_back(nextafterf(floorf(v) + 0.5f, 1e9f)); h.push_back(nextafterf(floorf(v) + 0.5f, -1e9f)); }
    int n = h.size();
    float *d; unsigned *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, d, o, n);
    std::vector<unsigned> r(n);
    hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        float v = h[i];
        float rr = nearbyintf(v);
        int want = std::isnan(v) ? 0 : rr < 0 ? 0 : rr > 255 ? 255 : (int)rr;
        if ((int)r[i] != want) { if (bad < 12) printf("x = %.6f (%a): got %u, cvRound+saturate gives %d\n", v, v, r[i], want); ++bad; }
    }
    printf("%d values, %d differ\n", n, bad);
    return 0;
}
