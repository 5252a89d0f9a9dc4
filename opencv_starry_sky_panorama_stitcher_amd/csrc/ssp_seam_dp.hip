// ssp_seam_dp.hip -- cv.detail_DpSeamFinder('COLOR' | 'COLOR_GRAD').find(images, corners, masks)
// (stitching_detailed_enhanced.py:243-249: the reference's default seam finder "dp_colorgrad", called at :1618 on the float32
// seam-scale warps of :1601-1604).  OpenCV 4.6.0 stitching/src/seam_finders.cpp, class DpSeamFinder.
//
// Division of labour.  The finder has a data-parallel part and a graph part:
//   device  * gradients of every image (cvtColor BGR2GRAY + Sobel 3x3, COLOR_GRAD)                         k_dp_gradients
//           * the edge costs of every overlapping pair over its whole overlap rectangle -- they depend on
//             the two images only, not on the masks, so all pairs are priced in ONE launch before any cut  k_dp_pair_costs
//           * the dynamic programme of estimateSeam: one work-group sweeps the component's bounding box
//             column by column (row by row), three predecessors per cell, then walks the control map back k_dp_seam
//   host    * the component graph of one pair (labelling, adjacency, conflict resolution, seam tips, relabelling after a
//             cut): a strictly sequential, data-dependent walk over <= a few 100 kPix of labels whose every step decides the next
//             one.  On the device it would be hundreds of dependent micro-launches per pair; it stays in C++ here, inside this
//             library (no oracle code, no Python).
// The masks are read once, cut on the host copy pair by pair in the order OpenCV visits the pairs (std::sort by centre distance,
// reversed), and written back once.  Results equal the CPU oracle (oracle/orc_seam.c) bit for bit; the two share no code and
// use different algorithms for the component labelling (scan-line run filling here, a pixel stack there) and the clustering.
#include "ssp_internal.hpp"

#include <algorithm>
#include <emmintrin.h>
#include <chrono>
#include <climits>
#include <cmath>
#include <numeric>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>
#include <utility>

using namespace ssp;

namespace {


// ---- device side -------------------------------------------------------------------------------------------------------------------
struct GradImg { const void *img; size_t pitch; int w, h, depth; float *gx, *gy; };

__device__ inline float gray_at(const GradImg &g, int y, int x)
{
    // BORDER_REFLECT_101 of Sobel's default border
    x = x < 0 ? (g.w > 1 ? -x : 0) : x >= g.w ? (g.w > 1 ? 2 * g.w - 2 - x : 0) : x;
    y = y < 0 ? (g.h > 1 ? -y : 0) : y >= g.h ? (g.h > 1 ? 2 * g.h - 2 - y : 0) : y;
    float b, gg, r;
    if (g.depth == SSP_F32) {
        const float *p = (const float *)((const char *)g.img + (size_t)y * g.pitch) + (size_t)x * 3;
        b = p[0]; gg = p[1]; r = p[2];
    } else {
        const uint8_t *p = (const uint8_t *)g.img + (size_t)y * g.pitch + (size_t)x * 3;
        b = (float)p[0]; gg = (float)p[1]; r = (float)p[2];
    }
    return b * 0.114f + gg * 0.587f + r * 0.299f;
}

// Sobel as sepFilter2D evaluates it: row filter first ((-1 0 1): c - a, (1 2 1): a + b*2 + c), then the column filter
__global__ __launch_bounds__(256) void k_dp_gradients(const GradImg *imgs, const int *first_block, int n)
{
    int z = 0;
    while (z + 1 < n && (int)blockIdx.x >= first_block[z + 1]) ++z;
    const GradImg g = imgs[z];
    const int t = ((int)blockIdx.x - first_block[z]) * 256 + threadIdx.x;
    if (t >= g.w * g.h) return;
    const int y = t / g.w, x = t - y * g.w;
    float rd[3], rs[3];
    for (int k = 0; k < 3; ++k) {
        const float a = gray_at(g, y + k - 1, x - 1), b = gray_at(g, y + k - 1, x), c = gray_at(g, y + k - 1, x + 1);
        rd[k] = c - a;
        rs[k] = a + b * 2 + c;
    }
    g.gx[t] = rd[0] + rd[1] * 2 + rd[2];
    g.gy[t] = rs[2] - rs[0];
}

struct PairCost {
    int a, b;                  // image indices
    int iw, ih;                // overlap rectangle size
    int ax, ay, bx, by;        // its origin inside image a / image b
    float *cv, *ch;            // iw x ih each: cost of the edge left of / above the pixel (undefined in column 0 / row 0)
};

__device__ inline void px3(const GradImg &g, int y, int x, float v[3])
{
    if (g.depth == SSP_F32) {
        const float *p = (const float *)((const char *)g.img + (size_t)y * g.pitch) + (size_t)x * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
    } else {
        const uint8_t *p = (const uint8_t *)g.img + (size_t)y * g.pitch + (size_t)x * 3;
        v[0] = (float)p[0]; v[1] = (float)p[1]; v[2] = (float)p[2];
    }
}
__device__ inline float diff_l2sq(const float a[3], const float b[3])
{
    const float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
    return d0 * d0 + d1 * d1 + d2 * d2;
}

// computeCosts without the label test (the host applies it when it asks for a seam): for the edge between (x-1, y) and (x, y)
// the colour term is the mean of the two cross differences, the gradient term 1 + |gx| of the four samples beside the edge.
__global__ __launch_bounds__(256) void k_dp_pair_costs(const GradImg *imgs, const PairCost *pairs, const int *first_block, int n_pairs, int grad)
{
    int z = 0;
    while (z + 1 < n_pairs && (int)blockIdx.x >= first_block[z + 1]) ++z;
    const PairCost p = pairs[z];
    const int t = ((int)blockIdx.x - first_block[z]) * 256 + threadIdx.x;
    if (t >= p.iw * p.ih) return;
    const int y = t / p.iw, x = t - y * p.iw;
    const GradImg A = imgs[p.a], B = imgs[p.b];
    const int ya = p.ay + y, xa = p.ax + x, yb = p.by + y, xb = p.bx + x;
    float a11[3], b11[3];
    px3(A, ya, xa, a11);
    px3(B, yb, xb, b11);
    if (x > 0) {
        float a10[3], b10[3];
        px3(A, ya, xa - 1, a10);
        px3(B, yb, xb - 1, b10);
        float c = (diff_l2sq(a10, b11) + diff_l2sq(a11, b10)) / 2;
        if (grad) {
            const float gsum = fabsf(A.gx[(size_t)ya * A.w + xa]) + fabsf(A.gx[(size_t)ya * A.w + xa - 1]) + fabsf(B.gx[(size_t)yb * B.w + xb]) +
                               fabsf(B.gx[(size_t)yb * B.w + xb - 1]) + 1.f;
            c = c / gsum;
        }
        p.cv[t] = c;
    }
    if (y > 0) {
        float a01[3], b01[3];
        px3(A, ya - 1, xa, a01);
        px3(B, yb - 1, xb, b01);
        float c = (diff_l2sq(a01, b11) + diff_l2sq(a11, b01)) / 2;
        if (grad) {
            const float gsum = fabsf(A.gy[(size_t)ya * A.w + xa]) + fabsf(A.gy[(size_t)(ya - 1) * A.w + xa]) + fabsf(B.gy[(size_t)yb * B.w + xb]) +
                               fabsf(B.gy[(size_t)(yb - 1) * B.w + xb]) + 1.f;
            c = c / gsum;
        }
        p.ch[t] = c;
    }
}

// estimateSeam's dynamic programme over the rw x rh bounding box of one intersection component.
//   inl    rw x rh bytes: 1 where the pixel belongs to the component
//   cv/ch  the pair's cost planes (pitch iw), (ox, oy) = position of the box inside the overlap rectangle
// The sweep direction has `len` cells across; one work-group, lanes over the cells of a line, a barrier per line.
struct SeamArgs {
    const uint8_t *inl; int rw, rh;
    const float *cv, *ch; int iw, ox, oy;
    int horizontal, sx, sy, dx, dy;     // source / destination cell, box coordinates
    uint8_t *control;                   // rw x rh scratch
    int *out;                           // out[0] = number of seam points (0: destination unreachable), then (x, y) pairs, box coordinates
};
#define DP_BAD 195075.f   // normL2(Point3f(255, 255, 255), Point3f(0, 0, 0)): OpenCV's normL2 is the squared norm

__device__ inline bool dp_in(const SeamArgs &a, int y, int x) { return x >= 0 && y >= 0 && x < a.rw && y < a.rh && a.inl[(size_t)y * a.rw + x]; }
__device__ inline float dp_cost_v(const SeamArgs &a, int y, int x)   // edge between (x-1, y) and (x, y); x may be rw
{
    return (dp_in(a, y, x) && dp_in(a, y, x - 1)) ? a.cv[(size_t)(a.oy + y) * a.iw + (a.ox + x)] : DP_BAD;
}
__device__ inline float dp_cost_h(const SeamArgs &a, int y, int x)   // edge between (x, y-1) and (x, y); y may be rh
{
    return (dp_in(a, y, x) && dp_in(a, y - 1, x)) ? a.ch[(size_t)(a.oy + y) * a.iw + (a.ox + x)] : DP_BAD;
}

#define DP_MAX_LINE 4096
// WIDE: lines of more than DP_MAX_LINE cells (an overlap of more than 4096 x 4096 pixels: full-size 8K frames) keep their two cost / reach lines in
// global memory behind the control map instead of LDS -- the work-group's waves share a CU and its L1, __syncthreads orders their accesses
template <bool WIDE>
__global__ __launch_bounds__(1024) void k_dp_seam(const SeamArgs *reqs)      // one work-group per request
{
    const SeamArgs a = reqs[blockIdx.x];
    __shared__ float l_cost[WIDE ? 1 : 2][WIDE ? 1 : DP_MAX_LINE];
    __shared__ uint8_t l_reach[WIDE ? 1 : 2][WIDE ? 1 : DP_MAX_LINE];
    const int len = a.horizontal ? a.rh : a.rw;        // cells of one line
    float *s_cost[2];
    uint8_t *s_reach[2];
    if (WIDE) {
        float *g = (float *)(a.control + (((size_t)a.rw * a.rh + 4 + 15) / 16) * 16);
        s_cost[0] = g; s_cost[1] = g + len;
        s_reach[0] = (uint8_t *)(g + 2 * (size_t)len); s_reach[1] = s_reach[0] + len;
    } else {
        s_cost[0] = l_cost[0]; s_cost[1] = l_cost[WIDE ? 0 : 1];
        s_reach[0] = l_reach[0]; s_reach[1] = l_reach[WIDE ? 0 : 1];
    }
    const int from = a.horizontal ? a.sx : a.sy, to = a.horizontal ? a.dx : a.dy;
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
        const bool src = i == (a.horizontal ? a.sy : a.sx);
        s_cost[0][i] = 0.f;
        s_reach[0][i] = src ? 1 : 0;
    }
    __syncthreads();
    int cur = 0;
    for (int line = from + 1; line <= to; ++line) {
        const int nxt = cur ^ 1;
        for (int i = threadIdx.x; i < len; i += blockDim.x) {
            const int x = a.horizontal ? line : i, y = a.horizontal ? i : line;
            int n = 0, code = 0;
            float best = 0.f;
            if (dp_in(a, y, x)) {
                // std::min_element over (cost, step) pairs: smaller cost, the earlier step on equal costs
                if (a.horizontal) {
                    if (s_reach[cur][i]) { best = s_cost[cur][i] + dp_cost_h(a, y, x - 1); code = 1; n = 1; }
                    if (i > 0 && s_reach[cur][i - 1]) {
                        const float c = s_cost[cur][i - 1] + dp_cost_h(a, y - 1, x - 1) + dp_cost_v(a, y - 1, x);
                        if (!n || c < best) { best = c; code = 2; }
                        n = 1;
                    }
                    if (i < len - 1 && s_reach[cur][i + 1]) {
                        const float c = s_cost[cur][i + 1] + dp_cost_h(a, y + 1, x - 1) + dp_cost_v(a, y, x);
                        if (!n || c < best) { best = c; code = 3; }
                        n = 1;
                    }
                } else {
                    if (s_reach[cur][i]) { best = s_cost[cur][i] + dp_cost_v(a, y - 1, x); code = 1; n = 1; }
                    if (i > 0 && s_reach[cur][i - 1]) {
                        const float c = s_cost[cur][i - 1] + dp_cost_v(a, y - 1, x - 1) + dp_cost_h(a, y, x - 1);
                        if (!n || c < best) { best = c; code = 2; }
                        n = 1;
                    }
                    if (i < len - 1 && s_reach[cur][i + 1]) {
                        const float c = s_cost[cur][i + 1] + dp_cost_v(a, y - 1, x + 1) + dp_cost_h(a, y, x);
                        if (!n || c < best) { best = c; code = 3; }
                        n = 1;
                    }
                }
            }
            s_cost[nxt][i] = best;
            s_reach[nxt][i] = n ? 255 : 0;
            a.control[(size_t)y * a.rw + x] = (uint8_t)code;
        }
        __syncthreads();
        cur = nxt;
    }
    if (threadIdx.x != 0) return;
    // the destination line is `cur` when the sweep ran at least one line; a sweep of zero lines means source == destination line
    const int di = a.horizontal ? a.dy : a.dx;
    if (!s_reach[cur][di]) { a.out[0] = 0; return; }
    int x = a.dx, y = a.dy, k = 0;
    a.out[1] = x; a.out[2] = y; k = 1;
    if (a.horizontal) {
        while (x != a.sx) {
            const int c = a.control[(size_t)y * a.rw + x];
            if (c == 2) --y; else if (c == 3) ++y;
            --x;
            a.out[1 + 2 * k] = x; a.out[2 + 2 * k] = y; ++k;
        }
    } else {
        while (y != a.sy) {
            const int c = a.control[(size_t)y * a.rw + x];
            if (c == 2) --x; else if (c == 3) ++x;
            --y;
            a.out[1 + 2 * k] = x; a.out[2 + 2 * k] = y; ++k;
        }
    }
    a.out[0] = k;
}

// ---- the same sweep for boxes of up to DPL_CELLS cells and lines of up to 1024 cells (every seam of the recorded runs) -------------------------
// k_dp_seam spends its time waiting: per line two dependent round trips to global memory (the component mask, then -- where it is set -- the
// cost samples; nothing in them depends on the dynamic programme), and afterwards ONE lane walks the control map back through ~250 dependent
// global loads (~175 us of a ~430 us seam).  Here the component mask is brought into LDS once (one bit per cell), the cost samples of line
// n+1 are loaded while line n is processed (one cell per lane, registers), and the control map lives in LDS as 2-bit codes (ds_or), so the
// walk back reads LDS.  Same candidates, same order, same float sums: same seam.
#define DPL_LINE 1024
#define DPL_CELLS 131072          // 32 KB of 2-bit codes + 16 KB of mask bits
struct DpCell { float h0, h1, v1, h2, v2; int in; };     // the cell's own flag and the cost samples of its three candidate steps
struct DpLds { const uint32_t *inl; int rw, rh; };
__device__ inline bool dpl_in(const DpLds &m, int y, int x)
{
    if (x < 0 || y < 0 || x >= m.rw || y >= m.rh) return false;
    const uint32_t cell = (uint32_t)y * (uint32_t)m.rw + (uint32_t)x;
    return (m.inl[cell >> 5] >> (cell & 31u)) & 1u;
}
__device__ inline float dpl_cost_v(const SeamArgs &a, const DpLds &m, int y, int x) { return (dpl_in(m, y, x) && dpl_in(m, y, x - 1)) ? a.cv[(size_t)(a.oy + y) * a.iw + (a.ox + x)] : DP_BAD; }
__device__ inline float dpl_cost_h(const SeamArgs &a, const DpLds &m, int y, int x) { return (dpl_in(m, y, x) && dpl_in(m, y - 1, x)) ? a.ch[(size_t)(a.oy + y) * a.iw + (a.ox + x)] : DP_BAD; }
__device__ inline DpCell dp_fetch(const SeamArgs &a, const DpLds &m, int line, int i, int len)
{
    DpCell c = {0.f, 0.f, 0.f, 0.f, 0.f, 0};
    if (i >= len) return c;
    const int x = a.horizontal ? line : i, y = a.horizontal ? i : line;
    c.in = dpl_in(m, y, x) ? 1 : 0;
    if (!c.in) return c;
    if (a.horizontal) {
        c.h0 = dpl_cost_h(a, m, y, x - 1);
        c.h1 = dpl_cost_h(a, m, y - 1, x - 1); c.v1 = dpl_cost_v(a, m, y - 1, x);
        c.h2 = dpl_cost_h(a, m, y + 1, x - 1); c.v2 = dpl_cost_v(a, m, y, x);
    } else {
        c.h0 = dpl_cost_v(a, m, y - 1, x);
        c.h1 = dpl_cost_v(a, m, y - 1, x - 1); c.v1 = dpl_cost_h(a, m, y, x - 1);
        c.h2 = dpl_cost_v(a, m, y - 1, x + 1); c.v2 = dpl_cost_h(a, m, y, x);
    }
    return c;
}
__global__ __launch_bounds__(1024) void k_dp_seam_lds(const SeamArgs *reqs)      // one work-group per request
{
    const SeamArgs a = reqs[blockIdx.x];
    __shared__ float s_cost[2][DPL_LINE];
    __shared__ uint8_t s_reach[2][DPL_LINE];
    __shared__ uint32_t s_ctl[DPL_CELLS / 16];
    __shared__ uint32_t s_inl[DPL_CELLS / 32];
    const int len = a.horizontal ? a.rh : a.rw;
    const int from = a.horizontal ? a.sx : a.sy, to = a.horizontal ? a.dx : a.dy;
    const int tid = threadIdx.x, cells = a.rw * a.rh;
    if (tid < len) {
        s_cost[0][tid] = 0.f;
        s_reach[0][tid] = tid == (a.horizontal ? a.sy : a.sx) ? 1 : 0;
    }
    for (int i = tid; i < (cells + 15) / 16; i += (int)blockDim.x) s_ctl[i] = 0u;
    // the component mask arrives bit-packed from the host (one word per 32 cells)
    for (int wd = tid; wd < (cells + 31) / 32; wd += (int)blockDim.x) s_inl[wd] = ((const uint32_t *)a.inl)[wd];
    __syncthreads();
    const DpLds m = {s_inl, a.rw, a.rh};
    // DPL_AHEAD lines of samples in flight: a line of the sweep is a few hundred ns (LDS + one barrier), a round trip to global memory from a
    // single work-group on an otherwise idle GPU 1-2 us
    constexpr int DPL_AHEAD = 8;
    const DpCell none = {0.f, 0.f, 0.f, 0.f, 0.f, 0};
    DpCell nx[DPL_AHEAD];
#pragma unroll
    for (int k = 0; k < DPL_AHEAD; ++k) nx[k] = from + 1 + k <= to ? dp_fetch(a, m, from + 1 + k, tid, len) : none;
    int cur = 0;
    for (int line = from + 1; line <= to; ++line) {
        const int nxt = cur ^ 1;
        const DpCell c = nx[0];
#pragma unroll
        for (int k = 0; k + 1 < DPL_AHEAD; ++k) nx[k] = nx[k + 1];
        nx[DPL_AHEAD - 1] = line + DPL_AHEAD <= to ? dp_fetch(a, m, line + DPL_AHEAD, tid, len) : none;
        const int i = tid;
        if (i < len) {
            int n = 0, code = 0;
            float best = 0.f;
            if (c.in) {
                // std::min_element over (cost, step) pairs: smaller cost, the earlier step on equal costs
                if (s_reach[cur][i]) { best = s_cost[cur][i] + c.h0; code = 1; n = 1; }
                if (i > 0 && s_reach[cur][i - 1]) {
                    const float v = s_cost[cur][i - 1] + c.h1 + c.v1;
                    if (!n || v < best) { best = v; code = 2; }
                    n = 1;
                }
                if (i < len - 1 && s_reach[cur][i + 1]) {
                    const float v = s_cost[cur][i + 1] + c.h2 + c.v2;
                    if (!n || v < best) { best = v; code = 3; }
                    n = 1;
                }
            }
            s_cost[nxt][i] = best;
            s_reach[nxt][i] = n ? 255 : 0;
            if (code) {
                const int x = a.horizontal ? line : i, y = a.horizontal ? i : line;
                const uint32_t cell = (uint32_t)y * (uint32_t)a.rw + (uint32_t)x;
                atomicOr(&s_ctl[cell >> 4], (uint32_t)code << (2 * (cell & 15u)));
            }
        }
        __syncthreads();
        cur = nxt;
    }
    if (tid != 0) return;
    const int di = a.horizontal ? a.dy : a.dx;
    if (!s_reach[cur][di]) { a.out[0] = 0; return; }
    int x = a.dx, y = a.dy, k = 0;
    a.out[1] = x; a.out[2] = y; k = 1;
    auto ctl = [&](int yy, int xx) { const uint32_t cell = (uint32_t)yy * (uint32_t)a.rw + (uint32_t)xx; return (int)((s_ctl[cell >> 4] >> (2 * (cell & 15u))) & 3u); };
    if (a.horizontal) {
        while (x != a.sx) {
            const int c = ctl(y, x);
            if (c == 2) --y; else if (c == 3) ++y;
            --x;
            a.out[1 + 2 * k] = x; a.out[2 + 2 * k] = y; ++k;
        }
    } else {
        while (y != a.sy) {
            const int c = ctl(y, x);
            if (c == 2) --x; else if (c == 3) ++x;
            --y;
            a.out[1 + 2 * k] = x; a.out[2 + 2 * k] = y; ++k;
        }
    }
    a.out[0] = k;
}

// ---- host side: one pair ------------------------------------------------------------------------------------------------------------
// SSP_SEAM_DP_TIMING: where the host time of the pairs goes (summed over the worker threads)
enum { T_CANVAS, T_COMPONENTS, T_TIPS, T_REQUEST, T_RELABEL, T_REFRESH, T_CUT, T_EDGES, T_COUNT };
static std::atomic<long long> g_tns[T_COUNT];
static const bool g_timing = getenv("SSP_SEAM_DP_TIMING") != nullptr;
struct Tick {
    int k; std::chrono::steady_clock::time_point t0;
    explicit Tick(int k_) : k(k_) { if (g_timing) t0 = std::chrono::steady_clock::now(); }
    ~Tick() { if (g_timing) g_tns[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
enum { FIRST = 1, SECOND = 2, INTERS = 4 };
struct Pt { int x, y; };
struct Box { int x0, y0, x1, y1; };   // [x0, x1) x [y0, y1)

struct PairState {
    int uw = 0, uh = 0, utlx = 0, utly = 0;
    Box win{0, 0, 0, 0};              // the overlap rectangle of the two images grown by 2 (canvas coordinates): labels and outlines are kept there
    std::vector<uint8_t> c1, c2;      // outline pixels of the two masks (valid inside win)
    std::vector<int> labels;          // component labels (valid inside win)
    std::vector<int> states;
    std::vector<Box> box;
    std::vector<std::vector<Pt>> contours;
    std::set<std::pair<int, int>> edges;
    int lbl(int y, int x) const { return labels[(size_t)y * uw + x]; }
    bool border_of(int y, int x, int l) const
    {
        return x == 0 || lbl(y, x - 1) != l || x == uw - 1 || lbl(y, x + 1) != l || y == 0 || lbl(y - 1, x) != l || y == uh - 1 || lbl(y + 1, x) != l;
    }
    bool touches(int y, int x, int l) const
    {
        return (x > 0 && lbl(y, x - 1) == l) || (y > 0 && lbl(y - 1, x) == l) || (x < uw - 1 && lbl(y, x + 1) == l) || (y < uh - 1 && lbl(y + 1, x) == l);
    }
};

struct UnionFind {
    std::vector<int> p;
    int make() { p.push_back((int)p.size()); return (int)p.size() - 1; }
    int find(int a) { while (p[a] != a) { p[a] = p[p[a]]; a = p[a]; } return a; }
    void unite(int a, int b) { a = find(a); b = find(b); if (a != b) p[std::max(a, b)] = std::min(a, b); }
};

// 4-connected components of equal classes, numbered 1.. in raster order of their first pixel that satisfies `seed` (the order in
// which cv::floodFill would be started there); regions without a seed pixel stay 0.  cls: one byte per pixel, 255 = belongs to no
// region.  Scan-line fill: a seed's whole run is labelled at once and the runs above / below it are queued -- every pixel is
// visited a small constant number of times, whatever the shape of the region.
template <typename SeedAt>
static int label_components(int w, int h, const uint8_t *cls, SeedAt seed, std::vector<int> &out)
{
    out.assign((size_t)w * h, 0);
    static thread_local std::vector<int> stack;
    int count = 0;
    for (int y0 = 0; y0 < h; ++y0)
        for (int x0 = 0; x0 < w; ++x0) {
            const size_t o0 = (size_t)y0 * w + x0;
            const uint8_t c = cls[o0];
            if (c == 255 || out[o0] || !seed(y0, x0)) continue;
            const int id = ++count;
            stack.clear();
            stack.push_back((int)o0);
            while (!stack.empty()) {
                const int o = stack.back();
                stack.pop_back();
                if (out[o]) continue;
                const int y = o / w;
                const uint8_t *crow = cls + (size_t)y * w;
                int *orow = out.data() + (size_t)y * w;
                int xl = o - y * w, xr = xl;
                while (xl > 0 && crow[xl - 1] == c && !orow[xl - 1]) --xl;
                while (xr + 1 < w && crow[xr + 1] == c && !orow[xr + 1]) ++xr;
                for (int x = xl; x <= xr; ++x) orow[x] = id;
                for (int dy = -1; dy <= 1; dy += 2) {
                    const int yy = y + dy;
                    if (yy < 0 || yy >= h) continue;
                    const uint8_t *cr = cls + (size_t)yy * w;
                    const int *orr = out.data() + (size_t)yy * w;
                    bool in_run = false;
                    for (int x = xl; x <= xr; ++x) {
                        const bool ok = cr[x] == c && !orr[x];
                        if (ok && !in_run) stack.push_back(yy * w + x);   // one seed per stretch
                        in_run = ok;
                    }
                }
            }
        }
    return count;
}

struct PairJob {
    int a, b;
    int tl1x, tl1y, w1, h1, tl2x, tl2y, w2, h2;
    int ix0, iy0, iw, ih;        // overlap rectangle (pano coordinates) -- the device cost planes cover it
    const float *cv, *ch;        // device
};

// findComponents + findEdges on runs.  The canvases of a pair are the two masks: a few runs of equal class (first only / second only / both)
// per row.  Components, their numbering (raster order of the first pixel, the order cv::floodFill is started in), boxes, outlines (in
// raster order, as the pixel scan lists them) and the adjacency set all follow from the runs and their overlaps with the rows above and
// below -- O(runs + outline pixels) instead of three passes over every pixel of the union canvas (74 + 20 ms of the recorded 21-frame run's
// 180 ms, tools/bench_seam_dp.py).  The dense label image that the later steps index is filled run by run.
struct Run { int xl, xr; int set; uint8_t c; };

// the maximal stretches of non-zero bytes of a mask row, as [first, last] column pairs shifted by `off` (16 bytes per step)
static inline void mask_runs(const uint8_t *row, int w, int off, std::vector<std::pair<int, int>> &out)
{
    out.clear();
    const __m128i zero = _mm_setzero_si128();
    int x = 0;
    while (x < w) {
        // next set byte
        for (;;) {
            if (x + 16 <= w) {
                const unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(row + x)), zero)) ^ 0xffffu;   // bits of the set bytes
                if (m) { x += __builtin_ctz(m); break; }
                x += 16;
            } else {
                while (x < w && !row[x]) ++x;
                break;
            }
        }
        if (x >= w) break;
        const int first = x;
        // next clear byte
        for (;;) {
            if (x + 16 <= w) {
                const unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(row + x)), zero));              // bits of the clear bytes
                if (m) { x += __builtin_ctz(m); break; }
                x += 16;
            } else {
                while (x < w && row[x]) ++x;
                break;
            }
        }
        out.push_back({first + off, x - 1 + off});
    }
}

static void find_components_and_edges(PairState &s, const PairJob &job, const std::vector<uint8_t> &mask1, const std::vector<uint8_t> &mask2)
{
    const int W = s.uw, H = s.uh;
    static thread_local std::vector<Run> runs;
    static thread_local std::vector<int> row_start;
    static thread_local std::vector<std::pair<int, int>> ra, rb;
    static thread_local std::vector<std::pair<int, int>> vadj;
    vadj.clear();
    runs.clear();
    row_start.assign((size_t)H + 1, 0);
    const int ox1 = job.tl1x - s.utlx, oy1 = job.tl1y - s.utly, ox2 = job.tl2x - s.utlx, oy2 = job.tl2y - s.utly;
    UnionFind uf;
    for (int y = 0; y < H; ++y) {
        row_start[y] = (int)runs.size();
        // the row's runs of equal class (1 = first mask only, 2 = second only, 3 = both): each mask's stretches of set bytes (a mask is zero
        // outside its image's rectangle), merged at their end points -- the masks are read once, 16 bytes at a time, and nothing else is
        ra.clear(); rb.clear();
        if (y >= oy1 && y < oy1 + job.h1) mask_runs(&mask1[(size_t)(y - oy1) * job.w1], job.w1, ox1, ra);
        if (y >= oy2 && y < oy2 + job.h2) mask_runs(&mask2[(size_t)(y - oy2) * job.w2], job.w2, ox2, rb);
        {
            size_t i = 0, j = 0;
            int x = 0;                                  // everything left of x is done
            while (i < ra.size() || j < rb.size()) {
                // skip stretches that ended
                if (i < ra.size() && ra[i].second < x) { ++i; continue; }
                if (j < rb.size() && rb[j].second < x) { ++j; continue; }
                const bool ha = i < ra.size(), hb = j < rb.size();
                const int la = ha ? std::max(ra[i].first, x) : INT_MAX, lb = hb ? std::max(rb[j].first, x) : INT_MAX;
                const int start = std::min(la, lb);
                const bool ina = ha && ra[i].first <= start, inb = hb && rb[j].first <= start;
                // the class holds until the nearest end of a covering stretch or start of the other one
                int end = INT_MAX;
                if (ina) end = std::min(end, ra[i].second); else if (ha) end = std::min(end, ra[i].first - 1);
                if (inb) end = std::min(end, rb[j].second); else if (hb) end = std::min(end, rb[j].first - 1);
                runs.push_back(Run{start, end, uf.make(), (uint8_t)((ina ? 1 : 0) | (inb ? 2 : 0))});
                x = end + 1;
            }
        }
        // 4-connectivity with the row above: same class, overlapping columns
        if (y > 0) {
            int i = row_start[y - 1], j = row_start[y];
            const int ie = row_start[y], je = (int)runs.size();
            while (i < ie && j < je) {
                if (runs[i].xl <= runs[j].xr && runs[j].xl <= runs[i].xr) {
                    if (runs[i].c == runs[j].c) uf.unite(runs[i].set, runs[j].set);
                    else if (vadj.empty() || vadj.back().first != runs[i].set || vadj.back().second != runs[j].set) vadj.push_back({runs[i].set, runs[j].set});   // vertical neighbours of another class
                }
                if (runs[i].xr < runs[j].xr) ++i; else ++j;
            }
        }
    }
    row_start[H] = (int)runs.size();
    // numbering: a component's first pixel in raster order is the start of its first run
    static thread_local std::vector<int> id_of;
    id_of.assign(uf.p.size(), 0);
    int n = 0;
    for (Run &r : runs) {
        const int root = uf.find(r.set);
        if (!id_of[root]) id_of[root] = ++n;
        r.set = id_of[root];         // from here on: the label
    }
    // The dense label image is only ever indexed within one pixel of an intersection component (outline tests, the relabelling and rescans of
    // its box, the final cut of the overlap rectangle), and an intersection component lies inside the overlap rectangle: the labels are filled
    // inside `win` (the overlap grown by 2), the rest of the array is never read.  The same holds for the outlines (contours) of the image-only
    // components: nothing reads them, they are not collected.
    static const bool poison = getenv("SSP_SEAM_DP_POISON") != nullptr;      // tests: whatever lies outside the window would change the result if read
    if (poison) s.labels.assign((size_t)W * H, 0x3fffffff); else s.labels.resize((size_t)W * H);
    const Box win = s.win;
    for (int y = win.y0; y < win.y1; ++y) std::fill(&s.labels[(size_t)y * W + win.x0], &s.labels[(size_t)y * W + win.x1], 0);
    s.states.assign(n, 0);
    s.box.assign(n, Box{INT_MAX, INT_MAX, INT_MIN, INT_MIN});
    if ((int)s.contours.size() < n) s.contours.resize(n);
    for (int k = 0; k < n; ++k) s.contours[k].clear();
    s.edges.clear();
    // adjacencies are met once per pair of touching runs, i.e. thousands of times for a handful of distinct edges: a bit matrix in front of the set
    static thread_local std::vector<uint8_t> met;
    const bool matrix = n <= 1024;
    if (matrix) met.assign((size_t)n * n, 0);
    int last_a = 0, last_b = 0;
    auto meet = [&](int a, int b) {
        if ((a == last_a && b == last_b) || (a == last_b && b == last_a)) return;
        last_a = a; last_b = b;
        if (matrix) {
            uint8_t &m = met[(size_t)(a - 1) * n + (b - 1)];
            if (m) return;
            m = 1; met[(size_t)(b - 1) * n + (a - 1)] = 1;
        }
        s.edges.insert({a - 1, b - 1}); s.edges.insert({b - 1, a - 1});
    };
    static thread_local std::vector<std::pair<int, int>> up, dn, in;      // same-label cover of a run by the row above / below, and their common interior part
    auto cover = [&](int row, const Run &r, std::vector<std::pair<int, int>> &out, int &cursor) {
        out.clear();
        if (row < 0 || row >= H) return;
        const int e = row_start[row + 1];
        int k = std::max(cursor, row_start[row]);
        while (k < e && runs[k].xr < r.xl) ++k;
        cursor = k;                                     // runs of a row are visited left to right: the cursor only moves forward
        for (; k < e && runs[k].xl <= r.xr; ++k)
            if (runs[k].c == r.c) out.push_back({std::max(runs[k].xl, r.xl), std::min(runs[k].xr, r.xr)});
    };
    for (const auto &v : vadj) meet(id_of[uf.find(v.first)], id_of[uf.find(v.second)]);      // (found while the rows were joined; the sets are labels only now)
    for (int y = 0; y < H; ++y) {
        int cu = 0, cd = 0;
        const bool yin = y >= win.y0 && y < win.y1;
        for (int k = row_start[y]; k < row_start[y + 1]; ++k) {
            const Run &r = runs[k];
            const int l = r.set;
            if (yin) {
                const int fx0 = std::max(r.xl, win.x0), fx1 = std::min(r.xr + 1, win.x1);
                if (fx0 < fx1) std::fill(&s.labels[(size_t)y * W + fx0], &s.labels[(size_t)y * W + fx1], l);
            }
            s.states[l - 1] = r.c == 3 ? INTERS : r.c == 1 ? FIRST : SECOND;
            Box &b = s.box[l - 1];
            b.x0 = std::min(b.x0, r.xl); b.y0 = std::min(b.y0, y); b.x1 = std::max(b.x1, r.xr + 1); b.y1 = std::max(b.y1, y + 1);
            if (k + 1 < row_start[y + 1] && runs[k + 1].xl == r.xr + 1) meet(l, runs[k + 1].set);
            if (r.c != 3) continue;
            cover(y - 1, r, up, cu);
            cover(y + 1, r, dn, cd);
            // outline pixels: the run's ends, and every pixel whose upper or lower neighbour is not the same component
            std::vector<Pt> &ct = s.contours[l - 1];
            in.clear();
            for (size_t i = 0, j = 0; i < up.size() && j < dn.size();) {
                const int lo = std::max(std::max(up[i].first, dn[j].first), r.xl + 1), hi = std::min(std::min(up[i].second, dn[j].second), r.xr - 1);
                if (lo <= hi) in.push_back({lo, hi});
                if (up[i].second < dn[j].second) ++i; else ++j;
            }
            int x = r.xl;
            for (const auto &iv : in) {
                for (; x < iv.first; ++x) ct.push_back(Pt{x, y});
                x = iv.second + 1;
            }
            for (; x <= r.xr; ++x) ct.push_back(Pt{x, y});
        }
    }
}

static bool near_contour(const PairState &s, int y, int x, const std::vector<uint8_t> &cm)
{
    for (int yy = std::max(0, y - 2); yy <= std::min(s.uh - 1, y + 2); ++yy)
        for (int xx = std::max(0, x - 2); xx <= std::min(s.uw - 1, x + 2); ++xx)
            if (cm[(size_t)yy * s.uw + xx]) return true;
    return false;
}

// getSeamTips: contour points of the intersection component that lie next to comp2 and near both images' outlines, clustered
// (cv::partition with |p - q| < 10); the two clusters whose rounded centres are farthest apart give one tip each.
static bool seam_tips(const PairState &s, int comp1, int comp2, Pt &p1, Pt &p2)
{
    std::vector<Pt> sp;
    for (const Pt &p : s.contours[comp1])
        if (near_contour(s, p.y, p.x, s.c1) && near_contour(s, p.y, p.x, s.c2) && s.touches(p.y, p.x, comp2 + 1)) sp.push_back(p);
    if (sp.size() < 2) return false;
    const int N = (int)sp.size();
    UnionFind uf;
    for (int i = 0; i < N; ++i) uf.make();
    for (int i = 0; i < N; ++i)
        for (int j = i + 1; j < N; ++j) {
            const int dx = sp[i].x - sp[j].x, dy = sp[i].y - sp[j].y;
            if (dx * dx + dy * dy < 100) uf.unite(i, j);
        }
    std::vector<int> cls(N), cls_of_root(N, -1);
    int nlabels = 0;
    for (int i = 0; i < N; ++i) {
        const int r = uf.find(i);
        if (cls_of_root[r] < 0) cls_of_root[r] = nlabels++;
        cls[i] = cls_of_root[r];
    }
    if (nlabels < 2) return false;
    std::vector<long long> sx(nlabels, 0), sy(nlabels, 0);
    std::vector<int> cnt(nlabels, 0);
    for (int i = 0; i < N; ++i) { sx[cls[i]] += sp[i].x; sy[cls[i]] += sp[i].y; ++cnt[cls[i]]; }
    auto centre = [&](int c, double &cx, double &cy) { cx = std::nearbyint(sx[c] / (double)cnt[c]); cy = std::nearbyint(sy[c] / (double)cnt[c]); };   // cvRound
    int idx[2] = {-1, -1};
    double best = -std::numeric_limits<double>::max();
    for (int i = 0; i < nlabels - 1; ++i)
        for (int j = i + 1; j < nlabels; ++j) {
            double ax, ay, bx, by;
            centre(i, ax, ay); centre(j, bx, by);
            const double d = (ax - bx) * (ax - bx) + (ay - by) * (ay - by);
            if (d > best) { best = d; idx[0] = i; idx[1] = j; }
        }
    Pt tip[2] = {{0, 0}, {0, 0}};
    for (int k = 0; k < 2; ++k) {
        double cx, cy, md = std::numeric_limits<double>::max();
        centre(idx[k], cx, cy);
        for (int i = 0; i < N; ++i) {
            if (cls[i] != idx[k]) continue;
            const double d = (sp[i].x - cx) * (sp[i].x - cx) + (sp[i].y - cy) * (sp[i].y - cy);
            if (d < md) { md = d; tip[k] = sp[i]; }
        }
    }
    p1 = tip[0]; p2 = tip[1];
    return true;
}

// estimateSeam, first half: the request for the device sweep of component `comp` from tip p1 to tip p2 (the sweeps of all pairs that are in
// flight together are ONE launch: run_round).  -> 1 request made, < 0 error
struct SweepReq {
    int comp = -1, rw = 0, rh = 0;
    Pt p1{0, 0}, p2{0, 0}, src{0, 0}, dst{0, 0};
    bool horizontal = false, swapped = false, lds_form = false, wide = false;
    size_t pts = 0, inl_bytes = 0;
    std::vector<uint8_t> inl;      // the component's mask: bits (lds form) or bytes
    SeamArgs args;                 // pointers filled in by the round
};
static int seam_request(const PairState &s, const PairJob &job, int comp, Pt p1, Pt p2, SweepReq &r)
{
    const Box &b = s.box[comp];
    const int rw = b.x1 - b.x0, rh = b.y1 - b.y0, l = comp + 1;
    r.comp = comp; r.rw = rw; r.rh = rh; r.p1 = p1; r.p2 = p2;
    r.src = Pt{p1.x - b.x0, p1.y - b.y0}; r.dst = Pt{p2.x - b.x0, p2.y - b.y0};
    r.swapped = false;
    r.horizontal = std::abs(r.dst.x - r.src.x) > std::abs(r.dst.y - r.src.y);
    if (r.horizontal ? r.src.x > r.dst.x : r.src.y > r.dst.y) { std::swap(r.src, r.dst); r.swapped = true; }
    r.wide = (r.horizontal ? rh : rw) > DP_MAX_LINE;      // k_dp_seam<true>: the two lines in global memory
    r.lds_form = (r.horizontal ? rh : rw) <= DPL_LINE && (size_t)rw * rh <= DPL_CELLS;      // k_dp_seam_lds: the mask travels as bits
    r.pts = (size_t)(r.horizontal ? rw : rh) + 1;
    if (r.lds_form) {
        r.inl_bytes = (((size_t)rw * rh + 31) / 32) * 4;
        r.inl.assign(r.inl_bytes, 0);
        uint32_t *w = (uint32_t *)r.inl.data();
        size_t cell = 0;
        for (int y = 0; y < rh; ++y) {
            const int *row = &s.labels[(size_t)(y + b.y0) * s.uw + b.x0];
            for (int x = 0; x < rw; ++x, ++cell)
                if (row[x] == l) w[cell >> 5] |= 1u << (cell & 31);
        }
    } else {
        r.inl_bytes = (size_t)rw * rh;
        r.inl.resize(r.inl_bytes);
        for (int y = 0; y < rh; ++y)
            for (int x = 0; x < rw; ++x) r.inl[(size_t)y * rw + x] = s.lbl(y + b.y0, x + b.x0) == l;
    }
    SeamArgs &a = r.args;
    a.inl = nullptr; a.rw = rw; a.rh = rh;
    a.cv = job.cv; a.ch = job.ch; a.iw = job.iw;
    a.ox = b.x0 + s.utlx - job.ix0; a.oy = b.y0 + s.utly - job.iy0;   // the component lies inside the overlap rectangle
    a.horizontal = r.horizontal; a.sx = r.src.x; a.sy = r.src.y; a.dx = r.dst.x; a.dy = r.dst.y;
    a.control = nullptr; a.out = nullptr;
    if (a.ox < 0 || a.oy < 0 || a.ox + rw > job.iw || a.oy + rh > job.ih) return set_error(SSP_ERR_STATE, "DpSeamFinder: an intersection component leaves the overlap rectangle");
    return 1;
}
// ... second half: the seam from the sweep's output (h_out[0] = number of points, destination first).  -> 1 seam, 0 destination unreachable, < 0 error
static int seam_result(const PairState &s, const SweepReq &r, const int *h_out, std::vector<Pt> &seam)
{
    const Box &b = s.box[r.comp];
    const int k = h_out[0];
    if (k <= 0) return 0;
    if ((size_t)k > r.pts) return set_error(SSP_ERR_STATE, "DpSeamFinder: the sweep returned %d seam points for a box that holds %zu", k, r.pts);
    seam.clear();
    for (int i = 0; i < k; ++i) seam.push_back(Pt{h_out[1 + 2 * i] + b.x0, h_out[2 + 2 * i] + b.y0});   // destination first
    if (!r.swapped) std::reverse(seam.begin(), seam.end());
    if (seam.front().x != r.p1.x || seam.front().y != r.p1.y || seam.back().x != r.p2.x || seam.back().y != r.p2.y)
        return set_error(SSP_ERR_STATE, "DpSeamFinder: the restored seam does not join its tips");
    return 1;
}

// updateLabelsUsingSeam: the seam and the component's outline split its box into parts; parts that border comp2 along more than
// 5 % of the outline and other components along less than 10 % go over to comp2.
static void relabel_along_seam(PairState &s, int comp1, int comp2, const std::vector<Pt> &seam, bool horizontal)
{
    const Box b = s.box[comp1];
    const int mw = b.x1 - b.x0, mh = b.y1 - b.y0, l1 = comp1 + 1, l2 = comp2 + 1;
    static thread_local std::vector<uint8_t> open;      // (scratch of this thread: a relabelling is a tenth of a millisecond, the allocations showed)
    static thread_local std::vector<int> part;
    // the walls (outline and seam pixels) are 255, everything else one open class
    open.assign((size_t)mw * mh, 0);
    const std::vector<Pt> &ct = s.contours[comp1];
    for (const Pt &p : ct) open[(size_t)(p.y - b.y0) * mw + (p.x - b.x0)] = 255;
    for (const Pt &p : seam) open[(size_t)(p.y - b.y0) * mw + (p.x - b.x0)] = 255;
    const int nparts = label_components(mw, mh, open.data(), [&](int y, int x) { return s.lbl(y + b.y0, x + b.x0) == l1; }, part);
    // walls are 255 in OpenCV's mask; a 255th part would be mistaken for one (never reached at seam scale, kept for fidelity)
    const int WALL = 255;
    auto at = [&](int y, int x) -> int & { return part[(size_t)y * mw + x]; };
    for (const Pt &p : ct) at(p.y - b.y0, p.x - b.x0) = WALL;
    for (const Pt &p : seam) at(p.y - b.y0, p.x - b.x0) = WALL;
    for (const Pt &p : ct) {   // outline pixels join a neighbouring part (8-neighbourhood, the last hit wins), in outline order
        const int x = p.x - b.x0, y = p.y - b.y0;
        static const int dx[] = {-1, +1, 0, 0, -1, +1, -1, +1}, dy[] = {0, 0, -1, +1, -1, -1, +1, +1};
        bool ok = false;
        for (int j = 0; j < 8; ++j) {
            const int c = x + dx[j], r = y + dy[j];
            if (c >= 0 && c < mw && r >= 0 && r < mh && at(r, c) && at(r, c) != WALL) { ok = true; at(y, x) = at(r, c); }
        }
        if (!ok) at(y, x) = 0;
    }
    for (const Pt &p : seam) {  // the seam runs along the upper (left) side of its pixels: they belong to the part below (right)
        const int x = p.x - b.x0, y = p.y - b.y0;
        const bool has = horizontal ? (y < mh - 1 && at(y + 1, x) && at(y + 1, x) != WALL) : (x < mw - 1 && at(y, x + 1) && at(y, x + 1) != WALL);
        at(y, x) = has ? (horizontal ? at(y + 1, x) : at(y, x + 1)) : 0;
    }
    std::vector<int> to2(nparts + 1, 0), to_other(nparts + 1, 0);
    for (const Pt &p : ct) {
        const int m = at(p.y - b.y0, p.x - b.x0);
        if (m < 0 || m > nparts) continue;
        if (s.touches(p.y, p.x, l2)) ++to2[m];
        const int x = p.x, y = p.y;
        auto foreign = [&](int yy, int xx) { const int v = s.lbl(yy, xx); return v != l1 && v != l2; };
        if ((x > 0 && foreign(y, x - 1)) || (y > 0 && foreign(y - 1, x)) || (x < s.uw - 1 && foreign(y, x + 1)) || (y < s.uh - 1 && foreign(y + 1, x))) ++to_other[m];
    }
    const double len = (double)ct.size();
    std::vector<uint8_t> moves(nparts + 1, 0);
    bool any = false;
    for (int k = 1; k <= nparts; ++k) { moves[k] = to2[k] / len > 0.05 && to_other[k] / len < 0.1; any = any || moves[k]; }
    if (!any) return;
    for (int y = 0; y < mh; ++y)
        for (int x = 0; x < mw; ++x) {
            const int m = at(y, x);
            if (m > 0 && m <= nparts && moves[m]) s.labels[(size_t)(y + b.y0) * s.uw + (x + b.x0)] = l2;
        }
}

static void refresh(PairState &s, int c)
{
    const Box old = s.box[c];
    const int l = c + 1, W = s.uw, H = s.uh;
    Box nb{INT_MAX, INT_MAX, INT_MIN, INT_MIN};
    std::vector<Pt> &ct = s.contours[c];
    ct.clear();
    for (int y = old.y0; y < old.y1; ++y) {
        const int *row = &s.labels[(size_t)y * W], *up = y > 0 ? row - W : nullptr, *dn = y < H - 1 ? row + W : nullptr;
        int first = -1, last = -1;
        for (int x = old.x0; x < old.x1; ++x) {
            if (row[x] != l) continue;
            if (first < 0) first = x;
            last = x;
            const bool inner = x > 0 && x < W - 1 && up && dn && row[x - 1] == l && row[x + 1] == l && up[x] == l && dn[x] == l;
            if (!inner) ct.push_back(Pt{x, y});
        }
        if (first >= 0) { nb.x0 = std::min(nb.x0, first); nb.x1 = std::max(nb.x1, last + 1); nb.y0 = std::min(nb.y0, y); nb.y1 = std::max(nb.y1, y + 1); }
    }
    s.box[c] = nb;
}

// One pair of DpSeamFinder::process as a resumable run: begin (canvases, outlines, components, edges), then step -- cut after cut through the
// intersection components -- which returns whenever it needs a device sweep (estimateSeam), and resume with the sweep's output.  The pairs
// whose images are disjoint run side by side (run_rounds): their host work on worker threads, their sweeps in one launch.
struct PairRun {
    const PairJob *job = nullptr;
    std::vector<uint8_t> *mask1 = nullptr, *mask2 = nullptr;
    PairState s;
    SweepReq req;
    int c1 = -1, c2 = -1;
    bool need_sweep = false, done = false;
    int rc = 0;
    std::string err;            // (worker threads: the thread-local error text is copied here and re-raised by the caller's thread)
};

static void pair_begin(PairRun &run)
{
    const PairJob &job = *run.job;
    PairState &s = run.s;
    std::vector<uint8_t> &mask1 = *run.mask1, &mask2 = *run.mask2;

    s.utlx = std::min(job.tl1x, job.tl2x); s.utly = std::min(job.tl1y, job.tl2y);
    s.uw = std::max(job.tl1x + job.w1, job.tl2x + job.w2) - s.utlx;
    s.uh = std::max(job.tl1y + job.h1, job.tl2y + job.h2) - s.utly;
    const size_t un = (size_t)s.uw * s.uh;
    const int W = s.uw, H = s.uh;
    s.win = Box{std::max(0, job.ix0 - s.utlx - 2), std::max(0, job.iy0 - s.utly - 2), std::min(W, job.ix0 - s.utlx + job.iw + 2), std::min(H, job.iy0 - s.utly + job.ih + 2)};
    {
        Tick tk(T_CANVAS);
        // outline pixels of a mask: set, with an unset (or no) 4-neighbour on the union canvas; a mask is zero outside its image's rectangle, so a
        // set pixel on the rectangle's edge is one.  The outlines are only ever asked about within 2 pixels of an intersection component's outline
        // (seam_tips / near_contour), i.e. inside `win`: they are computed there from the masks themselves (no canvas copies of the masks), the
        // rest of the arrays is never read.
        static const bool poison = getenv("SSP_SEAM_DP_POISON") != nullptr;
        if (poison) { s.c1.assign(un, 255); s.c2.assign(un, 255); } else { s.c1.resize(un); s.c2.resize(un); }
        auto outline = [&](const std::vector<uint8_t> &m, std::vector<uint8_t> &c, int rx, int ry, int mw, int mh) {
            for (int y = s.win.y0; y < s.win.y1; ++y) {
                uint8_t *out = &c[(size_t)y * W];
                memset(out + s.win.x0, 0, (size_t)(s.win.x1 - s.win.x0));
                const int my = y - ry;
                if (my < 0 || my >= mh) continue;
                const uint8_t *row = &m[(size_t)my * mw];
                const int X0 = std::max(s.win.x0, rx), X1 = std::min(s.win.x1, rx + mw);
                if (my == 0 || my == mh - 1) {
                    for (int x = X0; x < X1; ++x) out[x] = row[x - rx] ? 255 : 0;
                    continue;
                }
                const int xa = std::max(X0, rx + 1), xb = std::min(X1, rx + mw - 1);
                for (int x = X0; x < std::min(xa, X1); ++x) out[x] = row[x - rx] ? 255 : 0;
                {
                    // (no branches: the compiler makes 16 pixels per step of it)
                    const uint8_t *q = row - rx;
                    for (int x = xa; x < xb; ++x) {
                        const unsigned set = q[x] != 0, all4 = (unsigned)(q[x - 1] != 0) & (unsigned)(q[x + 1] != 0) & (unsigned)(q[x - mw] != 0) & (unsigned)(q[x + mw] != 0);
                        out[x] = (uint8_t)(0u - (set & (all4 ^ 1u)));
                    }
                }
                for (int x = std::max(xb, X0); x < X1; ++x) out[x] = row[x - rx] ? 255 : 0;
            }
        };
        outline(mask1, s.c1, job.tl1x - s.utlx, job.tl1y - s.utly, job.w1, job.h1);
        outline(mask2, s.c2, job.tl2x - s.utlx, job.tl2y - s.utly, job.w2, job.h2);
    }
    Tick t2(T_COMPONENTS);
    find_components_and_edges(s, job, mask1, mask2);
}

static void pair_cut(PairRun &run)
{
    Tick tk(T_CUT);
    const PairJob &job = *run.job;
    PairState &s = run.s;
    std::vector<uint8_t> &mask1 = *run.mask1, &mask2 = *run.mask2;
    // cut the masks: a pixel of one mask is cleared where the component it lies in went to the OTHER image and that image's mask is set -- only
    // inside the overlap rectangle of the two images.  A component's state holds FIRST or SECOND, never both: one pass does both masks.
    const int cx0 = job.ix0 - s.utlx, cy0 = job.iy0 - s.utly;
    for (int y = 0; y < job.ih; ++y) {
        const int *lrow = &s.labels[(size_t)(cy0 + y) * s.uw + cx0];
        uint8_t *r1 = &mask1[(size_t)(job.iy0 - job.tl1y + y) * job.w1 + (job.ix0 - job.tl1x)];
        uint8_t *r2 = &mask2[(size_t)(job.iy0 - job.tl2y + y) * job.w2 + (job.ix0 - job.tl2x)];
        for (int x = 0; x < job.iw; ++x) {
            // (stores only where a pixel changes: a table-driven form that rewrites every pixel of both masks measured 2-3 x slower)
            const int l = lrow[x];
            if (l <= 0 || !r1[x] || !r2[x]) continue;
            if (s.states[l - 1] & FIRST) r2[x] = 0; else if (s.states[l - 1] & SECOND) r1[x] = 0;
        }
    }
}

// after a cut of component c1 against c2 (or the hand-over of a component with one neighbour)
static void pair_after_cut(PairRun &run)
{
    PairState &s = run.s;
    { Tick tk(T_REFRESH); refresh(s, run.c1); }
    // OpenCV also rescans c2 -- within c2's OLD box, which misses the pixels it just gained; nothing reads c2's box or outline
    // afterwards (c2 is an image-only component: it is never the one that gets cut), so the rescan is left out here
    s.edges.erase({run.c1, run.c2});
    s.edges.erase({run.c2, run.c1});
}

// advance until the pair needs a sweep (need_sweep) or is finished (done; the masks are cut)
static void pair_step(PairRun &run)
{
    PairState &s = run.s;
    run.need_sweep = false;
    for (;;) {
        // the first edge (lexicographic order of the set) whose intersection component meets a component of the other side
        int c1 = -1, c2 = -1;
        {
            Tick tk(T_EDGES);
            for (const auto &e : s.edges)
                if ((s.states[e.first] & INTERS) && (s.states[e.first] & ~INTERS) != s.states[e.second]) { c1 = e.first; c2 = e.second; break; }
        }
        if (c1 < 0) break;
        run.c1 = c1; run.c2 = c2;
        const auto lo = s.edges.lower_bound({c1, INT_MIN}), hi = s.edges.upper_bound({c1, INT_MAX});
        if (std::distance(lo, hi) == 1) {   // hasOnlyOneNeighbor: the whole component goes over
            const Box &b = s.box[c1];
            for (int y = b.y0; y < b.y1; ++y)
                for (int x = b.x0; x < b.x1; ++x)
                    if (s.lbl(y, x) == c1 + 1) s.labels[(size_t)y * s.uw + x] = c2 + 1;
            s.states[c1] = s.states[c2] == FIRST ? SECOND : FIRST;
        } else {
            Pt p1, p2;
            bool tips;
            { Tick tk(T_TIPS); tips = seam_tips(s, c1, c2, p1, p2); }
            if (tips) {
                Tick tk(T_REQUEST);
                const int ok = seam_request(s, *run.job, c1, p1, p2, run.req);
                if (ok < 0) { run.rc = ok; run.err = ssp_last_error(); run.done = true; return; }
                run.need_sweep = true;
                return;                      // pair_resume continues behind the sweep
            }
            s.states[c1] = s.states[c2] == FIRST ? (INTERS | SECOND) : (INTERS | FIRST);
        }
        pair_after_cut(run);
    }
    pair_cut(run);
    run.done = true;
}

static void pair_resume(PairRun &run, const int *h_out)
{
    PairState &s = run.s;
    std::vector<Pt> seam;
    const int ok = seam_result(s, run.req, h_out, seam);
    if (ok < 0) { run.rc = ok; run.err = ssp_last_error(); run.done = true; run.need_sweep = false; return; }
    if (ok) { Tick tk(T_RELABEL); relabel_along_seam(s, run.c1, run.c2, seam, run.req.horizontal); }
    s.states[run.c1] = s.states[run.c2] == FIRST ? (INTERS | SECOND) : (INTERS | FIRST);
    pair_after_cut(run);
    pair_step(run);
}

// (WorkerPool: ssp_internal.hpp -- the pairs of a round share no image and no state)
template <typename F>
static void parallel_for(int n, int threads, F fn) { WorkerPool::get().run(n, threads, fn); }

// growable device / pinned scratch of the rounds
// Pinned host staging that outlives a call: hipHostMalloc / hipHostFree cost a good part of a millisecond each, a seam finder run is 25 ms.
// One set per host thread (the library is re-entrant across threads); grown, never shrunk, and left to the process's end (a destructor would run
// after the HIP runtime's own at exit).
struct Pinned {
    char *p = nullptr;
    size_t cap = 0;
    int need(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        if (hipHostMalloc((void **)&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return set_error(SSP_ERR_MEMORY, "DpSeamFinder: pinned staging of %zu bytes failed", bytes); }
        cap = want;
        return 0;
    }
};

// growable device scratch of the rounds (from the pool, returned at the end of the call) + this thread's pinned staging
struct RoundBuffers {
    char *d_in = nullptr, *d_ctl = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
    size_t cap_in = 0, cap_ctl = 0, cap_out = 0;
    Pinned *pin_in = nullptr, *pin_out = nullptr;
    int grow(char **d, char **h, Pinned *pin, size_t *cap, size_t need)
    {
        if (h) { SSP_TRY(pin->need(need)); *h = pin->p; }
        if (need <= *cap) return 0;
        if (*d) pool_free(*d);
        *d = nullptr;
        *cap = need + need / 2 + 4096;
        SSP_TRY(pool_alloc(*cap, (void **)d));
        return 0;
    }
    void release()
    {
        if (d_in) pool_free(d_in);
        if (d_ctl) pool_free(d_ctl);
        if (d_out) pool_free(d_out);
        d_in = d_ctl = d_out = h_in = h_out = nullptr;
        cap_in = cap_ctl = cap_out = 0;
    }
};

// The sweeps the active pairs are waiting for: one upload (requests + component masks), one launch per kernel form (one work-group per
// request), one read-back, one synchronisation -- instead of that per seam (82 of them on the reference's recorded 21-frame run).
static int launch_sweeps(std::vector<PairRun *> &waiting, RoundBuffers &rb, std::vector<const int *> &outs, hipEvent_t done)
{
    const size_t R = waiting.size();
    outs.assign(R, nullptr);
    if (!R) return 0;
    // order: lds-form requests first, then the plain ones (each kernel walks a contiguous run of the request table)
    auto form = [](const PairRun *r) { return r->req.lds_form ? 0 : r->req.wide ? 2 : 1; };
    std::stable_sort(waiting.begin(), waiting.end(), [&](const PairRun *a, const PairRun *b) { return form(a) < form(b); });
    size_t n_lds = 0, n_wide = 0;
    for (const PairRun *r : waiting) { n_lds += r->req.lds_form ? 1 : 0; n_wide += r->req.wide ? 1 : 0; }
    size_t in_bytes = align_up(sizeof(SeamArgs) * R, 16), ctl_bytes = 0, out_bytes = 0;
    std::vector<size_t> o_inl(R), o_ctl(R), o_out(R);
    for (size_t i = 0; i < R; ++i) {
        const SweepReq &q = waiting[i]->req;
        o_inl[i] = in_bytes; in_bytes += align_up(q.inl_bytes + 4, 16);
        o_ctl[i] = ctl_bytes;
        if (!q.lds_form) ctl_bytes += align_up((size_t)q.rw * q.rh + 4, 16) + (q.wide ? align_up((size_t)(q.horizontal ? q.rh : q.rw) * 10 + 16, 16) : 0);   // wide: 2 float + 2 byte lines
        o_out[i] = out_bytes; out_bytes += align_up(sizeof(int) * (1 + 2 * q.pts), 16);
    }
    SSP_TRY(rb.grow(&rb.d_in, &rb.h_in, rb.pin_in, &rb.cap_in, in_bytes));
    SSP_TRY(rb.grow(&rb.d_ctl, nullptr, nullptr, &rb.cap_ctl, std::max<size_t>(ctl_bytes, 16)));
    SSP_TRY(rb.grow(&rb.d_out, &rb.h_out, rb.pin_out, &rb.cap_out, out_bytes));
    int block = 256;
    for (size_t i = 0; i < R; ++i) {
        SweepReq &q = waiting[i]->req;
        q.args.inl = (const uint8_t *)(rb.d_in + o_inl[i]);
        q.args.control = (uint8_t *)(rb.d_ctl + o_ctl[i]);
        q.args.out = (int *)(rb.d_out + o_out[i]);
        memcpy(rb.h_in + sizeof(SeamArgs) * i, &q.args, sizeof(SeamArgs));
        memcpy(rb.h_in + o_inl[i], q.inl.data(), q.inl_bytes);
        if (q.lds_form) { const int len = q.horizontal ? q.rh : q.rw; block = std::max(block, len <= 256 ? 256 : len <= 512 ? 512 : 1024); }   // fewer waves at the barrier of every line
    }
    SSP_HIP(hipMemcpyAsync(rb.d_in, rb.h_in, in_bytes, hipMemcpyHostToDevice, stream()));
    {
        double cells = 0;
        for (const PairRun *r : waiting) cells += (double)r->req.rw * r->req.rh;
        ProfileScope ps("seam_dp_sweep", cells * 10);
        if (n_lds) hipLaunchKernelGGL(k_dp_seam_lds, dim3((unsigned)n_lds), dim3(block), 0, stream(), (const SeamArgs *)rb.d_in);
        if (R > n_lds + n_wide) hipLaunchKernelGGL(k_dp_seam<false>, dim3((unsigned)(R - n_lds - n_wide)), dim3(1024), 0, stream(), (const SeamArgs *)rb.d_in + n_lds);
        if (n_wide) hipLaunchKernelGGL(k_dp_seam<true>, dim3((unsigned)n_wide), dim3(1024), 0, stream(), (const SeamArgs *)rb.d_in + (R - n_wide));
    }
    SSP_HIP(hipMemcpyAsync(rb.h_out, rb.d_out, out_bytes, hipMemcpyDeviceToHost, stream()));
    SSP_HIP(hipEventRecord(done, stream()));          // (the caller waits for it when it comes back to this group: the other group's host work runs meanwhile)
    for (size_t i = 0; i < R; ++i) outs[i] = (const int *)(rb.h_out + o_out[i]);
    return 0;
}

// All pairs, in DpSeamFinder's order wherever the order can matter: a pair starts once no earlier pair that shares an image with it is still
// waiting or running -- pairs without a common image read and cut disjoint masks.  The running pairs form two groups that take turns: while the
// sweeps of one group run on the device, the host (worker threads) resumes, advances and starts the pairs of the other.
static int run_rounds(const std::vector<PairJob> &jobs, std::vector<std::vector<uint8_t>> &hm, int n_images, int *rounds_out, int *sweeps_out)
{
    static const int threads = []() { const char *e = getenv("SSP_SEAM_DP_THREADS"); const int hw = (int)std::thread::hardware_concurrency();
                                       return e ? std::max(1, atoi(e)) : std::max(1, std::min(hw > 0 ? hw : 1, 16)); }();
    static const bool serial = getenv("SSP_SEAM_DP_SERIAL") != nullptr;      // (A/B: one pair at a time, one sweep per round trip -- the round-3 order of work)
    static const bool timing = getenv("SSP_SEAM_DP_TIMING") != nullptr;
    std::vector<size_t> pending;
    for (size_t q = 0; q < jobs.size(); ++q)
        if (jobs[q].iw > 0 && jobs[q].ih > 0) pending.push_back(q);
    struct Group {
        std::vector<std::unique_ptr<PairRun>> active;
        std::vector<PairRun *> waiting;          // their sweeps are in flight
        std::vector<const int *> outs;
        RoundBuffers rb;
        hipEvent_t ev = nullptr;
    } grp[2];
    static thread_local Pinned pins[4];
    grp[0].rb.pin_in = &pins[0]; grp[0].rb.pin_out = &pins[1]; grp[1].rb.pin_in = &pins[2]; grp[1].rb.pin_out = &pins[3];
    static thread_local std::vector<std::unique_ptr<PairRun>> spare;       // (the pairs' canvases, reused from call to call)
    int rc = 0, rounds = 0, sweeps = 0, max_active = 0;
    std::string err;
    double t_host = 0, t_wait = 0;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    for (Group &g : grp)
        if (hipEventCreateWithFlags(&g.ev, hipEventDisableTiming) != hipSuccess) rc = set_error(SSP_ERR_DEVICE, "DpSeamFinder: event creation failed");
    for (int turn = 0; !rc && (!pending.empty() || !grp[0].active.empty() || !grp[1].active.empty()); ++turn) {
        Group &g = grp[turn & 1];
        Group &other = grp[(turn & 1) ^ 1];
        // 1. the sweeps this group asked for a turn ago have had the other group's host work to finish in
        if (!g.waiting.empty()) {
            const auto t0 = now();
            if (hipEventSynchronize(g.ev) != hipSuccess) { rc = set_error(SSP_ERR_DEVICE, "DpSeamFinder: a sweep launch failed"); break; }
            const auto t1 = now();
            t_wait += ms(t0, t1);
            parallel_for((int)g.waiting.size(), threads, [&](int i) { pair_resume(*g.waiting[i], g.outs[i]); });
            t_host += ms(t1, now());
            g.waiting.clear();
        }
        for (size_t i = 0; i < g.active.size();) {
            if (g.active[i]->done) {
                if (g.active[i]->rc && !rc) { rc = g.active[i]->rc; err = g.active[i]->err; }
                spare.push_back(std::move(g.active[i]));
                g.active.erase(g.active.begin() + (long)i);
            } else ++i;
        }
        if (rc) break;
        // 2. start what may start (into the group with fewer pairs when it is this one's turn: both get work)
        std::vector<char> blocked((size_t)n_images, 0);
        for (const Group &q : grp)
            for (const auto &r : q.active) { blocked[r->job->a] = 1; blocked[r->job->b] = 1; }
        std::vector<PairRun *> fresh;
        std::vector<size_t> startable;           // positions in `pending`, in order
        for (size_t k = 0; k < pending.size(); ++k) {
            const PairJob &j = jobs[pending[k]];
            if (!blocked[j.a] && !blocked[j.b]) startable.push_back(k);
            blocked[j.a] = 1; blocked[j.b] = 1;          // later pairs of these images wait behind this one whether it starts now or not
        }
        // this group takes its half of everything that runs or may start (the rest starts in the other group's turn, a sweep's duration later)
        const size_t total = g.active.size() + other.active.size() + startable.size();
        size_t quota = serial ? (total == startable.size() && !startable.empty() ? 1 : 0) : ((total + 1) / 2 > g.active.size() ? (total + 1) / 2 - g.active.size() : 0);
        quota = std::min(quota, startable.size());
        for (size_t q = quota; q-- > 0;) {               // (back to front: erasing keeps the earlier positions valid)
            const size_t k = startable[q];
            const PairJob &j = jobs[pending[k]];
            std::unique_ptr<PairRun> run;
            if (!spare.empty()) { run = std::move(spare.back()); spare.pop_back(); } else run.reset(new PairRun());
            run->job = &j; run->mask1 = &hm[j.a]; run->mask2 = &hm[j.b];
            run->need_sweep = run->done = false; run->rc = 0; run->c1 = run->c2 = -1;
            fresh.push_back(run.get());
            g.active.push_back(std::move(run));
            pending.erase(pending.begin() + (long)k);
        }
        {
            const auto t0 = now();
            parallel_for((int)fresh.size(), threads, [&](int i) { pair_begin(*fresh[i]); pair_step(*fresh[i]); });
            t_host += ms(t0, now());
        }
        max_active = std::max(max_active, (int)(g.active.size() + other.active.size()));
        // 3. one launch for every sweep this group now asks for; it runs while the other group has its turn
        for (const auto &r : g.active)
            if (!r->done && r->need_sweep) g.waiting.push_back(r.get());
        if (!g.waiting.empty()) {
            rc = launch_sweeps(g.waiting, g.rb, g.outs, g.ev);
            if (rc) break;
            ++rounds; sweeps += (int)g.waiting.size();
        }
    }
    for (Group &g : grp) {
        if (g.ev) { (void)hipEventSynchronize(g.ev); (void)hipEventDestroy(g.ev); }
        g.rb.release();
    }
    if (timing) {
        fprintf(stderr, "seam_dp: %d sweeps in %d launches; host work %.1f ms, waiting for sweeps %.1f ms; at most %d pairs side by side, %d threads\n", sweeps, rounds, t_host, t_wait,
                max_active, threads);
        static const char *names[T_COUNT] = {"canvases+outlines", "components", "tips", "request", "relabel", "refresh", "cut", "edge scan"};
        fprintf(stderr, "seam_dp: thread time");
        for (int k = 0; k < T_COUNT; ++k) fprintf(stderr, "  %s %.2f ms", names[k], (double)g_tns[k].exchange(0) / 1e6);
        fprintf(stderr, "\n");
    }
    if (rounds_out) *rounds_out = rounds;
    if (sweeps_out) *sweeps_out = sweeps;
    if (rc && !err.empty()) return set_error(rc, "%s", err.c_str());
    return rc;
}

}  // namespace

// cost_func: 0 = 'COLOR' (images 8UC3 or 32FC3 of the masks' sizes: the same colour costs), 1 = 'COLOR_GRAD' (32FC3 only -- the reference
// passes float32 copies of its 8-bit seam-scale warps, and cv2's 8-bit grey is another, fixed-point, number).  masks: 8UC1, cut in place.  pair_order (optional, n(n-1) ints): the pairs in
// the order they were processed.
SSP_API int ssp_seam_dp(int n, const int *corners_xy, ssp_image *const *images, ssp_image *const *masks, int cost_func, int *pair_order)
{
    SSP_TRY(ensure_init());
    SSP_REQUIRE(n >= 0 && (n == 0 || (corners_xy && images && masks)), "seam_dp: null argument");
    SSP_REQUIRE(cost_func == 0 || cost_func == 1, "seam_dp: cost function must be 0 (COLOR) or 1 (COLOR_GRAD)");
    if (n == 0) return 0;
    for (int i = 0; i < n; ++i) {
        SSP_REQUIRE(masks[i] && masks[i]->depth == SSP_U8 && masks[i]->cn == 1, "seam_dp: mask %d must be CV_8UC1", i);
        SSP_REQUIRE(images[i] && images[i]->cn == 3 && (images[i]->depth == SSP_U8 || images[i]->depth == SSP_F32), "seam_dp: image %d must be CV_8UC3 or CV_32FC3", i);
        // COLOR_GRAD: cv2 runs cvtColor(BGR2GRAY) before Sobel; on 8-bit images that is a fixed-point grey rounded to uint8, which is not
        // what gray_at() computes and which the oracle does not restate.  The reference hands over float32 (sde.py:1601-1604): only that.
        SSP_REQUIRE(!(cost_func == 1 && images[i]->depth != SSP_F32), "seam_dp: COLOR_GRAD takes CV_32FC3 images (image %d is 8-bit; the reference converts with astype(np.float32), sde.py:1601-1604)", i);
        SSP_REQUIRE(images[i]->w == masks[i]->w && images[i]->h == masks[i]->h, "seam_dp: image %d is %dx%d but its mask %dx%d", i, images[i]->w, images[i]->h, masks[i]->w,
                    masks[i]->h);
    }
    // ---- DpSeamFinder::find: all pairs, std::sort by centre distance (libstdc++, as in the reference's OpenCV wheel), reversed
    std::vector<std::pair<size_t, size_t>> pairs;
    for (size_t i = 0; i + 1 < (size_t)n; ++i)
        for (size_t j = i + 1; j < (size_t)n; ++j) pairs.push_back({i, j});
    auto centre_dist = [&](const std::pair<size_t, size_t> &p) {
        const int ax = corners_xy[2 * p.first] + masks[p.first]->w / 2, ay = corners_xy[2 * p.first + 1] + masks[p.first]->h / 2;
        const int bx = corners_xy[2 * p.second] + masks[p.second]->w / 2, by = corners_xy[2 * p.second + 1] + masks[p.second]->h / 2;
        return (ax - bx) * (ax - bx) + (ay - by) * (ay - by);
    };
    std::sort(pairs.begin(), pairs.end(), [&](const std::pair<size_t, size_t> &l, const std::pair<size_t, size_t> &r) { return centre_dist(l) < centre_dist(r); });
    std::reverse(pairs.begin(), pairs.end());
    if (pair_order)
        for (size_t q = 0; q < pairs.size(); ++q) { pair_order[2 * q] = (int)pairs[q].first; pair_order[2 * q + 1] = (int)pairs[q].second; }

    // ---- device: gradients of all images, cost planes of all overlapping pairs
    std::vector<GradImg> gi(n);
    std::vector<int> gblocks(n + 1, 0);
    size_t grad_floats = 0;
    for (int i = 0; i < n; ++i) grad_floats += (size_t)images[i]->w * images[i]->h;
    float *grad = nullptr;
    if (cost_func) SSP_TRY(pool_alloc(sizeof(float) * 2 * std::max<size_t>(grad_floats, 1), (void **)&grad));
    {
        size_t off = 0;
        for (int i = 0; i < n; ++i) {
            const size_t px = (size_t)images[i]->w * images[i]->h;
            gi[i] = GradImg{images[i]->data, images[i]->pitch, images[i]->w, images[i]->h, images[i]->depth, grad ? grad + off : nullptr, grad ? grad + grad_floats + off : nullptr};
            off += px;
            gblocks[i + 1] = gblocks[i] + (int)((px + 255) / 256);
        }
    }
    std::vector<PairJob> jobs;
    std::vector<PairCost> pc;
    std::vector<int> pblocks(1, 0);
    size_t cost_floats = 0;
    for (const auto &p : pairs) {
        const int a = (int)p.first, b = (int)p.second;
        PairJob j;
        j.a = a; j.b = b;
        j.tl1x = corners_xy[2 * a]; j.tl1y = corners_xy[2 * a + 1]; j.w1 = masks[a]->w; j.h1 = masks[a]->h;
        j.tl2x = corners_xy[2 * b]; j.tl2y = corners_xy[2 * b + 1]; j.w2 = masks[b]->w; j.h2 = masks[b]->h;
        j.ix0 = std::max(j.tl1x, j.tl2x); j.iy0 = std::max(j.tl1y, j.tl2y);
        j.iw = std::min(j.tl1x + j.w1, j.tl2x + j.w2) - j.ix0; j.ih = std::min(j.tl1y + j.h1, j.tl2y + j.h2) - j.iy0;
        j.cv = j.ch = nullptr;
        if (j.iw > 0 && j.ih > 0) cost_floats += 2 * (size_t)j.iw * j.ih;
        jobs.push_back(j);
    }
    float *costs = nullptr;
    {
        const int arc = pool_alloc(sizeof(float) * std::max<size_t>(cost_floats, 1), (void **)&costs);
        if (arc) { if (grad) pool_free(grad); return arc; }
    }
    {
        size_t off = 0;
        for (PairJob &j : jobs) {
            if (j.iw <= 0 || j.ih <= 0) continue;
            const size_t px = (size_t)j.iw * j.ih;
            PairCost c;
            c.a = j.a; c.b = j.b; c.iw = j.iw; c.ih = j.ih;
            c.ax = j.ix0 - j.tl1x; c.ay = j.iy0 - j.tl1y; c.bx = j.ix0 - j.tl2x; c.by = j.iy0 - j.tl2y;
            c.cv = costs + off; c.ch = costs + off + px;
            j.cv = c.cv; j.ch = c.ch;
            off += 2 * px;
            pc.push_back(c);
            pblocks.push_back(pblocks.back() + (int)((px + 255) / 256));
        }
    }
    // descriptor tables
    const size_t tb_gi = sizeof(GradImg) * n, tb_gb = sizeof(int) * (n + 1), tb_pc = sizeof(PairCost) * std::max<size_t>(pc.size(), 1), tb_pb = sizeof(int) * pblocks.size();
    char *tables = nullptr;
    const size_t o_gb = align_up(tb_gi, 16), o_pc = o_gb + align_up(tb_gb, 16), o_pb = o_pc + align_up(tb_pc, 16), tb_all = o_pb + align_up(tb_pb, 16);
    int rc = pool_alloc(tb_all, (void **)&tables);
    std::vector<char> host_tables(tb_all, 0);
    if (!rc) {
        memcpy(host_tables.data(), gi.data(), tb_gi);
        memcpy(host_tables.data() + o_gb, gblocks.data(), tb_gb);
        if (!pc.empty()) memcpy(host_tables.data() + o_pc, pc.data(), sizeof(PairCost) * pc.size());
        memcpy(host_tables.data() + o_pb, pblocks.data(), tb_pb);
        if (hipMemcpyAsync(tables, host_tables.data(), tb_all, hipMemcpyHostToDevice, stream()) != hipSuccess) rc = set_error(SSP_ERR_DEVICE, "seam_dp: descriptor upload failed");
    }
    if (!rc && cost_func && gblocks[n] > 0) {
        ProfileScope ps("seam_dp_gradients", (double)grad_floats * (12 + 8));
        hipLaunchKernelGGL(k_dp_gradients, dim3(gblocks[n]), dim3(256), 0, stream(), (const GradImg *)tables, (const int *)(tables + o_gb), n);
    }
    if (!rc && !pc.empty()) {
        ProfileScope ps("seam_dp_costs", (double)cost_floats * (4 + 24));
        hipLaunchKernelGGL(k_dp_pair_costs, dim3(pblocks.back()), dim3(256), 0, stream(), (const GradImg *)tables, (const PairCost *)(tables + o_pc), (const int *)(tables + o_pb),
                           (int)pc.size(), cost_func);
    }
    // ---- masks to the host (one synchronisation), pairs in order, masks back
    // (a pitched host <-> device copy goes row by row through the runtime's staging buffers; the padded planes travel as ONE linear
    // copy each and are packed / spread on the host)
    // All of them through ONE pinned staging buffer of this thread: copies to pageable memory are staged by the runtime and block, 2 x 21 of them
    // were 3-4 ms of the recorded run's 25.
    std::vector<std::vector<uint8_t>> hm(n);
    std::vector<size_t> off(n + 1, 0);
    std::vector<char> lin(n, 0);
    for (int i = 0; i < n; ++i) {
        lin[i] = masks[i]->pitch - (size_t)masks[i]->w <= 64;      // a view into a wider plane keeps the strided copy
        off[i + 1] = off[i] + align_up(lin[i] ? masks[i]->pitch * (size_t)(masks[i]->h - 1) + masks[i]->w : (size_t)masks[i]->w * masks[i]->h, 64);
    }
    static thread_local Pinned stage;
    if (!rc) rc = stage.need(off[n]);
    for (int i = 0; i < n && !rc; ++i) {
        hm[i].resize((size_t)masks[i]->w * masks[i]->h);
        const hipError_t e = lin[i] ? hipMemcpyAsync(stage.p + off[i], masks[i]->data, masks[i]->pitch * (size_t)(masks[i]->h - 1) + masks[i]->w, hipMemcpyDeviceToHost, stream())
                                    : hipMemcpy2DAsync(stage.p + off[i], masks[i]->w, masks[i]->data, masks[i]->pitch, masks[i]->w, masks[i]->h, hipMemcpyDeviceToHost, stream());
        if (e != hipSuccess) rc = set_error(SSP_ERR_DEVICE, "seam_dp: mask download failed");
    }
    if (!rc && hipStreamSynchronize(stream()) != hipSuccess) rc = set_error(SSP_ERR_DEVICE, "seam_dp: synchronisation failed");
    for (int i = 0; i < n && !rc; ++i) {
        const size_t hp = lin[i] ? masks[i]->pitch : (size_t)masks[i]->w;
        for (int y = 0; y < masks[i]->h; ++y) memcpy(&hm[i][(size_t)y * masks[i]->w], stage.p + off[i] + (size_t)y * hp, (size_t)masks[i]->w);
    }
    int rounds = 0, sweeps = 0;
    if (!rc) rc = run_rounds(jobs, hm, n, &rounds, &sweeps);
    for (int i = 0; i < n && !rc; ++i) {
        const size_t hp = lin[i] ? masks[i]->pitch : (size_t)masks[i]->w;
        for (int y = 0; y < masks[i]->h; ++y) memcpy(stage.p + off[i] + (size_t)y * hp, &hm[i][(size_t)y * masks[i]->w], (size_t)masks[i]->w);   // (the row padding keeps what came down)
        const hipError_t e = lin[i] ? hipMemcpyAsync(masks[i]->data, stage.p + off[i], masks[i]->pitch * (size_t)(masks[i]->h - 1) + masks[i]->w, hipMemcpyHostToDevice, stream())
                                    : hipMemcpy2DAsync(masks[i]->data, masks[i]->pitch, stage.p + off[i], masks[i]->w, masks[i]->w, masks[i]->h, hipMemcpyHostToDevice, stream());
        if (e != hipSuccess) rc = set_error(SSP_ERR_DEVICE, "seam_dp: mask upload failed");
    }
    // the staging buffer is this thread's next call's too: the uploads have to be through
    if (hipStreamSynchronize(stream()) != hipSuccess && !rc) rc = set_error(SSP_ERR_DEVICE, "seam_dp: synchronisation failed");
    for (int i = 0; i < n; ++i) image_note_read(images[i]);
    if (tables) pool_free(tables);
    pool_free(costs);
    if (grad) pool_free(grad);
    if (!rc && hipGetLastError() != hipSuccess) rc = set_error(SSP_ERR_DEVICE, "seam_dp: a kernel launch failed");
    return rc;
}
