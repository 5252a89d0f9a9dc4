// ssp_warp.hip -- cv.PyRotationWarper on gfx950: ROI scan, fused mapBackward + remap kernels.
//
// Replaces (reference call sites in stitching_detailed_enhanced.py):
//   :1545/:1684 cv.PyRotationWarper(type, scale)           -> ssp_warper_create
//   :1696       warper.warpRoi                              -> ssp_warper_roi          (k_roi_scan)
//   :1557/:1731 warper.warp(img, LINEAR|AREA, REFLECT)      -> ssp_warper_warp*        (k_warp_sep / k_warp_generic)
//   :1591/:1740 warper.warp(mask, NEAREST, CONSTANT)        -> fused into the same pass (ssp_warper_warp_with_mask)
// OpenCV builds float maps with a single-threaded buildMaps and then runs cv::remap; here the map is
// computed in registers and never written to HBM.  HBM-bound: B_warp = 3c*S + (3c+1)*D bytes per frame.
#include "ssp_internal.hpp"
#include "ssp_projector.hpp"
#include "ssp_warp_device.hpp"

using namespace ssp;

struct ssp_warper {
    Projector p;
    std::string type;
};
// The recent warpRois of the process: the reference makes a warper per panorama (sde.py:1684), asks for the rois of all images, then warps every
// image and its all-255 mask with the same sizes and cameras (sde.py:1696, :1731, :1740), panorama after panorama -- one device scan + read-back
// per camera instead of three per panorama (a pure function of projection, scale, frame size and camera); most recently used first
struct RoiEntry { int kind, w, h, val[4]; float scale, a, b, K[9], R[9], T[3]; };
static std::vector<RoiEntry> g_rois;
static std::mutex g_rois_mutex;

// ---- host: ProjectorBase::setCameraParams -----------------------------------------------------------------
namespace ssp {
static void mul3(const float *a, const float *b, float *d)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j];
            d[3 * i + j] = s + a[3 * i + 2] * b[6 + j];
        }
}
static void inv3(const float *m, float *d)
{
    // cv::invert, 3x3 float: cofactors / determinant evaluated in double, stored as float
    double a = m[0], b = m[1], c = m[2], e = m[3], f = m[4], g = m[5], h = m[6], i = m[7], j = m[8];
    double det = a * (f * j - g * i) - b * (e * j - g * h) + c * (e * i - f * h);
    if (det == 0.0) {
        for (int q = 0; q < 9; ++q) d[q] = 0.f;
        return;
    }
    double r = 1.0 / det;
    d[0] = (float)((f * j - g * i) * r);
    d[1] = (float)((c * i - b * j) * r);
    d[2] = (float)((b * g - c * f) * r);
    d[3] = (float)((g * h - e * j) * r);
    d[4] = (float)((a * j - c * h) * r);
    d[5] = (float)((c * e - a * g) * r);
    d[6] = (float)((e * i - f * h) * r);
    d[7] = (float)((b * h - a * i) * r);
    d[8] = (float)((a * f - b * e) * r);
}
void set_camera(Projector &p, const float K[9], const float Rin[9])
{
    float R[9], T[3] = {0.f, 0.f, 0.f};
    memcpy(R, Rin, sizeof R);
    if (p.kind == PK_AFFINE) {
        // AffineWarper::getRTfromHomogeneous: R = (H with the translation column zeroed)^T, T = -(R * (tx, ty, 0))
        float tx = R[2], ty = R[5];
        R[2] = 0.f;
        R[5] = 0.f;
        float Rt[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Rt[3 * i + j] = R[3 * j + i];
        memcpy(R, Rt, sizeof R);
        for (int i = 0; i < 3; ++i) {
            float s = R[3 * i] * tx + R[3 * i + 1] * ty;
            s = s + R[3 * i + 2] * 0.f;
            T[i] = s * -1.f;
        }
    }
    memcpy(p.k, K, sizeof p.k);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) p.rinv[3 * i + j] = R[3 * j + i];
    float kinv[9];
    inv3(K, kinv);
    mul3(R, kinv, p.r_kinv);
    mul3(K, p.rinv, p.k_rinv);
    memcpy(p.t, T, sizeof T);
}
}  // namespace ssp

// ---- ROI scan: min/max of mapForward over all / border / corner pixels -----------------------------------------
struct MinMax { float lo_u, lo_v, hi_u, hi_v; };

__device__ inline void mm_take(MinMax &m, float u, float v)
{
    // std::min(a, b) = (b < a) ? b : a : a NaN never replaces the running extreme
    m.lo_u = (u < m.lo_u) ? u : m.lo_u;
    m.lo_v = (v < m.lo_v) ? v : m.lo_v;
    m.hi_u = (m.hi_u < u) ? u : m.hi_u;
    m.hi_v = (m.hi_v < v) ? v : m.hi_v;
}

__global__ __launch_bounds__(256) void k_roi_scan(Projector p, int W, int H, int mode, long long npts, MinMax *partial)
{
    MinMax m = {3.402823466e+38f, 3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npts; i += (long long)gridDim.x * blockDim.x) {
        int x, y;
        if (mode == SCAN_FULL) {
            y = (int)(i / W);
            x = (int)(i - (long long)y * W);
        } else if (mode == SCAN_BORDER) {
            if (i < W) { x = (int)i; y = 0; }
            else if (i < 2LL * W) { x = (int)(i - W); y = H - 1; }
            else if (i < 2LL * W + H) { x = 0; y = (int)(i - 2LL * W); }
            else { x = W - 1; y = (int)(i - 2LL * W - H); }
        } else {
            x = (i & 2) ? W - 1 : 0;
            y = (i & 1) ? H - 1 : 0;
        }
        float u, v;
        map_forward(p, (float)x, (float)y, u, v);
        mm_take(m, u, v);
    }
    // wave reduction (64 lanes), then across the 4 waves through LDS
    for (int off = 32; off > 0; off >>= 1) {
        MinMax o;
        o.lo_u = __shfl_down(m.lo_u, off);
        o.lo_v = __shfl_down(m.lo_v, off);
        o.hi_u = __shfl_down(m.hi_u, off);
        o.hi_v = __shfl_down(m.hi_v, off);
        m.lo_u = (o.lo_u < m.lo_u) ? o.lo_u : m.lo_u;
        m.lo_v = (o.lo_v < m.lo_v) ? o.lo_v : m.lo_v;
        m.hi_u = (m.hi_u < o.hi_u) ? o.hi_u : m.hi_u;
        m.hi_v = (m.hi_v < o.hi_v) ? o.hi_v : m.hi_v;
    }
    __shared__ MinMax sm[4];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sm[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < 4; ++wv) {
            MinMax o = sm[wv];
            m.lo_u = (o.lo_u < m.lo_u) ? o.lo_u : m.lo_u;
            m.lo_v = (o.lo_v < m.lo_v) ? o.lo_v : m.lo_v;
            m.hi_u = (m.hi_u < o.hi_u) ? o.hi_u : m.hi_u;
            m.hi_v = (m.hi_v < o.hi_v) ? o.hi_v : m.hi_v;
        }
        partial[blockIdx.x] = m;
    }
}

namespace ssp {
int detect_roi(const Projector &p, int W, int H, int roi[4])
{
    SSP_TRY(ensure_init());
    SSP_REQUIRE(W > 0 && H > 0, "warpRoi: empty source size %dx%d", W, H);
    int mode = roi_scan_mode(p.kind);
    long long npts = mode == SCAN_FULL ? (long long)W * H : mode == SCAN_BORDER ? 2LL * W + 2LL * H : 4;
    int blocks = (int)std::min<long long>(2048, (npts + 255) / 256);
    MinMax *d_part = nullptr;
    SSP_TRY(pool_alloc(sizeof(MinMax) * blocks, (void **)&d_part));
    {
        ProfileScope ps("roi_scan", 0);
        hipLaunchKernelGGL(k_roi_scan, dim3(blocks), dim3(256), 0, stream(), p, W, H, mode, npts, d_part);
    }
    std::vector<MinMax> part(blocks);
    hipError_t e = hipMemcpyAsync(part.data(), d_part, sizeof(MinMax) * blocks, hipMemcpyDeviceToHost, stream());
    if (e == hipSuccess) e = hipStreamSynchronize(stream());
    pool_free(d_part);
    if (e != hipSuccess) SSP_FAIL(SSP_ERR_DEVICE, "roi scan failed: %s", hipGetErrorString(e));
    MinMax m = part[0];
    for (int i = 1; i < blocks; ++i) {
        const MinMax &o = part[i];
        m.lo_u = (o.lo_u < m.lo_u) ? o.lo_u : m.lo_u;
        m.lo_v = (o.lo_v < m.lo_v) ? o.lo_v : m.lo_v;
        m.hi_u = (m.hi_u < o.hi_u) ? o.hi_u : m.hi_u;
        m.hi_v = (m.hi_v < o.hi_v) ? o.hi_v : m.hi_v;
    }
    int tlx = (int)m.lo_u, tly = (int)m.lo_v, brx = (int)m.hi_u, bry = (int)m.hi_v;
    if (p.kind == PK_SPHERICAL) {
        // SphericalWarper::detectResultRoi: a pole that projects inside the frame extends the range to u = 0,
        // v = pi*scale (north) or v = 0 (south); applied after the truncation above
        float tl_uf = (float)tlx, tl_vf = (float)tly, br_uf = (float)brx, br_vf = (float)bry;
        for (int pass = 0; pass < 2; ++pass) {
            float x = p.rinv[1], y = pass == 0 ? p.rinv[4] : -p.rinv[4], z = p.rinv[7];
            if (y > 0.f) {
                float x_ = (p.k[0] * x + p.k[1] * y) / z + p.k[2];
                float y_ = p.k[4] * y / z + p.k[5];
                if (x_ > 0.f && x_ < W && y_ > 0.f && y_ < H) {
                    float pv = pass == 0 ? (float)(SSP_PI_D * p.scale) : 0.f;
                    tl_uf = (0.f < tl_uf) ? 0.f : tl_uf;
                    tl_vf = (pv < tl_vf) ? pv : tl_vf;
                    br_uf = (br_uf < 0.f) ? 0.f : br_uf;
                    br_vf = (br_vf < pv) ? pv : br_vf;
                }
            }
        }
        tlx = (int)tl_uf; tly = (int)tl_vf; brx = (int)br_uf; bry = (int)br_vf;
    }
    roi[0] = tlx;
    roi[1] = tly;
    roi[2] = brx - tlx + 1;
    roi[3] = bry - tly + 1;
    SSP_REQUIRE(roi[2] > 0 && roi[3] > 0 && (long long)roi[2] * roi[3] < (1LL << 33),
                "warpRoi: degenerate or absurd roi %dx%d (try another projection or wave correction, cf. sde.py:1576-1586)", roi[2], roi[3]);
    return 0;
}
}  // namespace ssp

// ---- live columns of a warped frame ------------------------------------------------------------------------------------
// col[x] = 1 when any pixel of column x of the roi carries a set warped mask (INTER_NEAREST + BORDER_CONSTANT on the all-255 mask,
// sde.py:1740).  A frame that straddles u = +-pi*scale gets OpenCV's full-circle roi (detectResultRoiByBorder sees u near both ends):
// its mask is set in two column ranges at the ends of the roi and nowhere in the 5/6 of the circle between them.
__global__ __launch_bounds__(256) void k_live_columns(Projector p, int W, int H, int dw, int dh, int tlx, int tly, int rows_per_block, uint8_t *col)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= dw) return;
    const int y0 = blockIdx.y * rows_per_block, y1 = min(y0 + rows_per_block, dh);
    bool live = false;
    for (int y = y0; y < y1 && !live; ++y) {
        float fx, fy;
        map_backward(p, (float)(x + tlx), (float)(y + tly), fx, fy);
        const int mx = sat_s16(cv_round(fx)), my = sat_s16(cv_round(fy));
        live = (unsigned)mx < (unsigned)W && (unsigned)my < (unsigned)H;
    }
    if (live) col[x] = 1;
}

namespace ssp {
// The column ranges of `roi` that the blender has to see -- the set mask columns grown by `reach` (beyond 4 * 2^bands pixels of every set
// mask pixel nothing of a fed image is ever multiplied by anything but 0: k_warp_records_far) -- as one or two rectangles `parts` (absolute
// warped coordinates, full height).  One rectangle equal to the roi unless the frame's live columns leave a dead run wide enough to pay:
// then the two ranges either side of the longest dead run (a straddling frame), or the trimmed range.  Only rois much larger than the
// frame are scanned at all (a roi is the bounding box of the frame's image: a dead run needs the box to be mostly empty).
int live_parts(const Projector &p, int W, int H, const int roi[4], int reach, int parts[2][4], int *n_parts)
{
    auto whole = [&]() { memcpy(parts[0], roi, 4 * sizeof(int)); *n_parts = 1; return 0; };
    const int dw = roi[2], dh = roi[3];
    const int margin = reach + 32;
    if (reach <= 0 || (long long)dw * dh < 2LL * W * H || dw <= 4 * margin + 512) return whole();
    SSP_TRY(ensure_init());
    uint8_t *d_col = nullptr;
    SSP_TRY(pool_alloc((size_t)dw, (void **)&d_col));
    std::vector<uint8_t> col((size_t)dw);
    hipError_t e = hipMemsetAsync(d_col, 0, (size_t)dw, stream());
    if (e == hipSuccess) {
        const int rows_per_block = 64;
        hipLaunchKernelGGL(k_live_columns, dim3((dw + 255) / 256, (dh + rows_per_block - 1) / rows_per_block), dim3(256), 0, stream(), p, W, H, dw, dh, roi[0], roi[1], rows_per_block, d_col);
        e = hipMemcpyAsync(col.data(), d_col, (size_t)dw, hipMemcpyDeviceToHost, stream());
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream());
    pool_free(d_col);
    if (e != hipSuccess) SSP_FAIL(SSP_ERR_DEVICE, "live column scan failed: %s", hipGetErrorString(e));
    int first = -1, last = -1;
    for (int x = 0; x < dw; ++x)
        if (col[x]) { if (first < 0) first = x; last = x; }
    if (first < 0) return whole();
    int g0 = -1, g1 = -2;   // the longest dead run strictly inside [first, last]
    for (int x = first; x <= last;) {
        if (col[x]) { ++x; continue; }
        int e2 = x;
        while (e2 + 1 <= last && !col[e2 + 1]) ++e2;
        if (e2 - x > g1 - g0) { g0 = x; g1 = e2; }
        x = e2 + 1;
    }
    auto put = [&](int k, int a0, int a1) { parts[k][0] = roi[0] + a0; parts[k][1] = roi[1]; parts[k][2] = a1 - a0 + 1; parts[k][3] = dh; };
    if (g1 - g0 + 1 > 2 * margin + 512) {
        put(0, std::max(0, first - margin), std::min(dw - 1, g0 - 1 + margin));
        put(1, std::max(0, g1 + 1 - margin), std::min(dw - 1, last + margin));
        *n_parts = 2;
        return 0;
    }
    const int a0 = std::max(0, first - margin), a1 = std::min(dw - 1, last + margin);
    if ((long long)(a1 - a0 + 1) * 10 < (long long)dw * 9) { put(0, a0, a1); *n_parts = 1; return 0; }
    return whole();
}
}  // namespace ssp

// ---- generic kernel: any projection, interpolation, border, u8/f32, 1 or 3 channels ---------------------------
template <typename T, int CN>
__global__ __launch_bounds__(256) void k_warp_generic(Projector p, SrcView src, void *dst, size_t dpitch, int dw, int dh, int tlx,
                                                      int tly, int interp, int border, float *xmap, float *ymap)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    float fx, fy;
    map_backward(p, (float)(x + tlx), (float)(y + tly), fx, fy);
    if (xmap) {
        xmap[(size_t)y * dw + x] = fx;
        ymap[(size_t)y * dw + x] = fy;
        if (!dst) return;
    }
    T out[CN];
    remap_pixel<T, CN>(src, fx, fy, interp, border, out);
    T *d = (T *)((char *)dst + (size_t)y * dpitch) + (size_t)x * CN;
#pragma unroll
    for (int c = 0; c < CN; ++c) d[c] = out[c];
}

// warpBackward: destination pixel (x, y) of the original frame reads the warped image at mapForward(x, y) - roi.tl
template <typename T, int CN>
__global__ __launch_bounds__(256) void k_warp_backward(Projector p, SrcView src, void *dst, size_t dpitch, int dw, int dh, int tlx, int tly, int interp, int border)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    float u, v;
    map_forward(p, (float)x, (float)y, u, v);
    T out[CN];
    remap_pixel<T, CN>(src, u - (float)tlx, v - (float)tly, interp, border, out);
    T *d = (T *)((char *)dst + (size_t)y * dpitch) + (size_t)x * CN;
#pragma unroll
    for (int c = 0; c < CN; ++c) d[c] = out[c];
}

// ---- separable projections: per-column / per-row trigonometry tables -----------------------------------------
__global__ void k_sep_tables(int kind, float scale, int tlx, int tly, int dw, int dh, float *colS, float *colC, float *rowA, float *rowB)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < dw) {
        float u = (float)(i + tlx);
        u /= scale;
        colS[i] = ssp_sinf(u);
        colC[i] = ssp_cosf(u);
    } else if (i < dw + dh) {
        int j = i - dw;
        float v = (float)(j + tly);
        v /= scale;
        float a, b;
        if (kind == PK_SPHERICAL) {
            a = ssp_sinf(SSP_PI_F - v);
            b = ssp_cosf(SSP_PI_F - v);
        } else if (kind == PK_CYLINDRICAL) {
            a = 1.0f;
            b = v;
        } else {  // PK_MERCATOR
            float lat = ssp_atanf(ssp_sinhf(v));
            a = ssp_cosf(lat);
            b = ssp_sinf(lat);
        }
        rowA[j] = a;
        rowB[j] = b;
    }
}

// ---- fused fast path: separable projection, u8c3 source, INTER_LINEAR + border, plus the NEAREST/CONSTANT mask
// Each lane produces 4 consecutive output pixels of one row; a 256-thread group covers a 256 x 4 tile.
struct SepArgs {
    SrcView src;
    uint8_t *dst; size_t dpitch;   // u8c3
    uint8_t *mask; size_t mpitch;  // u8 or null
    int dw, dh;
    const float *colS, *colC, *rowA, *rowB;
    float kr[9];
    int border;
    float hix, hiy;  // largest source coordinate whose cvRound is still inside the frame (see nearest_hi)
    int xshift;      // 0..3: lane groups cover columns [4j - xshift, 4j - xshift + 4) so that the stores are 4-byte aligned in the
                     // destination plane; the column tables are stored shifted by the same amount (entry t = column t - xshift, clamped)
};

typedef uint32_t u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint16_t u16_u1 __attribute__((aligned(1)));
typedef uint32_t u32_u1 __attribute__((aligned(1)));

// mask preparation fused into the warp epilogue (sde.py:1760-1772): the dilated seam-scale mask, resized with
// INTER_LINEAR_EXACT to the warped size, AND-ed with the warped all-255 mask.  Tables as in k_lin_exact_tab.
struct MaskPrep {
    const uint8_t *dil; size_t dpitch;
    const int *xo, *xc, *yo, *yc;  // xo/xc padded to a multiple of 4 entries
    const int *flags; int fgx;     // per (seam row yo, 256-column segment): 1 = every seam-mask sample the segment interpolates from rows yo, yo+1 is 255
};

// exposure compensation fused into the warp epilogue (sde.py:1754 compensator.apply on the warped frame): kind 1 = one gain per
// channel (Gain / ChannelsCompensator), kind 2 = gain map (Blocks*Compensator: resize(gain_map, frame size, INTER_LINEAR) in f32 --
// pixel-centre mapping, horizontal then vertical lerp -- then multiply, cvRound, saturate), same operation order as k_apply_map.
struct GainArgs {
    int kind;
    float g[3];
    const float *gm; int gw, gh, gcn;
    const int *xi; const float *xa;   // per column (shifted like the warp's tables): first tap index, weight of the second tap
    const int *yi; const float *yb;   // per row
};

__device__ inline void gain_lin_coord(int d, int ssize, int dsize, int &s0, float &f)
{
    double scale = 1.0 / ((double)dsize / (double)ssize);
    float fv = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(fv);
    fv -= s;
    if (s < 0) { fv = 0; s = 0; }
    if (s >= ssize - 1) { fv = 0; s = ssize - 1; }
    s0 = s;
    f = fv;
}

__device__ inline uint32_t gain_apply_px(uint32_t px, float g0, float g1, float g2)
{
    const float r0 = __builtin_rintf((float)(px & 0xffu) * g0), r1 = __builtin_rintf((float)((px >> 8) & 0xffu) * g1), r2 = __builtin_rintf((float)((px >> 16) & 0xffu) * g2);
    const uint32_t b = (uint32_t)(r0 < 0.f ? 0 : (r0 > 255.f ? 255 : (int)r0)), g = (uint32_t)(r1 < 0.f ? 0 : (r1 > 255.f ? 255 : (int)r1)),
                   r = (uint32_t)(r2 < 0.f ? 0 : (r2 > 255.f ? 255 : (int)r2));
    return b | (g << 8) | (r << 16);
}

// the prepared seam mask of four consecutive pixels (table index t0 .. t0 + 3 of row y), one byte each: the dilated seam-scale mask
// resized with INTER_LINEAR_EXACT (sde.py:1760-1768): (h0*(256-cy) + h1*cy + 2^15) >> 16 with h = p[o]*(256-cx) + p[o+1]*cx ; coefficient -1 =
// copy the edge sample
__device__ inline uint32_t seam_mask4(const MaskPrep &mpr, int y, int t0)
{
    const MaskPrep *mp = &mpr;
    uint32_t sm = 0;
    {
        const int4 o4 = *(const int4 *)(mp->xo + t0);
        const int o[4] = {o4.x, o4.y, o4.z, o4.w};
        const int cyv = mp->yc[y];
        const uint8_t *r0 = mp->dil + (size_t)mp->yo[y] * mp->dpitch;
        const uint8_t *r1 = cyv >= 0 ? r0 + mp->dpitch : r0;
        const uint32_t cy1 = cyv >= 0 ? (uint32_t)cyv : 0u;
        const u16x2 wy = __builtin_bit_cast(u16x2, (256u - cy1) | (cy1 << 16));
        // upscaling: the 4 pixels' sample pairs (o, o+1) lie within 4 consecutive samples -> one 4-byte read per row
        const bool narrow = o[3] - o[0] <= 2;
        const bool all_narrow = __ballot(!narrow) == 0ULL;
        uint32_t w0 = 0, w1 = 0;
        if (all_narrow) {
            w0 = *(const u32_u1 *)(r0 + o[0]);
            w1 = *(const u32_u1 *)(r1 + o[0]);
        }
        // inside the seam mask every sample is 255 and so is every interpolated value: nothing to compute (and no coefficients to load)
        if (all_narrow && __ballot((w0 & w1) != 0xffffffffu) == 0ULL) sm = 0xffffffffu;
        else {
        const int4 c4v = *(const int4 *)(mp->xc + t0);
        const int cxv[4] = {c4v.x, c4v.y, c4v.z, c4v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int oi = o[i], ci = cxv[i];
            uint32_t p0, p1;  // samples o and o+1 of both rows in the low 16 bits
            if (all_narrow) {
                const uint32_t sh = 8u * (uint32_t)(oi - o[0]);
                p0 = (w0 >> sh) & 0xffffu;
                p1 = (w1 >> sh) & 0xffffu;
            } else {
                p0 = *(const u16_u1 *)(r0 + oi);
                p1 = *(const u16_u1 *)(r1 + oi);
            }
            const uint32_t cx1 = ci >= 0 ? (uint32_t)ci : 0u;
            const u16x2 wxp = __builtin_bit_cast(u16x2, (256u - cx1) | (cx1 << 16));
            // bytes -> u16 pairs, then h = dot2(pair, (256-cx, cx)); h <= 255*256 fits 16 bits for the vertical dot2
            const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, (p0 & 0xffu) | ((p0 & 0xff00u) << 8)), wxp, 0u, false);
            const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, (p1 & 0xffu) | ((p1 & 0xff00u) << 8)), wxp, 0u, false);
            const uint32_t v = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, h0 | (h1 << 16)), wy, 1u << 15, false);
            sm |= (v >> 16) << (8 * i);
        }
        }
    }
    return sm;
}

// tile = (4*LX) pixels x (256/LX) rows per 256-thread group; LX = 64: 256x4, 32: 128x8, 16: 64x16
// GEN: the map of every pixel by map_backward(*gp, ...) (any projection; tlx / tly = the part's corner) instead of the separable tables
template <int LX, bool GAIN = false, bool GEN = false>
__device__ inline void warp_sep_body(const SepArgs &a, const bool prep, const MaskPrep mpv, int bx, int by, const GainArgs *ga = nullptr, const Projector *gp = nullptr, int tlx = 0,
                                     int tly = 0)
{
    const int lane = threadIdx.x & (LX - 1);
    int y = by * (256 / LX) + (threadIdx.x / LX);
    if (LX == 64) y = __builtin_amdgcn_readfirstlane(y);  // one row per wave: row tables come through scalar loads
    const int t0 = (bx * LX + lane) * 4, x0 = t0 - a.xshift;  // t0: table index (16-byte aligned), x0: first column (may be < 0)
    if (y >= a.dh || x0 >= a.dw) return;
    const float ra = GEN ? 0.f : a.rowA[y], rb = GEN ? 0.f : a.rowB[y];
    // row-constant parts of K*R^T*ray: kr[1]*y_, kr[4]*y_, kr[7]*y_
    const float c1 = a.kr[1] * rb, c4 = a.kr[4] * rb, c7 = a.kr[7] * rb;
    // the tables are padded (entries beyond the roi repeat valid columns): one 16-byte load per table, no special cases below;
    // columns outside [0, dw) are computed like the others and not stored
    const float4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float4 cs4 = GEN ? zero4 : *(const float4 *)(a.colS + t0), cc4 = GEN ? zero4 : *(const float4 *)(a.colC + t0);
    const bool full = x0 >= 0 && x0 + 4 <= a.dw;
    const f32x2 cs[2] = {{cs4.x, cs4.y}, {cs4.z, cs4.w}};
    const f32x2 cc[2] = {{cc4.x, cc4.y}, {cc4.z, cc4.w}};
    const uint32_t pitch = (uint32_t)a.src.pitch;  // < 2^24 and rows < 2^15: 24-bit multiplies, 32-bit byte offsets
    // K*R^T*ray for two pixels per instruction (v_pk_mul_f32 / v_pk_add_f32; no contraction: OpenCV's operation order)
    f32x2 X[2], Y[2], Z[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 rx = ra * cs[h], rz = ra * cc[h];
        X[h] = (a.kr[0] * rx + c1) + a.kr[2] * rz;
        Y[h] = (a.kr[3] * rx + c4) + a.kr[5] * rz;
        Z[h] = (a.kr[6] * rx + c7) + a.kr[8] * rz;
    }
    // operands in the range where the shared-reciprocal division is exact?  (wave-uniform decision)
    const float zlo = fminf(fminf(Z[0].x, Z[0].y), fminf(Z[1].x, Z[1].y));
    float big = fmaxf(fmaxf(Z[0].x, Z[0].y), fmaxf(Z[1].x, Z[1].y));
#pragma unroll
    for (int h = 0; h < 2; ++h) big = fmaxf(big, fmaxf(fmaxf(fabsf(X[h].x), fabsf(X[h].y)), fmaxf(fabsf(Y[h].x), fabsf(Y[h].y))));
    const bool plain_div = GEN || !(zlo > 8.6736174e-19f && big < 1.1529215e18f);  // 2^-60, 2^60; also true for NaN
    f32x2 QX[2], QY[2];
    const bool fast_div = !GEN && __ballot(plain_div) == 0ULL;  // wave-uniform
    int ix[4], iy[4];
    uint32_t ax[4], ay[4];
    uint32_t mk = 0xffffffffu;
    bool all_in = true;
    // fast-path window: the two aligned 12-byte row reads [floor4(3*ix), +12) stay inside the row
    const uint32_t wlim = (uint32_t)max(a.src.w - 3, 0), hlim = (uint32_t)(a.src.h - 1);
    if (fast_div) {
#pragma unroll
        for (int h = 0; h < 2; ++h) div2_exact(X[h], Y[h], Z[h], QX[h], QY[h]);
        // interior candidate: every z > 0 here, and a lane is only accepted with 0 <= ix < w-3, 0 <= iy < h-1 -- in that range
        // cvRound needs no overflow guard, the int16 saturation is the identity and the nearest-neighbour mask test is true
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float qx = (i & 1) ? QX[i >> 1].y : QX[i >> 1].x, qy = (i & 1) ? QY[i >> 1].y : QY[i >> 1].x;
            const int isx = (int)__builtin_rintf(qx * 32.f), isy = (int)__builtin_rintf(qy * 32.f);
            ix[i] = isx >> 5;
            iy[i] = isy >> 5;
            ax[i] = isx & 31;
            ay[i] = isy & 31;
            all_in = all_in && (uint32_t)ix[i] < wlim && (uint32_t)iy[i] < hlim;
        }
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            QX[h].x = X[h].x / Z[h].x; QX[h].y = X[h].y / Z[h].y;
            QY[h].x = Y[h].x / Z[h].x; QY[h].y = Y[h].y / Z[h].y;
        }
    }
    const bool interior = fast_div && __ballot(!all_in) == 0ULL;  // wave-uniform
    if (!interior) {
        // the general quantisation: z <= 0 -> (-1, -1), cvRound with its overflow value, int16 saturation, and the
        // INTER_NEAREST + BORDER_CONSTANT test on the all-255 mask without converting: cvRound(f) in [0, n-1] <=> -0.5 <= f <= n-0.5
        // (upper bound exclusive when n is even: the tie n-0.5 rounds to the even neighbour n)
        const float hix = a.hix, hiy = a.hiy;
        mk = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float z = (i & 1) ? Z[i >> 1].y : Z[i >> 1].x;
            const float qx = (i & 1) ? QX[i >> 1].y : QX[i >> 1].x, qy = (i & 1) ? QY[i >> 1].y : QY[i >> 1].x;
            float fx = z > 0 ? qx : -1.f, fy = z > 0 ? qy : -1.f;
            if (GEN) map_backward(*gp, (float)(min(max(x0 + i, 0), a.dw - 1) + tlx), (float)(y + tly), fx, fy);
            const int isx = cv_round_fast(fx * 32.f), isy = cv_round_fast(fy * 32.f);
            ix[i] = sat_s16(isx >> 5);
            iy[i] = sat_s16(isy >> 5);
            ax[i] = isx & 31;
            ay[i] = isy & 31;
            if (fx >= -0.5f && fx <= hix && fy >= -0.5f && fy <= hiy) mk |= 0xffu << (8 * i);
        }
    }
    uint32_t px[4];
    if (interior) {
        // wave-uniform fast path: all 8 gathers of the lane are issued before the first use.  A gather is a 4-byte ALIGNED 12-byte
        // read that covers the 6 tap bytes, then v_alignbyte: a misaligned 8-byte read costs twice as much in the texture addresser
        // (tools/ta_microbench.hip: 33 vs 18 cycles per wave instruction)
        u32x3_a4 g0[4], g1[4];
        uint32_t o0[4], o1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t off = __umul24((uint32_t)iy[i], pitch) + __umul24((uint32_t)ix[i], 3u), off1 = off + pitch;
            o0[i] = off & 3u; o1[i] = off1 & 3u;
            g0[i] = *(const u32x3_a4 *)(a.src.data + (off & ~3u));
            g1[i] = *(const u32x3_a4 *)(a.src.data + (off1 & ~3u));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2_unaligned q0, q1;
            q0.x = __builtin_amdgcn_alignbyte(g0[i].y, g0[i].x, o0[i]); q0.y = __builtin_amdgcn_alignbyte(g0[i].z, g0[i].y, o0[i]);
            q1.x = __builtin_amdgcn_alignbyte(g1[i].y, g1[i].x, o1[i]); q1.y = __builtin_amdgcn_alignbyte(g1[i].z, g1[i].y, o1[i]);
            px[i] = blend_taps_dot(q0, q1, ax[i], ay[i]);
        }
    } else {
        // Waves that straddle the frame outline, or lie outside it (the warped roi is a bounding box).  With a mirroring border
        // (REPLICATE / REFLECT / REFLECT_101) at most one reflection away, the taps x, x+1 map to neighbouring or equal
        // columns: still two 8-byte row reads per pixel, the columns are put back in tap order with v_perm_b32.
        const int w = a.src.w, h = a.src.h;
        const int mul = a.border == SSP_BORDER_REPLICATE ? 0 : 1;
        const int xa = a.border == SSP_BORDER_REFLECT ? -1 : 0, ya = xa;
        const int xb = a.border == SSP_BORDER_REFLECT ? 2 * w - 1 : (a.border == SSP_BORDER_REFLECT_101 ? 2 * w - 2 : w - 1);
        const int yb = a.border == SSP_BORDER_REFLECT ? 2 * h - 1 : (a.border == SSP_BORDER_REFLECT_101 ? 2 * h - 2 : h - 1);
        const bool mirror_border = (a.border == SSP_BORDER_REPLICATE || a.border == SSP_BORDER_REFLECT || a.border == SSP_BORDER_REFLECT_101) && w >= 3;
        int x0[4], x1[4], y0[4], y1[4];
        bool single = mirror_border;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int xs = ix[i], xt = ix[i] + 1, ys = iy[i], yt = iy[i] + 1;
            x0[i] = xs < 0 ? xa - xs * mul : (xs >= w ? xb - xs * mul : xs);
            x1[i] = xt < 0 ? xa - xt * mul : (xt >= w ? xb - xt * mul : xt);
            y0[i] = ys < 0 ? ya - ys * mul : (ys >= h ? yb - ys * mul : ys);
            y1[i] = yt < 0 ? ya - yt * mul : (yt >= h ? yb - yt * mul : yt);
            single = single && (uint32_t)x0[i] < (uint32_t)w && (uint32_t)x1[i] < (uint32_t)w && (uint32_t)y0[i] < (uint32_t)h && (uint32_t)y1[i] < (uint32_t)h;
        }
        if (__ballot(!single) == 0ULL) {
            u32x2_unaligned q0[4], q1[4];
            uint32_t sh[4];
            const uint32_t last = (uint32_t)(3 * w - 8);  // the 8-byte read must end inside the row
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t b3 = __umul24((uint32_t)min(x0[i], x1[i]), 3u), ob = min(b3, last);
                sh[i] = 8u * (b3 - ob);
                q0[i] = *(const u32x2_unaligned *)(a.src.data + __umul24((uint32_t)y0[i], pitch) + ob);
                q1[i] = *(const u32x2_unaligned *)(a.src.data + __umul24((uint32_t)y1[i], pitch) + ob);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint64_t r0 = (((uint64_t)q0[i].y << 32) | q0[i].x) >> sh[i], r1 = (((uint64_t)q1[i].y << 32) | q1[i].x) >> sh[i];
                // bytes 0-2: column min(x0, x1), bytes 3-5: the next column.  Tap order: (x0, x1)
                const bool same = x0[i] == x1[i], swapped = x1[i] < x0[i];
                const uint32_t selx = swapped ? 0x00050403u : (same ? 0x00020100u : 0x03020100u);
                const uint32_t sely = (swapped || same) ? 0x0c0c0201u : 0x0c0c0504u;
                u32x2_unaligned t0, t1;
                t0.x = __builtin_amdgcn_perm((uint32_t)(r0 >> 32), (uint32_t)r0, selx);
                t0.y = __builtin_amdgcn_perm((uint32_t)(r0 >> 32), (uint32_t)r0, sely);
                t1.x = __builtin_amdgcn_perm((uint32_t)(r1 >> 32), (uint32_t)r1, selx);
                t1.y = __builtin_amdgcn_perm((uint32_t)(r1 >> 32), (uint32_t)r1, sely);
                px[i] = blend_taps_dot(t0, t1, ax[i], ay[i]);
            }
        } else {
#pragma unroll 1
            for (int i = 0; i < 4; ++i) px[i] = bilinear_u8c3_at(a.src, ix[i], iy[i], ax[i], ay[i], a.border);
        }
    }
    if (GAIN && ga->kind) {
        if (ga->kind == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) px[i] = gain_apply_px(px[i], ga->g[0], ga->g[1], ga->g[2]);
        } else {
            const int gw = ga->gw, gcn = ga->gcn;
            const int y0 = ga->yi[y], y1 = min(y0 + 1, ga->gh - 1);
            const float b1 = ga->yb[y], b0 = 1.f - b1;
            const int4 xi4 = *(const int4 *)(ga->xi + t0);
            const float4 xa4 = *(const float4 *)(ga->xa + t0);
            const int xs[4] = {xi4.x, xi4.y, xi4.z, xi4.w};
            const float as[4] = {xa4.x, xa4.y, xa4.z, xa4.w};
            const float *r0 = ga->gm + (size_t)y0 * gw * gcn, *r1 = ga->gm + (size_t)y1 * gw * gcn;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int x0g = xs[i], x1g = min(x0g + 1, gw - 1);
                const float a1 = as[i], a0 = 1.f - a1;
                float gg[3];
                for (int c = 0; c < gcn; ++c) {
                    const float t0g = r0[x0g * gcn + c] * a0 + r0[x1g * gcn + c] * a1;
                    const float t1g = r1[x0g * gcn + c] * a0 + r1[x1g * gcn + c] * a1;
                    gg[c] = t0g * b0 + t1g * b1;
                }
                if (gcn == 1) gg[1] = gg[2] = gg[0];
                px[i] = gain_apply_px(px[i], gg[0], gg[1], gg[2]);
            }
        }
    }
    const MaskPrep *mp = &mpv;
    // inside the seam mask the prepared mask is 255 whatever the coefficients: one scalar flag per wave (LX == 64) instead of the loads
    const bool seam_inside = LX == 64 && prep && mp->flags && mp->flags[(size_t)mp->yo[y] * mp->fgx + bx] != 0;
    if (prep && mk && !seam_inside) mk &= seam_mask4(*mp, y, t0);
    uint8_t *d = a.dst + (ptrdiff_t)y * (ptrdiff_t)a.dpitch + (ptrdiff_t)x0 * 3;
    if (full) {
        // 12 bytes: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3   (4-byte aligned: plane rows are 16-byte aligned and x0 + xshift is a multiple of 4 columns from one)
        u32x3_a4 w;
        w.x = (px[0] & 0xffffffu) | (px[1] << 24);
        w.y = ((px[1] >> 8) & 0xffffu) | (px[2] << 16);
        w.z = ((px[2] >> 16) & 0xffu) | (px[3] << 8);
        *(u32x3_a4 *)d = w;
        if (a.mask) *(uint32_t *)(a.mask + (ptrdiff_t)y * (ptrdiff_t)a.mpitch + x0) = mk;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (x0 + i < 0 || x0 + i >= a.dw) continue;
            d[3 * i] = (uint8_t)px[i];
            d[3 * i + 1] = (uint8_t)(px[i] >> 8);
            d[3 * i + 2] = (uint8_t)(px[i] >> 16);
            if (a.mask) a.mask[(ptrdiff_t)y * (ptrdiff_t)a.mpitch + x0 + i] = (uint8_t)(mk >> (8 * i));
        }
    }
}

// ---- float frames (BASELINE config 5): the same separable map, float bilinear in OpenCV's operation order ---------------------------
// remap INTER_LINEAR on CV_32FC3: quantised coordinates as for 8-bit, weights (1 - fx/32)(1 - fy/32) ..., sum = ((a w00 + b w01) + d w10)
// + e w11 without contraction.  A lane produces 4 pixels; a pixel whose four taps lie inside the frame reads 2 x 24 bytes, any other
// goes through remap_pixel (border rules).  dst is float3 per pixel, mask as in the 8-bit kernel.
typedef float f32x4_w __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2_w __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x3_w __attribute__((ext_vector_type(3), aligned(4)));
// Lane mapping: a wave covers 256 consecutive columns of one row, lane L takes columns L, L+64, L+128, L+192.  Consecutive lanes thus
// read source windows 12 bytes apart and store 12 bytes apart: every vector-memory instruction touches ~768 contiguous bytes (7 cache
// lines) instead of 64 windows 48 bytes apart (24 lines) -- the texture-address path, not HBM, bounded the 4-adjacent-pixels form.
__global__ __launch_bounds__(256) void k_warp_sep_f32c3(SepArgs a)
{
    const int lane = threadIdx.x & 63;
    const int y = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * 4 + (threadIdx.x >> 6)));
    const int xb = blockIdx.x * 256 + lane;
    if (y >= a.dh || blockIdx.x * 256 >= (unsigned)a.dw) return;
    const float ra = a.rowA[y], rb = a.rowB[y];
    const float c1 = a.kr[1] * rb, c4 = a.kr[4] * rb, c7 = a.kr[7] * rb;
    const float hix = a.hix, hiy = a.hiy;
    float *drow = (float *)((char *)a.dst + (size_t)y * a.dpitch);
    uint8_t *mrow = a.mask ? a.mask + (size_t)y * a.mpitch : nullptr;
    float fxs[4], fys[4];
    int isxs[4], isys[4];
    bool live[4], valid[4];
    bool inner = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = xb + 64 * i;
        live[i] = x < a.dw;
        const int xc = live[i] ? x : a.dw - 1;      // the tables hold dw entries
        const float cs = a.colS[xc], cc = a.colC[xc];
        const float rx = ra * cs, rz = ra * cc;
        const float X = (a.kr[0] * rx + c1) + a.kr[2] * rz, Y = (a.kr[3] * rx + c4) + a.kr[5] * rz, Z = (a.kr[6] * rx + c7) + a.kr[8] * rz;
        const float fx = Z > 0 ? X / Z : -1.f, fy = Z > 0 ? Y / Z : -1.f;
        valid[i] = fx >= -0.5f && fx <= hix && fy >= -0.5f && fy <= hiy;
        fxs[i] = fx; fys[i] = fy;
        isxs[i] = cv_round(fx * 32.f); isys[i] = cv_round(fy * 32.f);
        const int ix = sat_s16(isxs[i] >> 5), iy = sat_s16(isys[i] >> 5);
        inner = inner && live[i] && (unsigned)ix < (unsigned)(a.src.w - 1) && (unsigned)iy < (unsigned)(a.src.h - 1);
    }
    if (__all(inner)) {
        // the whole wave is inside the frame: no branches between the 16 gathers of a lane, so they are all in flight together
        f32x4_w r0a[4], r1a[4];
        f32x2_w r0b[4], r1b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint8_t *p = a.src.data + (size_t)(isys[i] >> 5) * a.src.pitch + (size_t)(isxs[i] >> 5) * 12;
            r0a[i] = *(const f32x4_w *)p;
            r0b[i] = *(const f32x2_w *)(p + 16);
            r1a[i] = *(const f32x4_w *)(p + a.src.pitch);
            r1b[i] = *(const f32x2_w *)(p + a.src.pitch + 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int axi = isxs[i] & 31, ayi = isys[i] & 31;
            const float vx1 = (float)axi * (1.f / 32), vx0 = 1.f - vx1, vy1 = (float)ayi * (1.f / 32), vy0 = 1.f - vy1;
            const float w00 = vy0 * vx0, w01 = vy0 * vx1, w10 = vy1 * vx0, w11 = vy1 * vx1;
            const float A[3] = {r0a[i].x, r0a[i].y, r0a[i].z}, B[3] = {r0a[i].w, r0b[i].x, r0b[i].y}, D[3] = {r1a[i].x, r1a[i].y, r1a[i].z},
                        E[3] = {r1a[i].w, r1b[i].x, r1b[i].y};
            float o[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float t = A[c] * w00 + B[c] * w01;
                t = t + D[c] * w10;
                t = t + E[c] * w11;
                o[c] = t;
            }
            const int x = xb + 64 * i;
            const f32x3_w ov = {o[0], o[1], o[2]};
            *(f32x3_w *)(drow + (size_t)x * 3) = ov;
            if (mrow) mrow[x] = valid[i] ? 255 : 0;
        }
        return;
    }
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        if (!live[i]) continue;
        const float fx = fxs[i], fy = fys[i];
        const int isx = isxs[i], isy = isys[i];
        const int ix = sat_s16(isx >> 5), iy = sat_s16(isy >> 5);
        float o[3];
        if ((unsigned)ix < (unsigned)(a.src.w - 1) && (unsigned)iy < (unsigned)(a.src.h - 1)) {
            const int axi = isx & 31, ayi = isy & 31;
            const float vx1 = (float)axi * (1.f / 32), vx0 = 1.f - vx1, vy1 = (float)ayi * (1.f / 32), vy0 = 1.f - vy1;
            const float w00 = vy0 * vx0, w01 = vy0 * vx1, w10 = vy1 * vx0, w11 = vy1 * vx1;
            const uint8_t *p = a.src.data + (size_t)iy * a.src.pitch + (size_t)ix * 12;
            const f32x4_w r0a = *(const f32x4_w *)p;
            const f32x2_w r0b = *(const f32x2_w *)(p + 16);
            const f32x4_w r1a = *(const f32x4_w *)(p + a.src.pitch);
            const f32x2_w r1b = *(const f32x2_w *)(p + a.src.pitch + 16);
            const float A[3] = {r0a.x, r0a.y, r0a.z}, B[3] = {r0a.w, r0b.x, r0b.y}, D[3] = {r1a.x, r1a.y, r1a.z}, E[3] = {r1a.w, r1b.x, r1b.y};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float t = A[c] * w00 + B[c] * w01;
                t = t + D[c] * w10;
                t = t + E[c] * w11;
                o[c] = t;
            }
        } else {
            remap_pixel<float, 3>(a.src, fx, fy, SSP_INTER_LINEAR, a.border, o);
        }
        const int x = xb + 64 * i;
        drow[(size_t)x * 3] = o[0]; drow[(size_t)x * 3 + 1] = o[1]; drow[(size_t)x * 3 + 2] = o[2];
        if (mrow) mrow[x] = valid[i] ? 255 : 0;
    }
}

__global__ __launch_bounds__(256) void k_warp_sep_u8c3(SepArgs a) { warp_sep_body<64>(a, false, MaskPrep(), blockIdx.x, blockIdx.y); }

// ---- batched form: all frames of a panorama in two launches (tables/mask-prep inputs, then the fused warp) ------------
// One descriptor per PART: a frame, or one of the two live column ranges of a frame whose roi spans the full circle (a frame that
// straddles u = +-pi*scale; ssp_composer.hip).  Kept compact -- 32-bit pitches, table pointers derived from one base -- so that
// WARP_MAXB of them fit the kernel-argument segment.
struct WarpBatchCore {
    const uint8_t *sdata; uint8_t *dst; uint8_t *mask;   // source frame (u8c3), destination planes (u8c3, u8 or null)
    uint32_t spitch, dpitch, mpitch;
    int sw, sh, dw, dh;
    float kr[9];
    int border;
    float hix, hiy;  // largest source coordinate whose cvRound is still inside the frame (see nearest_hi)
    int xshift;      // as SepArgs::xshift
};
struct WarpBatchGain {                 // exposure compensation (kind 0: none); the coordinate tables are filled by k_warp_prep_batch
    int kind; float g[3];
    const float *gm; int gw, gh, gcn; int pad_;
    int *tabs;                         // xi (dw4 ints) | xa (dw4 floats) | yi (dh ints) | yb (dh floats)
};
struct WarpBatchDesc {
    WarpBatchCore a;
    WarpBatchGain gain;
    int kind; float scale; int tlx, tly, dw4;
    int full_dw, x_off;         // width of the frame's whole roi and the part's first column in it: the resize tables of the mask preparation and of
                                // the gain map are those of the WHOLE warped frame (cv.resize to the roi's size), read at columns x_off ...
    int prep;                   // 1: mask preparation fused
    float *tab;                 // colS | colC (dw4 each) | rowA | rowB (dh each)
    const uint8_t *seam; uint8_t *dil;                              // seam-scale warped mask (sde.py:1591-1599) and its 3x3 dilation (sde.py:1760)
    uint32_t seam_pitch, dil_pitch; int seam_w, seam_h;
    int *lin;                   // xo | xc (dw4 each) | yo | yc (dh each) | seam-interior flags per (seam row, 256-column segment), filled by k_warp_prep_batch
    int fgx, has_flags;
    int4 *tiles;                // two int4 per 64 x 16 output tile: the strip kernel's records, filled by k_warp_records_batch with the prep launch
    uint32_t *cmap;             // COORDINATE PLANE of the part (null: the map comes from the separable tables): WB_CMAP_HEAD words holding the part's Projector,
                                // then one word per pixel of every tile row (pitch warp_tiles_x(dw) * 64): the quantised map relative to the tile's tap origin,
                                // written once per geometry by k_warp_cmap_batch (see there)
};
#define WB_CMAP_HEAD 64
static_assert(sizeof(Projector) <= 4 * WB_CMAP_HEAD, "the projector must fit the head of a coordinate plane");
__host__ __device__ inline const Projector &wb_proj(const WarpBatchDesc &d) { return *(const Projector *)d.cmap; }
__host__ __device__ inline int *wb_flags(const WarpBatchDesc &d) { return d.has_flags ? d.lin + 2 * ((size_t)d.dw4 + d.a.dh) : nullptr; }
__host__ __device__ inline const int *wb_gxi(const WarpBatchDesc &d) { return d.gain.tabs; }
__host__ __device__ inline const float *wb_gxa(const WarpBatchDesc &d) { return (const float *)(d.gain.tabs + d.dw4); }
__host__ __device__ inline const int *wb_gyi(const WarpBatchDesc &d) { return d.gain.tabs + 2 * (size_t)d.dw4; }
__host__ __device__ inline const float *wb_gyb(const WarpBatchDesc &d) { return (const float *)(d.gain.tabs + 2 * (size_t)d.dw4 + d.a.dh); }
__device__ inline SrcView wb_src(const WarpBatchCore &a) { SrcView s = {a.sdata, (size_t)a.spitch, a.sw, a.sh}; return s; }
// the single-frame kernels' argument forms of a descriptor (the rest kernel runs their gather body)
__device__ inline SepArgs wb_sep(const WarpBatchDesc &d)
{
    SepArgs s;
    s.src = wb_src(d.a);
    s.dst = d.a.dst; s.dpitch = d.a.dpitch; s.mask = d.a.mask; s.mpitch = d.a.mpitch;
    s.dw = d.a.dw; s.dh = d.a.dh;
    s.colS = d.tab; s.colC = d.tab + d.dw4; s.rowA = d.tab + 2 * (size_t)d.dw4; s.rowB = s.rowA + d.a.dh;
#pragma unroll
    for (int q = 0; q < 9; ++q) s.kr[q] = d.a.kr[q];
    s.border = d.a.border; s.hix = d.a.hix; s.hiy = d.a.hiy; s.xshift = d.a.xshift;
    return s;
}
__device__ inline GainArgs wb_gainargs(const WarpBatchDesc &d)
{
    GainArgs g;
    g.kind = d.gain.kind; g.g[0] = d.gain.g[0]; g.g[1] = d.gain.g[1]; g.g[2] = d.gain.g[2];
    g.gm = d.gain.gm; g.gw = d.gain.gw; g.gh = d.gain.gh; g.gcn = d.gain.gcn;
    g.xi = wb_gxi(d); g.xa = wb_gxa(d); g.yi = wb_gyi(d); g.yb = wb_gyb(d);
    return g;
}

__device__ inline void lin_exact_entry(int ssize, int dsize, int d, int &ofs, int &coef)
{
    double scale = 1.0 / ((double)dsize / (double)ssize);
    double fval = scale * ((double)d + 0.5) - 0.5;
    int ival = (int)floor(fval);
    if (ival >= 0 && ssize > 1) {
        if (ival < ssize - 1) { ofs = ival; coef = (int)rint((fval - (double)ival) * 256.0); }
        else { ofs = ssize - 1; coef = -1; }
    } else { ofs = 0; coef = -1; }
}

// up to WARP_MAXB frames per launch: the descriptors travel in the kernel-argument segment, so every field is a scalar
// load and every pointer is known to be global memory (no FLAT accesses, no per-lane loads of uniform data).  15 descriptors
// are 4 KB of kernel arguments, which the runtime takes; BASELINE config 3's 12 frames -- 14 parts when the ring is closed -- are
// then one launch (A/B on one box, 8 + 4 against 12: step 0.978 -> 0.952 ms: one prep and one rest launch less, one kernel tail less)
#ifndef WARP_MAXB
#define WARP_MAXB 15
#endif
struct WarpBatchArgs {
    WarpBatchDesc d[WARP_MAXB];
    int *rest;      // LDS-staged variant: rest[0] = number of tiles that were not staged, rest[1 + i] = their linear index (z, by, bx); null: none
    int rest_cap;   // entries of the list; rest[1 + rest_cap] = 1 when some tile missed because its gain rows do not fit (never inlined)
    int rest_known; // 1: the list is the composer's stored one (static geometry): nobody zeroes or appends, the rest launch just walks it
};
static_assert(sizeof(WarpBatchArgs) <= 4096 - 48, "the warp descriptors (+ 48 bytes of scalar arguments) must fit the kernel-argument segment");

__global__ __launch_bounds__(256) void k_warp_prep_batch(const WarpBatchArgs args)
{
    const WarpBatchDesc &d = args.d[blockIdx.z];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (args.rest && !args.rest_known && i == 0 && blockIdx.z == 0) { args.rest[0] = 0; args.rest[1 + args.rest_cap] = 0; args.rest[2 + args.rest_cap] = 0; }   // the next launches append to the list / set the flags
    const int dw4 = d.dw4, dh = d.a.dh, dw = d.a.dw;
    // (1) trigonometry tables of the separable projection (none for a part that reads its map from a coordinate plane only)
    if (i < dw4 + dh) {
        if (!d.tab) return;
        float *colS = d.tab, *colC = d.tab + dw4, *rowA = d.tab + 2 * dw4, *rowB = rowA + dh;
        if (i < dw4) {
            float u = (float)(min(max(i - d.a.xshift, 0), dw - 1) + d.tlx);
            u /= d.scale;
            colS[i] = ssp_sinf(u);
            colC[i] = ssp_cosf(u);
        } else {
            int j = i - dw4;
            float v = (float)(j + d.tly);
            v /= d.scale;
            float av, bv;
            if (d.kind == PK_SPHERICAL) { av = ssp_sinf(SSP_PI_F - v); bv = ssp_cosf(SSP_PI_F - v); }
            else if (d.kind == PK_CYLINDRICAL) { av = 1.0f; bv = v; }
            else { float lat = ssp_atanf(ssp_sinhf(v)); av = ssp_cosf(lat); bv = ssp_sinf(lat); }
            rowA[j] = av;
            rowB[j] = bv;
        }
        return;
    }
    i -= dw4 + dh;
    // (1b) gain-map resize coordinates
    if (d.gain.kind == 2) {
        if (i < dw4 + dh) {
            int s0; float f;
            if (i < dw4) { gain_lin_coord(min(max(i - d.a.xshift, 0), dw - 1) + d.x_off, d.gain.gw, d.full_dw, s0, f); ((int *)wb_gxi(d))[i] = s0; ((float *)wb_gxa(d))[i] = f; }
            else { gain_lin_coord(i - dw4, d.gain.gh, dh, s0, f); ((int *)wb_gyi(d))[i - dw4] = s0; ((float *)wb_gyb(d))[i - dw4] = f; }
            return;
        }
        i -= dw4 + dh;
    }
    if (!d.prep) return;
    // (2) INTER_LINEAR_EXACT tables seam size -> warped size
    if (i < dw4 + dh) {
        int *xo = d.lin, *xc = d.lin + dw4, *yo = d.lin + 2 * dw4, *yc = yo + dh;
        int o, c;
        if (i < dw4) { lin_exact_entry(d.seam_w, d.full_dw, min(max(i - d.a.xshift, 0), dw - 1) + d.x_off, o, c); xo[i] = o; xc[i] = c; }
        else { lin_exact_entry(d.seam_h, dh, i - dw4, o, c); yo[i - dw4] = o; yc[i - dw4] = c; }
        return;
    }
    i -= dw4 + dh;
    // (3) cv.dilate(seam mask, None)
    if (i < d.seam_w * d.seam_h) {
        int y = i / d.seam_w, x = i - y * d.seam_w, m = 0;
        for (int dy = -1; dy <= 1; ++dy) {
            int yy = y + dy;
            if (yy < 0 || yy >= d.seam_h) continue;
            const uint8_t *r = d.seam + (size_t)yy * d.seam_pitch;
            for (int dx = -1; dx <= 1; ++dx) {
                int xx = x + dx;
                if (xx < 0 || xx >= d.seam_w) continue;
                m = max(m, (int)r[xx]);
            }
        }
        d.dil[(size_t)y * d.dil_pitch + x] = (uint8_t)m;
        return;
    }
    i -= d.seam_w * d.seam_h;
    // (4) seam-interior flags: the samples a 256-column segment of a row interpolates lie in seam rows yo..yo+1, columns xo(first)..xo(last)+1
    // of the DILATED mask; where the undilated mask is 255 the dilated one is too, so the undilated window being all 255 suffices
    if (d.has_flags && i < d.seam_h * d.fgx) {
        const int yo = i / d.fgx, seg = i - yo * d.fgx;   // all warped rows that interpolate from seam rows yo, yo+1 share the flag
        int xa, xb, c;
        lin_exact_entry(d.seam_w, d.full_dw, min(max(seg * 256 - d.a.xshift, 0), dw - 1) + d.x_off, xa, c);
        lin_exact_entry(d.seam_w, d.full_dw, min(max(seg * 256 + 255 - d.a.xshift, 0), dw - 1) + d.x_off, xb, c);
        const int y1 = min(yo + 1, d.seam_h - 1), x1 = min(xb + 1, d.seam_w - 1);
        uint32_t all = 0xffffffffu;   // AND of every sample of the window, four at a time (no early exit: the loads stay independent)
        for (int yy = yo; yy <= y1; ++yy) {
            const uint8_t *r = d.seam + (size_t)yy * d.seam_pitch;
            int xx = xa;
            for (; xx + 3 <= x1; xx += 4) all &= *(const u32_u1 *)(r + xx);
            for (; xx <= x1; ++xx) all &= 0xffffff00u | r[xx];
        }
        wb_flags(d)[i] = all == 0xffffffffu;
    }
}

// ---- float frames, batched (BASELINE config 5): every part of a panorama in ONE launch, image + prepared mask ---------------------------------
// k_warp_sep_f32c3's body per part (separable map from the part's tables, float bilinear in OpenCV's operation order, gathers straight from
// HBM: the LDS-staged form was measured 1.5x slower for 12-byte taps, profiles/r03_cfg5_warp_variants.txt) with the mask preparation of
// sde.py:1760-1772 in its epilogue -- the dilated seam-scale mask, resized with INTER_LINEAR_EXACT to the warped size, AND-ed with the validity
// mask -- from the tables k_warp_prep_batch leaves (rounds 1-3: per frame a table launch, a warp launch, a dilation and a resize+and pass).
__device__ inline uint32_t seam_mask1(const WarpBatchDesc &d, int y, int t)
{
    const int dw4 = d.dw4;
    const int *xo = d.lin, *xc = d.lin + dw4, *yo = d.lin + 2 * dw4, *yc = yo + d.a.dh;
    const int o = xo[t], c = xc[t], cyv = yc[y];
    const uint8_t *r0 = d.dil + (size_t)yo[y] * d.dil_pitch;
    const uint8_t *r1 = cyv >= 0 ? r0 + d.dil_pitch : r0;
    const uint32_t cy1 = cyv >= 0 ? (uint32_t)cyv : 0u, cx1 = c >= 0 ? (uint32_t)c : 0u;
    const uint32_t a0 = r0[o], a1 = cx1 ? r0[o + 1] : 0u, b0 = r1[o], b1 = cx1 ? r1[o + 1] : 0u;
    const uint32_t h0 = a0 * (256u - cx1) + a1 * cx1, h1 = b0 * (256u - cx1) + b1 * cx1;
    return (h0 * (256u - cy1) + h1 * cy1 + (1u << 15)) >> 16;
}
__global__ __launch_bounds__(256) void k_warp_f32_batch(const WarpBatchArgs args)
{
    const WarpBatchDesc &d = args.d[blockIdx.z];
    const WarpBatchCore &a = d.a;
    const int lane = threadIdx.x & 63;
    const int y = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * 4 + (threadIdx.x >> 6)));
    const int xb = blockIdx.x * 256 + lane;
    if (y >= a.dh || blockIdx.x * 256 >= (unsigned)a.dw) return;
    const float *colS = d.tab, *colC = d.tab + d.dw4, *rowA = d.tab + 2 * (size_t)d.dw4, *rowB = rowA + a.dh;
    const float ra = rowA[y], rb = rowB[y];
    const float c1 = a.kr[1] * rb, c4 = a.kr[4] * rb, c7 = a.kr[7] * rb;
    const float hix = a.hix, hiy = a.hiy;
    const SrcView src = wb_src(a);
    float *drow = (float *)((char *)a.dst + (size_t)y * a.dpitch);
    uint8_t *mrow = a.mask + (size_t)y * a.mpitch;
    // inside the seam mask the prepared mask is 255 whatever the coefficients: one flag per (seam row, 256-column segment)
    const bool seam_in = !d.prep || (d.has_flags && wb_flags(d)[(size_t)d.lin[2 * d.dw4 + y] * d.fgx + blockIdx.x] != 0);
    float fxs[4], fys[4];
    int isxs[4], isys[4];
    bool live[4], valid[4];
    bool inner = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = xb + 64 * i;
        live[i] = x < a.dw;
        const int xc = live[i] ? x : a.dw - 1;
        const float cs = colS[xc], cc = colC[xc];
        const float rx = ra * cs, rz = ra * cc;
        const float X = (a.kr[0] * rx + c1) + a.kr[2] * rz, Y = (a.kr[3] * rx + c4) + a.kr[5] * rz, Z = (a.kr[6] * rx + c7) + a.kr[8] * rz;
        const float fx = Z > 0 ? X / Z : -1.f, fy = Z > 0 ? Y / Z : -1.f;
        valid[i] = fx >= -0.5f && fx <= hix && fy >= -0.5f && fy <= hiy;
        fxs[i] = fx; fys[i] = fy;
        isxs[i] = cv_round(fx * 32.f); isys[i] = cv_round(fy * 32.f);
        const int ix = sat_s16(isxs[i] >> 5), iy = sat_s16(isys[i] >> 5);
        inner = inner && live[i] && (unsigned)ix < (unsigned)(a.sw - 1) && (unsigned)iy < (unsigned)(a.sh - 1);
    }
    const bool all_inner = __all(inner);
    f32x4_w r0a[4], r1a[4];
    f32x2_w r0b[4], r1b[4];
    if (all_inner) {
        // the whole wave is inside the frame: no branches between the 16 gathers of a lane, so they are all in flight together
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint8_t *p = a.sdata + (size_t)(isys[i] >> 5) * a.spitch + (size_t)(isxs[i] >> 5) * 12;
            r0a[i] = *(const f32x4_w *)p;
            r0b[i] = *(const f32x2_w *)(p + 16);
            r1a[i] = *(const f32x4_w *)(p + a.spitch);
            r1b[i] = *(const f32x2_w *)(p + a.spitch + 16);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!live[i]) continue;
        const int x = xb + 64 * i;
        const int isx = isxs[i], isy = isys[i];
        const int ix = sat_s16(isx >> 5), iy = sat_s16(isy >> 5);
        float o[3];
        const bool in = all_inner || ((unsigned)ix < (unsigned)(a.sw - 1) && (unsigned)iy < (unsigned)(a.sh - 1));
        if (in) {
            f32x4_w q0a, q1a; f32x2_w q0b, q1b;
            if (all_inner) { q0a = r0a[i]; q0b = r0b[i]; q1a = r1a[i]; q1b = r1b[i]; }
            else {
                const uint8_t *p = a.sdata + (size_t)iy * a.spitch + (size_t)ix * 12;
                q0a = *(const f32x4_w *)p; q0b = *(const f32x2_w *)(p + 16); q1a = *(const f32x4_w *)(p + a.spitch); q1b = *(const f32x2_w *)(p + a.spitch + 16);
            }
            const int axi = isx & 31, ayi = isy & 31;
            const float vx1 = (float)axi * (1.f / 32), vx0 = 1.f - vx1, vy1 = (float)ayi * (1.f / 32), vy0 = 1.f - vy1;
            const float w00 = vy0 * vx0, w01 = vy0 * vx1, w10 = vy1 * vx0, w11 = vy1 * vx1;
            const float A[3] = {q0a.x, q0a.y, q0a.z}, B[3] = {q0a.w, q0b.x, q0b.y}, D[3] = {q1a.x, q1a.y, q1a.z}, E[3] = {q1a.w, q1b.x, q1b.y};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float t = A[c] * w00 + B[c] * w01;
                t = t + D[c] * w10;
                t = t + E[c] * w11;
                o[c] = t;
            }
        } else {
            remap_pixel<float, 3>(src, fxs[i], fys[i], SSP_INTER_LINEAR, a.border, o);
        }
        const f32x3_w ov = {o[0], o[1], o[2]};
        *(f32x3_w *)(drow + (size_t)x * 3) = ov;
        uint32_t m = valid[i] ? 255u : 0u;
        if (m && !seam_in) m &= seam_mask1(d, y, x);
        mrow[x] = (uint8_t)m;
    }
}

// ---- tile geometry of the LDS-staged warp (k_warp_strip_batch below) and of its rest kernel: 64 x 16 output pixels per tile, 16 lanes x 4 pixels
// per row.  (The two earlier LDS forms -- one tile per work-group with a separate tile-record launch, and the round-1 gather batch -- lost
// against the strip form and are gone from the library; their numbers are kept in profiles/r02_warp_variants.txt.)
#define WT_W 64
#define WT_H 16
#define WT_GAIN_ROWS 4

__host__ __device__ inline int warp_tiles_x(int dw) { return (dw + 3 + WT_W - 1) / WT_W; }
__host__ __device__ inline int warp_tiles_y(int dh) { return (dh + WT_H - 1) / WT_H; }

// the tiles the strip kernel could not stage, through the gather body (same 64 x 16 tile shape): a fixed grid walks the list
template <bool GAIN>
__global__ __launch_bounds__(256) void k_warp_rest_batch(const WarpBatchArgs args, int gx, int gy, int n_tiles)
{
    const int n = min(args.rest[0], n_tiles), per_img = gx * gy;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int t = args.rest[1 + i];
        if (t < 0 || t >= n_tiles) continue;
        const int z = t / per_img, l = t - z * per_img, by = l / gx, bx = l - by * gx;
        const WarpBatchDesc &d = args.d[z];
        MaskPrep mp;
        mp.dil = d.dil; mp.dpitch = d.dil_pitch;
        mp.xo = d.lin; mp.xc = d.lin + d.dw4; mp.yo = d.lin + 2 * d.dw4; mp.yc = mp.yo + d.a.dh;
        mp.flags = nullptr; mp.fgx = 0;
        const SepArgs sa = wb_sep(d);
        const GainArgs gargs = wb_gainargs(d);
        if (d.cmap && !d.tab) warp_sep_body<16, GAIN, true>(sa, d.prep != 0, mp, bx, by, &gargs, &wb_proj(d), d.tlx, d.tly);     // no tables: the map per pixel
        else warp_sep_body<16, GAIN>(sa, d.prep != 0, mp, bx, by, &gargs);
    }
}

// bits of a float.  NOT __builtin_bit_cast(uint32_t, v.y) on a vector element: ROCm 7.2's clang reads element 0 for every element there
// (checked with a three-line probe kernel in round 2); through a by-value parameter the element is an ordinary scalar.
__device__ inline uint32_t fbits(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ inline float ufloat(uint32_t u) { return __builtin_bit_cast(float, u); }       // (the same the other way round: never bit_cast a vector ELEMENT)

// three 64*V + 32768 values of one pixel (V = the 2^10-scaled bilinear sum): byte 2 of each is the rounded 8-bit sample
struct Px3 { uint32_t b, g, r; };
__device__ inline Px3 blend_taps_v(uint32_t q0x, uint32_t q0y, uint32_t q1x, uint32_t q1y, uint32_t ax, uint32_t ay)
{
    const uint32_t wx = (32u - ax) | (ax << 24);                                   // weights for bytes 0 and 3
    const u16x2 wy = __builtin_bit_cast(u16x2, __umul24(ay, 4194240u) + 2048u);    // (2048 - 64 ay) | (64 ay) << 16
    const uint32_t g0 = __builtin_amdgcn_alignbit(q0y, q0x, 8), r0 = __builtin_amdgcn_alignbit(q0y, q0x, 16);
    const uint32_t g1 = __builtin_amdgcn_alignbit(q1y, q1x, 8), r1 = __builtin_amdgcn_alignbit(q1y, q1x, 16);
    const uint32_t hb0 = __builtin_amdgcn_udot4(q0x, wx, 0u, false), hb1 = __builtin_amdgcn_udot4(q1x, wx, 0u, false);
    const uint32_t hg0 = __builtin_amdgcn_udot4(g0, wx, 0u, false), hg1 = __builtin_amdgcn_udot4(g1, wx, 0u, false);
    const uint32_t hr0 = __builtin_amdgcn_udot4(r0, wx, 0u, false), hr1 = __builtin_amdgcn_udot4(r1, wx, 0u, false);
    Px3 o;
    o.b = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, hb0 | (hb1 << 16)), wy, 32768u, false);
    o.g = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, hg0 | (hg1 << 16)), wy, 32768u, false);
    o.r = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, hr0 | (hr1 << 16)), wy, 32768u, false);
    return o;
}

// saturate_cast<uchar>(cvRound(v)) packed into byte `pos` of `into`: v_cvt_pk_u8_f32 rounds to nearest even and saturates (NaN -> 0), checked
// against nearbyintf + clamp on 8 117 values incl. every tie (a probe kernel of round 2)
__device__ inline uint32_t pack_u8_rne(float v, uint32_t pos, uint32_t into)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(v, pos, into);
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// exact t / d for t * d < 2^32 and d > 1 with m = floor(2^32 / d) + 1; m = 0: the host could not guarantee that, plain division
__device__ inline uint32_t udiv_by_magic(uint32_t t, uint32_t d, uint32_t m) { return m ? __umulhi(t, m) : t / d; }


// ---- strip variant: four 64 x 16 tiles per work-group, source rectangles measured in the kernel, LDS-DMA double buffering -------------
// What the tile kernel above pays per tile -- a launch that measures the rectangles, a dependent global read of the record, the row
// values, the staging latency in front of the first tap -- is paid here once per 256 x 16 strip or hidden:
//   * set-up: the strip's column tables go to LDS; 36 lanes evaluate the map at a 3 x 3 grid of each of the 4 tiles, 4 lanes turn the
//     samples into the tiles' source rectangles (records in LDS); exposure compensation: the gain map's rows under the strip, resized
//     horizontally to the strip's 256 columns, go to LDS;
//   * per tile: `buffer_load_dwordx4 ... lds` (LDS-DMA: no registers, lane L of a wave lands at M0 + 16 L, out-of-range bytes arrive
//     as zeros) copies the NEXT tile's rectangle into the other LDS buffer while this tile's taps are interpolated -- one barrier per
//     tile.  The rectangle lies row-major in LDS with a pitch of exactly its own width in 16-byte chunks, so any shape of up to 640
//     chunks fits (the map magnifies towards the frame's sides: rectangles there are wider than 256 bytes);
//   * tiles whose taps leave the frame stay in LDS: the rectangle is the image of the tap range under BORDER_REFLECT (a fold at the
//     frame edge keeps it compact) and every tap goes through borderInterpolate before it is addressed (four separate pixels instead
//     of two 6-byte reads);
//   * what is left (rectangles beyond the buffers, pixels behind the camera, other border modes) goes to k_warp_rest_batch's list.
// Instruction count is what bounds these kernels (a gfx950 SIMD issues one VALU instruction per ~4 cycles whatever its kind, measured on
// every variant: profiles/r02_*), so the map, the two IEEE divisions and the quantisation run two pixels per instruction (v_pk_*_f32).
#ifndef WS_NT
#define WS_NT 4
#endif
#ifndef WS_BUF
#define WS_BUF 10240          // one staging buffer: 640 chunks of 16 bytes (608 would admit a 7th work-group per CU but sends 3x the tiles to the rest list: slower)
#endif
#define WS_BUF_CMAP 12288     // ... of the coordinate-plane variant: 768 chunks.  The projections it serves put frames at any angle (fisheye: around the zenith), and the
                              // source rectangle of a 64 x 16 tile turned by 45 degrees is 57 x 57 pixels = 60 rows x 12 chunks
#define WS_STAGE 1
#define WS_BORDER 2           // taps leave the frame: reflected addressing
#define WS_SKIP 8             // nothing of the tile lies inside the roi
#define WS_INLINE 16          // not stageable, and the plan says such tiles are rare: every lane takes the per-pixel general path right here

__device__ inline int reflect_idx(int v, int n) { return v < 0 ? -v - 1 : (v >= n ? 2 * n - 1 - v : v); }      // BORDER_REFLECT, |excursion| <= n

// two 6-byte reads of a pixel whose four taps lie inside the frame (4-byte aligned 12-byte windows) -> 64 V + 32768 per channel
__device__ inline Px3 taps_interior(const uint8_t *tile, uint32_t pitchl, uint32_t c0, uint32_t bx, uint32_t by)
{
    const uint32_t ixr = bx >> 5;
    // 24-bit multiply-adds on purpose: left to itself the compiler folds 3 * ix + c0 into a 64-bit v_mad_u64_u32 (quarter rate)
    const uint32_t ad = __umul24(by >> 5, pitchl) + (__umul24(ixr, 3u) + c0), o = ad & 3u;
    const uint32_t *pp = (const uint32_t *)(tile + (ad & ~3u)), *pq = (const uint32_t *)(tile + (ad & ~3u) + pitchl);
    const uint32_t w0 = pp[0], w1 = pp[1], w2 = pp[2], u0 = pq[0], u1 = pq[1], u2 = pq[2];
    return blend_taps_v(__builtin_amdgcn_alignbyte(w1, w0, o), __builtin_amdgcn_alignbyte(w2, w1, o), __builtin_amdgcn_alignbyte(u1, u0, o), __builtin_amdgcn_alignbyte(u2, u1, o),
                        bx & 31u, by & 31u);
}
// The same for a lane's four pixels with the LDS reads written out (ds_read2_b32 + ds_read_b32 per row and pixel, all sixteen in flight, one
// wait): for an LDS read that follows an LDS-DMA the compiler waits vmcnt(0) -- it cannot tell the addresses apart -- which would make every
// wave wait for the copy of the NEXT tile's rectangle that k_warp_strip_planes has just requested into another slot.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ inline void taps_interior4(uint32_t lds_tile, uint32_t pitchl, uint32_t c0, const uint32_t bx[4], const uint32_t by[4], Px3 v[4])
{
    uint32_t a0[4], o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t ad = __umul24(by[i] >> 5, pitchl) + (__umul24(bx[i] >> 5, 3u) + c0);
        o[i] = ad & 3u;
        a0[i] = lds_tile + (ad & ~3u);
    }
    u32x2_t w01[4], u01[4];
    uint32_t w2[4], u2[4];
    asm volatile("ds_read2_b32 %0, %16 offset1:1\n\tds_read_b32 %1, %16 offset:8\n\tds_read2_b32 %2, %17 offset1:1\n\tds_read_b32 %3, %17 offset:8\n\t"
                 "ds_read2_b32 %4, %18 offset1:1\n\tds_read_b32 %5, %18 offset:8\n\tds_read2_b32 %6, %19 offset1:1\n\tds_read_b32 %7, %19 offset:8\n\t"
                 "ds_read2_b32 %8, %20 offset1:1\n\tds_read_b32 %9, %20 offset:8\n\tds_read2_b32 %10, %21 offset1:1\n\tds_read_b32 %11, %21 offset:8\n\t"
                 "ds_read2_b32 %12, %22 offset1:1\n\tds_read_b32 %13, %22 offset:8\n\tds_read2_b32 %14, %23 offset1:1\n\tds_read_b32 %15, %23 offset:8\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(w01[0]), "=&v"(w2[0]), "=&v"(u01[0]), "=&v"(u2[0]), "=&v"(w01[1]), "=&v"(w2[1]), "=&v"(u01[1]), "=&v"(u2[1]),
                   "=&v"(w01[2]), "=&v"(w2[2]), "=&v"(u01[2]), "=&v"(u2[2]), "=&v"(w01[3]), "=&v"(w2[3]), "=&v"(u01[3]), "=&v"(u2[3])
                 : "v"(a0[0]), "v"(a0[0] + pitchl), "v"(a0[1]), "v"(a0[1] + pitchl), "v"(a0[2]), "v"(a0[2] + pitchl), "v"(a0[3]), "v"(a0[3] + pitchl));
#pragma unroll
    for (int i = 0; i < 4; ++i)
        v[i] = blend_taps_v(__builtin_amdgcn_alignbyte(w01[i].y, w01[i].x, o[i]), __builtin_amdgcn_alignbyte(w2[i], w01[i].y, o[i]), __builtin_amdgcn_alignbyte(u01[i].y, u01[i].x, o[i]),
                            __builtin_amdgcn_alignbyte(u2[i], u01[i].y, o[i]), bx[i] & 31u, by[i] & 31u);
}
// two float4 of LDS (the gain rows of a pixel group) without the compiler's vmcnt(0) in front (see taps_interior4)
typedef float asm_f32x4_t __attribute__((ext_vector_type(4)));
__device__ inline void lds_read_2x128(const float *p0, const float *p1, float4 &r0, float4 &r1)
{
    asm_f32x4_t t0, t1;
    const uint32_t a0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float *)p0, a1 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float *)p1;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t0), "=&v"(t1) : "v"(a0), "v"(a1));
    r0 = make_float4(t0.x, t0.y, t0.z, t0.w); r1 = make_float4(t1.x, t1.y, t1.z, t1.w);
}

// taps through borderInterpolate(BORDER_REFLECT), each on its own: four pixels of 3 bytes
__device__ inline Px3 taps_reflect(const uint8_t *tile, uint32_t pitchl, uint32_t c0, uint32_t bx, uint32_t by, int ux0, int uy0, int rx0, int ry0, int sw, int sh)
{
    const int ixu = ux0 + (int)(bx >> 5), iyu = uy0 + (int)(by >> 5);
    const uint32_t ca = 3u * (uint32_t)(reflect_idx(ixu, sw) - rx0) + c0, cb = 3u * (uint32_t)(reflect_idx(ixu + 1, sw) - rx0) + c0;
    const uint32_t ya = __umul24((uint32_t)(reflect_idx(iyu, sh) - ry0), pitchl), yb = __umul24((uint32_t)(reflect_idx(iyu + 1, sh) - ry0), pitchl);
    const uint32_t a00 = ya + ca, a01 = ya + cb, a10 = yb + ca, a11 = yb + cb;
    const uint32_t *p00 = (const uint32_t *)(tile + (a00 & ~3u)), *p01 = (const uint32_t *)(tile + (a01 & ~3u));
    const uint32_t *p10 = (const uint32_t *)(tile + (a10 & ~3u)), *p11 = (const uint32_t *)(tile + (a11 & ~3u));
    const uint32_t t00 = __builtin_amdgcn_alignbyte(p00[1], p00[0], a00 & 3u), t01 = __builtin_amdgcn_alignbyte(p01[1], p01[0], a01 & 3u);
    const uint32_t t10 = __builtin_amdgcn_alignbyte(p10[1], p10[0], a10 & 3u), t11 = __builtin_amdgcn_alignbyte(p11[1], p11[0], a11 & 3u);
    // [B0 G0 R0 .] [B1 G1 R1 .] -> the 6-byte layout of the interior form: B0 G0 R0 B1 | G1 R1 . .
    return blend_taps_v(__builtin_amdgcn_perm(t01, t00, 0x04020100u), __builtin_amdgcn_perm(t01, t01, 0x0c0c0201u), __builtin_amdgcn_perm(t11, t10, 0x04020100u),
                        __builtin_amdgcn_perm(t11, t11, 0x0c0c0201u), bx & 31u, by & 31u);
}

#define WS_GFIT 32            // record only: the gain rows under the tile's strip fit the LDS slice (else the tile can never be done inline)
#define WS_EMPTY 64           // record only: no pixel of the tile maps into the frame (all nine samples lie beyond one and the same side of it)
#define WS_FAR 128            // empty, and so is every tile within the blender's reach of it: only the mask (zeros) is written

// Tile records of the strip kernel, one lane per 64 x 16 tile: the source rectangle to stage (from nine samples of the map) and the tile's flags.
// Everything here is a function of the cameras, the rois and the gain-map SHAPE -- never of the frames -- so it runs with the prep launch on a
// composer's first panoramas only and the records stay in the frame's `tiles` buffer (two int4 per tile); the strip kernel just loads them.
// (Round 2 computed them inside the strip kernel: 36 + 4 lanes of wave 0 while the other three waves of the group waited at the barrier.)
// Tiles that cannot be staged go on the rest list here (the prep launch has zeroed its counter).
__global__ __launch_bounds__(256) void k_warp_records_batch(const WarpBatchArgs args, int gxt, int gyt, int n_tiles, int rest_cap)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tiles) return;
    const int per_img = gxt * gyt, z = t / per_img, l = t - z * per_img, by = l / gxt, bx = l - by * gxt;
    const WarpBatchDesc &d = args.d[z];
    const WarpBatchCore &a = d.a;
    const int dw = a.dw, dh = a.dh, dw4 = d.dw4, xshift = a.xshift, sw = a.sw, sh = a.sh;
    const int fgx = warp_tiles_x(dw), fgy = warp_tiles_y(dh);
    if (bx >= fgx || by >= fgy) return;
    const float *colS = d.tab, *colC = d.tab + dw4, *rowA = d.tab + 2 * (size_t)dw4, *rowB = rowA + dh;
    const bool plane_like = d.kind == PK_PLANE || d.kind == PK_AFFINE;      // (no Z > 0 rule: a sample behind the camera is not "beyond the frame")
    bool gain_fits = true;
    if (d.gain.kind == 2) {
        const int *gyi = wb_gyi(d);
        const int gbase = gyi[min(by * WT_H, dh - 1)], glast = min(gyi[min(by * WT_H + WT_H - 1, dh - 1)] + 1, d.gain.gh - 1);
        gain_fits = glast - gbase + 1 <= WT_GAIN_ROWS;
    }
    int flags = 0;
    int4 r0 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0};
    const int X0 = max(bx * WT_W - xshift, 0), X1 = min(bx * WT_W - xshift + WT_W - 1, dw - 1), Y0 = by * WT_H, Y1 = min(Y0 + WT_H - 1, dh - 1);
    if (X0 > X1) flags = WS_SKIP;
    else {
        // the map at the corners, edge midpoints and centre of the part of the tile inside the roi
        float lox = 3.0e38f, hix = -3.0e38f, loy = 3.0e38f, hiy = -3.0e38f;
        bool okall = true;
        int side = 31;      // sides of the frame ALL samples lie beyond (8 px of margin): left, right, above, below, behind the camera
        for (int s9 = 0; s9 < 9; ++s9) {
            const int j = s9 / 3, i = s9 - 3 * j;
            const int px = i == 0 ? X0 : (i == 1 ? (X0 + X1) >> 1 : X1), py = j == 0 ? Y0 : (j == 1 ? (Y0 + Y1) >> 1 : Y1);
            float X, Y, Z;
            if (d.cmap) map_backward_xyz(wb_proj(d), (float)(px + d.tlx), (float)(py + d.tly), X, Y, Z);      // any projection
            else {
                const float cs = colS[px + xshift], cc = colC[px + xshift], sa = rowA[py], sb = rowB[py];
                const float rx = sa * cs, rz = sa * cc;
                X = (a.kr[0] * rx + a.kr[1] * sb) + a.kr[2] * rz; Y = (a.kr[3] * rx + a.kr[4] * sb) + a.kr[5] * rz; Z = (a.kr[6] * rx + a.kr[7] * sb) + a.kr[8] * rz;
            }
            const bool v = Z > 8.6736174e-19f && Z < 1.1529215e18f && fabsf(X) < 1.1529215e18f && fabsf(Y) < 1.1529215e18f;
            const float qx = v ? X / Z : 0.f, qy = v ? Y / Z : 0.f;
            okall = okall && v && fabsf(qx) < 60000.f && fabsf(qy) < 60000.f;
            lox = fminf(lox, qx); hix = fmaxf(hix, qx); loy = fminf(loy, qy); hiy = fmaxf(hiy, qy);
            const float fxs = Z > 0 ? X / Z : 0.f, fys = Z > 0 ? Y / Z : 0.f;
            side &= Z > 0 ? ((fxs < -8.f ? 1 : 0) | (fxs > (float)sw + 8.f ? 2 : 0) | (fys < -8.f ? 4 : 0) | (fys > (float)sh + 8.f ? 8 : 0)) : (plane_like ? 0 : 16);
        }
        // The map is projective with Z of one sign over the tile: nine samples beyond one side of the frame (or all behind the camera) put every
        // pixel of the tile there -- its warped mask is 0 throughout.
        if (side) flags |= WS_EMPTY;
        // unreflected tap range (ix .. ix + 1 of every pixel, one pixel of margin: over a 64 x 16 tile the map departs from its affine
        // interpolation by well under a pixel -- curvature ~ 1 / focal length -- and every lane of the strip kernel re-checks)
        const int ux0 = (int)floorf(lox) - 1, ux1 = (int)floorf(hix) + 2, uy0 = (int)floorf(loy) - 1, uy1 = (int)floorf(hiy) + 2;
        const bool interior = ux0 >= 0 && uy0 >= 0 && ux1 <= sw - 1 && uy1 <= sh - 1;
        int rx0 = ux0, rx1 = ux1, ry0 = uy0, ry1 = uy1;
        bool can = okall && gain_fits;
        if (!interior) {
            // the image of [u0, u1] under BORDER_REFLECT, at most one fold per side and not both sides at once
            can = can && a.border == SSP_BORDER_REFLECT && ux0 >= -sw && ux1 <= 2 * sw - 1 && uy0 >= -sh && uy1 <= 2 * sh - 1 && !(ux0 < 0 && ux1 > sw - 1) && !(uy0 < 0 && uy1 > sh - 1);
            if (ux1 < 0) { rx0 = -ux1 - 1; rx1 = -ux0 - 1; } else if (ux0 < 0) { rx0 = 0; rx1 = max(ux1, -ux0 - 1); }
            else if (ux0 > sw - 1) { rx0 = 2 * sw - 1 - ux1; rx1 = 2 * sw - 1 - ux0; } else if (ux1 > sw - 1) { rx0 = min(ux0, 2 * sw - 1 - ux1); rx1 = sw - 1; }
            if (uy1 < 0) { ry0 = -uy1 - 1; ry1 = -uy0 - 1; } else if (uy0 < 0) { ry0 = 0; ry1 = max(uy1, -uy0 - 1); }
            else if (uy0 > sh - 1) { ry0 = 2 * sh - 1 - uy1; ry1 = 2 * sh - 1 - uy0; } else if (uy1 > sh - 1) { ry0 = min(uy0, 2 * sh - 1 - uy1); ry1 = sh - 1; }
            can = can && rx0 >= 0 && ry0 >= 0 && rx1 <= sw - 1 && ry1 <= sh - 1;
            flags |= WS_BORDER;
        }
        const int rowbytes = 3 * (rx1 + 1) - ((3 * rx0) & ~15), rows = ry1 - ry0 + 1, nch = (rowbytes + 15) >> 4;
        can = can && nch >= 1 && nch <= 40 && rows >= 2 && rows * nch <= (d.cmap ? WS_BUF_CMAP : WS_BUF) / 16;
        if (can) flags |= WS_STAGE;
        if (gain_fits) flags |= WS_GFIT;      // (tiles that are neither staged nor far go on the rest list in k_warp_records_far)
        r0 = make_int4(rx0, ry0, (rx1 - rx0 + 1) | (rows << 16), flags);
        // chunk index e of the rectangle -> row e / nch by multiplication: exact for e * nch < 2^16 (e < 768, nch <= 40)
        r1 = make_int4(ux0, uy0, (ux1 - ux0 + 1) | ((uy1 - uy0 + 1) << 16), can ? (nch | ((65536 / nch + 1) << 8)) : 0);
    }
    r0.w = flags;
    d.tiles[2 * (by * fgx + bx)] = r0;
    d.tiles[2 * (by * fgx + bx) + 1] = r1;
}

// Coordinate planes (one work-group per 64 x 16 tile, 4 pixels per lane as in the strip kernel).  For every pixel of a staged tile: the map
// by ITS DEFINITION -- map_backward of the part's projector, the very function the single-frame kernels and the oracle evaluate per pixel --
// quantised as cv::remap does (cvRound(32 x), cvRound(32 y)) and stored relative to the tile's unreflected tap origin (ux0, uy0) of its record:
//     bits 0-12 cvRound(32 x) - 32 ux0      bits 13-27 cvRound(32 y) - 32 uy0      bit 28 the INTER_NEAREST / BORDER_CONSTANT mask of the pixel
// Like the tables of the separable projections and the tile records, the plane is a function of the cameras only: written on a composer's first
// panoramas, read by every later one (4 bytes per warped pixel instead of the per-pixel map: 35 instructions for the separable projections,
// ~350 with the binary64 transcendentals of the other thirteen).  A pixel whose taps are not all inside the staged rectangle (the record's nine
// samples missed a bulge of the map) takes its whole tile off the staged path -- onto the rest list, which evaluates the map per pixel -- so the
// strip kernel needs no per-pixel fallback; and a tile the samples called EMPTY is only left so when no pixel of it has a set mask (the
// nine-sample argument is exact for projective maps only).
__global__ __launch_bounds__(256) void k_warp_cmap_batch(const WarpBatchArgs args, int gxt, int gyt, int n_tiles)
{
    const int t = blockIdx.x;
    const int per_img = gxt * gyt, z = t / per_img, l = t - z * per_img, by = l / gxt, bx = l - by * gxt;
    const WarpBatchDesc &d = args.d[z];
    if (!d.cmap) return;
    const WarpBatchCore &a = d.a;
    const int dw = a.dw, dh = a.dh, xshift = a.xshift;
    const int fgx = warp_tiles_x(dw), fgy = warp_tiles_y(dh);
    if (bx >= fgx || by >= fgy) return;
    const int4 r0 = d.tiles[2 * (by * fgx + bx)], r1 = d.tiles[2 * (by * fgx + bx) + 1];
    const int fl = r0.w;
    if (fl & WS_SKIP) return;
    const Projector &P = wb_proj(d);
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int y = by * WT_H + ly, t0 = bx * WT_W + 4 * lx, x0 = t0 - xshift;
    if (y >= dh) return;
    const int ux0 = r1.x, uy0 = r1.y, uw = r1.z & 0xffff, uh = r1.z >> 16;
    const int limx = (uw - 1) << 5, limy = (uh - 1) << 5;
    uint32_t cm[4];
    bool bad = false, any_valid = false;
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        const int x = min(max(x0 + i, 0), dw - 1);        // columns of the group outside the roi repeat the edge column (never stored by the strip kernel)
        float fx, fy;
        map_backward(P, (float)(x + d.tlx), (float)(y + d.tly), fx, fy);
        const bool valid = fx >= -0.5f && fx <= a.hix && fy >= -0.5f && fy <= a.hiy;
        const long long qx = (long long)cv_round(fx * 32.f) - 32LL * ux0, qy = (long long)cv_round(fy * 32.f) - 32LL * uy0;
        const bool in = qx >= 0 && qx < limx && qy >= 0 && qy < limy;
        const bool real = x0 + i >= 0 && x0 + i < dw;
        bad = bad || (real && !in);
        any_valid = any_valid || (real && valid);
        cm[i] = in ? ((uint32_t)qx | ((uint32_t)qy << 13) | (valid ? 1u << 28 : 0u)) : 0u;
    }
    if ((fl & WS_STAGE) && bad) atomicAnd(&d.tiles[2 * (by * fgx + bx)].w, ~WS_STAGE);
    if ((fl & WS_EMPTY) && any_valid) atomicAnd(&d.tiles[2 * (by * fgx + bx)].w, ~WS_EMPTY);
    if (fl & WS_STAGE) *(uint4 *)(d.cmap + WB_CMAP_HEAD + (size_t)y * (size_t)(fgx * WT_W) + t0) = make_uint4(cm[0], cm[1], cm[2], cm[3]);
}

// Second pass over the tile records (one lane per tile): FAR tiles and the rest list.
// A level-0 pixel can reach the blended panorama only through a pyramid sample that carries weight, and the weight maps are Gaussians of the
// warped mask: a sample of level l has weight only within 2 * 2^l pixels of a set mask pixel, and its Laplacian value draws on level 0 within
// 6 * 2^l -- so nothing beyond 4 * 2^bands pixels of the nearest set mask pixel is ever multiplied by anything but 0 (the same reach that
// bounds the multi-GPU strips, parallel.plan_strips).  `far_px` is that distance (0: feature off).  An EMPTY tile whose whole neighbourhood
// within far_px is EMPTY too (tiles beyond the roi count as empty) is FAR: the strip kernel writes its mask (zeros) and leaves the image
// bytes alone -- OpenCV computes reflected garbage there, which no output depends on.  What this buys: a frame that straddles u = +-pi*scale gets
// OpenCV's full-circle roi, 5/6 of it empty (bench.py ring360: 68 043 such tiles went through the gather kernel, 0.83 of 2.04 ms).
// rest[2 + rest_cap] is set when any tile is far: panoramas of a geometry without far tiles run the strip kernel compiled without that path.
__global__ __launch_bounds__(256) void k_warp_records_far(const WarpBatchArgs args, int gxt, int gyt, int n_tiles, int rest_cap, int far_px)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tiles) return;
    const int per_img = gxt * gyt, z = t / per_img, l = t - z * per_img, by = l / gxt, bx = l - by * gxt;
    const WarpBatchDesc &d = args.d[z];
    const int fgx = warp_tiles_x(d.a.dw), fgy = warp_tiles_y(d.a.dh);
    if (bx >= fgx || by >= fgy) return;
    const int flags = d.tiles[2 * (by * fgx + bx)].w;
    bool far = false;
    if ((flags & WS_EMPTY) && far_px > 0 && d.a.mask) {
        const int rx = (far_px + WT_W - 1) / WT_W + 1, ry = (far_px + WT_H - 1) / WT_H + 1;     // (+1: the column shift of the tile grid, rounding)
        far = true;
        for (int yy = max(by - ry, 0); yy <= min(by + ry, fgy - 1) && far; ++yy)
            for (int xx = max(bx - rx, 0); xx <= min(bx + rx, fgx - 1); ++xx)
                if (!(d.tiles[2 * (yy * fgx + xx)].w & (WS_EMPTY | WS_SKIP))) { far = false; break; }
    }
    if (far) {
        if (args.rest && !args.rest_known) args.rest[2 + rest_cap] = 1;
    } else if (!(flags & (WS_STAGE | WS_SKIP)) && args.rest && !args.rest_known) {
        const int slot = atomicAdd(args.rest, 1);
        if (slot < rest_cap) args.rest[1 + slot] = t;
        if (!(flags & WS_GFIT)) args.rest[1 + rest_cap] = 1;   // such tiles can never go inline (the gain rows of the strip are not in LDS)
    }
    // FAR goes to a word nobody else reads in this launch (bit 30 of the second int4's .w): the neighbours still see EMPTY in the first
    int4 *r1 = &d.tiles[2 * (by * fgx + bx) + 1];
    r1->w = (r1->w & 0x3fffffff) | (far ? 0x40000000 : 0);
}

// GAIN 0: none, 1: one gain per channel, 2: gain map with one channel, 3: gain map with three channels; FAR: the geometry has far tiles.
// (The map comes from the separable projections' tables; the coordinate-plane form for all projections is k_warp_strip_planes below.)
#ifdef WS_TRACE
// debug build only (tools/trace_warp.py): s_memtime stamps of every wave of the last launch, 32 slots per wave.  The stamps go to LDS and leave for
// global memory at the wave's end: a global store per stamp would sit in front of the kernel's own vmcnt waits and be what they measure
#define WS_TRACE_WAVES (32768 * 4)
__device__ unsigned int g_ws_trace[WS_TRACE_WAVES * 32];
#define WS_TRACE_DECL __shared__ unsigned int s_trace[4][32]; if ((threadIdx.x & 63) < 32) s_trace[threadIdx.x >> 6][threadIdx.x & 31] = 0u
#define WS_STAMP(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) s_trace[threadIdx.x >> 6][(slot)] = (unsigned int)t_; } while (0)
#define WS_TRACE_OUT do { if ((threadIdx.x & 63) < 32 && blockIdx.x < 32768) g_ws_trace[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + (threadIdx.x & 31)] = s_trace[threadIdx.x >> 6][threadIdx.x & 31]; } while (0)
extern "C" __attribute__((visibility("default"))) int ssp_debug_warp_trace(void *host, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ws_trace), bytes, 0, hipMemcpyDeviceToHost);
}
#else
#define WS_TRACE_DECL do { } while (0)
#define WS_STAMP(slot) do { } while (0)
#define WS_TRACE_OUT do { } while (0)
#endif
template <int GAIN, bool FAR>
__global__ __launch_bounds__(256) void k_warp_strip_batch(const WarpBatchArgs args, int gxt, int gyt, int sgx, int n_strips, int xcd_remap, uint32_t m_per_img, uint32_t m_sgx, int rest_cap, int inline_rest)
{
    constexpr int GCN = GAIN == 3 ? 3 : 1;
    constexpr int BUF = WS_BUF;
    __shared__ __attribute__((aligned(16))) uint8_t s_buf[2][BUF + 16];
#ifndef WS_TAB_LDS
#define WS_TAB_LDS 1          // 1: the strip's 256 column-table entries go through LDS; 0: a lane loads its 2 x 4 entries per tile straight from the tables
#endif                        // (2 KB of LDS and 6-10 VGPRs less, but two dependent global loads in front of every tile's map: 270 -> 310 us, profiles/r04_warp_variants.txt)
#if WS_TAB_LDS
    __shared__ __attribute__((aligned(16))) float s_cs[256], s_cc[256];
#endif
    __shared__ __attribute__((aligned(16))) float s_gain[GAIN >= 2 ? GCN * WT_GAIN_ROWS * 256 : 4];
    __shared__ __attribute__((aligned(16))) int s_rec[WS_NT * 8];
    int t = blockIdx.x;
    if (xcd_remap == 1) {
        const int xcd = t & 7, idx = t >> 3, q = n_strips >> 3, r = n_strips & 7;
        t = xcd * q + min(xcd, r) + idx;
    } else if (xcd_remap > 1) {
        // chunks of `xcd_remap` consecutive strips go round the XCDs (work-group t runs on XCD t & 7): an XCD still reads contiguous runs of the
        // frames into its own L2, and all XCDs finish together whatever the parts cost (contiguous eighths gave the two XCDs that own a closed
        // ring's four small parts half the work of the others); the launch is rounded up to whole rounds of chunks
        const int xcd = t & 7, idx = t >> 3, c = idx / xcd_remap, within = idx - c * xcd_remap;
        t = (c * 8 + xcd) * xcd_remap + within;
        if (t >= n_strips) return;
    }
    const int per_img = sgx * gyt, z = (int)udiv_by_magic((uint32_t)t, (uint32_t)per_img, m_per_img), l = t - z * per_img;
    const int by = (int)udiv_by_magic((uint32_t)l, (uint32_t)sgx, m_sgx), sx = l - by * sgx;
    const WarpBatchDesc &d = args.d[z];
    const WarpBatchCore &a = d.a;
    const int dw = a.dw, dh = a.dh, dw4 = d.dw4, xshift = a.xshift, sw = a.sw, sh = a.sh;
    const int fgx = warp_tiles_x(dw), fgy = warp_tiles_y(dh);
    if (by >= fgy || WS_NT * sx >= fgx) return;
    const int nt = min(WS_NT, fgx - WS_NT * sx);
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    WS_TRACE_DECL;
    WS_STAMP(0);
    const int y = by * WT_H + ly, yc = min(y, dh - 1);
    // the seam row this row interpolates from (mask preparation): asked for now, used behind the set-up fence -- with the flag it indexes two dependent
    // round trips that every wave would otherwise make on its own behind the fence (tools/trace_warp.py: 2 400 of a wave's 30 000 cycles)
    int seam_row = 0;
    if (d.prep) seam_row = d.lin[2 * dw4 + yc];
    const uint32_t pitch = a.spitch;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.sdata, (short)0, (int)(pitch * (uint32_t)sh), 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)d.tab, (short)0, (int)(8 * (dw4 + dh)), 0x00020000);   // colS | colC | rowA | rowB
    // ---- strips without a live tile (FAR variant): in a full-circle roi most strips consist of far tiles only -- their masks, and out
    if (FAR) {
        int any_live = 0, any_far = 0;
#pragma unroll
        for (int k = 0; k < WS_NT; ++k) {
            if (k >= nt) break;
            const int bxk = WS_NT * sx + k;
            const int f0 = d.tiles[2 * (by * fgx + bxk)].w, f1 = d.tiles[2 * (by * fgx + bxk) + 1].w;      // (uniform: scalar loads)
            const bool far = (f1 & 0x40000000) != 0;
            any_far |= far ? (1 << k) : 0;
            any_live |= (!far && !(f0 & WS_SKIP)) ? 1 : 0;
        }
        if (!any_live) {
            if (a.mask && y < dh)
                for (int k = 0; k < nt; ++k) {
                    if (!((any_far >> k) & 1)) continue;
                    const int t0f = (WS_NT * sx + k) * WT_W + 4 * lx, x0f = t0f - xshift;
                    for (int i = 0; i < 4; ++i)
                        if (x0f + i >= 0 && x0f + i < dw) a.mask[(ptrdiff_t)y * (ptrdiff_t)a.mpitch + x0f + i] = 0;
                }
            return;
        }
    }
    // ---- set-up -------------------------------------------------------------------------------------------------------------------
    float ra = 0.f, rb = 0.f;
    {
        ra = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, 4u * (uint32_t)yc, 8u * (uint32_t)dw4, 0));
        rb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, 4u * (uint32_t)yc, 8u * (uint32_t)dw4 + 4u * (uint32_t)dh, 0));
#if WS_TAB_LDS
        const uint32_t tc = 4u * (uint32_t)min(256 * sx + tid, dw4 - 1);
        s_cs[tid] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, tc, 0, 0));
        s_cc[tid] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, tc, 4u * (uint32_t)dw4, 0));
#endif
    }
    // exposure compensation: gain rows under the strip resized horizontally to this thread's column (first half of resize(gain_map, frame
    // size, INTER_LINEAR) in OpenCV's order; the vertical half follows per pixel); one gain per channel: nothing to prepare
    int grow0 = 0, grow1 = 0;
    float gb1 = 0.f;
    if (GAIN >= 2) {
        const WarpBatchGain &ga = d.gain;
        const int *gyi = wb_gyi(d);
        const int gbase = gyi[min(by * WT_H, dh - 1)];
        const int gy0 = gyi[yc];
        grow0 = gy0 - gbase; grow1 = min(gy0 + 1, ga.gh - 1) - gbase; gb1 = wb_gyb(d)[yc];
        const int tj = min(256 * sx + tid, dw4 - 1);
        const int xg0 = wb_gxi(d)[tj], xg1 = min(xg0 + 1, ga.gw - 1);
        const float a1 = wb_gxa(d)[tj], a0f = 1.f - a1;
#pragma unroll
        for (int gr = 0; gr < WT_GAIN_ROWS; ++gr) {
            const float *row = ga.gm + (min(gbase + gr, ga.gh - 1) * ga.gw) * GCN;
#pragma unroll
            for (int c = 0; c < GCN; ++c) s_gain[(c * WT_GAIN_ROWS + gr) * 256 + tid] = row[xg0 * GCN + c] * a0f + row[xg1 * GCN + c] * a1;
        }
    }
    if (tid < WS_NT) {
        // tile `tid`: its record (source rectangle to stage, unreflected tap range, flags) comes from k_warp_records_batch
        const int k = tid, bx = WS_NT * sx + k;
        int4 r0 = {0, 0, 0, WS_SKIP}, r1 = {0, 0, 0, 0};
        if (k < nt) {
            r0 = d.tiles[2 * (by * fgx + bx)];
            r1 = d.tiles[2 * (by * fgx + bx) + 1];
            const bool far = (r1.w & 0x40000000) != 0;          // k_warp_records_far: nothing of this tile can reach the panorama
            r1.w &= 0x3fffffff;
            if (FAR && far) r0.w = WS_FAR;
            // not stageable: inline when the plan says such tiles are rare (and the strip's gain rows are in LDS), else it is on the rest list
            else if (!(r0.w & (WS_STAGE | WS_SKIP)) && inline_rest && (r0.w & WS_GFIT)) r0.w |= WS_INLINE;
        }
        *(int4 *)(s_rec + 8 * k) = r0;
        *(int4 *)(s_rec + 8 * k + 4) = r1;
    }
    __syncthreads();
    WS_STAMP(1);
    // LDS-DMA of tile k's rectangle into buffer b: chunk e = 256 p + tid of the rectangle (row-major, nch chunks per row) per pass p, so that a
    // wave's 64 chunks are consecutive in LDS
    auto stage = [&](int k, int b) {
        const int rx0 = __builtin_amdgcn_readfirstlane(s_rec[8 * k]), ry0 = __builtin_amdgcn_readfirstlane(s_rec[8 * k + 1]);
        const int wh = __builtin_amdgcn_readfirstlane(s_rec[8 * k + 2]), fl = __builtin_amdgcn_readfirstlane(s_rec[8 * k + 3]);
        const int nm = __builtin_amdgcn_readfirstlane(s_rec[8 * k + 7]);
        if (!(fl & WS_STAGE)) return;
        const int rows = wh >> 16, nch = nm & 0xff;
        const uint32_t mg = (uint32_t)nm >> 8, a0 = (3u * (uint32_t)rx0) & ~15u;
        const int total = rows * nch;
#pragma unroll
        for (int p = 0; p < (BUF + 4095) / 4096; ++p)
            if (256 * p < total) {
                const uint32_t e = 256u * p + (uint32_t)tid, row = __umul24(e, mg) >> 16, chunk = e - row * (uint32_t)nch;
                if ((int)e < total)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(s_buf[b] + p * 4096 + wave * 1024), 16,
                                                             __umul24((uint32_t)ry0 + row, pitch) + a0 + 16u * chunk, 0, 0, 0);
            }
    };
    stage(0, 0);
    const f32x2 c1 = {a.kr[1] * rb, a.kr[1] * rb}, c4 = {a.kr[4] * rb, a.kr[4] * rb}, c7 = {a.kr[7] * rb, a.kr[7] * rb};
    const bool row_live = y < dh;
    // per-row part of the mask preparation: is everything this row of the strip interpolates from inside the seam mask?  (first use: tile 0's step 5)
    bool seam_in = true;
    if (d.prep && row_live) seam_in = d.has_flags && wb_flags(d)[seam_row * d.fgx + sx] != 0;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)(a.dst - 3 * xshift), (short)0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void *)(a.mask ? a.mask - xshift : a.dst), (short)0, 0x7ffffff0, 0x00020000);
    const uint32_t drow = __umul24((uint32_t)yc, (uint32_t)a.dpitch), mrow = __umul24((uint32_t)yc, (uint32_t)a.mpitch);
    // ---- the tiles of the strip -----------------------------------------------------------------------------------------------------------
    // Order inside a tile (round 3): the copy of tile k+1's rectangle is issued AFTER the last LDS read of tile k.  For an LDS read that
    // follows an LDS-DMA the compiler inserts s_waitcnt vmcnt(0) (it cannot tell the addresses apart); with the copy issued before the taps,
    // as in round 2, every wave waited for the rectangle it had just requested.  Now the copy flies during the stores of tile k and the map
    // of tile k+1; the LDS reads that come before the explicit wait -- tile record, column tables -- are made where the compiler cannot
    // see a pending copy (record: loaded one tile ahead; tables: inline assembly).
    int n_rx0 = __builtin_amdgcn_readfirstlane(s_rec[0]), n_ry0 = __builtin_amdgcn_readfirstlane(s_rec[1]), n_fl = __builtin_amdgcn_readfirstlane(s_rec[3]);
    int n_ux0 = __builtin_amdgcn_readfirstlane(s_rec[4]), n_uy0 = __builtin_amdgcn_readfirstlane(s_rec[5]), n_uwh = __builtin_amdgcn_readfirstlane(s_rec[6]), n_nm = __builtin_amdgcn_readfirstlane(s_rec[7]);
    WS_STAMP(2);
    for (int k = 0; k < nt; ++k) {
        const int b = k & 1;
        const int rx0 = n_rx0, ry0 = n_ry0, fl = n_fl, ux0 = n_ux0, uy0 = n_uy0, uwh = n_uwh, nm = n_nm;
        const bool staged = (fl & WS_STAGE) != 0;
        const int t0 = (WS_NT * sx + k) * WT_W + 4 * lx, x0 = t0 - xshift;
        const bool live = (fl & (WS_STAGE | WS_INLINE)) != 0 && row_live && x0 < dw;
        // -- 1. this lane's four pixels, two per instruction: K R^T ray in OpenCV's operation order, the two IEEE divisions (shared reciprocal,
        // the refinement sequence of a correctly rounded division: exact for 2^-60 <= Z < 2^60 and quotients that pass the range test), and
        // cvRound(32 q) relative to the (unreflected) tap range: 32 q + 1.5 * 2^23 - 32 * origin rounded once to an integer (ties to even, the
        // constant is even) leaves cvRound(32 q) - 32 * origin in the mantissa; anything outside [0, 2^22) -- negative, huge, NaN -- leaves bits
        // above it set after the xor
        uint32_t bxr[4], byr[4];
        uint32_t mk = 0xffffffffu;
        bool bad = !staged;
        if (live && staged) {
            float4 cs4, cc4;
#if !WS_TAB_LDS
            {
                // (the table entries of columns beyond the roi read as zeros: buffer bounds; such lanes are not live)
                const u32x4_t t_s = __builtin_amdgcn_raw_buffer_load_b128(rt, 4u * (uint32_t)t0, 0, 0), t_c = __builtin_amdgcn_raw_buffer_load_b128(rt, 4u * (uint32_t)t0, 4u * (uint32_t)dw4, 0);
                cs4 = make_float4(ufloat(t_s.x), ufloat(t_s.y), ufloat(t_s.z), ufloat(t_s.w));
                cc4 = make_float4(ufloat(t_c.x), ufloat(t_c.y), ufloat(t_c.z), ufloat(t_c.w));
            }
#else
            {
                typedef float asm_f32x4 __attribute__((ext_vector_type(4)));
                asm_f32x4 t_s, t_c;
                const uint32_t as = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)(s_cs + WT_W * k + 4 * lx);
                const uint32_t ac = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)(s_cc + WT_W * k + 4 * lx);
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t_s), "=&v"(t_c) : "v"(as), "v"(ac));
                cs4 = make_float4(t_s.x, t_s.y, t_s.z, t_s.w); cc4 = make_float4(t_c.x, t_c.y, t_c.z, t_c.w);
            }
#endif
            const f32x2 cs[2] = {{cs4.x, cs4.y}, {cs4.z, cs4.w}}, cc[2] = {{cc4.x, cc4.y}, {cc4.z, cc4.w}};
            const float MXs = (float)(12582912 - 32 * ux0), MYs = (float)(12582912 - 32 * uy0);
            const f32x2 MX = {MXs, MXs}, MY = {MYs, MYs}, k32 = {32.f, 32.f}, one = {1.f, 1.f};
            // scalar f32: add / mul / fma issue at about 2.7 cycles per wave instruction against 5.5 for their packed forms (profiles/r02_valu_microbench.txt)
            float Zv[4], QXv[4], QYv[4];
            const float csv[4] = {cs4.x, cs4.y, cs4.z, cs4.w}, ccv[4] = {cc4.x, cc4.y, cc4.z, cc4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float rx = ra * csv[i], rz = ra * ccv[i];
                const float X = (a.kr[0] * rx + c1.x) + a.kr[2] * rz, Y = (a.kr[3] * rx + c4.x) + a.kr[5] * rz, Z = (a.kr[6] * rx + c7.x) + a.kr[8] * rz;
                float r = __builtin_amdgcn_rcpf(Z);
                const float e = __builtin_fmaf(-Z, r, 1.f);
                r = __builtin_fmaf(e, r, r);
                float q = X * r;
                float tt = __builtin_fmaf(-Z, q, X);
                q = __builtin_fmaf(tt, r, q);
                tt = __builtin_fmaf(-Z, q, X);
                QXv[i] = __builtin_fmaf(tt, r, q);
                q = Y * r;
                tt = __builtin_fmaf(-Z, q, Y);
                q = __builtin_fmaf(tt, r, q);
                tt = __builtin_fmaf(-Z, q, Y);
                QYv[i] = __builtin_fmaf(tt, r, q);
                Zv[i] = Z;
                bxr[i] = fbits(__builtin_fmaf(QXv[i], 32.f, MXs)) ^ 0x4B400000u;
                byr[i] = fbits(__builtin_fmaf(QYv[i], 32.f, MYs)) ^ 0x4B400000u;
            }
            const f32x2 Zs[2] = {{Zv[0], Zv[1]}, {Zv[2], Zv[3]}}, QX[2] = {{QXv[0], QXv[1]}, {QXv[2], QXv[3]}}, QY[2] = {{QYv[0], QYv[1]}, {QYv[2], QYv[3]}};
            (void)cs; (void)cc; (void)MX; (void)MY; (void)k32; (void)one;
            const uint32_t mxx = max(max(bxr[0], bxr[1]), max(bxr[2], bxr[3])), mxy = max(max(byr[0], byr[1]), max(byr[2], byr[3]));
            // 2^-60 <= Z < 2^60  <=>  bits(Z) - 0x21800000 < 0x3C000000 (unsigned)
            const uint32_t mxz = max(max(fbits(Zs[0].x) - 0x21800000u, fbits(Zs[0].y) - 0x21800000u), max(fbits(Zs[1].x) - 0x21800000u, fbits(Zs[1].y) - 0x21800000u));
            bad = mxx >= (uint32_t)(((uwh & 0xffff) - 1) << 5) || mxy >= (uint32_t)(((uwh >> 16) - 1) << 5) || mxz >= 0x3C000000u;
            if (fl & WS_BORDER) {
                // INTER_NEAREST + BORDER_CONSTANT on the all-255 mask without converting: cvRound(f) in [0, n-1] <=> -0.5 <= f <= n-0.5 (see nearest_hi)
                mk = 0u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float qx = (i & 1) ? QX[i >> 1].y : QX[i >> 1].x, qy = (i & 1) ? QY[i >> 1].y : QY[i >> 1].x;
                    if (qx >= -0.5f && qx <= a.hix && qy >= -0.5f && qy <= a.hiy) mk |= 0xffu << (8 * i);
                }
            }
        }
        // -- 2. the rectangle has landed (DMA issued one tile ago); everybody is done with the other buffer: refill it for the next tile
        asm volatile("" ::"v"(bxr[0]), "v"(byr[3]) : "memory");
        WS_STAMP(4 + 6 * k);
        __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
        WS_STAMP(5 + 6 * k);
        __syncthreads();
        WS_STAMP(6 + 6 * k);
        uint32_t o0 = 0, o1 = 0, o2 = 0;
        if (live) {
        // -- 3. taps from LDS, fixed-point bilinear
        const uint8_t *tile = s_buf[b];
        const uint32_t c0 = (3u * (uint32_t)rx0) & 15u, pitchl = 16u * (uint32_t)(nm & 0xff);
        Px3 v[4];
        if (!bad) {
            if (!(fl & WS_BORDER)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = taps_interior(tile, pitchl, c0, bxr[i], byr[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = taps_reflect(tile, pitchl, c0, bxr[i], byr[i], ux0, uy0, rx0, ry0, sw, sh);
            }
        } else {
            // taps outside what was staged (the samples missed them) or an operand out of the exact division's range: the general per-pixel form
            mk = 0u;
#if WS_TAB_LDS
            const float4 cs4 = *(const float4 *)(s_cs + WT_W * k + 4 * lx), cc4 = *(const float4 *)(s_cc + WT_W * k + 4 * lx);
            const float csv[4] = {cs4.x, cs4.y, cs4.z, cs4.w}, ccv[4] = {cc4.x, cc4.y, cc4.z, cc4.w};
#else
            const u32x4_t t_s = __builtin_amdgcn_raw_buffer_load_b128(rt, 4u * (uint32_t)t0, 0, 0), t_c = __builtin_amdgcn_raw_buffer_load_b128(rt, 4u * (uint32_t)t0, 4u * (uint32_t)dw4, 0);
            const float csv[4] = {ufloat(t_s.x), ufloat(t_s.y), ufloat(t_s.z), ufloat(t_s.w)};
            const float ccv[4] = {ufloat(t_c.x), ufloat(t_c.y), ufloat(t_c.z), ufloat(t_c.w)};
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float rx = ra * csv[i], rz = ra * ccv[i];
                const float X = (a.kr[0] * rx + c1.x) + a.kr[2] * rz, Y = (a.kr[3] * rx + c4.x) + a.kr[5] * rz, Z = (a.kr[6] * rx + c7.x) + a.kr[8] * rz;
                const float fx = Z > 0 ? X / Z : -1.f, fy = Z > 0 ? Y / Z : -1.f;
                const uint32_t q = bilinear_u8c3(wb_src(a), fx, fy, a.border);
                v[i].b = (q & 0xffu) << 16; v[i].g = ((q >> 8) & 0xffu) << 16; v[i].r = q & 0xff0000u;
                if (fx >= -0.5f && fx <= a.hix && fy >= -0.5f && fy <= a.hiy) mk |= 0xffu << (8 * i);
            }
        }
        asm volatile("" ::"v"(v[0].b), "v"(v[3].r) : "memory");
        WS_STAMP(7 + 6 * k);
        // -- 4. exposure compensation and packing: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
        if (GAIN) {
            float g[4][3];
            if (GAIN >= 2) {
                const float b1 = gb1, b0 = 1.f - b1;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (c < GCN) {
                        float4 t0g, t1g;
                        t0g = *(const float4 *)(s_gain + (c * WT_GAIN_ROWS + grow0) * 256 + WT_W * k + 4 * lx); t1g = *(const float4 *)(s_gain + (c * WT_GAIN_ROWS + grow1) * 256 + WT_W * k + 4 * lx);
                        g[0][c] = t0g.x * b0 + t1g.x * b1; g[1][c] = t0g.y * b0 + t1g.y * b1; g[2][c] = t0g.z * b0 + t1g.z * b1; g[3][c] = t0g.w * b0 + t1g.w * b1;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) g[i][c] = g[i][0];
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { g[i][0] = d.gain.g[0]; g[i][1] = d.gain.g[1]; g[i][2] = d.gain.g[2]; }
            }
            // multiply(image, gain): saturate_cast<uchar>(cvRound(sample * gain)) per channel (sde.py:1754), straight into the output words
#define GPX(i, c) ((float)((c == 0 ? v[i].b : c == 1 ? v[i].g : v[i].r) >> 16 & 0xffu) * g[i][c])
            o0 = pack_u8_rne(GPX(0, 0), 0, o0); o0 = pack_u8_rne(GPX(0, 1), 1, o0); o0 = pack_u8_rne(GPX(0, 2), 2, o0); o0 = pack_u8_rne(GPX(1, 0), 3, o0);
            o1 = pack_u8_rne(GPX(1, 1), 0, o1); o1 = pack_u8_rne(GPX(1, 2), 1, o1); o1 = pack_u8_rne(GPX(2, 0), 2, o1); o1 = pack_u8_rne(GPX(2, 1), 3, o1);
            o2 = pack_u8_rne(GPX(2, 2), 0, o2); o2 = pack_u8_rne(GPX(3, 0), 1, o2); o2 = pack_u8_rne(GPX(3, 1), 2, o2); o2 = pack_u8_rne(GPX(3, 2), 3, o2);
#undef GPX
        } else {
            // byte 2 of each value, gathered with v_perm_b32 (selector bytes 0-3: second operand, 4-7: first operand)
            const uint32_t t0p = __builtin_amdgcn_perm(v[0].g, v[0].b, 0x0c0c0602u), u0p = __builtin_amdgcn_perm(v[1].b, v[0].r, 0x0c0c0602u);
            const uint32_t t1p = __builtin_amdgcn_perm(v[1].r, v[1].g, 0x0c0c0602u), u1p = __builtin_amdgcn_perm(v[2].g, v[2].b, 0x0c0c0602u);
            const uint32_t t2p = __builtin_amdgcn_perm(v[3].b, v[2].r, 0x0c0c0602u), u2p = __builtin_amdgcn_perm(v[3].r, v[3].g, 0x0c0c0602u);
            o0 = __builtin_amdgcn_perm(u0p, t0p, 0x05040100u);
            o1 = __builtin_amdgcn_perm(u1p, t1p, 0x05040100u);
            o2 = __builtin_amdgcn_perm(u2p, t2p, 0x05040100u);
        }
        // -- 5. mask preparation (sde.py:1760-1772) unless this row's share of the strip lies inside the seam mask
        if (d.prep && mk && !seam_in) {
            MaskPrep mp;
            mp.dil = d.dil; mp.dpitch = d.dil_pitch;
            mp.xo = d.lin; mp.xc = d.lin + dw4; mp.yo = d.lin + 2 * dw4; mp.yc = mp.yo + dh;
            mp.flags = nullptr; mp.fgx = 0;
            mk &= seam_mask4(mp, y, t0);
        }
        }   // live
        asm volatile("" ::"v"(o0), "v"(o2), "v"(mk) : "memory");
        WS_STAMP(8 + 6 * k);
        // -- 5b. the next tile: its record into scalars, then the copy of its rectangle into the other buffer (every lane carries chunks);
        // all LDS reads of this tile are behind us, and every wave has finished with the other buffer (it passed this tile's barrier)
        if (k + 1 < nt) {
            n_rx0 = __builtin_amdgcn_readfirstlane(s_rec[8 * (k + 1)]); n_ry0 = __builtin_amdgcn_readfirstlane(s_rec[8 * (k + 1) + 1]); n_fl = __builtin_amdgcn_readfirstlane(s_rec[8 * (k + 1) + 3]);
            n_ux0 = __builtin_amdgcn_readfirstlane(s_rec[8 * (k + 1) + 4]); n_uy0 = __builtin_amdgcn_readfirstlane(s_rec[8 * (k + 1) + 5]);
            n_uwh = __builtin_amdgcn_readfirstlane(s_rec[8 * (k + 1) + 6]); n_nm = __builtin_amdgcn_readfirstlane(s_rec[8 * (k + 1) + 7]);
            stage(k + 1, b ^ 1);
        }
        if (FAR && (fl & WS_FAR) && row_live && x0 < dw && a.mask) {
            // a far tile: the mask is all there is to write (the image bytes under it are never multiplied by anything but 0)
            if (x0 >= 0 && x0 + 4 <= dw) __builtin_amdgcn_raw_buffer_store_b32(0u, rm, mrow + (uint32_t)t0, 0, 0);
            else
                for (int i = 0; i < 4; ++i)
                    if (x0 + i >= 0 && x0 + i < dw) a.mask[(ptrdiff_t)y * (ptrdiff_t)a.mpitch + x0 + i] = 0;
        }
        if (!live) { WS_STAMP(9 + 6 * k); continue; }
        // -- 6. stores (rows of the blender's planes: 4-byte aligned groups, see xshift)
        if (x0 >= 0 && x0 + 4 <= dw) {
            u32x3_a4 w;
            w.x = o0; w.y = o1; w.z = o2;
            __builtin_amdgcn_raw_buffer_store_b96(w, rd, drow + 3u * (uint32_t)t0, 0, 0);
            if (a.mask) __builtin_amdgcn_raw_buffer_store_b32(mk, rm, mrow + (uint32_t)t0, 0, 0);
        } else {
            uint8_t *dp = a.dst + (ptrdiff_t)y * (ptrdiff_t)a.dpitch + (ptrdiff_t)x0 * 3;
            const uint32_t ww[3] = {o0, o1, o2};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (x0 + i < 0 || x0 + i >= dw) continue;
#pragma unroll
                for (int c = 0; c < 3; ++c) { const int bidx = 3 * i + c; dp[bidx] = (uint8_t)(ww[bidx >> 2] >> (8 * (bidx & 3))); }
                if (a.mask) a.mask[(ptrdiff_t)y * (ptrdiff_t)a.mpitch + x0 + i] = (uint8_t)(mk >> (8 * i));
            }
        }
        WS_STAMP(9 + 6 * k);
    }
    WS_STAMP(3);
    WS_TRACE_OUT;
}

// ---- the coordinate-plane variant with TWO rectangle copies in flight per work-group (round 4) ---------------------------------------------------
// PMC says the plane variant of k_warp_strip_batch idles: 906 VALU instructions per wave, waves parked 62 % of their cycles, one rectangle copy
// outstanding per work-group.  Here the staging area is ONE arena of 24 KB cut into three slots of 8 KB when every rectangle of the strip fits
// that (the usual case) or two of 12 KB, and the copy of tile k + 2 (k + 1) is requested right behind the barrier of tile k -- into the slot tile
// k - 1 has just been read out of.  Nothing is counted by hand: a tile's four coordinate words are loaded right BEHIND the copy of that tile
// was requested (one tile ahead of their use), vmcnt retires in order, so the wait the compiler places in front of their first use is the wait for
// that tile's rectangle -- and not for the younger copies behind it.  What must not happen is an LDS read the compiler can see behind an LDS-DMA
// (it would wait vmcnt(0)): taps and gain rows are hand-written reads, the tile records live in registers (lane k of every wave holds tile k's
// record, v_readlane), the barrier is a bare s_barrier (each wave has waited for its own chunks of the rectangle; no fence needed for LDS).
template <int GAIN, bool FAR>
__global__ __launch_bounds__(256) void k_warp_strip_planes(const WarpBatchArgs args, int gxt, int gyt, int sgx, int n_strips, int xcd_remap, uint32_t m_per_img, uint32_t m_sgx)
{
    constexpr int GCN = GAIN == 3 ? 3 : 1;
    constexpr int ARENA = 2 * WS_BUF_CMAP;
    __shared__ __attribute__((aligned(16))) uint8_t s_arena[ARENA + 64];
    __shared__ __attribute__((aligned(16))) float s_gain[GAIN >= 2 ? GCN * WT_GAIN_ROWS * 256 : 4];
    int t = blockIdx.x;
    if (xcd_remap == 1) {
        const int xcd = t & 7, idx = t >> 3, q = n_strips >> 3, r = n_strips & 7;
        t = xcd * q + min(xcd, r) + idx;
    } else if (xcd_remap > 1) {
        const int xcd = t & 7, idx = t >> 3, c = idx / xcd_remap, within = idx - c * xcd_remap;
        t = (c * 8 + xcd) * xcd_remap + within;
        if (t >= n_strips) return;
    }
    const int per_img = sgx * gyt, z = (int)udiv_by_magic((uint32_t)t, (uint32_t)per_img, m_per_img), l = t - z * per_img;
    const int by = (int)udiv_by_magic((uint32_t)l, (uint32_t)sgx, m_sgx), sx = l - by * sgx;
    const WarpBatchDesc &d = args.d[z];
    const WarpBatchCore &a = d.a;
    const int dw = a.dw, dh = a.dh, dw4 = d.dw4, xshift = a.xshift, sw = a.sw, sh = a.sh;
    const int fgx = warp_tiles_x(dw), fgy = warp_tiles_y(dh);
    if (by >= fgy || WS_NT * sx >= fgx) return;
    const int nt = min(WS_NT, fgx - WS_NT * sx);
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    WS_TRACE_DECL;
    WS_STAMP(0);
    const int y = by * WT_H + ly, yc = min(y, dh - 1);
    int seam_row = 0;                                     // (asked for early, as in k_warp_strip_batch)
    if (d.prep) seam_row = d.lin[2 * dw4 + yc];
    const uint32_t pitch = a.spitch;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.sdata, (short)0, (int)(pitch * (uint32_t)sh), 0x00020000);
    const uint32_t cpitch = 4u * (uint32_t)(fgx * WT_W);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(d.cmap + WB_CMAP_HEAD), (short)0, (int)(cpitch * (uint32_t)dh), 0x00020000);
    // ---- the strip's tile records: lane k of every wave holds tile k's (two int4)
    int4 q0 = {0, 0, 0, WS_SKIP}, q1 = {0, 0, 0, 0};
    if (lane < nt) {
        q0 = d.tiles[2 * (by * fgx + WS_NT * sx + lane)];
        q1 = d.tiles[2 * (by * fgx + WS_NT * sx + lane) + 1];
        const bool far = (q1.w & 0x40000000) != 0;
        q1.w &= 0x3fffffff;
        if (FAR && far) q0.w = WS_FAR;
    }
    if (FAR) {
        // strips without a live tile: their far tiles' masks, and out
        const unsigned long long live_b = __ballot(lane < nt && !(q0.w & (WS_SKIP | WS_FAR))), far_b = __ballot(lane < nt && (q0.w & WS_FAR));
        if ((live_b & 15ULL) == 0ULL) {
            if (a.mask && y < dh)
                for (int k = 0; k < nt; ++k) {
                    if (!((far_b >> k) & 1ULL)) continue;
                    const int t0f = (WS_NT * sx + k) * WT_W + 4 * lx, x0f = t0f - xshift;
                    for (int i = 0; i < 4; ++i)
                        if (x0f + i >= 0 && x0f + i < dw) a.mask[(ptrdiff_t)y * (ptrdiff_t)a.mpitch + x0f + i] = 0;
                }
            return;
        }
    }
    // ---- set-up: gain rows under the strip (as in k_warp_strip_batch)
    int grow0 = 0, grow1 = 0;
    float gb1 = 0.f;
    if (GAIN >= 2) {
        const WarpBatchGain &ga = d.gain;
        const int *gyi = wb_gyi(d);
        const int gbase = gyi[min(by * WT_H, dh - 1)];
        const int gy0 = gyi[yc];
        grow0 = gy0 - gbase; grow1 = min(gy0 + 1, ga.gh - 1) - gbase; gb1 = wb_gyb(d)[yc];
        const int tj = min(256 * sx + tid, dw4 - 1);
        const int xg0 = wb_gxi(d)[tj], xg1 = min(xg0 + 1, ga.gw - 1);
        const float a1 = wb_gxa(d)[tj], a0f = 1.f - a1;
#pragma unroll
        for (int gr = 0; gr < WT_GAIN_ROWS; ++gr) {
            const float *row = ga.gm + (min(gbase + gr, ga.gh - 1) * ga.gw) * GCN;
#pragma unroll
            for (int c = 0; c < GCN; ++c) s_gain[(c * WT_GAIN_ROWS + gr) * 256 + tid] = row[xg0 * GCN + c] * a0f + row[xg1 * GCN + c] * a1;
        }
    }
    const bool row_live = y < dh;
    bool seam_in = true;
    if (d.prep && row_live) seam_in = d.has_flags && wb_flags(d)[seam_row * d.fgx + sx] != 0;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)(a.dst - 3 * xshift), (short)0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void *)(a.mask ? a.mask - xshift : a.dst), (short)0, 0x7ffffff0, 0x00020000);
    const uint32_t drow = __umul24((uint32_t)yc, (uint32_t)a.dpitch), mrow = __umul24((uint32_t)yc, (uint32_t)a.mpitch);
    __syncthreads();          // the gain rows are in LDS (the last fence of this kernel: nothing is in flight yet)
    WS_STAMP(1);
    // three slots when every rectangle of the strip fits a third of the arena
    int biggest = 0;
    for (int k = 0; k < nt; ++k) {
        const int fl = __builtin_amdgcn_readlane(q0.w, k), wh = __builtin_amdgcn_readlane(q0.z, k), nm = __builtin_amdgcn_readlane(q1.w, k);
        if (fl & WS_STAGE) biggest = max(biggest, (wh >> 16) * (nm & 0xff));
    }
    const int nslot = biggest * 16 <= ARENA / 3 ? 3 : 2, depth = nslot - 1;
    const uint32_t slot_bytes = (uint32_t)(ARENA / nslot) & ~15u;
    const uint32_t arena = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s_arena;
    auto slot_of = [&](int k) { return (uint32_t)(k >= nslot ? k - nslot : k) * slot_bytes; };       // k < 2 * nslot (k <= 3)
    // request the copy of tile k's rectangle: chunk e = 256 p + tid of the rectangle per pass; a pass is issued by the waves that carry chunks of it (wave-uniform)
    auto stage = [&](int k) {
        const int fl = __builtin_amdgcn_readlane(q0.w, k);
        if (!(fl & WS_STAGE)) return;
        const int rx0 = __builtin_amdgcn_readlane(q0.x, k), ry0 = __builtin_amdgcn_readlane(q0.y, k), wh = __builtin_amdgcn_readlane(q0.z, k), nm = __builtin_amdgcn_readlane(q1.w, k);
        const int rows = wh >> 16, nch = nm & 0xff, total = rows * nch;
        const uint32_t mg = (uint32_t)nm >> 8, a0 = (3u * (uint32_t)rx0) & ~15u, base = slot_of(k);
#pragma unroll
        for (int p = 0; p < (WS_BUF_CMAP + 4095) / 4096; ++p)
            if (256 * p + 64 * wave < total) {
                const uint32_t e = min(256u * p + (uint32_t)tid, (uint32_t)total - 1u), row = __umul24(e, mg) >> 16, chunk = e - row * (uint32_t)nch;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(s_arena + base + p * 4096 + wave * 1024), 16,
                                                         __umul24((uint32_t)ry0 + row, pitch) + a0 + 16u * chunk, 0, 0, 0);
            }
    };
    auto coords = [&](int k) { return __builtin_amdgcn_raw_buffer_load_b128(rc, __umul24((uint32_t)yc, cpitch) + 4u * (uint32_t)((WS_NT * sx + k) * WT_W + 4 * lx), 0, 0); };
    // prologue: tile 0's rectangle, then its coordinates; tile 1's rectangle behind them when three slots are in use
    u32x4_t cm_next;
    stage(0);
    cm_next = coords(0);
    if (depth == 2 && nt > 1) stage(1);
    WS_STAMP(2);
    // A tile's stores are issued one tile late, behind the next tile's barrier: the wait in front of a tile's coordinates is for everything this
    // wave has in flight (the compiler cannot count the copies and stores of a loop whose tiles differ: vmcnt(0)), and stores issued just before it
    // would put a write's whole round trip, ~1 700 cycles, in front of every tile (tools/trace_warp.py)
    uint32_t p0 = 0, p1 = 0, p2 = 0, pmk = 0;
    int pt0 = 0;
    bool plive = false;
    auto flush = [&]() {
        if (!plive) return;
        const int x0 = pt0 - xshift;
        if (x0 >= 0 && x0 + 4 <= dw) {
            u32x3_a4 w;
            w.x = p0; w.y = p1; w.z = p2;
            __builtin_amdgcn_raw_buffer_store_b96(w, rd, drow + 3u * (uint32_t)pt0, 0, 0);
            if (a.mask) __builtin_amdgcn_raw_buffer_store_b32(pmk, rm, mrow + (uint32_t)pt0, 0, 0);
        } else {
            uint8_t *dp = a.dst + (ptrdiff_t)y * (ptrdiff_t)a.dpitch + (ptrdiff_t)x0 * 3;
            const uint32_t ww[3] = {p0, p1, p2};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (x0 + i < 0 || x0 + i >= dw) continue;
#pragma unroll
                for (int c = 0; c < 3; ++c) { const int bidx = 3 * i + c; dp[bidx] = (uint8_t)(ww[bidx >> 2] >> (8 * (bidx & 3))); }
                if (a.mask) a.mask[(ptrdiff_t)y * (ptrdiff_t)a.mpitch + x0 + i] = (uint8_t)(pmk >> (8 * i));
            }
        }
        plive = false;
    };
    for (int k = 0; k < nt; ++k) {
        const int rx0 = __builtin_amdgcn_readlane(q0.x, k), ry0 = __builtin_amdgcn_readlane(q0.y, k), fl = __builtin_amdgcn_readlane(q0.w, k);
        const int ux0 = __builtin_amdgcn_readlane(q1.x, k), uy0 = __builtin_amdgcn_readlane(q1.y, k), nm = __builtin_amdgcn_readlane(q1.w, k);
        const bool staged = (fl & WS_STAGE) != 0;
        const int t0 = (WS_NT * sx + k) * WT_W + 4 * lx, x0 = t0 - xshift;
        const bool live = staged && row_live && x0 < dw;
        // -- 1. this lane's coordinates of tile k: their first use is where the compiler waits for them -- and with them, in order, for tile k's rectangle
        const u32x4_t cm = cm_next;
        const uint32_t bxr[4] = {cm.x & 0x1fffu, cm.y & 0x1fffu, cm.z & 0x1fffu, cm.w & 0x1fffu};
        const uint32_t byr[4] = {(cm.x >> 13) & 0x7fffu, (cm.y >> 13) & 0x7fffu, (cm.z >> 13) & 0x7fffu, (cm.w >> 13) & 0x7fffu};
        uint32_t mk = 0xffffffffu;
        if (fl & WS_BORDER) mk = ((cm.x >> 28) & 1u) * 0xffu | ((cm.y >> 28) & 1u) * 0xff00u | ((cm.z >> 28) & 1u) * 0xff0000u | ((cm.w >> 28) & 1u) * 0xff000000u;
        asm volatile("" ::"v"(bxr[0]), "v"(byr[3]) : "memory");      // (the unpacking stays in front of the barrier)
        WS_STAMP(4 + 6 * k);
        // -- 2. everybody's chunks of tile k's rectangle have landed, everybody is done with tile k - 1's slot
        __builtin_amdgcn_s_barrier();
        WS_STAMP(5 + 6 * k);
        // -- 2b. the copy that goes into the freed slot (tile k + depth) and the next tile's coordinates, in the order of the tiles: a tile's
        // coordinates are requested behind ITS rectangle and in front of the next tile's (the issue order DMA(0) c(0) DMA(1) c(1) DMA(2) ... is what
        // makes "the coordinates of tile k have arrived" mean "the rectangle of tile k has arrived, the one of tile k + 1 may still be on its way")
        if (depth == 1) {
            if (k + 1 < nt) { stage(k + 1); cm_next = coords(k + 1); }
        } else {
            if (k + 1 < nt) cm_next = coords(k + 1);
            if (k + 2 < nt) stage(k + 2);
        }
        flush();                  // tile k - 1's pixels
        WS_STAMP(6 + 6 * k);
        uint32_t o0 = 0, o1 = 0, o2 = 0;
        if (live) {
            // -- 3. taps from LDS, fixed-point bilinear
            const uint32_t c0 = (3u * (uint32_t)rx0) & 15u, pitchl = 16u * (uint32_t)(nm & 0xff);
            Px3 v[4];
            if (!(fl & WS_BORDER)) taps_interior4(arena + slot_of(k), pitchl, c0, bxr, byr, v);
            else {
                const uint8_t *tile = s_arena + slot_of(k);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = taps_reflect(tile, pitchl, c0, bxr[i], byr[i], ux0, uy0, rx0, ry0, sw, sh);
            }
            asm volatile("" ::"v"(v[0].b), "v"(v[3].r) : "memory");
            WS_STAMP(7 + 6 * k);
            // -- 4. exposure compensation and packing
            if (GAIN) {
                float g[4][3];
                if (GAIN >= 2) {
                    const float b1 = gb1, b0 = 1.f - b1;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if (c < GCN) {
                            float4 t0g, t1g;
                            lds_read_2x128(s_gain + (c * WT_GAIN_ROWS + grow0) * 256 + WT_W * k + 4 * lx, s_gain + (c * WT_GAIN_ROWS + grow1) * 256 + WT_W * k + 4 * lx, t0g, t1g);
                            g[0][c] = t0g.x * b0 + t1g.x * b1; g[1][c] = t0g.y * b0 + t1g.y * b1; g[2][c] = t0g.z * b0 + t1g.z * b1; g[3][c] = t0g.w * b0 + t1g.w * b1;
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) g[i][c] = g[i][0];
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { g[i][0] = d.gain.g[0]; g[i][1] = d.gain.g[1]; g[i][2] = d.gain.g[2]; }
                }
#define GPX(i, c) ((float)((c == 0 ? v[i].b : c == 1 ? v[i].g : v[i].r) >> 16 & 0xffu) * g[i][c])
                o0 = pack_u8_rne(GPX(0, 0), 0, o0); o0 = pack_u8_rne(GPX(0, 1), 1, o0); o0 = pack_u8_rne(GPX(0, 2), 2, o0); o0 = pack_u8_rne(GPX(1, 0), 3, o0);
                o1 = pack_u8_rne(GPX(1, 1), 0, o1); o1 = pack_u8_rne(GPX(1, 2), 1, o1); o1 = pack_u8_rne(GPX(2, 0), 2, o1); o1 = pack_u8_rne(GPX(2, 1), 3, o1);
                o2 = pack_u8_rne(GPX(2, 2), 0, o2); o2 = pack_u8_rne(GPX(3, 0), 1, o2); o2 = pack_u8_rne(GPX(3, 1), 2, o2); o2 = pack_u8_rne(GPX(3, 2), 3, o2);
#undef GPX
            } else {
                const uint32_t t0p = __builtin_amdgcn_perm(v[0].g, v[0].b, 0x0c0c0602u), u0p = __builtin_amdgcn_perm(v[1].b, v[0].r, 0x0c0c0602u);
                const uint32_t t1p = __builtin_amdgcn_perm(v[1].r, v[1].g, 0x0c0c0602u), u1p = __builtin_amdgcn_perm(v[2].g, v[2].b, 0x0c0c0602u);
                const uint32_t t2p = __builtin_amdgcn_perm(v[3].b, v[2].r, 0x0c0c0602u), u2p = __builtin_amdgcn_perm(v[3].r, v[3].g, 0x0c0c0602u);
                o0 = __builtin_amdgcn_perm(u0p, t0p, 0x05040100u);
                o1 = __builtin_amdgcn_perm(u1p, t1p, 0x05040100u);
                o2 = __builtin_amdgcn_perm(u2p, t2p, 0x05040100u);
            }
            // -- 5. mask preparation (sde.py:1760-1772) unless this row's share of the strip lies inside the seam mask
            if (d.prep && mk && !seam_in) {
                MaskPrep mp;
                mp.dil = d.dil; mp.dpitch = d.dil_pitch;
                mp.xo = d.lin; mp.xc = d.lin + dw4; mp.yo = d.lin + 2 * dw4; mp.yc = mp.yo + dh;
                mp.flags = nullptr; mp.fgx = 0;
                mk &= seam_mask4(mp, y, t0);
            }
        }
        if (FAR && (fl & WS_FAR) && row_live && x0 < dw && a.mask) {
            if (x0 >= 0 && x0 + 4 <= dw) __builtin_amdgcn_raw_buffer_store_b32(0u, rm, mrow + (uint32_t)t0, 0, 0);
            else
                for (int i = 0; i < 4; ++i)
                    if (x0 + i >= 0 && x0 + i < dw) a.mask[(ptrdiff_t)y * (ptrdiff_t)a.mpitch + x0 + i] = 0;
        }
        asm volatile("" ::"v"(o0), "v"(o2), "v"(mk) : "memory");
        WS_STAMP(8 + 6 * k);
        // -- 6. the stores wait for the next tile's barrier (or the end of the strip)
        p0 = o0; p1 = o1; p2 = o2; pmk = mk; pt0 = t0; plive = live;
        WS_STAMP(9 + 6 * k);
    }
    flush();
    WS_STAMP(3);
    WS_TRACE_OUT;
}

// nearest-neighbour mask for non-separable projections (src is the all-255 mask of sde.py:1739)
__global__ __launch_bounds__(256) void k_warp_generic_with_mask(Projector p, SrcView src, uint8_t *dst, size_t dpitch, uint8_t *mask,
                                                                size_t mpitch, int dw, int dh, int tlx, int tly, int border)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    float fx, fy;
    map_backward(p, (float)(x + tlx), (float)(y + tly), fx, fy);
    uint32_t v = bilinear_u8c3(src, fx, fy, border);
    uint8_t *d = dst + (size_t)y * dpitch + (size_t)x * 3;
    d[0] = (uint8_t)v;
    d[1] = (uint8_t)(v >> 8);
    d[2] = (uint8_t)(v >> 16);
    if (mask) {
        int mx = sat_s16(cv_round(fx)), my = sat_s16(cv_round(fy));
        mask[(size_t)y * mpitch + x] = ((unsigned)mx < (unsigned)src.w && (unsigned)my < (unsigned)src.h) ? 255 : 0;
    }
}

// float frames (BASELINE config 5): image (INTER_LINEAR, float weights in OpenCV's order) and validity mask from one map evaluation
__global__ __launch_bounds__(256) void k_warp_generic_f32c3_with_mask(Projector p, SrcView src, float *dst, size_t dpitch, uint8_t *mask, size_t mpitch, int dw,
                                                                      int dh, int tlx, int tly, int border)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    float fx, fy;
    map_backward(p, (float)(x + tlx), (float)(y + tly), fx, fy);
    float out[3];
    remap_pixel<float, 3>(src, fx, fy, SSP_INTER_LINEAR, border, out);
    float *d = (float *)((char *)dst + (size_t)y * dpitch) + (size_t)x * 3;
    d[0] = out[0]; d[1] = out[1]; d[2] = out[2];
    int mx = sat_s16(cv_round(fx)), my = sat_s16(cv_round(fy));
    mask[(size_t)y * mpitch + x] = ((unsigned)mx < (unsigned)src.w && (unsigned)my < (unsigned)src.h) ? 255 : 0;
}

// ---- host side ---------------------------------------------------------------------------------------------------
namespace ssp {

// cvRound(f) <= n-1  <=>  f <= n-0.5, except that the tie n-0.5 rounds to the even neighbour: exclusive when n is even
static float nearest_hi(int n)
{
    float t = (float)n - 0.5f;
    return (n & 1) ? t : nextafterf(t, 0.f);
}

static int check_kr(const float K[9], const float R[9])
{
    SSP_REQUIRE(K && R, "K and R must be 3x3 float32 (CV_32F) arrays");
    return 0;
}

static double warp_algo_bytes(const ssp_image *src, int dw, int dh, bool with_mask)
{
    double c = depth_size(src->depth);
    return c * src->cn * (double)src->w * src->h + (c * src->cn + (with_mask ? 1 : 0)) * (double)dw * dh;
}

int warp_table_cols(int dw);
int warp_flag_cols(int dw);

// image + (optional) mask in one pass.  interp/border as cv2; mask only with u8c3 LINEAR.
int warp_launch(const Projector &p, const ssp_image *src, const int roi[4], int interp, int border, ssp_image *dst, ssp_image *mask)
{
    const int dw = roi[2], dh = roi[3];
    SrcView sv = {(const uint8_t *)src->data, src->pitch, src->w, src->h};
    if (interp == SSP_INTER_AREA) interp = SSP_INTER_LINEAR;  // cv::remap does the same
    SSP_REQUIRE(interp == SSP_INTER_NEAREST || interp == SSP_INTER_LINEAR, "warp: interpolation %d not supported by remap here", interp);
    SSP_REQUIRE(border >= 0 && border <= 4, "warp: border mode %d not supported", border);
    dim3 grid((dw + 63) / 64, (dh + 3) / 4), block(256);
    const bool u8c3lin = src->depth == SSP_U8 && src->cn == 3 && interp == SSP_INTER_LINEAR;
    // the fused kernel computes source byte offsets with 24-bit multiplies in 32 bits: rows < 2^15 (int16 coordinates anyway),
    // pitch < 2^24, frame < 4 GiB; anything larger takes the generic kernel
    const bool fits_fast = src->h <= 32767 && src->pitch < ((size_t)1 << 24) && (size_t)src->pitch * (size_t)src->h < ((size_t)1 << 32);
    if (u8c3lin && is_separable(p.kind) && fits_fast) {
        float *tab = nullptr;
        const size_t dw4 = (size_t)warp_table_cols(dw);  // column tables padded: the kernel reads them as float4
        SSP_TRY(pool_alloc(sizeof(float) * 2 * (dw4 + dh), (void **)&tab));
        SepArgs a;
        a.src = sv;
        a.dst = (uint8_t *)dst->data; a.dpitch = dst->pitch;
        a.mask = mask ? (uint8_t *)mask->data : nullptr; a.mpitch = mask ? mask->pitch : 0;
        a.dw = dw; a.dh = dh;
        a.colS = tab; a.colC = tab + dw4; a.rowA = tab + 2 * dw4; a.rowB = a.rowA + dh;
        memcpy(a.kr, p.k_rinv, sizeof a.kr);
        a.border = border;
        a.hix = nearest_hi(src->w); a.hiy = nearest_hi(src->h);
        a.xshift = 0;
        {
            ProfileScope ps("warp_tables", 0);
            hipLaunchKernelGGL(k_sep_tables, dim3(((int)dw4 + dh + 255) / 256), dim3(256), 0, stream(), p.kind, p.scale, roi[0], roi[1], (int)dw4, dh,
                               (float *)a.colS, (float *)a.colC, (float *)a.rowA, (float *)a.rowB);
        }
        {
            ProfileScope ps("warp_fused", warp_algo_bytes(src, dw, dh, mask != nullptr));
            dim3 g2((dw + 255) / 256, (dh + 3) / 4);
            hipLaunchKernelGGL(k_warp_sep_u8c3, g2, block, 0, stream(), a);
        }
        pool_free(tab);
    } else if (u8c3lin && mask) {
        ProfileScope ps("warp_generic", warp_algo_bytes(src, dw, dh, true));
        hipLaunchKernelGGL(k_warp_generic_with_mask, grid, block, 0, stream(), p, sv, (uint8_t *)dst->data, dst->pitch, (uint8_t *)mask->data,
                           mask->pitch, dw, dh, roi[0], roi[1], border);
    } else if (src->depth == SSP_F32 && src->cn == 3 && interp == SSP_INTER_LINEAR && is_separable(p.kind)) {
        // float frames, separable projection: table-based map like the 8-bit kernel
        float *tab = nullptr;
        const size_t dw4 = (size_t)warp_table_cols(dw);
        SSP_TRY(pool_alloc(sizeof(float) * 2 * (dw4 + dh), (void **)&tab));
        SepArgs a;
        memset(&a, 0, sizeof a);
        a.src = sv;
        a.dst = (uint8_t *)dst->data; a.dpitch = dst->pitch;
        a.mask = mask ? (uint8_t *)mask->data : nullptr; a.mpitch = mask ? mask->pitch : 0;
        a.dw = dw; a.dh = dh;
        a.colS = tab; a.colC = tab + dw4; a.rowA = tab + 2 * dw4; a.rowB = a.rowA + dh;
        memcpy(a.kr, p.k_rinv, sizeof a.kr);
        a.border = border;
        a.hix = nearest_hi(src->w); a.hiy = nearest_hi(src->h);
        {
            ProfileScope ps("warp_tables", 0);
            hipLaunchKernelGGL(k_sep_tables, dim3(((int)dw4 + dh + 255) / 256), dim3(256), 0, stream(), p.kind, p.scale, roi[0], roi[1], (int)dw4, dh,
                               (float *)a.colS, (float *)a.colC, (float *)a.rowA, (float *)a.rowB);
        }
        {
            ProfileScope ps("warp_fused", warp_algo_bytes(src, dw, dh, mask != nullptr));
            hipLaunchKernelGGL(k_warp_sep_f32c3, dim3((dw + 255) / 256, (dh + 3) / 4), block, 0, stream(), a);
        }
        pool_free(tab);
    } else if (mask && src->depth == SSP_F32 && src->cn == 3 && interp == SSP_INTER_LINEAR) {
        ProfileScope ps("warp_generic", warp_algo_bytes(src, dw, dh, true));
        hipLaunchKernelGGL(k_warp_generic_f32c3_with_mask, grid, block, 0, stream(), p, sv, (float *)dst->data, dst->pitch, (uint8_t *)mask->data, mask->pitch, dw,
                           dh, roi[0], roi[1], border);
    } else {
        SSP_REQUIRE(!mask, "warp: fused mask output needs an 8UC3 or 32FC3 source with INTER_LINEAR");
        ProfileScope ps("warp_generic", warp_algo_bytes(src, dw, dh, false));
#define LAUNCH_GENERIC(T, CN)                                                                                                     \
    hipLaunchKernelGGL((k_warp_generic<T, CN>), grid, block, 0, stream(), p, sv, dst->data, dst->pitch, dw, dh, roi[0], roi[1], interp, \
                       border, (float *)nullptr, (float *)nullptr)
        if (src->depth == SSP_U8 && src->cn == 1) LAUNCH_GENERIC(uint8_t, 1);
        else if (src->depth == SSP_U8 && src->cn == 3) LAUNCH_GENERIC(uint8_t, 3);
        else if (src->depth == SSP_F32 && src->cn == 1) LAUNCH_GENERIC(float, 1);
        else if (src->depth == SSP_F32 && src->cn == 3) LAUNCH_GENERIC(float, 3);
        else SSP_FAIL(SSP_ERR_ARG, "warp: unsupported source type (depth %d, %d channels); 8U/32F with 1 or 3 channels", src->depth, src->cn);
#undef LAUNCH_GENERIC
    }
    SSP_HIP(hipGetLastError());
    return 0;
}


// ---- batched warp of n frames (composer) ---------------------------------------------------------------------------------
size_t warp_batch_desc_size() { return sizeof(WarpBatchDesc); }
// entries per column table: the roi width plus room for the alignment shift, as whole float4 groups
int warp_table_cols(int dw) { return (int)align_up((size_t)dw, 4) + 4; }
// 256-column segments per row (tiles of the LX = 64 launch, alignment shift included), and the size of a frame's `lin` buffer in ints:
// xo | xc | yo | yc | seam-interior flags
int warp_flag_cols(int dw) { return (dw + 3 + 255) / 256; }
// records of the LDS-staged variant: one int4 per 64 x 16 tile
size_t warp_tile_bytes(int dw, int dh) { return 2 * sizeof(int4) * (size_t)warp_tiles_x(dw) * warp_tiles_y(dh); }
size_t warp_lin_ints(int dw, int dh, int seam_h) { return 2 * ((size_t)warp_table_cols(dw) + dh) + (size_t)seam_h * warp_flag_cols(dw); }
// items of the prep launch for one frame with mask preparation: tables, INTER_LINEAR_EXACT tables, dilation, flags
int warp_prep_items(int dw, int dh, int seam_w, int seam_h) { return 2 * (warp_table_cols(dw) + dh) + seam_w * seam_h + seam_h * warp_flag_cols(dw); }

// fills one descriptor; tab/lin/dil are caller-owned persistent device buffers
// `roi` is the part's rectangle (absolute warped coordinates); full_dw / x_off place it inside its frame's whole roi (a whole frame: roi[2], 0)
// words of a part's coordinate plane (head + one word per pixel of every tile row); 0: the part is too large for 32-bit offsets
size_t warp_cmap_words(int dw, int dh)
{
    const size_t px = (size_t)warp_tiles_x(dw) * WT_W * (size_t)dh;
    return px < ((size_t)1 << 30) ? WB_CMAP_HEAD + px : 0;
}
// the head of a coordinate plane: the part's projector (camera set), for the kernels that evaluate the map itself (records, cmap, rest)
int warp_cmap_set_projector(void *cmap, const Projector &p)
{
    SSP_HIP(hipMemcpyAsync(cmap, &p, sizeof p, hipMemcpyHostToDevice, stream()));     // (pageable source: the runtime stages it before returning)
    return 0;
}

void warp_batch_fill(void *desc_, const Projector &p, const ssp_image *src, const int roi[4], int full_dw, int x_off, int border, uint8_t *dst, size_t dst_pitch, uint8_t *mask,
                     size_t mask_pitch, int xshift, float *tab, int prep, const ssp_image *seam, ssp_image *dil, int *lin, void *tiles, void *cmap)
{
    WarpBatchDesc &d = *(WarpBatchDesc *)desc_;
    memset(&d, 0, sizeof d);
    d.tiles = (int4 *)tiles;
    const int dw = roi[2], dh = roi[3];
    const int dw4 = warp_table_cols(dw);
    d.a.xshift = xshift & 3;
    d.a.sdata = (const uint8_t *)src->data; d.a.spitch = (uint32_t)src->pitch; d.a.sw = src->w; d.a.sh = src->h;
    d.a.dst = dst; d.a.dpitch = (uint32_t)dst_pitch;
    d.a.mask = mask; d.a.mpitch = (uint32_t)mask_pitch;
    d.a.dw = dw; d.a.dh = dh;
    memcpy(d.a.kr, p.k_rinv, sizeof d.a.kr);
    d.a.border = border;
    d.a.hix = nearest_hi(src->w); d.a.hiy = nearest_hi(src->h);
    d.kind = p.kind; d.scale = p.scale; d.tlx = roi[0]; d.tly = roi[1]; d.dw4 = dw4;
    d.full_dw = full_dw; d.x_off = x_off;
    d.tab = tab;
    d.cmap = (uint32_t *)cmap;
    d.prep = prep;
    if (prep) {
        d.seam = (const uint8_t *)seam->data; d.seam_pitch = (uint32_t)seam->pitch; d.seam_w = seam->w; d.seam_h = seam->h;
        d.dil = (uint8_t *)dil->data; d.dil_pitch = (uint32_t)dil->pitch;
        d.lin = lin;
        d.fgx = warp_flag_cols(dw);
        d.has_flags = 1;   // behind the tables in `lin`: the caller sizes it with warp_lin_ints()
    }
}

// exposure compensation of one frame: tabs = dw4 ints | dw4 floats | dh ints | dh floats (caller-owned, kind 2 only)
void warp_batch_set_gain(void *desc_, int kind, const float g[3], const float *d_map, int gw, int gh, int gcn, void *tabs)
{
    WarpBatchDesc &d = *(WarpBatchDesc *)desc_;
    memset(&d.gain, 0, sizeof d.gain);
    d.gain.kind = kind;
    if (kind == 1) memcpy(d.gain.g, g, sizeof d.gain.g);
    if (kind == 2) {
        d.gain.gm = d_map; d.gain.gw = gw; d.gain.gh = gh; d.gain.gcn = gcn;
        d.gain.tabs = (int *)tabs;
    }
}

// What a composer learns about its (static) geometry from its first panorama: how many tiles the strip kernel cannot stage.  When they are
// few, later panoramas let the strip kernel do them inline (per-pixel general path) and skip the rest launch altogether.
void warp_rest_plan_release(WarpRestPlan *p)
{
    if (p->h_count) (void)hipHostFree(p->h_count);
    if (p->ev) (void)hipEventDestroy(p->ev);
    if (p->d_list) pool_free(p->d_list);
    *p = WarpRestPlan();
}

// The count of the first panorama's rest list has arrived: take it over, and refuse a count beyond the list's capacity.  The kernels clamp
// (a slot >= capacity is not written, the rest kernel walks min(count, capacity) entries) so that an indexing bug cannot fault the GPU --
// but a clamped list means dropped tiles, i.e. stale pixels in the blender's planes, so the host turns it into an error instead of a picture.
// (gpurun_out/r2h/bench.err, round 2: the prep launch that zeroes rest[0] was issued from an args copy made BEFORE args.rest was assigned,
// the counter started from pool garbage and `rest[1 + atomicAdd(rest, 1)] = tile` wrote past the allocation -> "Memory access fault by GPU";
// fixed in 93e8185 by launching after the assignment, with the clamps added then.)
int warp_rest_plan_settle(WarpRestPlan *plan, bool wait)
{
    if (plan->state != 1) return 0;
    if (wait) SSP_HIP(hipEventSynchronize(plan->ev));
    else if (hipEventQuery(plan->ev) != hipSuccess) return 0;
    plan->state = 2;
    plan->count = plan->h_count[0]; plan->misfit = plan->h_count[1]; plan->has_far = plan->h_count[2] != 0;
    if (plan->count < 0 || plan->count > plan->capacity) {
        const int bad = plan->count;
        plan->state = 0; plan->count = 0;
        if (plan->d_list) { pool_free(plan->d_list); plan->d_list = nullptr; }
        SSP_FAIL(SSP_ERR_STATE, "fused warp: the rest list of the previous panorama holds %d tiles but the launch has only %d (list counter corrupted: tiles were dropped)", bad, plan->capacity);
    }
    return 0;
}

// float frames: the prep launch (tables, INTER_LINEAR_EXACT tables, dilated seam masks, interior flags) while the geometry is new, then ONE
// warp launch for all parts (descriptors as filled by warp_batch_fill with xshift 0; float planes)
int warp_batch_launch_f32(const void *h_descs, int n, int max_dw, int max_dh, int max_prep_items, double algo_bytes, double prep_bytes, WarpRestPlan *plan)
{
    const WarpBatchDesc *hd = (const WarpBatchDesc *)h_descs;
    for (int base = 0; base < n; base += WARP_MAXB) {
        const int cnt = std::min(WARP_MAXB, n - base);
        WarpBatchArgs args;
        memset(&args, 0, sizeof args);
        memcpy(args.d, hd + base, sizeof(WarpBatchDesc) * cnt);
        for (int i = 0; i < cnt; ++i) SSP_REQUIRE(args.d[i].tab != nullptr && args.d[i].a.mask != nullptr && args.d[i].a.xshift == 0, "float warp: part %d has no tables / mask plane", base + i);
        bool need_prep = true;
        if (plan && base == 0 && cnt == n) {
            std::vector<char> key(sizeof(WarpBatchDesc) * (size_t)cnt);
            memcpy(key.data(), args.d, key.size());
            for (int i = 0; i < cnt; ++i) {
                WarpBatchDesc &kd = ((WarpBatchDesc *)key.data())[i];
                kd.a.sdata = nullptr; kd.a.spitch = 0; kd.a.dst = nullptr; kd.a.dpitch = 0; kd.a.mask = nullptr; kd.a.mpitch = 0;
            }
            if (plan->prep_key == key) need_prep = false;
            else plan->prep_key.swap(key);
        } else if (plan) plan->prep_key.clear();
        const double share = (double)cnt / n;
        if (need_prep) {
            ProfileScope ps("warp_prep", prep_bytes * share);
            hipLaunchKernelGGL(k_warp_prep_batch, dim3((max_prep_items + 255) / 256, 1, cnt), dim3(256), 0, stream(), args);
        }
        {
            ProfileScope ps("warp_fused", algo_bytes * share);
            hipLaunchKernelGGL(k_warp_f32_batch, dim3((max_dw + 255) / 256, (max_dh + 3) / 4, cnt), dim3(256), 0, stream(), args);
        }
    }
    SSP_HIP(hipGetLastError());
    return 0;
}

int warp_batch_launch(const void *h_descs, int n, int max_dw, int max_dh, int max_prep_items, double algo_bytes, double prep_bytes, WarpRestPlan *plan, int far_px)
{
    // SSP_WARP_REST=inline (tools/fuzz_sweep.py): once the first panorama has shown that no tile misses for its gain rows, every non-stageable
    // tile takes the strip kernel's inline path however many there are; default: inline only when they are few
    static const bool force_inline = getenv("SSP_WARP_REST") && !strcmp(getenv("SSP_WARP_REST"), "inline");
    const WarpBatchDesc *hd = (const WarpBatchDesc *)h_descs;
    for (int base = 0; base < n; base += WARP_MAXB) {
        const int cnt = std::min(WARP_MAXB, n - base);
        WarpBatchArgs args;
        memset(&args, 0, sizeof args);
        memcpy(args.d, hd + base, sizeof(WarpBatchDesc) * cnt);
        const double share = (double)cnt / n;
        bool gain = false;
        for (int i = 0; i < cnt; ++i) {
            gain = gain || args.d[i].gain.kind != 0;
            SSP_REQUIRE((args.d[i].tab != nullptr || args.d[i].cmap != nullptr) && args.d[i].tiles != nullptr, "fused warp: frame %d has no table / tile-record buffer", base + i);
            SSP_REQUIRE((args.d[i].cmap != nullptr) == (args.d[0].cmap != nullptr), "fused warp: the parts of one launch must all have coordinate planes or none");
        }
        const bool cmap_mode = args.d[0].cmap != nullptr;
        const int gxt = warp_tiles_x(max_dw), gyt = warp_tiles_y(max_dh), nt = gxt * gyt * cnt;
        // exposure compensation mode of the batch (one compensator feeds every frame): 0 none, 1 gains, 2 / 3 gain map with 1 / 3 channels
        int gmode = 0;
        for (int i = 0; i < cnt; ++i) {
            const WarpBatchGain &g = args.d[i].gain;
            const int m = g.kind == 0 ? 0 : g.kind == 1 ? 1 : (g.gcn == 3 ? 3 : (g.gcn == 1 ? 2 : -1));
            gmode = i == 0 ? m : (gmode == m ? m : -1);
        }
        SSP_REQUIRE(gmode >= 0, "fused warp: the frames of one launch carry different kinds of gains (the composer applies such compensators in a separate pass)");
        // rest policy of this launch
        const bool whole = base == 0 && cnt == n;          // the plan describes single-batch panoramas only
        bool inline_rest = false, list_known = false;      // list_known: the composer kept the list its first panorama produced
        if (plan && whole) {
            // everything that decides whether a tile can be staged: launch shape, every frame's roi and gain-map shape (FNV-1a)
            unsigned long long sig = 1469598103934665603ULL;
            auto mix = [&sig](long long v) { for (int b = 0; b < 8; ++b) { sig ^= (unsigned long long)(v >> (8 * b)) & 0xffu; sig *= 1099511628211ULL; } };
            mix(nt); mix(gmode); mix(max_dw); mix(max_dh); mix(far_px); mix(cmap_mode);
            for (int i = 0; i < cnt; ++i) {
                const WarpBatchDesc &dd = args.d[i];
                mix(dd.a.dw); mix(dd.a.dh); mix(dd.full_dw); mix(dd.x_off); mix(dd.a.sw); mix(dd.a.sh); mix(dd.a.border); mix(dd.gain.kind); mix(dd.gain.gw); mix(dd.gain.gh); mix(dd.gain.gcn);
                for (int q = 0; q < 9; ++q) mix(__builtin_bit_cast(int, dd.a.kr[q]));
            }
            if (plan->sig != (long long)sig || plan->state == 0) {
                plan->sig = (long long)sig; plan->state = 0;
                if (plan->d_list) { pool_free(plan->d_list); plan->d_list = nullptr; }
            }
            SSP_TRY(warp_rest_plan_settle(plan, false));
            inline_rest = !cmap_mode && plan->state == 2 && !plan->misfit && (force_inline || plan->count <= std::max(64, nt / 256));
            list_known = plan->state == 2 && !inline_rest && plan->d_list != nullptr;
        }
        int *rest = nullptr;
        if (list_known) {
            args.rest = plan->d_list; args.rest_cap = nt; args.rest_known = 1;
        } else if (!inline_rest) {
            SSP_TRY(pool_alloc(sizeof(int) * ((size_t)nt + 3), (void **)&rest));
            args.rest = rest;      // rest[0] and the misfit flag behind the list are zeroed by the prep launch, the list is written by the next one
            args.rest_cap = nt;
        }
        // Everything the prep launch writes -- trigonometry tables, resize tables, the dilated seam mask and its interior flags, gain-map
        // coordinates -- depends on the composer's fixed geometry only, never on the frames.  Once it has run for exactly these descriptors
        // (compared with the per-panorama fields blanked) and no list counter needs zeroing, later panoramas skip it.
        bool need_prep = true;
        if (plan && whole && (inline_rest || list_known)) {
            std::vector<char> key(sizeof(WarpBatchDesc) * (size_t)cnt);
            memcpy(key.data(), args.d, key.size());
            for (int i = 0; i < cnt; ++i) {
                WarpBatchDesc &kd = ((WarpBatchDesc *)key.data())[i];
                kd.a.sdata = nullptr; kd.a.spitch = 0;
                kd.a.dst = nullptr; kd.a.dpitch = 0; kd.a.mask = nullptr; kd.a.mpitch = 0;
                kd.gain.g[0] = kd.gain.g[1] = kd.gain.g[2] = 0.f; kd.gain.gm = nullptr;
            }
            if (plan->prep_key == key) need_prep = false;
            else plan->prep_key.swap(key);
        } else if (plan) plan->prep_key.clear();
        if (need_prep) {
            // (after args.rest is set: this launch zeroes the list counter -- see warp_rest_plan_settle)
            ProfileScope ps("warp_prep", prep_bytes * share);
            hipLaunchKernelGGL(k_warp_prep_batch, dim3((max_prep_items + 255) / 256, 1, cnt), dim3(256), 0, stream(), args);
            hipLaunchKernelGGL(k_warp_records_batch, dim3((nt + 255) / 256), dim3(256), 0, stream(), args, gxt, gyt, nt, nt);     // reads the tables just built
            if (cmap_mode) {
                ProfileScope pc("warp_cmap", 0);
                hipLaunchKernelGGL(k_warp_cmap_batch, dim3(nt), dim3(256), 0, stream(), args, gxt, gyt, nt);                      // coordinate planes; may take tiles off the staged path
            }
            hipLaunchKernelGGL(k_warp_records_far, dim3((nt + 255) / 256), dim3(256), 0, stream(), args, gxt, gyt, nt, nt, far_px);   // far tiles, rest list
        }
        const int sgx = (gxt + WS_NT - 1) / WS_NT, ns = sgx * gyt * cnt;
        const uint64_t pi = (uint64_t)sgx * gyt;
        const uint32_t m_per_img = (pi > 1 && (uint64_t)ns * pi < (1ULL << 32)) ? (uint32_t)((1ULL << 32) / pi) + 1u : 0u;
        const uint32_t m_sgx = (sgx > 1 && pi * (uint64_t)sgx < (1ULL << 32)) ? (uint32_t)((1ULL << 32) / (uint64_t)sgx) + 1u : 0u;
        {
            // XCD-aware strip order (xcd_remap = 1): every XCD (own L2) gets a contiguous run of strips, which brings the source reads down from
            // 2x to 1.02x of the frame (PMC: 314 -> 154 MB per 6 frames, profiles/r01_*)
            ProfileScope ps("warp_fused", algo_bytes * share);
            // far tiles exist (or are not known not to exist yet): the variant that writes their masks; else the kernel without that path
            const bool with_far = far_px > 0 && !(plan && whole && plan->state == 2 && !plan->has_far);
            // strip order over the XCDs: chunks of four strip rows (SSP_WARP_XCD_CHUNK: strips per chunk; 1 = one contiguous eighth per XCD, rounds 1-3)
            static const int chunk_env = getenv("SSP_WARP_XCD_CHUNK") ? atoi(getenv("SSP_WARP_XCD_CHUNK")) : -1;
            const int xcd_order = chunk_env >= 0 ? chunk_env : std::max(2, 4 * sgx);
            const int ns_launch = xcd_order > 1 ? (ns + 8 * xcd_order - 1) / (8 * xcd_order) * (8 * xcd_order) : ns;
#define LAUNCH_STRIP2(G, F) do { if (cmap_mode) hipLaunchKernelGGL((k_warp_strip_planes<G, F>), dim3(ns_launch), dim3(256), 0, stream(), args, gxt, gyt, sgx, ns, xcd_order, m_per_img, m_sgx); \
                                 else hipLaunchKernelGGL((k_warp_strip_batch<G, F>), dim3(ns_launch), dim3(256), 0, stream(), args, gxt, gyt, sgx, ns, xcd_order, m_per_img, m_sgx, nt, inline_rest ? 1 : 0); } while (0)
#define LAUNCH_STRIP(G) do { if (with_far) LAUNCH_STRIP2(G, true); else LAUNCH_STRIP2(G, false); } while (0)
            if (gmode == 0) LAUNCH_STRIP(0); else if (gmode == 1) LAUNCH_STRIP(1); else if (gmode == 2) LAUNCH_STRIP(2); else LAUNCH_STRIP(3);
#undef LAUNCH_STRIP
#undef LAUNCH_STRIP2
        }
        if (!inline_rest && !(list_known && plan->count == 0)) {
            // what the strips did not stage (rectangles beyond the LDS buffers, pixels behind the camera, other border modes)
            {
                ProfileScope ps("warp_rest", 0);
                // (known list: one tile per work-group -- a tile through the gather body is a chain of several microseconds, and a frame that straddles
                // u = +-pi*scale leaves thousands of them at the rim of its footprint: bench.py ring360 85 -> 70 us)
                const int rest_grid = list_known ? std::max(1, std::min(plan->count, 65536)) : std::min(nt, 1024);
                if (gain) hipLaunchKernelGGL(k_warp_rest_batch<true>, dim3(rest_grid), dim3(256), 0, stream(), args, gxt, gyt, nt);
                else hipLaunchKernelGGL(k_warp_rest_batch<false>, dim3(rest_grid), dim3(256), 0, stream(), args, gxt, gyt, nt);
            }
            if (plan && whole && plan->state == 0) {
                // learn the count (and whether any tile missed for its gain rows) for the following panoramas: two small asynchronous copies
                if (!plan->h_count) SSP_HIP(hipHostMalloc((void **)&plan->h_count, 3 * sizeof(int)));
                if (!plan->ev) SSP_HIP(hipEventCreateWithFlags(&plan->ev, hipEventDisableTiming));
                SSP_HIP(hipMemcpyAsync(plan->h_count, rest, sizeof(int), hipMemcpyDeviceToHost, stream()));
                SSP_HIP(hipMemcpyAsync(plan->h_count + 1, rest + 1 + nt, 2 * sizeof(int), hipMemcpyDeviceToHost, stream()));     // misfit flag, far flag
                SSP_HIP(hipEventRecord(plan->ev, stream()));
                plan->state = 1;
                plan->capacity = nt;
                plan->d_list = rest;     // the geometry is static, so is the list: kept for the panoramas to come (neither rebuilt nor zeroed)
                rest = nullptr;
            }
            if (rest) pool_free(rest);
        }
    }
    SSP_HIP(hipGetLastError());
    return 0;
}

}  // namespace ssp

// ---- C ABI -------------------------------------------------------------------------------------------------------
namespace ssp {
// PyRotationWarper::PyRotationWarper(String type, float scale): type string -> projector
int make_projector(const char *type, float scale, Projector &p)
{
    static const struct { const char *name; int kind; float a, b; } tab[] = {
        {"plane", PK_PLANE, 0, 0}, {"affine", PK_AFFINE, 0, 0}, {"cylindrical", PK_CYLINDRICAL, 0, 0}, {"spherical", PK_SPHERICAL, 0, 0},
        {"fisheye", PK_FISHEYE, 0, 0}, {"stereographic", PK_STEREOGRAPHIC, 0, 0},
        {"compressedPlaneA2B1", PK_COMPRESSED, 2.f, 1.f}, {"compressedPlaneA1.5B1", PK_COMPRESSED, 1.5f, 1.f},
        {"compressedPlanePortraitA2B1", PK_COMPRESSED_PORTRAIT, 2.f, 1.f}, {"compressedPlanePortraitA1.5B1", PK_COMPRESSED_PORTRAIT, 1.5f, 1.f},
        {"paniniA2B1", PK_PANINI, 2.f, 1.f}, {"paniniA1.5B1", PK_PANINI, 1.5f, 1.f},
        {"paniniPortraitA2B1", PK_PANINI_PORTRAIT, 2.f, 1.f}, {"paniniPortraitA1.5B1", PK_PANINI_PORTRAIT, 1.5f, 1.f},
        {"mercator", PK_MERCATOR, 0, 0}, {"transverseMercator", PK_TRANSVERSE_MERCATOR, 0, 0},
    };
    SSP_REQUIRE(type, "warper: null type");
    for (const auto &e : tab)
        if (strcmp(type, e.name) == 0) {
            memset(&p, 0, sizeof p);
            p.kind = e.kind;
            p.scale = scale;
            p.a = e.a;
            p.b = e.b;
            return 0;
        }
    SSP_FAIL(SSP_ERR_ARG, "unknown warper :%s", type);
}
}  // namespace ssp

SSP_API int ssp_warper_create(const char *type, float scale, ssp_warper **out)
{
    SSP_REQUIRE(out, "warper: null argument");
    Projector p;
    SSP_TRY(make_projector(type, scale, p));
    ssp_warper *w = new ssp_warper();
    w->p = p;
    w->type = type;
    *out = w;
    return 0;
}
SSP_API int ssp_warper_destroy(ssp_warper *w) { delete w; return 0; }
SSP_API int ssp_warper_get_scale(const ssp_warper *w, float *s) { SSP_REQUIRE(w && s, "null"); *s = w->p.scale; return 0; }
SSP_API int ssp_warper_set_scale(ssp_warper *w, float s) { SSP_REQUIRE(w, "null"); w->p.scale = s; return 0; }

SSP_API int ssp_warper_roi(ssp_warper *w, int sw, int sh, const float K[9], const float R[9], int roi[4])
{
    SSP_REQUIRE(w && roi, "warpRoi: null argument");
    SSP_TRY(check_kr(K, R));
    set_camera(w->p, K, R);
    {
        std::lock_guard<std::mutex> lock(g_rois_mutex);
        for (size_t i = 0; i < g_rois.size(); ++i) {
            const RoiEntry &e = g_rois[i];
            if (e.kind == w->p.kind && e.w == sw && e.h == sh && e.scale == w->p.scale && e.a == w->p.a && e.b == w->p.b && !memcmp(e.K, K, sizeof e.K) && !memcmp(e.R, R, sizeof e.R) &&
                !memcmp(e.T, w->p.t, sizeof e.T)) {
                memcpy(roi, e.val, sizeof e.val);
                if (i) { const RoiEntry hit = e; g_rois.erase(g_rois.begin() + (long)i); g_rois.insert(g_rois.begin(), hit); }
                return 0;
            }
        }
    }
    SSP_TRY(detect_roi(w->p, sw, sh, roi));
    RoiEntry e;
    e.kind = w->p.kind; e.w = sw; e.h = sh; e.scale = w->p.scale; e.a = w->p.a; e.b = w->p.b;
    memcpy(e.K, K, sizeof e.K); memcpy(e.R, R, sizeof e.R); memcpy(e.T, w->p.t, sizeof e.T);
    memcpy(e.val, roi, sizeof e.val);
    std::lock_guard<std::mutex> lock(g_rois_mutex);
    g_rois.insert(g_rois.begin(), e);
    if (g_rois.size() > 256) g_rois.pop_back();
    return 0;
}

// The rectangles of a frame's roi that the multiband blender has to see (one, or two for a frame that straddles u = +-pi*scale), as the
// composer feeds them: parallel.plan_strips shards a closed ring by these instead of by whole rois
SSP_API int ssp_warper_live_parts(ssp_warper *w, int sw, int sh, const float K[9], const float R[9], int num_bands, int *parts_xywh, int capacity, int *count)
{
    SSP_REQUIRE(w && parts_xywh && count && capacity >= 2, "live parts: bad arguments (capacity >= 2 rectangles)");
    int roi[4];
    SSP_TRY(ssp_warper_roi(w, sw, sh, K, R, roi));
    int parts[2][4];
    SSP_TRY(live_parts(w->p, sw, sh, roi, live_reach(num_bands), parts, count));
    memcpy(parts_xywh, parts, sizeof(int) * 4 * (size_t)*count);
    return 0;
}

SSP_API int ssp_warper_warp_image(ssp_warper *w, const ssp_image *src, const float K[9], const float R[9], int interp, int border,
                                  ssp_image **dst, int corner[2])
{
    SSP_REQUIRE(w && src && dst, "warp: null argument");
    int roi[4];
    SSP_TRY(ssp_warper_roi(w, src->w, src->h, K, R, roi));
    ssp_image *d = nullptr;
    SSP_TRY(image_new(roi[2], roi[3], src->cn, src->depth, &d));
    int rc = warp_launch(w->p, src, roi, interp, border, d, nullptr);
    if (rc) { image_unref(d); return rc; }
    *dst = d;
    if (corner) { corner[0] = roi[0]; corner[1] = roi[1]; }
    return 0;
}

SSP_API int ssp_warper_warp_with_mask(ssp_warper *w, const ssp_image *src, const float K[9], const float R[9], int border, ssp_image **dst,
                                      ssp_image **mask, int corner[2])
{
    SSP_REQUIRE(w && src && dst, "warp: null argument");
    SSP_REQUIRE(src->depth == SSP_U8 && src->cn == 3, "warp_with_mask: source must be 8UC3");
    int roi[4];
    SSP_TRY(ssp_warper_roi(w, src->w, src->h, K, R, roi));
    ssp_image *d = nullptr, *m = nullptr;
    SSP_TRY(image_new(roi[2], roi[3], 3, SSP_U8, &d));
    if (mask) {
        int rc = image_new(roi[2], roi[3], 1, SSP_U8, &m);
        if (rc) { image_unref(d); return rc; }
    }
    int rc = warp_launch(w->p, src, roi, SSP_INTER_LINEAR, border, d, m);
    if (rc) { image_unref(d); image_unref(m); return rc; }
    *dst = d;
    if (mask) *mask = m;
    if (corner) { corner[0] = roi[0]; corner[1] = roi[1]; }
    return 0;
}

SSP_API int ssp_warper_warp(ssp_warper *w, const void *src, int sw, int sh, int cn, int depth, const float K[9], const float R[9], int interp,
                            int border, void *dst, int dw, int dh, int corner[2])
{
    SSP_REQUIRE(w && src && dst, "warp: null argument");
    ssp_image *s = nullptr, *d = nullptr;
    SSP_TRY(ssp_image_upload(src, sw, sh, cn, depth, &s));
    int c[2];
    int rc = ssp_warper_warp_image(w, s, K, R, interp, border, &d, c);
    image_unref(s);
    if (rc) return rc;
    if (d->w != dw || d->h != dh) {
        int gw = d->w, gh = d->h;
        image_unref(d);
        SSP_FAIL(SSP_ERR_ARG, "warp: dst is %dx%d but the warped roi is %dx%d (size it with ssp_warper_roi)", dw, dh, gw, gh);
    }
    rc = ssp_image_download(d, dst);
    image_unref(d);
    if (corner) { corner[0] = c[0]; corner[1] = c[1]; }
    return rc;
}

// PyRotationWarper::warpBackward(src, K, R, interp, border, dst_size): src has the size of warpRoi(dst_size, K, R)
SSP_API int ssp_warper_warp_backward(ssp_warper *w, const ssp_image *src, const float K[9], const float R[9], int interp, int border, int dst_w, int dst_h,
                                     ssp_image **dst)
{
    SSP_REQUIRE(w && src && dst && dst_w > 0 && dst_h > 0, "warpBackward: bad arguments");
    int roi[4];
    SSP_TRY(ssp_warper_roi(w, dst_w, dst_h, K, R, roi));
    SSP_REQUIRE(roi[2] == src->w && roi[3] == src->h, "warpBackward: src is %dx%d but warpRoi(dst_size) is %dx%d", src->w, src->h, roi[2], roi[3]);
    if (interp == SSP_INTER_AREA) interp = SSP_INTER_LINEAR;
    SSP_REQUIRE(interp == SSP_INTER_NEAREST || interp == SSP_INTER_LINEAR, "warpBackward: interpolation %d not supported by remap here", interp);
    SSP_REQUIRE(border >= 0 && border <= 4, "warpBackward: border mode %d not supported", border);
    ssp_image *d = nullptr;
    SSP_TRY(image_new(dst_w, dst_h, src->cn, src->depth, &d));
    SrcView sv = {(const uint8_t *)src->data, src->pitch, src->w, src->h};
    dim3 grid((dst_w + 63) / 64, (dst_h + 3) / 4), block(256);
    ProfileScope ps("warp_backward", (double)depth_size(src->depth) * src->cn * ((double)src->w * src->h + (double)dst_w * dst_h));
#define LAUNCH_BACK(T, CN) hipLaunchKernelGGL((k_warp_backward<T, CN>), grid, block, 0, stream(), w->p, sv, d->data, d->pitch, dst_w, dst_h, roi[0], roi[1], interp, border)
    if (src->depth == SSP_U8 && src->cn == 1) LAUNCH_BACK(uint8_t, 1);
    else if (src->depth == SSP_U8 && src->cn == 3) LAUNCH_BACK(uint8_t, 3);
    else if (src->depth == SSP_F32 && src->cn == 1) LAUNCH_BACK(float, 1);
    else if (src->depth == SSP_F32 && src->cn == 3) LAUNCH_BACK(float, 3);
    else { image_unref(d); SSP_FAIL(SSP_ERR_ARG, "warpBackward: unsupported source type (depth %d, %d channels); 8U/32F with 1 or 3 channels", src->depth, src->cn); }
#undef LAUNCH_BACK
    SSP_HIP(hipGetLastError());
    *dst = d;
    return 0;
}

SSP_API int ssp_warper_build_maps(ssp_warper *w, int sw, int sh, const float K[9], const float R[9], float *xmap, float *ymap, int dw, int dh,
                                  int roi[4])
{
    SSP_REQUIRE(w && xmap && ymap && roi, "buildMaps: null argument");
    SSP_TRY(ssp_warper_roi(w, sw, sh, K, R, roi));
    SSP_REQUIRE(roi[2] == dw && roi[3] == dh, "buildMaps: maps are %dx%d but the roi is %dx%d", dw, dh, roi[2], roi[3]);
    float *dx = nullptr, *dy = nullptr;
    size_t n = (size_t)dw * dh;
    SSP_TRY(pool_alloc(n * sizeof(float), (void **)&dx));
    int rc = pool_alloc(n * sizeof(float), (void **)&dy);
    if (rc) { pool_free(dx); return rc; }
    SrcView sv = {nullptr, 0, sw, sh};
    dim3 grid((dw + 63) / 64, (dh + 3) / 4), block(256);
    hipLaunchKernelGGL((k_warp_generic<uint8_t, 1>), grid, block, 0, stream(), w->p, sv, (void *)nullptr, (size_t)0, dw, dh, roi[0], roi[1], 0, 0, dx, dy);
    hipError_t e = hipMemcpyAsync(xmap, dx, n * sizeof(float), hipMemcpyDeviceToHost, stream());
    if (e == hipSuccess) e = hipMemcpyAsync(ymap, dy, n * sizeof(float), hipMemcpyDeviceToHost, stream());
    if (e == hipSuccess) e = hipStreamSynchronize(stream());
    pool_free(dx);
    pool_free(dy);
    if (e != hipSuccess) SSP_FAIL(SSP_ERR_DEVICE, "buildMaps failed: %s", hipGetErrorString(e));
    return 0;
}

SSP_API int ssp_warper_warp_point(ssp_warper *w, float x, float y, const float K[9], const float R[9], float uv[2])
{
    SSP_REQUIRE(w && uv, "warpPoint: null argument");
    SSP_TRY(check_kr(K, R));
    set_camera(w->p, K, R);
    map_forward(w->p, x, y, uv[0], uv[1]);  // host instantiation of the same inline functions
    return 0;
}
SSP_API int ssp_warper_warp_point_backward(ssp_warper *w, float u, float v, const float K[9], const float R[9], float xy[2])
{
    SSP_REQUIRE(w && xy, "warpPointBackward: null argument");
    SSP_TRY(check_kr(K, R));
    set_camera(w->p, K, R);
    map_backward(w->p, u, v, xy[0], xy[1]);
    return 0;
}

SSP_API int ssp_result_roi(int n, const int *c, const int *s, int roi[4])
{
    SSP_REQUIRE(n > 0 && c && s && roi, "resultRoi: bad arguments");
    int tlx = INT32_MAX, tly = INT32_MAX, brx = INT32_MIN, bry = INT32_MIN;
    for (int i = 0; i < n; ++i) {
        tlx = std::min(tlx, c[2 * i]);
        tly = std::min(tly, c[2 * i + 1]);
        brx = std::max(brx, c[2 * i] + s[2 * i]);
        bry = std::max(bry, c[2 * i + 1] + s[2 * i + 1]);
    }
    roi[0] = tlx; roi[1] = tly; roi[2] = brx - tlx; roi[3] = bry - tly;
    return 0;
}
