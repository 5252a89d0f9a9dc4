// ssp_seam.hip -- the two steps next to the hot path that work on device-resident warps.
//
// Replaces (stitching_detailed_enhanced.py):
//   :243-249, :1618   cv.detail.SeamFinder_createDefault(SeamFinder_VORONOI_SEAM).find(images, corners, masks)
//                     -> ssp_seam_voronoi          (stitching/src/seam_finders.cpp: PairwiseSeamFinder::run, VoronoiSeamFinder::findInPair)
//   :1822-1871        cv.detail.Timelapser_createDefault(type) .initialize / .process / .getDst
//                     -> ssp_timelapser_*          (stitching/src/timelapsers.cpp)
//   :1842             cv.bitwise_and(img, img, mask=mask)   -> ssp_bitwise_and_masked
// The Voronoi finder runs on seam-scale masks (~0.1 MPix each): it is a chain of small dependent launches per overlapping pair,
// kept on the device so that the masks never travel to the host.  DpSeamFinder (the reference's default) lives in ssp_seam_dp.hip.
#include "ssp_internal.hpp"

#include <vector>

using namespace ssp;

// ---- Voronoi seam finder -----------------------------------------------------------------------------------------------------
#define VORONOI_GAP 10
#define VORONOI_INF 65534  // what distanceTransform(DIST_L1, 3) returns where no zero pixel exists

struct VoronoiPair {
    uint8_t *m1; size_t p1; int w1, h1, ox1, oy1;  // mask of the first image, offset of the overlap inside it
    uint8_t *m2; size_t p2; int w2, h2, ox2, oy2;
    int rw, rh, sw, sh;                            // overlap size, and with the 10-pixel margin
    int *d1, *d2;                                  // distance maps, sw x sh each
};

// unique_k = submask_k without the pixels both cover; distance maps start as 0 on unique_k and "far" elsewhere
__global__ void k_voronoi_cut(const VoronoiPair p)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= p.sw || y >= p.sh) return;
    const int xa = p.ox1 + x - VORONOI_GAP, ya = p.oy1 + y - VORONOI_GAP, xb = p.ox2 + x - VORONOI_GAP, yb = p.oy2 + y - VORONOI_GAP;
    const bool s1 = xa >= 0 && ya >= 0 && xa < p.w1 && ya < p.h1 && p.m1[(size_t)ya * p.p1 + xa] != 0;
    const bool s2 = xb >= 0 && yb >= 0 && xb < p.w2 && yb < p.h2 && p.m2[(size_t)yb * p.p2 + xb] != 0;
    p.d1[(size_t)y * p.sw + x] = (s1 && !s2) ? 0 : VORONOI_INF;
    p.d2[(size_t)y * p.sw + x] = (s2 && !s1) ? 0 : VORONOI_INF;
}
// exact L1 distance to the nearest zero entry (what the 3x3 chamfer with a = 1, b = 2 computes): row scans ...
__global__ void k_l1_rows(int *d, int w, int h, int n_maps, size_t map_stride)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= h * n_maps) return;
    int *r = d + (size_t)(t / h) * map_stride + (size_t)(t % h) * w;
    int cur = VORONOI_INF;
    for (int x = 0; x < w; ++x) { cur = min(r[x], min(cur + 1, VORONOI_INF)); r[x] = cur; }
    cur = VORONOI_INF;
    for (int x = w - 1; x >= 0; --x) { cur = min(r[x], min(cur + 1, VORONOI_INF)); r[x] = cur; }
}
// ... then column scans
__global__ void k_l1_cols(int *d, int w, int h, int n_maps, size_t map_stride)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= w * n_maps) return;
    int *c = d + (size_t)(t / w) * map_stride + (t % w);
    int cur = VORONOI_INF;
    for (int y = 0; y < h; ++y) { cur = min(c[(size_t)y * w], min(cur + 1, VORONOI_INF)); c[(size_t)y * w] = cur; }
    cur = VORONOI_INF;
    for (int y = h - 1; y >= 0; --y) { cur = min(c[(size_t)y * w], min(cur + 1, VORONOI_INF)); c[(size_t)y * w] = cur; }
}
// seam = dist1 < dist2: the pixel stays with the first image (second mask cleared), otherwise with the second
__global__ void k_voronoi_apply(const VoronoiPair p)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= p.rw || y >= p.rh) return;
    const size_t o = (size_t)(y + VORONOI_GAP) * p.sw + (x + VORONOI_GAP);
    if (p.d1[o] < p.d2[o]) p.m2[(size_t)(p.oy2 + y) * p.p2 + (p.ox2 + x)] = 0;
    else p.m1[(size_t)(p.oy1 + y) * p.p1 + (p.ox1 + x)] = 0;
}

SSP_API int ssp_seam_voronoi(int n, const int *corners_xy, ssp_image *const *masks)
{
    SSP_TRY(ensure_init());
    SSP_REQUIRE(n >= 0 && (n == 0 || (corners_xy && masks)), "seam_voronoi: null argument");
    for (int i = 0; i < n; ++i)
        SSP_REQUIRE(masks[i] && masks[i]->depth == SSP_U8 && masks[i]->cn == 1, "seam_voronoi: mask %d must be CV_8UC1", i);
    // PairwiseSeamFinder::run: pairs in order, each one sees the cuts of the earlier ones (stream order)
    for (int i = 0; i + 1 < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const int x1 = corners_xy[2 * i], y1 = corners_xy[2 * i + 1], w1 = masks[i]->w, h1 = masks[i]->h;
            const int x2 = corners_xy[2 * j], y2 = corners_xy[2 * j + 1], w2 = masks[j]->w, h2 = masks[j]->h;
            const int rx = std::max(x1, x2), ry = std::max(y1, y2), rbx = std::min(x1 + w1, x2 + w2), rby = std::min(y1 + h1, y2 + h2);
            if (!(rx < rbx && ry < rby)) continue;  // overlapRoi
            VoronoiPair p;
            p.m1 = (uint8_t *)masks[i]->data; p.p1 = masks[i]->pitch; p.w1 = w1; p.h1 = h1; p.ox1 = rx - x1; p.oy1 = ry - y1;
            p.m2 = (uint8_t *)masks[j]->data; p.p2 = masks[j]->pitch; p.w2 = w2; p.h2 = h2; p.ox2 = rx - x2; p.oy2 = ry - y2;
            p.rw = rbx - rx; p.rh = rby - ry; p.sw = p.rw + 2 * VORONOI_GAP; p.sh = p.rh + 2 * VORONOI_GAP;
            const size_t map = (size_t)p.sw * p.sh;
            int *d = nullptr;
            SSP_TRY(pool_alloc(sizeof(int) * 2 * map, (void **)&d));
            p.d1 = d; p.d2 = d + map;
            ProfileScope ps("seam_voronoi_pair", (double)map * (2 + 2 * 4 * 5) + (double)p.rw * p.rh);
            hipLaunchKernelGGL(k_voronoi_cut, dim3((p.sw + 255) / 256, p.sh), dim3(256), 0, stream(), p);
            hipLaunchKernelGGL(k_l1_rows, dim3((2 * p.sh + 63) / 64), dim3(64), 0, stream(), d, p.sw, p.sh, 2, map);
            hipLaunchKernelGGL(k_l1_cols, dim3((2 * p.sw + 63) / 64), dim3(64), 0, stream(), d, p.sw, p.sh, 2, map);
            hipLaunchKernelGGL(k_voronoi_apply, dim3((p.rw + 255) / 256, p.rh), dim3(256), 0, stream(), p);
            pool_free(d);
        }
    SSP_HIP(hipGetLastError());
    return 0;
}

// ---- timelapser ----------------------------------------------------------------------------------------------------------------
struct ssp_timelapser {
    int type = SSP_TIMELAPSER_AS_IS;
    int roi[4] = {0, 0, 0, 0};
    ssp_image *dst = nullptr;
};

// dst.setTo(0) and the paste in one pass: every canvas pixel is written once
__global__ void k_timelapse(const int16_t *src, size_t sp, int sw, int sh, int dx, int dy, int16_t *dst, size_t dp, int dw, int dh)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    const int sx = x - dx, sy = y - dy;
    int16_t *d = (int16_t *)((char *)dst + (size_t)y * dp) + (size_t)x * 3;
    if (sx >= 0 && sy >= 0 && sx < sw && sy < sh) {
        const int16_t *s = (const int16_t *)((const char *)src + (size_t)sy * sp) + (size_t)sx * 3;
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    } else {
        d[0] = 0; d[1] = 0; d[2] = 0;
    }
}

SSP_API int ssp_timelapser_create(int type, ssp_timelapser **out)
{
    SSP_REQUIRE(out, "timelapser: null output");
    SSP_REQUIRE(type == SSP_TIMELAPSER_AS_IS || type == SSP_TIMELAPSER_CROP, "Timelapser_createDefault: unknown type %d", type);
    ssp_timelapser *t = new ssp_timelapser();
    t->type = type;
    *out = t;
    return 0;
}

SSP_API int ssp_timelapser_destroy(ssp_timelapser *t)
{
    if (t) { image_unref(t->dst); delete t; }
    return 0;
}

SSP_API int ssp_timelapser_initialize(ssp_timelapser *t, int n, const int *corners_xy, const int *sizes_wh)
{
    SSP_TRY(ensure_init());
    SSP_REQUIRE(t && n > 0 && corners_xy && sizes_wh, "timelapser.initialize: bad arguments");
    int x0, y0, x1, y1;
    if (t->type == SSP_TIMELAPSER_AS_IS) {  // resultRoi
        x0 = y0 = INT32_MAX; x1 = y1 = INT32_MIN;
        for (int i = 0; i < n; ++i) {
            x0 = std::min(x0, corners_xy[2 * i]); y0 = std::min(y0, corners_xy[2 * i + 1]);
            x1 = std::max(x1, corners_xy[2 * i] + sizes_wh[2 * i]); y1 = std::max(y1, corners_xy[2 * i + 1] + sizes_wh[2 * i + 1]);
        }
    } else {  // resultRoiIntersection
        x0 = y0 = INT32_MIN; x1 = y1 = INT32_MAX;
        for (int i = 0; i < n; ++i) {
            x0 = std::max(x0, corners_xy[2 * i]); y0 = std::max(y0, corners_xy[2 * i + 1]);
            x1 = std::min(x1, corners_xy[2 * i] + sizes_wh[2 * i]); y1 = std::min(y1, corners_xy[2 * i + 1] + sizes_wh[2 * i + 1]);
        }
    }
    SSP_REQUIRE(x1 > x0 && y1 > y0, "timelapser.initialize: empty canvas (%d x %d)", x1 - x0, y1 - y0);
    image_unref(t->dst);
    t->dst = nullptr;
    t->roi[0] = x0; t->roi[1] = y0; t->roi[2] = x1 - x0; t->roi[3] = y1 - y0;
    SSP_TRY(image_new(t->roi[2], t->roi[3], 3, SSP_S16, &t->dst));
    return 0;
}

SSP_API int ssp_timelapser_process(ssp_timelapser *t, const ssp_image *img, int tlx, int tly)
{
    SSP_REQUIRE(t && t->dst, "timelapser.process before initialize");
    SSP_REQUIRE(img && img->depth == SSP_S16 && img->cn == 3, "timelapser.process: image must be CV_16SC3");
    ProfileScope ps("timelapse", 6.0 * t->roi[2] * t->roi[3] + 6.0 * img->w * img->h);
    hipLaunchKernelGGL(k_timelapse, dim3((t->roi[2] + 255) / 256, t->roi[3]), dim3(256), 0, stream(), (const int16_t *)img->data, img->pitch, img->w, img->h,
                       tlx - t->roi[0], tly - t->roi[1], (int16_t *)t->dst->data, t->dst->pitch, t->roi[2], t->roi[3]);
    SSP_HIP(hipGetLastError());
    return 0;
}

SSP_API int ssp_timelapser_get_dst(ssp_timelapser *t, ssp_image **out)
{
    SSP_REQUIRE(t && t->dst && out, "timelapser.getDst before initialize");
    t->dst->refs++;
    *out = t->dst;
    return 0;
}

SSP_API int ssp_timelapser_dst_roi(const ssp_timelapser *t, int roi[4])
{
    SSP_REQUIRE(t && roi, "timelapser: null argument");
    memcpy(roi, t->roi, sizeof t->roi);
    return 0;
}

// ---- cv.bitwise_and(a, b, mask=m) ------------------------------------------------------------------------------------------------
__global__ void k_and_masked(const uint8_t *a, size_t ap, const uint8_t *b, size_t bp, const uint8_t *m, size_t mp, uint8_t *d, size_t dp, int w, int h, int bpp)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const bool on = m[(size_t)y * mp + x] != 0;
    const uint8_t *pa = a + (size_t)y * ap + (size_t)x * bpp, *pb = b + (size_t)y * bp + (size_t)x * bpp;
    uint8_t *pd = d + (size_t)y * dp + (size_t)x * bpp;
    for (int k = 0; k < bpp; ++k) pd[k] = on ? (uint8_t)(pa[k] & pb[k]) : (uint8_t)0;  // cv2 allocates a zeroed dst when none is passed
}

SSP_API int ssp_bitwise_and_masked(const ssp_image *a, const ssp_image *b, const ssp_image *mask, ssp_image **out)
{
    SSP_REQUIRE(a && b && mask && out, "bitwise_and: null argument");
    SSP_REQUIRE(a->w == b->w && a->h == b->h && a->cn == b->cn && a->depth == b->depth, "bitwise_and: operands differ in size or type");
    SSP_REQUIRE(mask->depth == SSP_U8 && mask->cn == 1 && mask->w == a->w && mask->h == a->h, "bitwise_and: mask must be CV_8UC1 of the operands' size");
    ssp_image *d = nullptr;
    SSP_TRY(image_new(a->w, a->h, a->cn, a->depth, &d));
    const int bpp = a->cn * depth_size(a->depth);
    ProfileScope ps("mask_and", (3.0 * bpp + 1) * a->w * a->h);
    hipLaunchKernelGGL(k_and_masked, dim3((a->w + 255) / 256, a->h), dim3(256), 0, stream(), (const uint8_t *)a->data, a->pitch, (const uint8_t *)b->data, b->pitch,
                       (const uint8_t *)mask->data, mask->pitch, (uint8_t *)d->data, d->pitch, a->w, a->h, bpp);
    SSP_HIP(hipGetLastError());
    *out = d;
    return 0;
}
