// ssp_comp.hip -- cv.detail.ExposureCompensator family on gfx950.
//
// Replaces (stitching_detailed_enhanced.py):
//   :649-665  get_compensator(): ExposureCompensator_createDefault(type) / detail_ChannelsCompensator / detail_BlocksChannelsCompensator
//   :1613     compensator.feed(corners, images_warped, masks_warped)   -- seam-scale u8c3 warps + u8 masks
//   :1754     compensator.apply(idx, corner, image_warped, mask_warped) -- in place
//
// feed(): the per-pair overlap statistics (count, sum of |BGR| or per-channel sums) are the O(pairs x overlap area)
// part; they run as one workgroup per overlapping pair (or block pair).  The (blocks x blocks) normal equations are
// assembled and solved on the host in double with the same partial-pivot LU as cv::solve -- a few kB of data.
// apply(): one streaming pass, gain (or bilinearly resized gain map) times pixel, round-half-even, saturate.
#include "ssp_internal.hpp"

using namespace ssp;

struct ViewDev {
    const uint8_t *img; size_t ip;   // first pixel of the view
    const uint8_t *mask; size_t mp;
    int w, h, cx, cy;
};
struct PairDev {
    int a, b;                // view indices
    int x_tl, y_tl, w, h;    // overlap rectangle (absolute coordinates)
};
struct PairOut {
    int count;
    int pad;
    double s1[3], s2[3];     // GAIN: s[0] = sum of sqrt(b^2+g^2+r^2); CHANNELS: per-channel sums (exact integers)
};

template <bool CHANNELS>
__global__ __launch_bounds__(256) void k_pair_stats(const ViewDev *views, const PairDev *pairs, PairOut *out)
{
    const PairDev p = pairs[blockIdx.x];
    const ViewDev va = views[p.a], vb = views[p.b];
    int cnt = 0;
    double s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0};
    const int n = p.w * p.h;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int y = i / p.w, x = i - y * p.w;
        int ax = p.x_tl - va.cx + x, ay = p.y_tl - va.cy + y, bx = p.x_tl - vb.cx + x, by = p.y_tl - vb.cy + y;
        if (va.mask[(size_t)ay * va.mp + ax] == 255 && vb.mask[(size_t)by * vb.mp + bx] == 255) {
            ++cnt;
            const uint8_t *q1 = va.img + (size_t)ay * va.ip + (size_t)ax * 3, *q2 = vb.img + (size_t)by * vb.ip + (size_t)bx * 3;
            if (CHANNELS) {
                for (int c = 0; c < 3; ++c) { s1[c] += q1[c]; s2[c] += q2[c]; }
            } else {
                s1[0] += sqrt((double)q1[0] * q1[0] + (double)q1[1] * q1[1] + (double)q1[2] * q1[2]);
                s2[0] += sqrt((double)q2[0] * q2[0] + (double)q2[1] * q2[1] + (double)q2[2] * q2[2]);
            }
        }
    }
    __shared__ int sc[256];
    __shared__ double sa[256], sb[256];
    const int NCH = CHANNELS ? 3 : 1;
    sc[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sc[threadIdx.x] += sc[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[blockIdx.x].count = sc[0];
    for (int c = 0; c < NCH; ++c) {
        __syncthreads();
        sa[threadIdx.x] = s1[c];
        sb[threadIdx.x] = s2[c];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) { sa[threadIdx.x] += sa[threadIdx.x + o]; sb[threadIdx.x] += sb[threadIdx.x + o]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) { out[blockIdx.x].s1[c] = sa[0]; out[blockIdx.x].s2[c] = sb[0]; }
    }
}

// image = saturate_u8(cvRound(image * gain)), gain per channel (GainCompensator / ChannelsCompensator::apply)
__global__ void k_apply_scalar(uint8_t *img, size_t ip, int w, int h, float g0, float g1, float g2)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w * 3 || y >= h) return;
    int c = x % 3;
    float g = c == 0 ? g0 : (c == 1 ? g1 : g2);
    uint8_t *p = img + (size_t)y * ip + x;
    float r = __builtin_rintf((float)*p * g);
    *p = (uint8_t)(r < 0.f ? 0 : (r > 255.f ? 255 : (int)r));
}

// BlocksCompensator::apply: resize(gain_map, image size, INTER_LINEAR) (pixel-centre mapping, float weights,
// horizontal pass then vertical pass), then multiply and round
__device__ inline void lin_coord(int d, int ssize, int dsize, int &s0, int &s1, float &f)
{
    double scale = 1.0 / ((double)dsize / (double)ssize);
    float fv = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(fv);
    fv -= s;
    if (s < 0) { fv = 0; s = 0; }
    if (s >= ssize - 1) { fv = 0; s = ssize - 1; }
    s0 = s;
    s1 = min(s + 1, ssize - 1);
    f = fv;
}
__global__ void k_apply_map(uint8_t *img, size_t ip, int w, int h, const float *gm, int gw, int gh, int gcn)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    uint8_t *p = img + (size_t)y * ip + (size_t)x * 3;
    float g[3];
    if (gw == w && gh == h) {
        for (int c = 0; c < 3; ++c) g[c] = gm[((size_t)y * gw + x) * gcn + (gcn == 3 ? c : 0)];
    } else {
        int x0, x1, y0, y1;
        float a1, b1;
        lin_coord(x, gw, w, x0, x1, a1);
        lin_coord(y, gh, h, y0, y1, b1);
        float a0 = 1.f - a1, b0 = 1.f - b1;
        for (int c = 0; c < gcn; ++c) {
            float t0 = gm[((size_t)y0 * gw + x0) * gcn + c] * a0 + gm[((size_t)y0 * gw + x1) * gcn + c] * a1;
            float t1 = gm[((size_t)y1 * gw + x0) * gcn + c] * a0 + gm[((size_t)y1 * gw + x1) * gcn + c] * a1;
            g[c] = t0 * b0 + t1 * b1;
        }
        if (gcn == 1) g[1] = g[2] = g[0];
    }
    for (int c = 0; c < 3; ++c) {
        float r = __builtin_rintf((float)p[c] * g[c]);
        p[c] = (uint8_t)(r < 0.f ? 0 : (r > 255.f ? 255 : (int)r));
    }
}

// ---- host --------------------------------------------------------------------------------------------------------
struct ssp_compensator {
    int type = SSP_COMP_NO;
    int bl_w = 32, bl_h = 32, nr_feeds = 1, nr_filter = 2;
    int n = 0;
    std::vector<double> gains;                 // n (GAIN) or 3n (CHANNELS)
    std::vector<int> gm_w, gm_h;
    int gm_cn = 1;
    std::vector<std::vector<float>> gmap;      // host copies
    std::vector<float *> d_gmap;               // device copies for apply
};

namespace ssp {

struct ViewHost {
    int img, x0, y0, w, h, cx, cy;
};

// cv::solve(A, b, x, DECOMP_LU) on doubles (hal::LU64f: partial pivoting, eps = DBL_EPSILON * 100)
static bool lu_solve(std::vector<double> &A, std::vector<double> &b, int m)
{
    const double eps = 2.220446049250313e-16 * 100;
    for (int i = 0; i < m; ++i) {
        int k = i;
        for (int j = i + 1; j < m; ++j)
            if (std::abs(A[(size_t)j * m + i]) > std::abs(A[(size_t)k * m + i])) k = j;
        if (std::abs(A[(size_t)k * m + i]) < eps) return false;
        if (k != i) {
            for (int j = i; j < m; ++j) std::swap(A[(size_t)i * m + j], A[(size_t)k * m + j]);
            std::swap(b[i], b[k]);
        }
        double d = -1 / A[(size_t)i * m + i];
        for (int j = i + 1; j < m; ++j) {
            double alpha = A[(size_t)j * m + i] * d;
            if (alpha == 0.0) continue;  // adding alpha*row == 0 leaves the row unchanged bit for bit
            double *rj = &A[(size_t)j * m], *ri = &A[(size_t)i * m];
            for (int q = i + 1; q < m; ++q) rj[q] += alpha * ri[q];
            b[j] += alpha * b[i];
        }
    }
    for (int i = m - 1; i >= 0; --i) {
        double s = b[i];
        for (int q = i + 1; q < m; ++q) s -= A[(size_t)i * m + q] * b[q];
        b[i] = s / A[(size_t)i * m + i];
    }
    return true;
}

// GainCompensator::singleFeed's normal equations for one channel
static void solve_gains(int nv, const std::vector<int> &N, const std::vector<double> &I, const std::vector<uint8_t> &skip, double *gains, int stride)
{
    const double alpha = 0.01, beta = 100;
    int num_eq = 0;
    for (int i = 0; i < nv; ++i) { gains[(size_t)i * stride] = 1.0; if (!skip[i]) ++num_eq; }
    if (num_eq == 0) return;
    std::vector<double> A((size_t)num_eq * num_eq, 0.0), b(num_eq, 0.0);
    for (int i = 0, ki = 0; i < nv; ++i) {
        if (skip[i]) continue;
        for (int j = 0, kj = 0; j < nv; ++j) {
            if (skip[j]) continue;
            int Nij = N[(size_t)i * nv + j];
            b[ki] += beta * Nij;
            A[(size_t)ki * num_eq + ki] += beta * Nij;
            if (j != i) {
                A[(size_t)ki * num_eq + ki] += 2 * alpha * I[(size_t)i * nv + j] * I[(size_t)i * nv + j] * Nij;
                A[(size_t)ki * num_eq + kj] -= 2 * alpha * I[(size_t)i * nv + j] * I[(size_t)j * nv + i] * Nij;
            }
            ++kj;
        }
        ++ki;
    }
    if (!lu_solve(A, b, num_eq)) std::fill(b.begin(), b.end(), 0.0);
    for (int i = 0, j = 0; i < nv; ++i)
        if (!skip[i]) gains[(size_t)i * stride] = b[j++];
}

// sepFilter2D(map, CV_32F, [.25 .5 .25], [.25 .5 .25]), BORDER_REFLECT_101
static int r101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}
static void filter_map(std::vector<float> &m, int w, int h, int cn)
{
    std::vector<float> t(m.size());
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int xl = r101(x - 1, w), xr = r101(x + 1, w);
            for (int c = 0; c < cn; ++c) {
                float side = m[((size_t)y * w + xl) * cn + c] + m[((size_t)y * w + xr) * cn + c];
                t[((size_t)y * w + x) * cn + c] = m[((size_t)y * w + x) * cn + c] * 0.5f + side * 0.25f;
            }
        }
    for (int y = 0; y < h; ++y) {
        int yu = r101(y - 1, h), yd = r101(y + 1, h);
        for (int x = 0; x < w * cn; ++x) {
            float side = t[(size_t)yu * w * cn + x] + t[(size_t)yd * w * cn + x];
            m[(size_t)y * w * cn + x] = t[(size_t)y * w * cn + x] * 0.5f + side * 0.25f;
        }
    }
}

static void comp_clear(ssp_compensator *c)
{
    for (float *p : c->d_gmap) pool_free(p);
    c->d_gmap.clear();
    c->gmap.clear();
    c->gm_w.clear();
    c->gm_h.clear();
    c->gains.clear();
    c->n = 0;
}

}  // namespace ssp

SSP_API int ssp_comp_create(int type, ssp_compensator **out)
{
    SSP_REQUIRE(out && type >= SSP_COMP_NO && type <= SSP_COMP_CHANNELS_BLOCKS, "compensator: unknown type %d", type);
    ssp_compensator *c = new ssp_compensator();
    c->type = type;
    *out = c;
    return 0;
}
SSP_API int ssp_comp_destroy(ssp_compensator *c)
{
    if (c) { comp_clear(c); delete c; }
    return 0;
}
SSP_API int ssp_comp_set_nr_feeds(ssp_compensator *c, int n) { SSP_REQUIRE(c && n >= 1, "setNrFeeds: bad value"); c->nr_feeds = n; return 0; }
SSP_API int ssp_comp_set_block_size(ssp_compensator *c, int w, int h) { SSP_REQUIRE(c && w > 0 && h > 0, "setBlockSize: bad value"); c->bl_w = w; c->bl_h = h; return 0; }
SSP_API int ssp_comp_set_nr_filtering(ssp_compensator *c, int n) { SSP_REQUIRE(c && n >= 0, "setNrGainsFilteringIterations: bad value"); c->nr_filter = n; return 0; }
SSP_API int ssp_comp_num_images(const ssp_compensator *c, int *n) { SSP_REQUIRE(c && n, "null"); *n = c->n; return 0; }

SSP_API int ssp_comp_feed(ssp_compensator *c, int n, const int *corners, ssp_image *const *images, ssp_image *const *masks)
{
    SSP_REQUIRE(c && n >= 0, "feed: bad arguments");
    comp_clear(c);
    c->n = n;
    if (c->type == SSP_COMP_NO || n == 0) return 0;
    SSP_REQUIRE(corners && images && masks, "feed: null argument");
    SSP_TRY(ensure_init());
    for (int i = 0; i < n; ++i) {
        SSP_REQUIRE(images[i] && masks[i], "feed: null image %d", i);
        SSP_REQUIRE(images[i]->depth == SSP_U8 && images[i]->cn == 3, "feed: image %d must be CV_8UC3", i);
        SSP_REQUIRE(masks[i]->depth == SSP_U8 && masks[i]->cn == 1 && masks[i]->w == images[i]->w && masks[i]->h == images[i]->h, "feed: mask %d mismatch", i);
    }
    const bool blocks = c->type == SSP_COMP_GAIN_BLOCKS || c->type == SSP_COMP_CHANNELS_BLOCKS;
    const bool channels = c->type == SSP_COMP_CHANNELS || c->type == SSP_COMP_CHANNELS_BLOCKS;
    const int gcn = channels ? 3 : 1;

    // private working copies (feed never modifies the caller's arrays; nr_feeds > 1 re-applies gains in place)
    std::vector<ssp_image *> work(n, nullptr);
    auto cleanup = [&]() { for (auto *p : work) image_unref(p); };
    for (int i = 0; i < n; ++i) {
        if (c->nr_feeds > 1) {
            int rc = ssp_image_convert(images[i], SSP_U8, &work[i]);
            if (rc) { cleanup(); return rc; }
        } else {
            work[i] = images[i];
            images[i]->refs++;
        }
    }

    // views: whole images, or BlocksCompensator's equalised blocks
    std::vector<ViewHost> views;
    std::vector<int> blw(n, 1), blh(n, 1);
    for (int i = 0; i < n; ++i) {
        const int W = images[i]->w, H = images[i]->h;
        if (!blocks) { views.push_back({i, 0, 0, W, H, corners[2 * i], corners[2 * i + 1]}); continue; }
        blw[i] = (W + c->bl_w - 1) / c->bl_w;
        blh[i] = (H + c->bl_h - 1) / c->bl_h;
        int bw = (W + blw[i] - 1) / blw[i], bh = (H + blh[i] - 1) / blh[i];
        for (int by = 0; by < blh[i]; ++by)
            for (int bx = 0; bx < blw[i]; ++bx) {
                int tx = bx * bw, ty = by * bh;
                int brx = std::min(tx + bw, W), bry = std::min(ty + bh, H);
                views.push_back({i, tx, ty, brx - tx, bry - ty, corners[2 * i] + tx, corners[2 * i + 1] + ty});
            }
    }
    const int nv = (int)views.size();
    // overlapping pairs (i <= j), overlapRoi
    std::vector<PairDev> pairs;
    for (int i = 0; i < nv; ++i)
        for (int j = i; j < nv; ++j) {
            const ViewHost &a = views[i], &b = views[j];
            int x_tl = std::max(a.cx, b.cx), y_tl = std::max(a.cy, b.cy);
            int x_br = std::min(a.cx + a.w, b.cx + b.w), y_br = std::min(a.cy + a.h, b.cy + b.h);
            if (x_tl < x_br && y_tl < y_br) pairs.push_back({i, j, x_tl, y_tl, x_br - x_tl, y_br - y_tl});
        }
    const int np = (int)pairs.size();
    ViewDev *d_views = nullptr;
    PairDev *d_pairs = nullptr;
    PairOut *d_out = nullptr;
    int rc = pool_alloc(sizeof(ViewDev) * nv, (void **)&d_views);
    if (!rc) rc = pool_alloc(sizeof(PairDev) * std::max(np, 1), (void **)&d_pairs);
    if (!rc) rc = pool_alloc(sizeof(PairOut) * std::max(np, 1), (void **)&d_out);
    auto cleanup_dev = [&]() { pool_free(d_views); pool_free(d_pairs); pool_free(d_out); };
    if (rc) { cleanup(); cleanup_dev(); return rc; }
    std::vector<ViewDev> hv(nv);
    for (int v = 0; v < nv; ++v) {
        const ViewHost &vh = views[v];
        const ssp_image *im = work[vh.img], *mk = masks[vh.img];
        hv[v].img = (const uint8_t *)im->data + (size_t)vh.y0 * im->pitch + (size_t)vh.x0 * 3;
        hv[v].ip = im->pitch;
        hv[v].mask = (const uint8_t *)mk->data + (size_t)vh.y0 * mk->pitch + vh.x0;
        hv[v].mp = mk->pitch;
        hv[v].w = vh.w; hv[v].h = vh.h; hv[v].cx = vh.cx; hv[v].cy = vh.cy;
    }
    hipError_t e = hipMemcpyAsync(d_views, hv.data(), sizeof(ViewDev) * nv, hipMemcpyHostToDevice, stream());
    if (e == hipSuccess && np) e = hipMemcpyAsync(d_pairs, pairs.data(), sizeof(PairDev) * np, hipMemcpyHostToDevice, stream());
    if (e == hipSuccess) e = hipStreamSynchronize(stream());
    if (e != hipSuccess) { cleanup(); cleanup_dev(); SSP_FAIL(SSP_ERR_DEVICE, "compensator feed: upload failed: %s", hipGetErrorString(e)); }

    std::vector<double> acc((size_t)nv * gcn, 1.0), g((size_t)nv * gcn, 1.0);
    std::vector<PairOut> po(std::max(np, 1));
    for (int it = 0; it < c->nr_feeds; ++it) {
        if (it > 0) {
            // apply the gains of the previous feed to the working copies (per view; views tile the images)
            for (int v = 0; v < nv; ++v) {
                const ViewHost &vh = views[v];
                ssp_image *im = work[vh.img];
                uint8_t *base = (uint8_t *)im->data + (size_t)vh.y0 * im->pitch + (size_t)vh.x0 * 3;
                float g0 = (float)g[(size_t)v * gcn], g1 = (float)g[(size_t)v * gcn + (gcn == 3 ? 1 : 0)], g2 = (float)g[(size_t)v * gcn + (gcn == 3 ? 2 : 0)];
                hipLaunchKernelGGL(k_apply_scalar, dim3((vh.w * 3 + 255) / 256, vh.h), dim3(256), 0, stream(), base, im->pitch, vh.w, vh.h, g0, g1, g2);
            }
        }
        if (np) {
            ProfileScope ps("comp_pair_stats", 0);
            if (channels) hipLaunchKernelGGL(k_pair_stats<true>, dim3(np), dim3(256), 0, stream(), d_views, d_pairs, d_out);
            else hipLaunchKernelGGL(k_pair_stats<false>, dim3(np), dim3(256), 0, stream(), d_views, d_pairs, d_out);
        }
        e = np ? hipMemcpyAsync(po.data(), d_out, sizeof(PairOut) * np, hipMemcpyDeviceToHost, stream()) : hipSuccess;
        if (e == hipSuccess) e = hipStreamSynchronize(stream());
        if (e != hipSuccess) { cleanup(); cleanup_dev(); SSP_FAIL(SSP_ERR_DEVICE, "compensator feed: pair statistics failed: %s", hipGetErrorString(e)); }
        std::vector<int> N((size_t)nv * nv, 0);
        std::vector<uint8_t> skip(nv, 1);
        std::vector<double> I((size_t)nv * nv, 0.0);
        for (int ch = 0; ch < gcn; ++ch) {
            std::fill(I.begin(), I.end(), 0.0);
            for (int k = 0; k < np; ++k) {
                const PairDev &p = pairs[k];
                int cnt = std::max(1, po[k].count);
                N[(size_t)p.a * nv + p.b] = N[(size_t)p.b * nv + p.a] = cnt;
                if (p.a != p.b) { skip[p.a] = 0; skip[p.b] = 0; }
                I[(size_t)p.a * nv + p.b] = po[k].s1[ch] / cnt;
                I[(size_t)p.b * nv + p.a] = po[k].s2[ch] / cnt;
            }
            solve_gains(nv, N, I, skip, g.data() + ch, gcn);
        }
        for (size_t q = 0; q < acc.size(); ++q) acc[q] = it == 0 ? g[q] : acc[q] * g[q];
    }
    cleanup();
    cleanup_dev();

    if (!blocks) {
        c->gains = acc;
    } else {
        c->gm_cn = gcn;
        int idx = 0;
        for (int i = 0; i < n; ++i) {
            int gw = blw[i], gh = blh[i];
            std::vector<float> m((size_t)gw * gh * gcn);
            for (int q = 0; q < gw * gh; ++q, ++idx)
                for (int ch = 0; ch < gcn; ++ch) m[(size_t)q * gcn + ch] = (float)acc[(size_t)idx * gcn + ch];
            for (int it = 0; it < c->nr_filter; ++it) filter_map(m, gw, gh, gcn);
            float *d = nullptr;
            SSP_TRY(pool_alloc(m.size() * sizeof(float), (void **)&d));
            SSP_HIP(hipMemcpyAsync(d, m.data(), m.size() * sizeof(float), hipMemcpyHostToDevice, stream()));
            SSP_HIP(hipStreamSynchronize(stream()));
            c->gm_w.push_back(gw);
            c->gm_h.push_back(gh);
            c->gmap.push_back(std::move(m));
            c->d_gmap.push_back(d);
        }
    }
    return 0;
}

SSP_API int ssp_comp_apply(ssp_compensator *c, int index, ssp_image *image)
{
    SSP_REQUIRE(c && image, "apply: null argument");
    if (c->type == SSP_COMP_NO) return 0;
    SSP_REQUIRE(index >= 0 && index < c->n, "apply: index %d out of range (fed %d images)", index, c->n);
    SSP_REQUIRE(image->depth == SSP_U8 && image->cn == 3, "apply: image must be CV_8UC3");
    const double px = (double)image->w * image->h;
    if (c->type == SSP_COMP_GAIN || c->type == SSP_COMP_CHANNELS) {
        float g0, g1, g2;
        if (c->type == SSP_COMP_GAIN) g0 = g1 = g2 = (float)c->gains[index];
        else { g0 = (float)c->gains[(size_t)index * 3]; g1 = (float)c->gains[(size_t)index * 3 + 1]; g2 = (float)c->gains[(size_t)index * 3 + 2]; }
        ProfileScope ps("comp_apply", 6 * px);
        hipLaunchKernelGGL(k_apply_scalar, dim3((image->w * 3 + 255) / 256, image->h), dim3(256), 0, stream(), (uint8_t *)image->data, image->pitch, image->w, image->h, g0,
                           g1, g2);
    } else {
        ProfileScope ps("comp_apply", 6 * px);
        hipLaunchKernelGGL(k_apply_map, dim3((image->w + 255) / 256, image->h), dim3(256), 0, stream(), (uint8_t *)image->data, image->pitch, image->w, image->h,
                           c->d_gmap[index], c->gm_w[index], c->gm_h[index], c->gm_cn);
    }
    SSP_HIP(hipGetLastError());
    ++image->version;
    return 0;
}

SSP_API int ssp_comp_get_gains(const ssp_compensator *c, double *gains, int capacity, int *count)
{
    SSP_REQUIRE(c && count, "getMatGains: null argument");
    *count = (int)c->gains.size();
    if (gains) {
        SSP_REQUIRE(capacity >= *count, "getMatGains: buffer too small");
        memcpy(gains, c->gains.data(), sizeof(double) * c->gains.size());
    }
    return 0;
}

SSP_API int ssp_comp_get_gain_map(const ssp_compensator *c, int index, float *map, int capacity, int *w, int *h, int *cn)
{
    SSP_REQUIRE(c && index >= 0 && index < (int)c->gmap.size(), "getGainMap: no gain map %d", index);
    if (w) *w = c->gm_w[index];
    if (h) *h = c->gm_h[index];
    if (cn) *cn = c->gm_cn;
    if (map) {
        SSP_REQUIRE(capacity >= (int)c->gmap[index].size(), "getGainMap: buffer too small");
        memcpy(map, c->gmap[index].data(), sizeof(float) * c->gmap[index].size());
    }
    return 0;
}

// ExposureCompensator::setMatGains: install gains without a feed (exposure_compensate.cpp: GainCompensator::setMatGains takes one
// 1x1 CV_64F per image, ChannelsCompensator one 3x1 per image, the block compensators one CV_32F map per image)
SSP_API int ssp_comp_set_gains(ssp_compensator *c, const double *gains, int count)
{
    SSP_REQUIRE(c && gains && count > 0, "setMatGains: null or empty argument");
    SSP_REQUIRE(c->type == SSP_COMP_GAIN || c->type == SSP_COMP_CHANNELS, "setMatGains: scalar gains need a GAIN or CHANNELS compensator");
    const int per = c->type == SSP_COMP_CHANNELS ? 3 : 1;
    SSP_REQUIRE(count % per == 0, "setMatGains: %d values are not a whole number of images (x%d)", count, per);
    comp_clear(c);
    c->gains.assign(gains, gains + count);
    c->n = count / per;
    return 0;
}

SSP_API int ssp_comp_set_gain_map(ssp_compensator *c, int index, const float *map, int w, int h, int cn)
{
    SSP_REQUIRE(c && map && w > 0 && h > 0, "setMatGains: null or empty gain map");
    SSP_REQUIRE(c->type == SSP_COMP_GAIN_BLOCKS || c->type == SSP_COMP_CHANNELS_BLOCKS, "setMatGains: gain maps need a block compensator");
    SSP_REQUIRE(cn == (c->type == SSP_COMP_CHANNELS_BLOCKS ? 3 : 1), "setMatGains: gain map has %d channels", cn);
    SSP_REQUIRE(index >= 0 && index <= (int)c->gmap.size(), "setMatGains: maps must be installed in image order (index %d, %d present)", index, (int)c->gmap.size());
    if (index == 0) comp_clear(c);                 // a new set of maps replaces the old one
    std::vector<float> m(map, map + (size_t)w * h * cn);
    float *d = nullptr;
    SSP_TRY(pool_alloc(m.size() * sizeof(float), (void **)&d));
    SSP_HIP(hipMemcpyAsync(d, m.data(), m.size() * sizeof(float), hipMemcpyHostToDevice, stream()));
    SSP_HIP(hipStreamSynchronize(stream()));
    c->gm_cn = cn;
    c->gm_w.push_back(w);
    c->gm_h.push_back(h);
    c->gmap.push_back(std::move(m));
    c->d_gmap.push_back(d);
    c->n = (int)c->gmap.size();
    return 0;
}

namespace ssp {
// used by the composer: device gain description of image `index`
int comp_gain_desc(const ssp_compensator *c, int index, int *kind, float g[3], const float **d_map, int *gw, int *gh, int *gcn)
{
    *kind = 0;
    if (!c || c->type == SSP_COMP_NO) return 0;
    SSP_REQUIRE(index >= 0 && index < c->n, "compensator has no gains for image %d", index);
    if (c->type == SSP_COMP_GAIN) { *kind = 1; g[0] = g[1] = g[2] = (float)c->gains[index]; }
    else if (c->type == SSP_COMP_CHANNELS) { *kind = 1; for (int q = 0; q < 3; ++q) g[q] = (float)c->gains[(size_t)index * 3 + q]; }
    else { *kind = 2; *d_map = c->d_gmap[index]; *gw = c->gm_w[index]; *gh = c->gm_h[index]; *gcn = c->gm_cn; }
    return 0;
}
}  // namespace ssp
