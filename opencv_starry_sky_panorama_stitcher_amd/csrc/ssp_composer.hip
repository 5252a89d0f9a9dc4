// ssp_composer.hip -- the compose loop of sde.py:1673-1930 as one device-resident plan.
//
// The reference's per-image loop (warp image :1731, warp mask :1740, compensator.apply :1754, astype :1755,
// dilate/resize/and :1760-1772, blender.feed :1886) followed by blender.blend (:1930), driven from ONE C call so
// that neither Python nor the host allocator sits between kernels.  Same kernels and arithmetic as the object
// API (ssp_warper_* / ssp_comp_apply / ssp_blender_*); geometry (warpRoi, resultRoi, seam-scale masks) is
// resolved once at creation, exactly where the reference resolves it (:1689-1698 and :1543-1599).
#include "ssp_blender.hpp"
#include "ssp_projector.hpp"

using namespace ssp;

namespace ssp {
int make_projector(const char *type, float scale, Projector &p);
void set_camera(Projector &p, const float K[9], const float R[9]);
int detect_roi(const Projector &p, int W, int H, int roi[4]);
int warp_launch(const Projector &p, const ssp_image *src, const int roi[4], int interp, int border, ssp_image *dst, ssp_image *mask);
int resize_linear_exact(const ssp_image *src, int dw, int dh, const ssp_image *and_with, ssp_image **out);
size_t warp_batch_desc_size();
void warp_batch_fill(void *desc, const Projector &p, const ssp_image *src, const int roi[4], int full_dw, int x_off, int border, uint8_t *dst, size_t dst_pitch, uint8_t *mask,
                     size_t mask_pitch, int xshift, float *tab, int prep, const ssp_image *seam, ssp_image *dil, int *lin, void *tiles, void *cmap);
size_t warp_cmap_words(int dw, int dh);
int warp_cmap_set_projector(void *cmap, const Projector &p);
int live_parts(const Projector &p, int W, int H, const int roi[4], int reach, int parts[2][4], int *n_parts);
size_t warp_tile_bytes(int dw, int dh);
int warp_table_cols(int dw);
size_t warp_lin_ints(int dw, int dh, int seam_h);
int warp_prep_items(int dw, int dh, int seam_w, int seam_h);
void warp_batch_set_gain(void *desc, int kind, const float g[3], const float *d_map, int gw, int gh, int gcn, void *tabs);
int comp_gain_desc(const ssp_compensator *c, int index, int *kind, float g[3], const float **d_map, int *gw, int *gh, int *gcn);
int warp_batch_launch(const void *d_descs, int n, int max_dw, int max_dh, int max_prep_items, double algo_bytes, double prep_bytes, WarpRestPlan *plan, int far_px);
int warp_batch_launch_f32(const void *d_descs, int n, int max_dw, int max_dh, int max_prep_items, double algo_bytes, double prep_bytes, WarpRestPlan *plan);
void warp_rest_plan_release(WarpRestPlan *p);
int warp_rest_plan_settle(WarpRestPlan *plan, bool wait);
}  // namespace ssp

struct ComposeImage {
    Projector proj;
    int roi[4];                      // warper.warpRoi: what the caller sees (corners / sizes / resultRoi are OpenCV's)
    ssp_image *seam_mask = nullptr;  // seam-scale warped all-255 mask (sde.py:1591-1599), before dilation
    int n_live = 1, live[2][4];      // the rectangles of the roi the blender has to see (ssp_warp.hip live_parts): two for a frame that straddles u = +-pi*scale
};
// Batched path: what is warped and fed is a PART -- a whole frame, or one of the two live column ranges of a frame whose roi spans the full
// circle.  OpenCV warps and feeds such a frame's whole roi, 5/6 of it reflected garbage under a zero mask; nothing farther than 4 * 2^bands
// pixels from a set mask pixel reaches the panorama (k_warp_records_far), so feeding the live ranges grown by that reach as separate images
// -- same place in the feed order, same pixels, same pano-relative 2^bands grid -- gives the same sums bit for bit.
// Persistent per-part tables (the warped part and its mask are written straight into the blender's planes).
struct ComposePart {
    int img = 0;
    int roi[4] = {0, 0, 0, 0};
    ssp_image *dil = nullptr;
    float *tab = nullptr;
    int *lin = nullptr;
    void *gtab = nullptr;   // gain-map resize tables (exposure compensation fused into the warp)
    void *tiles = nullptr;  // per-tile source rectangles of the LDS-staged warp (written with the prep launch)
    void *cmap = nullptr;   // coordinate plane: the quantised map of every warped pixel, written with the prep launch (projections without separable tables; ssp_warp.hip)
};

struct ssp_composer {
    ssp_compose_config cfg;
    std::string warp_type;
    std::vector<float> K, R;
    std::vector<ComposeImage> imgs;
    int pano[4] = {0, 0, 0, 0};
    ssp_compensator *comp = nullptr;  // borrowed
    ssp_blender *blender = nullptr;
    ssp_image *mosaic = nullptr, *rmask = nullptr, *result = nullptr;
    double bytes_warp = 0, bytes_pyr = 0, bytes_blend = 0;
    bool batched = false;  // separable projection + 8UC3 frames: two launches warp every frame (mask prep fused)
    bool batched_f32 = false;                     // float frames, separable projection, float pyramids: one warp launch for all parts (k_warp_f32_batch)
    bool use_tables = false, use_cmap = false;   // where the fused warp's map comes from: the separable projections' tables / coordinate planes (ssp_warp.hip)
    std::vector<ComposePart> parts;   // batched path: feed units, image by image
    bool parts_split = false;         // parts follow the frames' live ranges (else: one part per frame, its whole roi)
    DescRing ring;
    WarpRestPlan rest_plan;  // learnt from the first panorama: few non-stageable tiles -> later panoramas skip the rest launch (ssp_warp.hip)
};

static void composer_free_results(ssp_composer *c)
{
    image_unref(c->mosaic); image_unref(c->rmask); image_unref(c->result);
    c->mosaic = c->rmask = c->result = nullptr;
}

static void composer_free_parts(ssp_composer *c)
{
    for (auto &pt : c->parts) {
        image_unref(pt.dil);
        pool_free(pt.tab); pool_free(pt.lin); pool_free(pt.gtab); pool_free(pt.tiles); pool_free(pt.cmap);
    }
    c->parts.clear();
}

// (re)build the feed units of the batched path and their persistent tables: split = by live ranges, else one part per frame
static int composer_build_parts(ssp_composer *c, bool split)
{
    composer_free_parts(c);
    c->parts_split = split;
    warp_rest_plan_release(&c->rest_plan);   // the tile records and the rest list belong to the parts
    for (int i = 0; i < (int)c->imgs.size(); ++i) {
        const ComposeImage &im = c->imgs[i];
        const int np = split ? im.n_live : 1;
        for (int k = 0; k < np; ++k) {
            ComposePart pt;
            pt.img = i;
            memcpy(pt.roi, split ? im.live[k] : im.roi, sizeof pt.roi);
            c->parts.push_back(pt);
        }
    }
    for (auto &pt : c->parts) {
        const ComposeImage &im = c->imgs[pt.img];
        const size_t dw4 = (size_t)warp_table_cols(pt.roi[2]);
        if (c->use_tables) SSP_TRY(pool_alloc(sizeof(float) * 2 * (dw4 + pt.roi[3]), (void **)&pt.tab));
        SSP_TRY(pool_alloc(warp_tile_bytes(pt.roi[2], pt.roi[3]), &pt.tiles));
        if (c->use_cmap) {
            SSP_TRY(pool_alloc(sizeof(uint32_t) * warp_cmap_words(pt.roi[2], pt.roi[3]), &pt.cmap));
            SSP_TRY(warp_cmap_set_projector(pt.cmap, im.proj));
        }
        if (c->cfg.mask_prep && im.seam_mask) {      // (external seam masks not handed in yet: ssp_composer_set_seam_masks builds these)
            SSP_TRY(image_new(im.seam_mask->w, im.seam_mask->h, 1, SSP_U8, &pt.dil));
            SSP_TRY(pool_alloc(sizeof(int) * warp_lin_ints(pt.roi[2], pt.roi[3], im.seam_mask->h), (void **)&pt.lin));
        }
    }
    return 0;
}

SSP_API int ssp_composer_destroy(ssp_composer *c)
{
    if (!c) return 0;
    composer_free_results(c);
    for (auto &im : c->imgs) image_unref(im.seam_mask);
    composer_free_parts(c);
    c->ring.destroy();
    warp_rest_plan_release(&c->rest_plan);
    if (c->blender) ssp_blender_destroy(c->blender);
    delete c;
    return 0;
}

SSP_API int ssp_composer_create(const ssp_compose_config *cfg, ssp_composer **out)
{
    SSP_REQUIRE(cfg && out, "composer: null argument");
    SSP_REQUIRE(cfg->n_images > 0 && cfg->src_w > 0 && cfg->src_h > 0 && cfg->K && cfg->R && cfg->warp_type, "composer: incomplete config");
    SSP_REQUIRE(cfg->src_depth == SSP_U8 || cfg->src_depth == SSP_F32, "composer: frames must be 8UC3 or 32FC3");
    SSP_REQUIRE(cfg->use_graph == 0, "composer: use_graph is reserved and must be 0 (launches are eager; see DESIGN.md)");
    SSP_REQUIRE(cfg->src_depth == SSP_U8 || cfg->blend_type == SSP_BLEND_MULTIBAND, "composer: float frames need the multiband blender (float mode)");
    SSP_TRY(ensure_init());
    ssp_composer *c = new ssp_composer();
    c->cfg = *cfg;
    c->warp_type = cfg->warp_type;
    c->cfg.warp_type = c->warp_type.c_str();
    c->K.assign(cfg->K, cfg->K + 9 * (size_t)cfg->n_images);
    c->R.assign(cfg->R, cfg->R + 9 * (size_t)cfg->n_images);
    c->cfg.K = c->K.data();
    c->cfg.R = c->R.data();
    ssp_warper *ws = nullptr;
    int rc = 0;
    std::vector<int> corners, sizes;
    c->imgs.resize(cfg->n_images);
    for (int i = 0; i < cfg->n_images && !rc; ++i) {
        ComposeImage &im = c->imgs[i];
        rc = make_projector(cfg->warp_type, cfg->warper_scale, im.proj);
        if (rc) break;
        set_camera(im.proj, &c->K[9 * (size_t)i], &c->R[9 * (size_t)i]);
        rc = detect_roi(im.proj, cfg->src_w, cfg->src_h, im.roi);  // warper.warpRoi, sde.py:1696
        if (rc) break;
        corners.push_back(im.roi[0]); corners.push_back(im.roi[1]);
        sizes.push_back(im.roi[2]); sizes.push_back(im.roi[3]);
    }
    if (!rc) rc = ssp_result_roi(cfg->n_images, corners.data(), sizes.data(), c->pano);  // sde.py:1807
    // seam-scale masks (sde.py:1539-1546, :1591-1599): all-255 mask of the seam-scale frame, NEAREST/CONSTANT warp
    if (!rc && cfg->mask_prep && !cfg->external_seam_masks) {
        if (!(cfg->seam_w > 0 && cfg->seam_h > 0 && cfg->seam_aspect > 0)) { ssp_composer_destroy(c); SSP_FAIL(SSP_ERR_ARG, "composer: mask_prep needs seam_w/seam_h/seam_aspect"); }
        rc = ssp_warper_create(cfg->warp_type, cfg->warper_scale * cfg->seam_aspect, &ws);
        ssp_image *ones = nullptr;
        if (!rc) rc = image_new(cfg->seam_w, cfg->seam_h, 1, SSP_U8, &ones);
        if (!rc) rc = ssp_image_fill(ones, 255);
        for (int i = 0; i < cfg->n_images && !rc; ++i) {
            float Ks[9];
            memcpy(Ks, &c->K[9 * (size_t)i], sizeof Ks);
            Ks[0] *= cfg->seam_aspect; Ks[2] *= cfg->seam_aspect; Ks[4] *= cfg->seam_aspect; Ks[5] *= cfg->seam_aspect;  // sde.py:1550-1555
            int corner[2];
            rc = ssp_warper_warp_image(ws, ones, Ks, &c->R[9 * (size_t)i], SSP_INTER_NEAREST, SSP_BORDER_CONSTANT, &c->imgs[i].seam_mask, corner);
        }
        image_unref(ones);
    }
    if (!rc) rc = ssp_blender_create(cfg->blend_type, &c->blender);
    if (!rc && cfg->blend_type == SSP_BLEND_MULTIBAND) {
        rc = ssp_blender_set_num_bands(c->blender, cfg->num_bands);
        if (!rc && cfg->src_depth == SSP_F32) rc = ssp_blender_set_float_mode(c->blender, 1);
    }
    if (!rc && cfg->blend_type == SSP_BLEND_FEATHER) rc = ssp_blender_set_sharpness(c->blender, cfg->sharpness);
    if (ws) ssp_warper_destroy(ws);
    // batched path: persistent outputs and tables for every frame
    // (frames beyond the fused warp kernel's 32-bit source offsets -- pitch >= 2^24 or >= 4 GiB -- take the per-image path)
    c->batched = !rc && cfg->src_depth == SSP_U8 && cfg->blend_type == SSP_BLEND_MULTIBAND &&
                 cfg->src_h <= 32767 && (size_t)cfg->src_w * 3 + 256 < ((size_t)1 << 24) && ((size_t)cfg->src_w * 3 + 256) * (size_t)cfg->src_h < ((size_t)1 << 32);
    if (c->batched) {
        // the separable projections (spherical / cylindrical / mercator) compute their map from per-column / per-row tables; the other thirteen
        // read it from coordinate planes built once per geometry (cfg->coordinate_planes / SSP_WARP_CMAP=1: those for the separable ones too)
        const bool sep = is_separable(c->imgs[0].proj.kind);
        c->use_tables = sep;
        c->use_cmap = !sep || cfg->coordinate_planes != 0 || (getenv("SSP_WARP_CMAP") && atoi(getenv("SSP_WARP_CMAP")) != 0);
        if (c->use_cmap)
            for (const auto &im : c->imgs)
                if (!warp_cmap_words(im.roi[2], im.roi[3])) c->use_cmap = false;       // a roi beyond 2^30 pixels
        if (!c->use_cmap && !sep) c->batched = false;
        if (getenv("SSP_NO_BATCH_GENERIC") && !sep) c->batched = false;                // (A/B: the per-frame path of rounds 1-3 for the non-separable projections)
    }
    c->batched_f32 = !rc && !c->batched && cfg->src_depth == SSP_F32 && cfg->blend_type == SSP_BLEND_MULTIBAND && is_separable(c->imgs[0].proj.kind) && !getenv("SSP_NO_BATCH_F32");
    if (c->batched_f32) { c->use_tables = true; c->use_cmap = false; }
    // the live ranges of every frame: a frame that straddles u = +-pi*scale (every closed 360-degree ring has some) is fed as its two ends
    for (int i = 0; i < cfg->n_images && !rc && (c->batched || c->batched_f32); ++i) {
        ComposeImage &im = c->imgs[i];
        rc = live_parts(im.proj, cfg->src_w, cfg->src_h, im.roi, getenv("SSP_NO_SPLIT") ? 0 : live_reach(cfg->num_bands), im.live, &im.n_live);
    }
    if (!rc && (c->batched || c->batched_f32)) rc = composer_build_parts(c, true);
    if (rc) { ssp_composer_destroy(c); return rc; }
    *out = c;
    return 0;
}

SSP_API int ssp_composer_set_compensator(ssp_composer *c, ssp_compensator *comp)
{
    SSP_REQUIRE(c, "composer: null");
    // cv's compensators multiply 8-bit images in place (apply -> saturating multiply / convertTo); the reference never hands them
    // float frames (sde.py:1754 runs before astype(np.float32) feeds nothing but int16).  Refuse instead of ignoring the gains.
    int kind = 0; float g3[3]; const float *dm = nullptr; int gw = 0, gh = 0, gcn = 0;
    const bool identity = !comp || (comp_gain_desc(comp, 0, &kind, g3, &dm, &gw, &gh, &gcn) == 0 && kind == 0);
    SSP_REQUIRE(!(c->cfg.src_depth == SSP_F32 && !identity), "composer: exposure compensation applies to 8-bit frames only; float frames must be compensated by the caller");
    c->comp = comp;      // (what of the gains decides whether a tile can be staged -- kind and gain-map shape -- is part of the rest plan's signature)
    return 0;
}
// Seam-scale masks from the caller -- the masks a seam finder has cut (sde.py:1618 -> :1760), 8UC1, one per frame, any size -- in place of the
// warped all-255 masks the composer makes itself at creation (which is what the reference feeds with --seam no).  Retained; call again when
// their contents change (the dilated copies and the interior flags are rebuilt by the next panorama).
SSP_API int ssp_composer_set_seam_masks(ssp_composer *c, int n, ssp_image *const *masks)
{
    SSP_REQUIRE(c && masks && n == (int)c->imgs.size(), "composer: one seam mask per frame");
    SSP_REQUIRE(c->cfg.mask_prep, "composer: seam masks need mask_prep (sde.py:1760-1772)");
    bool resized = false;
    for (int i = 0; i < n; ++i) {
        SSP_REQUIRE(masks[i] && masks[i]->cn == 1 && masks[i]->depth == SSP_U8 && masks[i]->w > 0 && masks[i]->h > 0, "composer: seam mask %d must be a non-empty 8UC1 image", i);
        ssp_image *old = c->imgs[i].seam_mask;
        resized = resized || !old || old->w != masks[i]->w || old->h != masks[i]->h;
    }
    for (int i = 0; i < n; ++i) {
        ++masks[i]->refs;
        image_unref(c->imgs[i].seam_mask);
        c->imgs[i].seam_mask = masks[i];
    }
    c->rest_plan.prep_key.clear();                      // the prep launch dilates them and flags their interiors
    if (resized && (c->batched || c->batched_f32)) return composer_build_parts(c, c->parts_split);
    return 0;
}
SSP_API int ssp_composer_warp_rest_tiles(ssp_composer *c, int *state, int *count)
{
    SSP_REQUIRE(c && state && count, "null");
    SSP_TRY(warp_rest_plan_settle(&c->rest_plan, true));     // waits for the first panorama's read-back; SSP_ERR_STATE if the list overflowed
    *state = c->rest_plan.state; *count = c->rest_plan.state == 2 ? c->rest_plan.count : -1;
    return 0;
}
// the same without blocking: *state 1 = the first panorama's read-back is still on its way (the blocking form stalls a host pipeline)
SSP_API int ssp_composer_warp_rest_tiles_nowait(ssp_composer *c, int *state, int *count)
{
    SSP_REQUIRE(c && state && count, "null");
    SSP_TRY(warp_rest_plan_settle(&c->rest_plan, false));
    *state = c->rest_plan.state; *count = c->rest_plan.state == 2 ? c->rest_plan.count : -1;
    return 0;
}
// Drop everything the composer has learnt from its geometry (tables of the prep launch, the rest list): the next panorama rebuilds it, as the
// reference's OpenCV rebuilds its maps in every warp call.  bench.py's `tables_rebuilt` figure calls this before every step.
SSP_API int ssp_composer_forget_geometry(ssp_composer *c)
{
    SSP_REQUIRE(c, "composer: null");
    SSP_TRY(warp_rest_plan_settle(&c->rest_plan, true));
    c->rest_plan.state = 0;
    c->rest_plan.prep_key.clear();
    return 0;
}
SSP_API int ssp_composer_pano_roi(const ssp_composer *c, int roi[4]) { SSP_REQUIRE(c && roi, "null"); memcpy(roi, c->pano, sizeof c->pano); return 0; }
SSP_API int ssp_composer_image_roi(const ssp_composer *c, int index, int roi[4])
{
    SSP_REQUIRE(c && roi && index >= 0 && index < (int)c->imgs.size(), "composer: image index out of range");
    memcpy(roi, c->imgs[index].roi, sizeof c->imgs[index].roi);
    return 0;
}

// The feed units of the batched path: a frame's whole roi, or the two live column ranges of a frame that straddles u = +-pi*scale.
// (Frames outside the batched path, and a compensator that needs a separate pass, feed whole rois: one part per frame.)
SSP_API int ssp_composer_num_parts(const ssp_composer *c, int *count)
{
    SSP_REQUIRE(c && count, "null");
    *count = (c->batched || c->batched_f32) ? (int)c->parts.size() : (int)c->imgs.size();
    return 0;
}
SSP_API int ssp_composer_part(const ssp_composer *c, int part, int *image_index, int roi[4])
{
    SSP_REQUIRE(c && roi, "null");
    const bool has_parts = c->batched || c->batched_f32;
    const int n = has_parts ? (int)c->parts.size() : (int)c->imgs.size();
    SSP_REQUIRE(part >= 0 && part < n, "composer: part index out of range");
    if (has_parts) { memcpy(roi, c->parts[part].roi, 4 * sizeof(int)); if (image_index) *image_index = c->parts[part].img; }
    else { memcpy(roi, c->imgs[part].roi, 4 * sizeof(int)); if (image_index) *image_index = part; }
    return 0;
}

// warp (+apply, +mask prep) and pyramid build for every frame: everything of the step except blender.blend
static int composer_feed_impl(ssp_composer *c, ssp_image *const *frames, bool planes_only);

SSP_API int ssp_composer_feed(ssp_composer *c, ssp_image *const *frames) { return composer_feed_impl(c, frames, false); }

// multi-GPU: stop after the level-0 planes are complete (warp, apply, border) so that strips can be exported and sent while
// ssp_composer_feed_pyramids builds this GPU's own pyramids
SSP_API int ssp_composer_feed_planes(ssp_composer *c, ssp_image *const *frames)
{
    SSP_REQUIRE(c && (c->batched || (c->cfg.src_depth == SSP_F32 && c->cfg.blend_type == SSP_BLEND_MULTIBAND)),
                "composer feed_planes: needs the batched path (8-bit frames, multiband, separable projection) or float frames with float pyramids");
    return composer_feed_impl(c, frames, true);
}
SSP_API int ssp_composer_feed_pyramids(ssp_composer *c)
{
    SSP_REQUIRE(c, "composer: null");
    return mb_feed_end(c->blender);
}

static int composer_feed_impl(ssp_composer *c, ssp_image *const *frames, bool planes_only)
{
    SSP_REQUIRE(c && frames, "composer feed: null argument");
    const ssp_compose_config &cfg = c->cfg;
    composer_free_results(c);
    if (cfg.mask_prep)
        for (const auto &im : c->imgs)
            if (!im.seam_mask) SSP_FAIL(SSP_ERR_STATE, "composer: external seam masks were announced but not handed in (ssp_composer_set_seam_masks)");
    int rc = ssp_blender_prepare(c->blender, c->pano[0], c->pano[1], c->pano[2], c->pano[3]);  // sde.py:1820
    c->bytes_warp = c->bytes_pyr = c->bytes_blend = 0;
    if (!rc && c->batched) {
        const int n = cfg.n_images;
        for (int i = 0; i < n; ++i) {
            const ssp_image *src = frames[i];
            if (!src || src->w != cfg.src_w || src->h != cfg.src_h || src->cn != 3 || src->depth != cfg.src_depth)
                return set_error(SSP_ERR_ARG, "composer run: frame %d does not match the configured %dx%d 3-channel frames", i, cfg.src_w, cfg.src_h);
            if (src->pitch >= ((size_t)1 << 24) || (size_t)src->pitch * (size_t)src->h >= ((size_t)1 << 32))
                return set_error(SSP_ERR_ARG, "composer run: frame %d has a row pitch of %zu bytes; the fused warp needs pitch < 2^24 and frames < 4 GiB", i, src->pitch);
        }
        // wrapped caller buffers with a tight pitch or an odd base are repacked first (ssp_runtime: image_aligned_source)
        std::vector<const ssp_image *> srcs(n, nullptr);
        struct TmpImages { std::vector<ssp_image *> v; ~TmpImages() { for (ssp_image *t : v) image_unref(t); } } staged;   // stream-ordered pool: safe to release once launched
        staged.v.assign(n, nullptr);
        for (int i = 0; i < n; ++i) SSP_TRY(image_aligned_source(frames[i], &srcs[i], &staged.v[i]));
        // exposure compensation (:1754) rides in the warp epilogue unless a gain map has the frame's own size (tiny frames)
        bool fused_gain = c->comp != nullptr;
        std::vector<int> gkind(n, 0), ggw(n, 0), ggh(n, 0), ggcn(n, 0);
        std::vector<const float *> gmap(n, nullptr);
        std::vector<float> gval(3 * (size_t)n, 1.f);
        for (int i = 0; i < n && c->comp; ++i) {
            SSP_TRY(comp_gain_desc(c->comp, i, &gkind[i], &gval[3 * (size_t)i], &gmap[i], &ggw[i], &ggh[i], &ggcn[i]));
            if (gkind[i] == 2 && ggw[i] == c->imgs[i].roi[2] && ggh[i] == c->imgs[i].roi[3]) fused_gain = false;
            if (gkind[i] != gkind[0] || (gkind[i] == 2 && ggcn[i] != ggcn[0])) fused_gain = false;      // one kind of gain per fused launch
        }
        // a compensator that has to run as a separate pass over whole warped frames (ssp_comp_apply) needs whole frames in the planes
        const bool want_split = !(c->comp && !fused_gain);
        if (want_split != c->parts_split) SSP_TRY(composer_build_parts(c, want_split));
        const int np = (int)c->parts.size();
        // the blender hands out the interiors of its bordered level-0 planes: the warp writes every part and its mask in place
        std::vector<int> tls(2 * np), sizes(2 * np);
        for (int k = 0; k < np; ++k) {
            tls[2 * k] = c->parts[k].roi[0]; tls[2 * k + 1] = c->parts[k].roi[1];
            sizes[2 * k] = c->parts[k].roi[2]; sizes[2 * k + 1] = c->parts[k].roi[3];
        }
        std::vector<FeedSlot> slots(np);
        SSP_TRY(mb_feed_begin(c->blender, np, tls.data(), sizes.data(), SSP_U8, slots.data()));
        const size_t dsz = warp_batch_desc_size();
        std::vector<char> hbuf(dsz * np);  // descriptors travel by value in the kernel arguments
        void *hv = hbuf.data();
        int max_dw = 0, max_dh = 0, max_items = 0, gain_items = 0;
        double prep_bytes = 0;
        for (int i = 0; i < n; ++i) c->bytes_warp += 3.0 * cfg.src_w * cfg.src_h;      // every source frame is read once ...
        for (int k = 0; k < np; ++k) {
            ComposePart &pt = c->parts[k];
            const ComposeImage &ci = c->imgs[pt.img];
            warp_batch_fill((char *)hv + dsz * k, ci.proj, srcs[pt.img], pt.roi, ci.roi[2], pt.roi[0] - ci.roi[0], SSP_BORDER_REFLECT, slots[k].img, slots[k].ipitch, slots[k].mask,
                            slots[k].mpitch, slots[k].xshift, pt.tab, cfg.mask_prep, ci.seam_mask, pt.dil, pt.lin, pt.tiles, pt.cmap);  // :1731 + :1740 (+ :1760-1772) in one pass
            const int dw4 = warp_table_cols(pt.roi[2]);
            int items = dw4 + pt.roi[3];
            if (cfg.mask_prep) items = warp_prep_items(pt.roi[2], pt.roi[3], ci.seam_mask->w, ci.seam_mask->h);
            max_dw = std::max(max_dw, pt.roi[2]); max_dh = std::max(max_dh, pt.roi[3]); max_items = std::max(max_items, items);
            c->bytes_warp += 4.0 * pt.roi[2] * pt.roi[3];                               // ... and every warped pixel and mask byte written once
            if (cfg.mask_prep) prep_bytes += 2.0 * ci.seam_mask->w * ci.seam_mask->h;
            if (fused_gain) {
                const int i = pt.img;
                if (gkind[i] == 2) {
                    gain_items = std::max(gain_items, dw4 + pt.roi[3]);   // the prep launch is sized by the largest per-part item count
                    if (!pt.gtab) SSP_TRY(pool_alloc(sizeof(int) * 2 * ((size_t)dw4 + pt.roi[3]), &pt.gtab));
                }
                warp_batch_set_gain((char *)hv + dsz * k, gkind[i], &gval[3 * (size_t)i], gmap[i], ggw[i], ggh[i], ggcn[i], pt.gtab);
            }
        }
        max_items += gain_items;
        // pixels farther than 4 * 2^bands from every set mask pixel never reach the panorama (k_warp_records_far): the warp skips such tiles and
        // leaves their image bytes unwritten.  Sound for INTEGER pyramids only -- (short)(L * 0) = 0 whatever L holds; a NaN left in a float plane
        // times a zero weight would poison the sums -- which is what this path feeds (8-bit frames into integer pyramids)
        SSP_REQUIRE(!c->blender->float_mode && cfg.src_depth == SSP_U8, "composer: the batched 8-bit path feeds integer pyramids only");
        const int far_px = getenv("SSP_WARP_NO_FAR") ? 0 : live_reach(c->blender->num_bands);
        SSP_TRY(warp_batch_launch(hv, np, max_dw, max_dh, max_items, c->bytes_warp, prep_bytes, &c->rest_plan, far_px));
        for (int i = 0; i < n; ++i) image_note_read(frames[i]);   // frames uploaded on another stream: the pool must not recycle them under this warp
        if (c->comp && !fused_gain) {
            for (int i = 0; i < n; ++i) {     // (parts are whole frames here)
                ssp_image view;  // the warped frame inside the blender's plane
                view.data = slots[i].img; view.pitch = slots[i].ipitch; view.w = c->imgs[i].roi[2]; view.h = c->imgs[i].roi[3]; view.cn = 3; view.depth = SSP_U8;
                view.owned = false;
                SSP_TRY(ssp_comp_apply(c->comp, i, &view));  // :1754
            }
        }
        if (planes_only) return mb_feed_border(c->blender);
        return mb_feed_end(c->blender);  // border + Gaussian pyramids (:1886 x n)
    }
    if (!rc && c->batched_f32) {
        // float frames (config 5), batched: every part is warped straight into its bordered level-0 planes (image and prepared mask) by ONE launch
        const int n = cfg.n_images;
        for (int i = 0; i < n; ++i) {
            const ssp_image *src = frames[i];
            if (!src || src->w != cfg.src_w || src->h != cfg.src_h || src->cn != 3 || src->depth != SSP_F32)
                return set_error(SSP_ERR_ARG, "composer run: frame %d does not match the configured %dx%d 3-channel frames", i, cfg.src_w, cfg.src_h);
            SSP_REQUIRE(src->pitch < ((size_t)1 << 32), "composer run: frame %d has a row pitch beyond 32 bits", i);
        }
        if (!c->parts_split) SSP_TRY(composer_build_parts(c, true));
        const int np = (int)c->parts.size();
        std::vector<int> tls(2 * np), sizes(2 * np);
        for (int k = 0; k < np; ++k) {
            tls[2 * k] = c->parts[k].roi[0]; tls[2 * k + 1] = c->parts[k].roi[1];
            sizes[2 * k] = c->parts[k].roi[2]; sizes[2 * k + 1] = c->parts[k].roi[3];
        }
        std::vector<FeedSlot> slots(np);
        SSP_TRY(mb_feed_begin(c->blender, np, tls.data(), sizes.data(), SSP_F32, slots.data()));
        const size_t dsz = warp_batch_desc_size();
        std::vector<char> hbuf(dsz * np);
        int max_dw = 0, max_dh = 0, max_items = 0;
        double prep_bytes = 0;
        for (int i = 0; i < n; ++i) c->bytes_warp += 12.0 * cfg.src_w * cfg.src_h;
        for (int k = 0; k < np; ++k) {
            ComposePart &pt = c->parts[k];
            const ComposeImage &ci = c->imgs[pt.img];
            warp_batch_fill(hbuf.data() + dsz * k, ci.proj, frames[pt.img], pt.roi, ci.roi[2], pt.roi[0] - ci.roi[0], SSP_BORDER_REFLECT, slots[k].img, slots[k].ipitch, slots[k].mask,
                            slots[k].mpitch, 0, pt.tab, cfg.mask_prep, ci.seam_mask, pt.dil, pt.lin, pt.tiles, nullptr);      // :1731 + :1740 (+ :1760-1772) in one pass
            const int dw4 = warp_table_cols(pt.roi[2]);
            int items = dw4 + pt.roi[3];
            if (cfg.mask_prep) items = warp_prep_items(pt.roi[2], pt.roi[3], ci.seam_mask->w, ci.seam_mask->h);
            max_dw = std::max(max_dw, pt.roi[2]); max_dh = std::max(max_dh, pt.roi[3]); max_items = std::max(max_items, items);
            c->bytes_warp += 13.0 * pt.roi[2] * pt.roi[3];
            if (cfg.mask_prep) prep_bytes += 2.0 * ci.seam_mask->w * ci.seam_mask->h;
        }
        SSP_TRY(warp_batch_launch_f32(hbuf.data(), np, max_dw, max_dh, max_items, c->bytes_warp, prep_bytes, &c->rest_plan));
        for (int i = 0; i < n; ++i) image_note_read(frames[i]);
        if (planes_only) return mb_feed_border(c->blender);
        return mb_feed_end(c->blender);
    }
    if (!rc && cfg.src_depth == SSP_F32 && cfg.blend_type == SSP_BLEND_MULTIBAND) {
        // float frames (config 5): every frame is warped straight into its bordered level-0 plane (image and validity mask from one map
        // evaluation), the prepared mask is copied in, and the pyramids of all frames are built by one batch of launches
        const int n = cfg.n_images;
        for (int i = 0; i < n; ++i) {
            const ssp_image *src = frames[i];
            if (!src || src->w != cfg.src_w || src->h != cfg.src_h || src->cn != 3 || src->depth != SSP_F32)
                return set_error(SSP_ERR_ARG, "composer run: frame %d does not match the configured %dx%d 3-channel frames", i, cfg.src_w, cfg.src_h);
        }
        std::vector<int> tls(2 * n), sizes(2 * n);
        for (int i = 0; i < n; ++i) {
            tls[2 * i] = c->imgs[i].roi[0]; tls[2 * i + 1] = c->imgs[i].roi[1];
            sizes[2 * i] = c->imgs[i].roi[2]; sizes[2 * i + 1] = c->imgs[i].roi[3];
        }
        std::vector<FeedSlot> slots(n);
        SSP_TRY(mb_feed_begin(c->blender, n, tls.data(), sizes.data(), SSP_F32, slots.data()));
        for (int i = 0; i < n && !rc; ++i) {
            const ComposeImage &ci = c->imgs[i];
            ssp_image view;   // the frame's interior inside the blender's plane
            view.data = slots[i].img; view.pitch = slots[i].ipitch; view.w = ci.roi[2]; view.h = ci.roi[3]; view.cn = 3; view.depth = SSP_F32; view.owned = false;
            ssp_image mview;  // ... and its mask's: the warp writes the validity mask there, the mask preparation works on it in place
            mview.data = slots[i].mask; mview.pitch = slots[i].mpitch; mview.w = ci.roi[2]; mview.h = ci.roi[3]; mview.cn = 1; mview.depth = SSP_U8; mview.owned = false;
            rc = warp_launch(ci.proj, frames[i], ci.roi, SSP_INTER_LINEAR, SSP_BORDER_REFLECT, &view, &mview);  // :1731 + :1740 in one pass
            image_note_read(frames[i]);
            if (!rc && cfg.mask_prep) {
                ssp_image *dil = nullptr;
                rc = ssp_dilate3x3(ci.seam_mask, &dil);                                                   // :1760
                if (!rc) rc = resize_linear_exact(dil, mview.w, mview.h, &mview, nullptr);                // :1767 + :1772, in place
                image_unref(dil);
            }
            if (!rc) c->bytes_warp += 12.0 * cfg.src_w * cfg.src_h + 13.0 * ci.roi[2] * ci.roi[3];
        }
        if (rc) return rc;
        if (planes_only) return mb_feed_border(c->blender);
        return mb_feed_end(c->blender);
    }
    for (int i = 0; i < cfg.n_images && !rc; ++i) {
        ssp_image *src = frames[i];
        if (!src || src->w != cfg.src_w || src->h != cfg.src_h || src->cn != 3 || src->depth != cfg.src_depth) {
            rc = set_error(SSP_ERR_ARG, "composer run: frame %d does not match the configured %dx%d 3-channel frames", i, cfg.src_w, cfg.src_h);
            break;
        }
        ssp_image *warped = nullptr, *mask = nullptr, *fmask = nullptr;
        const ComposeImage &ci = c->imgs[i];
        const int corner[2] = {ci.roi[0], ci.roi[1]};
        rc = image_new(ci.roi[2], ci.roi[3], 3, cfg.src_depth, &warped);
        if (!rc) rc = image_new(ci.roi[2], ci.roi[3], 1, SSP_U8, &mask);
        if (!rc && cfg.src_depth == SSP_U8) {
            rc = warp_launch(ci.proj, src, ci.roi, SSP_INTER_LINEAR, SSP_BORDER_REFLECT, warped, mask);  // :1731 + :1740 in one pass
        } else if (!rc) {
            rc = warp_launch(ci.proj, src, ci.roi, SSP_INTER_LINEAR, SSP_BORDER_REFLECT, warped, nullptr);
            if (!rc) {
                // the mask warp of the float configuration: same NEAREST/CONSTANT rule on an all-255 source
                ssp_image *ones = nullptr;
                rc = image_new(cfg.src_w, cfg.src_h, 1, SSP_U8, &ones);
                if (!rc) rc = ssp_image_fill(ones, 255);
                if (!rc) rc = warp_launch(ci.proj, ones, ci.roi, SSP_INTER_NEAREST, SSP_BORDER_CONSTANT, mask, nullptr);
                image_unref(ones);
            }
        }
        if (!rc && c->comp && cfg.src_depth == SSP_U8) rc = ssp_comp_apply(c->comp, i, warped);  // :1754
        if (!rc && cfg.mask_prep) {
            ssp_image *dil = nullptr;
            rc = ssp_dilate3x3(c->imgs[i].seam_mask, &dil);                                       // :1760
            if (!rc) rc = resize_linear_exact(dil, mask->w, mask->h, mask, &fmask);               // :1767 + :1772
            image_unref(dil);
        }
        if (!rc) rc = ssp_blender_feed(c->blender, warped, fmask ? fmask : mask, corner[0], corner[1]);  // :1886
        if (!rc) {
            double S = (double)src->w * src->h, D = (double)warped->w * warped->h, cb = depth_size(cfg.src_depth);
            c->bytes_warp += 3 * cb * S + (3 * cb + 1) * D;
        }
        image_unref(warped); image_unref(mask); image_unref(fmask);
        image_note_read(src);
    }
    return rc;
}

SSP_API int ssp_composer_run(ssp_composer *c, ssp_image *const *frames)
{
    SSP_TRY(ssp_composer_feed(c, frames));
    return ssp_blender_blend(c->blender, c->cfg.want_result_s16 ? &c->result : nullptr, &c->rmask, &c->mosaic);  // :1930 (+ :1938 saturation)
}

// multi-GPU: the panorama roi is the union over ALL GPUs' frames; this GPU's blender is prepared with it
SSP_API int ssp_composer_set_pano_roi(ssp_composer *c, const int roi[4])
{
    SSP_REQUIRE(c && roi && roi[2] > 0 && roi[3] > 0, "composer: bad pano roi");
    for (const auto &im : c->imgs)
        SSP_REQUIRE(im.roi[0] >= roi[0] && im.roi[1] >= roi[1] && im.roi[0] + im.roi[2] <= roi[0] + roi[2] && im.roi[1] + im.roi[3] <= roi[1] + roi[3],
                    "composer: the pano roi does not contain every frame's roi");
    memcpy(c->pano, roi, sizeof c->pano);
    return 0;
}
SSP_API int ssp_composer_blender(ssp_composer *c, ssp_blender **out)
{
    SSP_REQUIRE(c && out, "composer: null");
    *out = c->blender;  // borrowed: valid while the composer lives
    return 0;
}
// blend only the rectangle (pano-relative, aligned to 2^bands) that this GPU is responsible for
SSP_API int ssp_composer_finish_region(ssp_composer *c, int x0, int y0, int w, int h)
{
    SSP_REQUIRE(c, "composer: null");
    composer_free_results(c);
    return ssp_blender_blend_region(c->blender, x0, y0, w, h, c->cfg.want_result_s16 ? &c->result : nullptr, &c->rmask, &c->mosaic);
}

SSP_API int ssp_composer_result(ssp_composer *c, ssp_image **mosaic, ssp_image **rmask, ssp_image **result)
{
    SSP_REQUIRE(c, "composer: null");
    if (!c->mosaic) SSP_FAIL(SSP_ERR_STATE, "composer: no result yet (call ssp_composer_run)");
    if (mosaic) *mosaic = c->mosaic;
    if (rmask) *rmask = c->rmask;
    if (result) *result = c->result;
    return 0;
}

SSP_API int ssp_composer_algorithmic_bytes(const ssp_composer *c, double *warp, double *pyramid, double *blend)
{
    SSP_REQUIRE(c, "composer: null");
    if (warp) *warp = c->bytes_warp;
    if (pyramid) *pyramid = c->bytes_pyr;
    if (blend) *blend = c->bytes_blend;
    return 0;
}
