// ssp_blender.hpp -- blender state shared by ssp_blend.hip (ABI glue, NO / feather blenders) and ssp_multiband.hip.
#pragma once
#include "ssp_internal.hpp"

#define SSP_MAX_BANDS 16
#define SSP_APRON 4  // pixels of border stored around every pyramid level

namespace ssp {

// A 2-D array in HBM whose element (0,0) is `base`; indices [-APRON, w+APRON) x [-APRON, h+APRON) are valid memory.
struct Plane {
    char *base = nullptr;
    size_t pitch = 0;
    void *alloc = nullptr;
};

// One fed image of the multiband blender.
//   MultiBandBlender::feed pads the image to a rectangle snapped to multiples of 2^bands (border: BORDER_REFLECT for the
//   image, 0 for the weight map).  Level 0 is stored WITH that border (the warp kernel writes the interior in place, a
//   border kernel fills the rest), and every level carries an APRON-pixel BORDER_REFLECT_101 apron -- the border pyrDown
//   applies -- so the pyramid kernels contain no border logic at all.
struct FeedRec {
    int iw = 0, ih = 0, left = 0, top = 0;                   // image size and its position inside the padded rectangle
    int pw[SSP_MAX_BANDS + 1], ph[SSP_MAX_BANDS + 1];        // padded level sizes
    int rx[SSP_MAX_BANDS + 1], ry[SSP_MAX_BANDS + 1];        // rectangle origin per level (pano level coordinates)
    Plane G[SSP_MAX_BANDS + 1];                              // level 0: g0_depth x3 ; levels >= 1: int16x3 (f32x3 in float mode; u8x3 with lvl8)
    Plane W[SSP_MAX_BANDS + 1];                              // level 0: u8 mask     ; levels >= 1: f32 weights
    int g0_depth = SSP_U8;
    bool lvl8 = false;                                       // fed 8-bit into integer pyramids: every Gaussian level stays within [0, 255] and is stored as u8x3
    int level_bytes() const { return lvl8 ? 3 : (g0_depth == SSP_F32 ? 12 : 6); }   // bytes per pixel of G[l], l >= 1 (float frames only exist in float mode)
};

// where a producer (the warp kernel, or a copy) writes one image and its mask
struct FeedSlot {
    uint8_t *img; size_t ipitch;   // first pixel of the image interior inside the bordered level-0 plane
    uint8_t *mask; size_t mpitch;
    int xshift;                    // (column of the interior inside the padded rectangle) mod 4: the planes are 4-byte aligned at multiples of 4 columns
};

}  // namespace ssp

struct ssp_blender {
    int type = SSP_BLEND_NO;
    int want_bands = 5, num_bands = 0;
    float sharpness = 0.02f;
    bool float_mode = false;
    bool prepared = false;
    int roi[4] = {0, 0, 0, 0}, final_roi[4] = {0, 0, 0, 0};
    // NO / FEATHER accumulators
    ssp_image *dst = nullptr, *dst_mask = nullptr, *dst_weight = nullptr;
    // MULTIBAND
    int lw[SSP_MAX_BANDS + 1], lh[SSP_MAX_BANDS + 1];
    std::vector<ssp::FeedRec> feeds;
    int pending = 0;  // feeds handed out by mb_feed_begin whose pyramids are not built yet
    bool obj_pending = false;  // ... by ssp_blender_feed (object API): their pyramids are built together by whoever needs them first (blend)
    bool border_done = false;  // ... and whether their level-0 borders are already filled (mb_feed_border)
    bool strip_planes = false; // multi-GPU strips travel in the layout of a level-0 plane: the receive buffer IS the plane (no import copy)
    ssp_image *ext_lap[SSP_MAX_BANDS + 1] = {nullptr}, *ext_w[SSP_MAX_BANDS + 1] = {nullptr};
    ssp::DescRing ring;  // per-level image descriptors of the blend kernels
    // the descriptor tables of the last few blends, by content: a stream of panoramas with fixed geometry gets the same planes from the pool
    // step after step, so the table it needs is already on the device and the per-step upload (a runtime copy kernel on the stream) goes away
    struct DescCache { std::vector<char> host; size_t key_bytes = 0; void *dev = nullptr; hipEvent_t last_use = nullptr; hipStream_t used_on = nullptr; unsigned long long stamp = 0; };
    DescCache desc_cache[4];
    unsigned long long desc_stamp = 0;
};

namespace ssp {
// multiband implementation (ssp_multiband.hip)
void mb_release(ssp_blender *b);
// reserve the bordered level-0 planes of n images; the caller fills the interiors through `slots`, then calls mb_feed_end
int mb_feed_begin(ssp_blender *b, int n, const int *tls_xy, const int *sizes_wh, int depth, FeedSlot *slots, bool append = false);
int mb_flush(ssp_blender *b);        // pyramids of the images fed one by one through the object API (no-op otherwise)
int mb_feed_border(ssp_blender *b);  // level-0 planes complete (exportable); mb_feed_end builds the pyramids
int mb_feed_end(ssp_blender *b);
int mb_feed_end_pair(ssp_blender *a, ssp_blender *b);  // pending images of both blenders in one chain of launches (b may be null)
// feed n device images by copying them into the bordered planes (object API)
int mb_feed_images(ssp_blender *b, int n, ssp_image *const *imgs, ssp_image *const *masks, const int *tls_xy);
int mb_run_levels(ssp_blender *b, ssp_image *result, ssp_image *rmask, ssp_image *mosaic, int export_level, const int *region, void *exp_lap, float *exp_w);
int mb_import_partial(ssp_blender *b, int level, int x0, int y0, int w, int h, const void *lap, const void *wgt);
int mb_export_strips(ssp_blender *b, int n, const int *feeds, const int *rects_xywh, void *const *imgs, void *const *masks);
int mb_feed_strips(ssp_blender *b, int n, const int *rects_xywh, const void *const *imgs, const void *const *masks, bool defer = false);
int mb_order_feeds(ssp_blender *b, const int *keys, int n);
size_t mb_strip_buffer_bytes(int w, int h, int bytes_per_px, bool planes);
// all-level strips: the same rectangle of every pyramid level in one buffer; the receiver launches nothing
size_t mb_level_strip_buffer_bytes(int num_bands, bool float_mode, int w, int h);
int mb_export_level_strips(ssp_blender *b, int n, const int *feeds, const int *rects_xywh, void *const *bufs);
int mb_feed_level_strips(ssp_blender *b, int n, const int *rects_xywh, const int *origins_x, const void *const *bufs);
}  // namespace ssp
