/*
 * ssp.h -- C ABI of the MI355X-native warp / compensate / blend hot path (libssp_hip.so).
 *
 * This is the drop-in boundary for the cv2 object protocol that the reference drives from
 * stitching_detailed_enhanced.py (sde.py) compose_imgs_to_panorama (:1355-1954).  Every entry
 * point names the reference call site it replaces.  Plain pointers and sizes only: no torch, no
 * C++ types.  Host pointers are borrowed for the duration of a call; "image" handles own device
 * memory (HBM) and let warp -> apply -> feed -> blend run without leaving the GPU.
 *
 * Error convention (cv2 raises cv2.error; sde.py:1567-1586 catches it): every function returns
 * 0 on success or a non-zero ssp_status and stores a thread-local message readable through
 * ssp_last_error().  Nothing aborts the process.  There is NO CPU fallback: if no gfx950 device
 * or the kernels are missing, calls fail with SSP_ERR_DEVICE.
 */
#ifndef SSP_H
#define SSP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    SSP_OK = 0,
    SSP_ERR_ARG = 1,      /* bad type/shape/value (cv2: error -215 assertion failed) */
    SSP_ERR_DEVICE = 2,   /* HIP runtime / no device */
    SSP_ERR_MEMORY = 3,   /* allocation failure (sde.py:1573 records OpenCL's equivalent) */
    SSP_ERR_STATE = 4     /* call order violated (feed before prepare, second blend, ...) */
} ssp_status;

/* cv2 constants used by the caller (sde.py:755-766, :1562-1564, :1595-1597) */
enum { SSP_INTER_NEAREST = 0, SSP_INTER_LINEAR = 1, SSP_INTER_AREA = 3 };
enum { SSP_BORDER_CONSTANT = 0, SSP_BORDER_REPLICATE = 1, SSP_BORDER_REFLECT = 2, SSP_BORDER_WRAP = 3,
       SSP_BORDER_REFLECT_101 = 4 };
/* element depths = OpenCV depth codes */
enum { SSP_U8 = 0, SSP_S16 = 3, SSP_F32 = 5 };

typedef struct ssp_image ssp_image;             /* device-resident 2-D array (cv.UMat stand-in, sde.py:1539, :1886) */
typedef struct ssp_warper ssp_warper;           /* cv.PyRotationWarper */
typedef struct ssp_compensator ssp_compensator; /* cv.detail.ExposureCompensator family */
typedef struct ssp_blender ssp_blender;         /* cv.detail.Blender / FeatherBlender / MultiBandBlender */
typedef struct ssp_composer ssp_composer;       /* batched warp->apply->feed->blend plan (one launch sequence / hipGraph) */
typedef struct ssp_timer ssp_timer;             /* hipEvent pair on the library stream */

/* ---- runtime ------------------------------------------------------------------------------------- */
const char *ssp_last_error(void);
int ssp_version(void);
int ssp_init(int device);                 /* select device, create the stream and the HBM pool */
int ssp_device_count(int *count);
int ssp_device_name(char *buf, int len);
int ssp_sync(void);                       /* hipStreamSynchronize on the library stream */
int ssp_set_stream(void *hip_stream);     /* run on a caller-owned hipStream_t (e.g. torch's current stream) */
int ssp_device_copy(void *dst_dev, const void *src_dev, size_t bytes);   /* D2D on the library stream */
int ssp_device_copy_kernel(void *dst_dev, const void *src_dev, size_t bytes);   /* the same as a 16-byte-per-lane kernel: what HBM yields to a kernel (bench.py's copy ceiling) */
/* more than one panorama in flight: extra streams; ssp_use_stream switches the stream of every following call without
 * synchronising (NULL = back to the library's own stream); the pool keeps separate free lists per stream */
int ssp_stream_create(void **out_hip_stream);
int ssp_stream_destroy(void *hip_stream);
int ssp_stream_sync(void *hip_stream);
int ssp_use_stream(void *hip_stream);
int ssp_current_stream(void **out_hip_stream);   /* the stream the next call will launch on */
/* Threading: the library keeps ONE current stream per process (the last ssp_use_stream / ssp_set_stream wins) and every entry
 * point launches on it.  Drive the library from one host thread, or serialise the calls of several threads yourself; the HBM
 * pool is mutex-protected, the current-stream selection is not.  Parallelism comes from streams (several panoramas in flight
 * from one thread), not from host threads. */
int ssp_pool_stats(size_t *bytes_in_use, size_t *bytes_cached);
int ssp_pool_trim(void);
int ssp_timer_create(ssp_timer **t);
int ssp_timer_start(ssp_timer *t);
int ssp_timer_stop(ssp_timer *t);
int ssp_timer_elapsed_ms(ssp_timer *t, float *ms);   /* synchronises on the stop event */
int ssp_timer_destroy(ssp_timer *t);
/* per-kernel hipEvent timing (bench.py's live roofline measurement); names are the kernel families */
int ssp_profile_enable(int on);
int ssp_profile_reset(void);
int ssp_profile_count(int *n);
int ssp_profile_get(int idx, char *name, int name_len, int *launches, float *total_ms, double *algo_bytes);

/* PMC calibration: stream `total_bytes` `reps` times at 4 / 8 / 16 bytes per lane (kernels k_calib_read / k_calib_write);
 * run under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE to learn the factor between the counter and real bytes */
int ssp_calibrate_stream(int bytes_per_lane, size_t total_bytes, int reps);

/* ---- device images (cv.UMat stand-in: sde.py:1539-1541, :1599 .get(), :1886) ------------------------ */
int ssp_image_create(int width, int height, int channels, int depth, ssp_image **out);
int ssp_image_upload(const void *host, int width, int height, int channels, int depth, ssp_image **out);
int ssp_image_wrap(void *dev_ptr, size_t pitch_bytes, int width, int height, int channels, int depth, ssp_image **out);
int ssp_image_download(const ssp_image *img, void *host);          /* tightly packed rows; synchronises */
int ssp_image_info(const ssp_image *img, int *width, int *height, int *channels, int *depth, size_t *pitch, void **dev_ptr);
int ssp_image_retain(ssp_image *img);
int ssp_image_release(ssp_image *img);
int ssp_image_fill(ssp_image *img, double value);
int ssp_image_convert(const ssp_image *src, int depth, ssp_image **out);   /* ndarray.astype(int16) at sde.py:1755 (saturating) */
int ssp_image_all_equal(const ssp_image *img, int value, int *flag);        /* 8-bit: *flag = every sample == value (the all-255 mask of sde.py:1739); synchronises */

/* ---- warper: cv.PyRotationWarper (sde.py:1545-1546, :1684-1688) ------------------------------------- */
int ssp_warper_create(const char *type, float scale, ssp_warper **out);   /* 16 type strings, sde.py:218-237 */
int ssp_warper_destroy(ssp_warper *w);
int ssp_warper_get_scale(const ssp_warper *w, float *scale);
int ssp_warper_set_scale(ssp_warper *w, float scale);
/* warper.warpRoi((w,h), K, R) -> (x,y,w,h)                                   sde.py:1696 */
int ssp_warper_roi(ssp_warper *w, int src_w, int src_h, const float K[9], const float R[9], int roi[4]);
/* warper.warp(src, K, R, interp, border) -> (corner, dst)                     sde.py:1557, :1591, :1731, :1740
 * host form: dst is caller-allocated roi[3] x roi[2] x channels (size from ssp_warper_roi) */
int ssp_warper_warp(ssp_warper *w, const void *src, int src_w, int src_h, int channels, int depth, const float K[9],
                    const float R[9], int interp, int border, void *dst, int dst_w, int dst_h, int corner[2]);
/* device form: returns a new image handle; stays in HBM */
int ssp_warper_warp_image(ssp_warper *w, const ssp_image *src, const float K[9], const float R[9], int interp, int border,
                          ssp_image **dst, int corner[2]);
/* fused form of the pair of calls at sde.py:1731 + :1740 (image LINEAR/REFLECT + all-255 mask NEAREST/CONSTANT):
 * one pass, maps never materialised.  mask may be NULL. */
int ssp_warper_warp_with_mask(ssp_warper *w, const ssp_image *src, const float K[9], const float R[9], int border,
                              ssp_image **dst, ssp_image **mask, int corner[2]);
/* What of a frame's roi the multiband blender has to see (no reference counterpart): one rectangle (x, y, w, h in
 * warped coordinates) -- the roi itself for an ordinary frame -- or two for a frame that straddles u = +-pi*scale, whose roi from
 * warpRoi (sde.py:1696) spans the full circle while its mask is set at the two ends only: the set columns grown by 4 * 2^num_bands.
 * The composer feeds these rectangles; parallel.plan_strips shards closed rings by them.  capacity >= 2 rectangles. */
int ssp_warper_live_parts(ssp_warper *w, int src_w, int src_h, const float K[9], const float R[9], int num_bands, int *parts_xywh, int capacity, int *count);
/* cv2 extras (unused by the reference): buildMaps, warpPoint, warpPointBackward */
/* PyRotationWarper::warpBackward(src, K, R, interp, border, dst_size) -- not used by the reference (SURVEY 8(b) nice-to-have) */
int ssp_warper_warp_backward(ssp_warper *w, const ssp_image *src, const float K[9], const float R[9], int interp, int border, int dst_w, int dst_h,
                             ssp_image **dst);
int ssp_warper_build_maps(ssp_warper *w, int src_w, int src_h, const float K[9], const float R[9], float *xmap, float *ymap,
                          int dst_w, int dst_h, int roi[4]);
int ssp_warper_warp_point(ssp_warper *w, float x, float y, const float K[9], const float R[9], float uv[2]);
int ssp_warper_warp_point_backward(ssp_warper *w, float u, float v, const float K[9], const float R[9], float xy[2]);

/* ---- helpers on the path (sde.py:1760-1772, :1807) --------------------------------------------------- */
int ssp_result_roi(int n, const int *corners_xy, const int *sizes_wh, int roi[4]);   /* cv.detail.resultRoi */
int ssp_dilate3x3(const ssp_image *mask, ssp_image **out);                           /* cv.dilate(mask, None) */
int ssp_resize_linear_exact(const ssp_image *mask, int dst_w, int dst_h, ssp_image **out); /* cv.resize(INTER_LINEAR_EXACT) */
int ssp_bitwise_and(const ssp_image *a, const ssp_image *b, ssp_image **out);        /* cv.bitwise_and */

/* ---- frame prologue of the compose loop (sde.py:1699-1711) --------------------------- */
/* cv.resize(img, None, fx=fx, fy=fy, interpolation=cv.INTER_AREA) for decimation (sde.py:1701-1707), 8-bit, any channel count;
 * dsize = (cvRound(w*fx), cvRound(h*fy)).  `lut` (256 entries, host memory, may be NULL) is applied to the result in the same
 * pass: with ssp_bw_point_lut that is adjust_black_and_white_point (sde.py:1711, image_processors.py:32-41). */
int ssp_resize_area(const ssp_image *src, double fx, double fy, const uint8_t *lut, ssp_image **out);
/* the 256-entry table of ((clip(v, black, white) - black) * (255 / (white - black))).astype(uint8) (image_processors.py:35-39) */
int ssp_bw_point_lut(int black, int white, uint8_t lut[256]);
/* adjust_black_and_white_point on its own (no resize: sde.py:1708-1711 when compose_scale ~ 1) */
int ssp_apply_lut(const ssp_image *src, const uint8_t lut[256], ssp_image **out);

/* ---- seam estimation and timelapse on device-resident warps ------------------- */
/* cv.detail.SeamFinder_createDefault(cv.detail.SeamFinder_VORONOI_SEAM).find(images, corners, masks) (sde.py:243-249, :1618):
 * the 8UC1 masks are cut in place, pairs visited in PairwiseSeamFinder::run's order.  (SeamFinder_NO leaves the masks as they
 * are and needs no call.) */
int ssp_seam_voronoi(int n, const int *corners_xy, ssp_image *const *masks);
/* cv.detail_DpSeamFinder(costFunc).find(images, corners, masks) (sde.py:243-249 "dp_color" / "dp_colorgrad" -- the reference's
 * default -- called at :1618 with the float32 seam-scale warps of :1601-1604): OpenCV seam_finders.cpp DpSeamFinder::find.
 * cost_func 0 = 'COLOR' (images 8UC3 or 32FC3: the colour differences are the same numbers), 1 = 'COLOR_GRAD' (32FC3 only: cv2's
 * 8-bit BGR2GRAY is a fixed-point grey that is not restated; the reference passes float32); images of their masks' sizes; the 8UC1
 * masks are cut in place.  Components of any size: sweep lines of more than 4096 cells keep their cost lines in global memory instead of LDS.
 * pair_order (may be NULL, n(n-1) ints) receives the image pairs in the order they were processed.  Gradients, edge costs and
 * the dynamic programme run on the device, the component graph of each pair on the host (csrc/ssp_seam_dp.hip). */
int ssp_seam_dp(int n, const int *corners_xy, ssp_image *const *images, ssp_image *const *masks, int cost_func, int *pair_order);
/* cv.detail.Timelapser_createDefault(type) (sde.py:1822-1851) */
enum { SSP_TIMELAPSER_AS_IS = 0, SSP_TIMELAPSER_CROP = 1 };
typedef struct ssp_timelapser ssp_timelapser;
int ssp_timelapser_create(int type, ssp_timelapser **out);
int ssp_timelapser_destroy(ssp_timelapser *t);
int ssp_timelapser_initialize(ssp_timelapser *t, int n, const int *corners_xy, const int *sizes_wh);  /* .initialize(corners, sizes) */
int ssp_timelapser_process(ssp_timelapser *t, const ssp_image *img_s16c3, int tl_x, int tl_y);         /* .process(img, mask, tl): the mask is unused by OpenCV */
int ssp_timelapser_get_dst(ssp_timelapser *t, ssp_image **out);                                        /* .getDst(): retained 16SC3 canvas */
int ssp_timelapser_dst_roi(const ssp_timelapser *t, int roi[4]);
/* cv.bitwise_and(a, b, mask=mask) (sde.py:1842): zero where the mask is zero */
int ssp_bitwise_and_masked(const ssp_image *a, const ssp_image *b, const ssp_image *mask, ssp_image **out);

/* ---- exposure compensation (sde.py:649-665, :1613, :1754) -------------------------------------------- */
enum { SSP_COMP_NO = 0, SSP_COMP_GAIN = 1, SSP_COMP_GAIN_BLOCKS = 2, SSP_COMP_CHANNELS = 3, SSP_COMP_CHANNELS_BLOCKS = 4 };
int ssp_comp_create(int type, ssp_compensator **out);           /* ExposureCompensator_createDefault(type) */
int ssp_comp_destroy(ssp_compensator *c);
int ssp_comp_set_nr_feeds(ssp_compensator *c, int n);
int ssp_comp_set_block_size(ssp_compensator *c, int w, int h);
int ssp_comp_set_nr_filtering(ssp_compensator *c, int n);
/* compensator.feed(corners, images, masks): seam-scale warped u8c3 images + u8 masks (device handles) */
int ssp_comp_feed(ssp_compensator *c, int n, const int *corners_xy, ssp_image *const *images, ssp_image *const *masks);
/* compensator.apply(index, corner, image, mask): mutates image in place (u8c3) */
int ssp_comp_apply(ssp_compensator *c, int index, ssp_image *image);
int ssp_comp_num_images(const ssp_compensator *c, int *n);
int ssp_comp_get_gains(const ssp_compensator *c, double *gains, int capacity, int *count);   /* getMatGains */
int ssp_comp_get_gain_map(const ssp_compensator *c, int index, float *map, int capacity, int *w, int *h, int *cn);
/* ExposureCompensator::setMatGains (cv2 API; the reference never calls it): gains without a feed.  Scalar kinds take n (GAIN) or
 * 3n (CHANNELS, b g r per image) doubles; block kinds take one float32 map per image, installed in image order from index 0. */
int ssp_comp_set_gains(ssp_compensator *c, const double *gains, int count);
int ssp_comp_set_gain_map(ssp_compensator *c, int index, const float *map, int w, int h, int cn);

/* ---- blenders (sde.py:1806-1820, :1886-1889, :1930) --------------------------------------------------- */
enum { SSP_BLEND_NO = 0, SSP_BLEND_FEATHER = 1, SSP_BLEND_MULTIBAND = 2 };
int ssp_blender_create(int type, ssp_blender **out);            /* Blender_createDefault / detail_MultiBandBlender / detail_FeatherBlender */
int ssp_blender_destroy(ssp_blender *b);
int ssp_blender_set_num_bands(ssp_blender *b, int n);           /* setNumBands, sde.py:1815 */
int ssp_blender_get_num_bands(const ssp_blender *b, int *n);
int ssp_blender_set_sharpness(ssp_blender *b, float s);         /* setSharpness, sde.py:1819 */
int ssp_blender_set_float_mode(ssp_blender *b, int on);         /* f32 pyramids (BASELINE config 5; no OpenCV counterpart) */
int ssp_blender_prepare(ssp_blender *b, int x, int y, int w, int h);       /* prepare(resultRoi) */
/* blender.feed(img, mask, tl): img s16c3 (or u8c3 holding the same values, or f32c3 in float mode), mask u8 */
int ssp_blender_feed(ssp_blender *b, ssp_image *img, ssp_image *mask, int tl_x, int tl_y);
/* n feeds at once, in order (same result as n ssp_blender_feed calls; every pyramid level of all images is one launch) */
int ssp_blender_feed_batch(ssp_blender *b, int n, ssp_image *const *imgs, ssp_image *const *masks, const int *tls_xy);
/* blender.blend() -> (result s16c3 | f32c3, result_mask u8).  mosaic_u8 (optional) is the saturated 8-bit
 * panorama that cv.imwrite produces from the int16 result (sde.py:1938). */
int ssp_blender_blend(ssp_blender *b, ssp_image **result, ssp_image **result_mask, ssp_image **mosaic_u8);
/* multi-GPU: export / import the partial sums (sum of (short)(L*w), sum of w; before normalisation) of one pyramid level.
 * (x0, y0, w, h) is a rectangle of the PADDED pano in level-0 pixels relative to the pano corner, aligned to 2^num_bands;
 * the buffers are tightly packed device arrays of (h >> level) x (w >> level) samples (s16x3 | f32x3, and f32). */
int ssp_blender_level_info(const ssp_blender *b, int level, int *w, int *h);
int ssp_blender_export_partial(ssp_blender *b, int level, int x0, int y0, int w, int h, void *lap_dev, void *weight_f32_dev);
int ssp_blender_import_partial(ssp_blender *b, int level, int x0, int y0, int w, int h, const void *lap_dev, const void *weight_f32_dev);
/* Multi-GPU strip exchange, level-0 protocol (parallel.py plan_strips(levels=False); no reference counterpart; the all-level
 * protocol further down is the default): instead of partial pyramid sums
 * (13.3 B/px) a GPU sends the part of a fed frame's bordered level-0 planes that another GPU needs (8UC3 image with its
 * BORDER_REFLECT band + 8UC1 mask, 4 B/px, tightly packed device buffers); the receiver feeds it as an image that fills its
 * rectangle and rebuilds the pyramids.  Rectangles are pano-relative level-0 coordinates; fed strips must be aligned to
 * 2^bands (>= 2 bands), exported ones to 4 columns inside the frame's padded rectangle.  order_feeds sorts all fed images by key (the global image index) so that the f32 weight sums run in the same
 * order as on one GPU. */
int ssp_blender_export_strips(ssp_blender *b, int n, const int *feed_indices, const int *rects_xywh, void *const *imgs_u8c3, void *const *masks_u8);
int ssp_blender_feed_strips(ssp_blender *b, int n, const int *rects_xywh, const void *const *imgs_u8c3, const void *const *masks_u8);
/* Double-buffered multi-GPU step (parallel.HipStripPipeline; no reference counterpart): feed_strips_begin
 * takes the received strips like ssp_blender_feed_strips but leaves their pyramids pending; feed_end_pair then builds the
 * pending pyramids of two blenders (the strips of panorama k, the own frames of panorama k+1; b may be NULL) in one chain. */
/* Strip buffers in PLANE layout (planes != 0): the exporter writes each strip with the row pitch and apron offset of a level-0
 * plane, so the receiver's buffer IS the plane (no import copy); buffers are ssp_strip_buffer_bytes long, 16-byte aligned, and must
 * stay untouched until the panorama they were fed to is blended.  Both ends of an exchange use the same setting. */
int ssp_blender_set_strip_layout(ssp_blender *b, int planes);
int ssp_strip_buffer_bytes(int w, int h, int bytes_per_px, int planes, size_t *bytes);   /* 1: mask, 3: 8UC3 strip, 12: 32FC3 strip (float pyramids) */
int ssp_blender_feed_strips_begin(ssp_blender *b, int n, const int *rects_xywh, const void *const *imgs_u8c3, const void *const *masks_u8);
/* All-level strips (parallel.plan_strips(levels=True)): the same rectangle of EVERY level of a fed image's planes
 * (Gaussian levels and weight levels, aprons included) in one buffer of ssp_level_strip_buffer_bytes; the receiver uses the buffer as the
 * planes themselves and builds nothing, so the rectangle only has to cover its region grown by 2^bands.  export needs the pyramids built
 * (after ssp_blender_feed / ssp_composer_feed_pyramids); buffers are 16-byte aligned and stay untouched until the panorama is blended.
 * origins_x[i]: pano-relative column where the padded rectangle of strip i's image starts on its owner (StripPlan.prect): rows travel in whole
 * 16-byte chunks of the owner's planes, so the strip's first column sits (column distance x sample size) mod 16 bytes into the buffer's rows. */
int ssp_blender_export_level_strips(ssp_blender *b, int n, const int *feed_indices, const int *rects_xywh, void *const *bufs, int num_bands);
int ssp_blender_feed_level_strips(ssp_blender *b, int n, const int *rects_xywh, const int *origins_x, const void *const *bufs, int num_bands);   /* num_bands: what the buffers were sized for; must be the prepared blender's */
int ssp_level_strip_buffer_bytes(int w, int h, int num_bands, int float_pyramids, size_t *bytes);   /* num_bands: the blender's effective band count */
int ssp_blender_feed_end_pair(ssp_blender *a, ssp_blender *b);
int ssp_blender_order_feeds(ssp_blender *b, const int *keys, int n);

/* blend() restricted to such a rectangle: outputs have the rectangle's size clipped to the final roi (consumes the state) */
int ssp_blender_blend_region(ssp_blender *b, int x0, int y0, int w, int h, ssp_image **result, ssp_image **result_mask, ssp_image **mosaic_u8);

/* ---- composer: the whole compose loop of sde.py:1673-1930 as one device-resident plan ---------------- */
typedef struct {
    const char *warp_type;    /* config.warp */
    float warper_scale;       /* warped_image_scale * compose_work_aspect, sde.py:1687 */
    int n_images;
    int src_w, src_h;         /* compose-scale frame size (all frames equal) */
    int src_depth;            /* SSP_U8 (configs 1-4) or SSP_F32 (config 5) */
    const float *K;           /* n_images x 9 */
    const float *R;           /* n_images x 9 */
    int blend_type;           /* SSP_BLEND_* */
    int num_bands;
    float sharpness;
    int mask_prep;            /* 1: dilate + resize + and with the seam-scale masks (sde.py:1760-1772) */
    int seam_w, seam_h;       /* seam-scale frame size (sde.py:1539: the all-255 masks have this size) */
    float seam_aspect;        /* seam scale / compose scale: K and the warper scale are multiplied by it (sde.py:1546-1555) */
    int want_result_s16;      /* also produce the int16 result of blend() (the 8-bit mosaic and mask always are) */
    int use_graph;            /* reserved, must be 0: the step is GPU-bound with eager launches */
    int external_seam_masks;  /* 1 (with mask_prep): the seam-scale masks come from the caller (ssp_composer_set_seam_masks, before the first run) */
    int coordinate_planes;    /* 1: the separable projections (spherical / cylindrical / mercator) read their map from coordinate planes too, as the other
                               * thirteen always do: 4 bytes per warped pixel and a plane-building launch per geometry buy a warp kernel with a third
                               * fewer instructions -- for callers that compose many panoramas with the same cameras.  0: tables (one panorama per camera set) */
} ssp_compose_config;
int ssp_composer_create(const ssp_compose_config *cfg, ssp_composer **out);
int ssp_composer_destroy(ssp_composer *c);
int ssp_composer_set_compensator(ssp_composer *c, ssp_compensator *comp);   /* gains from a prior feed (sde.py:1613) */
/* seam-scale masks from the caller (what a seam finder returned, sde.py:1618; 8UC1, one per frame) in place of the warped all-255 masks the
 * composer makes itself with mask_prep (the reference's masks with --seam no); retained; call again when their contents change */
int ssp_composer_set_seam_masks(ssp_composer *c, int n, ssp_image *const *masks);
/* what the composer has learnt about its geometry from its first panorama: *state 0 unknown, 2 known (the call waits for a read-back that
 * is on its way); *count = tiles of the LDS-staged warp that cannot be staged (-1 until known).  Few of them: later panoramas do them
 * inline and skip one launch.  SSP_ERR_STATE when the device-side list counted more tiles than the launch has (tiles were dropped). */
int ssp_composer_warp_rest_tiles(ssp_composer *c, int *state, int *count);
int ssp_composer_warp_rest_tiles_nowait(ssp_composer *c, int *state, int *count);   /* the same without waiting: *state 1 = the read-back is still on its way */
/* forget it all (trigonometry / resize tables of the prep launch, the rest list): the next panorama rebuilds the geometry, as OpenCV's
 * warper rebuilds its maps in every warp call (sde.py:1731, :1740) -- the like-for-like cost, bench.py's `tables_rebuilt` */
int ssp_composer_forget_geometry(ssp_composer *c);
int ssp_composer_pano_roi(const ssp_composer *c, int roi[4]);
int ssp_composer_image_roi(const ssp_composer *c, int index, int roi[4]);
/* the composer's feed units ("parts"): one per frame, two for a frame that straddles u = +-pi*scale (ssp_warper_live_parts); image_roi / pano_roi
 * stay OpenCV's (sde.py:1696-1698, :1807) */
int ssp_composer_num_parts(const ssp_composer *c, int *count);
int ssp_composer_part(const ssp_composer *c, int part, int *image_index, int roi[4]);
/* one step: all frames warp+mask (+apply) -> pyramids -> blend; result handles are owned by the composer */
int ssp_composer_run(ssp_composer *c, ssp_image *const *frames);
int ssp_composer_result(ssp_composer *c, ssp_image **mosaic_u8, ssp_image **result_mask, ssp_image **result_s16);
/* multi-GPU form of a step: feed (warp + pyramids of this GPU's frames into a blender prepared with the GLOBAL pano
 * roi), exchange partial sums through the borrowed blender (ssp_blender_export/import_partial), finish the own region */
int ssp_composer_set_pano_roi(ssp_composer *c, const int roi[4]);
int ssp_composer_feed(ssp_composer *c, ssp_image *const *frames);
/* multi-GPU: ssp_composer_feed in two halves -- level-0 planes (warp, apply, border: strips can be exported), then the pyramids */
int ssp_composer_feed_planes(ssp_composer *c, ssp_image *const *frames);
int ssp_composer_feed_pyramids(ssp_composer *c);
int ssp_composer_blender(ssp_composer *c, ssp_blender **borrowed);
int ssp_composer_finish_region(ssp_composer *c, int x0, int y0, int w, int h);
int ssp_composer_algorithmic_bytes(const ssp_composer *c, double *warp, double *pyramid, double *blend);

#ifdef __cplusplus
}
#endif
#endif /* SSP_H */
