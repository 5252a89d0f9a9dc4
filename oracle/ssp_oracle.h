/*
 * ssp_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the OpenCV 4.6.0 arithmetic that the reference reaches from
 * stitching_detailed_enhanced.py:1355-1954 (compose_imgs_to_panorama) through the cv2 objects
 * PyRotationWarper / ExposureCompensator / Blender.  OpenCV's source is a third-party dependency
 * (opencv-python==4.6.0.66, /root/reference/requirements.txt:2) that is absent from
 * /root/reference and from this image, so the restatement follows the published algorithm
 *  and the reference's own call sites.
 *
 * PARITY STATUS: geometry (warpRoi / resultRoi / camera prep / num_bands) is pinned by the 30
 * known-answer tests recovered from the reference's recorded runs (tests/golden/kat_*.json).
 * PIXEL VALUES ARE "PARITY UNPINNED": the reference holds no pixel-level fixture for remap,
 * pyramids, blending or exposure compensation.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef SSP_ORACLE_H
#define SSP_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* interpolation / border codes = cv2 constants (stitching_detailed_enhanced.py:755-766) */
enum { ORC_INTER_NEAREST = 0, ORC_INTER_LINEAR = 1, ORC_INTER_AREA = 3 };
enum { ORC_BORDER_CONSTANT = 0, ORC_BORDER_REPLICATE = 1, ORC_BORDER_REFLECT = 2, ORC_BORDER_WRAP = 3,
       ORC_BORDER_REFLECT_101 = 4 };
enum { ORC_U8 = 0, ORC_S16 = 3, ORC_F32 = 5 };  /* CV_8U, CV_16S, CV_32F depth codes */

typedef struct orc_warper orc_warper;
typedef struct orc_blender orc_blender;
typedef struct orc_comp orc_comp;

const char *orc_last_error(void);
int orc_uses_libm(void);

/* ---- warper (sde.py:1545, :1684 cv.PyRotationWarper) ---- */
orc_warper *orc_warper_create(const char *type, float scale);
void orc_warper_destroy(orc_warper *w);
int orc_warper_set_camera(orc_warper *w, const float K[9], const float R[9]);
void orc_warper_get_projector(const orc_warper *w, float k[9], float rinv[9], float r_kinv[9], float k_rinv[9], float t[3]);
void orc_warper_map_forward(const orc_warper *w, float x, float y, float *u, float *v);
void orc_warper_map_backward(const orc_warper *w, float u, float v, float *x, float *y);
/* sde.py:1696 warpRoi -> (x, y, w, h) */
int orc_warper_roi(orc_warper *w, int src_w, int src_h, const float K[9], const float R[9], int roi[4]);
/* buildMaps: xmap/ymap of roi size (caller-allocated from orc_warper_roi) */
int orc_warper_build_maps(orc_warper *w, int src_w, int src_h, const float K[9], const float R[9],
                          float *xmap, float *ymap, int roi[4]);
/* sde.py:1557/:1591/:1731/:1740 warp; dst is roi[3] x roi[2] x cn of the same depth */
int orc_warper_warp(orc_warper *w, const void *src, int src_w, int src_h, int cn, int depth,
                    const float K[9], const float R[9], int interp, int border, void *dst, int roi[4]);

/* PyRotationWarper::warpBackward: src has the size of warpRoi(dst size); dst is dst_h x dst_w x cn */
int orc_warper_warp_backward(orc_warper *w, const void *src, int src_w, int src_h, int cn, int depth, const float K[9], const float R[9], int interp,
                             int border, int dst_w, int dst_h, void *dst);

/* cv::remap with two float maps */
int orc_remap(const void *src, int src_w, int src_h, int cn, int depth, const float *xmap, const float *ymap,
              int dst_w, int dst_h, int interp, int border, void *dst);

/* ---- pyramids (imgproc/pyramids.cpp) ---- */
void orc_pyr_down_s16(const int16_t *src, int w, int h, int cn, int16_t *dst);        /* dst ((w+1)/2, (h+1)/2) */
void orc_pyr_down_f32(const float *src, int w, int h, int cn, float *dst);
void orc_pyr_up_s16(const int16_t *src, int w, int h, int cn, int16_t *dst, int dw, int dh);
void orc_pyr_up_f32(const float *src, int w, int h, int cn, float *dst, int dw, int dh);

/* ---- helpers on the path (sde.py:1760-1772, :1701) ---- */
void orc_dilate3x3_u8(const uint8_t *src, int w, int h, uint8_t *dst);
void orc_resize_linear_exact_u8(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh);
void orc_resize_linear_f32(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh);
void orc_resize_area_u8(const uint8_t *src, int sw, int sh, int cn, uint8_t *dst, int dw, int dh);
void orc_resize_area_u8_scale(const uint8_t *src, int sw, int sh, int cn, double fx, double fy, const uint8_t *lut, uint8_t *dst, int dw, int dh);
void orc_bw_point_lut(int black, int white, uint8_t lut[256]);
void orc_distance_l1(const uint8_t *mask, int w, int h, float *dist);
void orc_result_roi(int n, const int *corners, const int *sizes, int roi[4]);
/* sde.py:243-249, :1618 SeamFinder_VORONOI_SEAM: masks (u8, sizes (w,h)) are cut in place */
void orc_seam_voronoi(int n, const int *corners, const int *sizes, uint8_t *const *masks);
/* sde.py:243-249, :1618 cv.detail_DpSeamFinder(costFunc): images float32 BGR of the masks' sizes; cost_func 0 COLOR, 1 COLOR_GRAD;
 * order_out (optional): the n(n-1)/2 pairs in processing order.  Returns 0, or -1 with orc_last_error set. */
int orc_seam_dp(int n, const int *corners, const int *sizes, const float *const *images, uint8_t *const *masks, int cost_func, int *order_out);
void orc_seam_dp_gradients(const float *img_bgr, int w, int h, float *gradx, float *grady);

/* ---- blenders (sde.py:1806-1820, :1886, :1930) ---- */
enum { ORC_BLEND_NO = 0, ORC_BLEND_FEATHER = 1, ORC_BLEND_MULTIBAND = 2 };
orc_blender *orc_blender_create(int type);
void orc_blender_destroy(orc_blender *b);
void orc_blender_set_num_bands(orc_blender *b, int n);
int orc_blender_num_bands(const orc_blender *b);
void orc_blender_set_sharpness(orc_blender *b, float s);
void orc_blender_set_float_mode(orc_blender *b, int on);   /* f32 pyramid variant (config 5; no OpenCV counterpart) */
int orc_blender_prepare(orc_blender *b, int x, int y, int w, int h);
int orc_blender_feed(orc_blender *b, const void *img, const uint8_t *mask, int w, int h, int tlx, int tly);
int orc_blender_blend(orc_blender *b, void *dst, uint8_t *dst_mask);   /* dst: final roi size, s16c3 (or f32c3) */
/* introspection for multi-GPU tests: accumulated (un-normalised) pyramids */
int orc_blender_level_size(const orc_blender *b, int level, int *w, int *h);
const int16_t *orc_blender_level_lap(const orc_blender *b, int level);
const float *orc_blender_level_weight(const orc_blender *b, int level);
int orc_blender_add_partial(orc_blender *b, int level, const int32_t *lap, const float *wgt);

/* ---- exposure compensators (sde.py:649-665, :1613, :1754) ---- */
enum { ORC_COMP_NO = 0, ORC_COMP_GAIN = 1, ORC_COMP_GAIN_BLOCKS = 2, ORC_COMP_CHANNELS = 3, ORC_COMP_CHANNELS_BLOCKS = 4 };
orc_comp *orc_comp_create(int type, int bl_w, int bl_h, int nr_feeds, int nr_filter);
void orc_comp_destroy(orc_comp *c);
int orc_comp_feed(orc_comp *c, int n, const int *corners, const int *sizes, const uint8_t *const *images,
                  const uint8_t *const *masks);                      /* images u8c3, sizes (w,h) */
int orc_comp_apply(orc_comp *c, int index, uint8_t *image, int w, int h);   /* in place, u8c3 */
int orc_comp_num_images(const orc_comp *c);
int orc_comp_gains(const orc_comp *c, double *out);                 /* n (GAIN) or 3n (CHANNELS) */
int orc_comp_gain_map_size(const orc_comp *c, int index, int *w, int *h, int *cn);
int orc_comp_gain_map(const orc_comp *c, int index, float *out);

int orc_set_threads(int n);   /* OpenMP flavour only: threads for the marked row loops; returns the count in effect */

#ifdef __cplusplus
}
#endif

#endif
