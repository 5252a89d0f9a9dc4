/* orc_internal.h -- shared helpers of the CPU oracle (test infrastructure only). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
/* Row loops marked ORC_PAR_FOR run in parallel in the OpenMP flavour (liborc_omp.so: the multi-core CPU baseline of bench.py).
 * Every marked loop writes disjoint rows and reads only data finished before it, so the flavours are bit-identical
 * (tests/test_oracle_pixels.py compares them); without -fopenmp the macros vanish. */
#ifdef _OPENMP
#include <omp.h>
#define ORC_PRAGMA(x) _Pragma(#x)
#define ORC_PAR_FOR ORC_PRAGMA(omp parallel for schedule(static))
#define ORC_PAR ORC_PRAGMA(omp parallel)
#define ORC_FOR ORC_PRAGMA(omp for schedule(static))
#else
#define ORC_PAR_FOR
#define ORC_PAR
#define ORC_FOR
#endif
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ssp_oracle.h"
#include "../include/ssp_math.h"

#ifdef SSP_ORACLE_LIBM
/* what a glibc-linked OpenCV build calls (warpers_inl.hpp uses sinf/cosf/... on floats) */
#define M_SIN(x) sinf(x)
#define M_COS(x) cosf(x)
#define M_TAN(x) tanf(x)
#define M_ATAN(x) atanf(x)
#define M_ATAN2(y, x) atan2f(y, x)
#define M_ASIN(x) asinf(x)
#define M_ACOS(x) acosf(x)
#define M_LOG(x) logf(x)
#define M_SINH(x) sinhf(x)
#define M_COSH(x) coshf(x)
#else
#define M_SIN(x) ssp_sinf(x)
#define M_COS(x) ssp_cosf(x)
#define M_TAN(x) ssp_tanf(x)
#define M_ATAN(x) ssp_atanf(x)
#define M_ATAN2(y, x) ssp_atan2f(y, x)
#define M_ASIN(x) ssp_asinf(x)
#define M_ACOS(x) ssp_acosf(x)
#define M_LOG(x) ssp_logf(x)
#define M_SINH(x) ssp_sinhf(x)
#define M_COSH(x) ssp_coshf(x)
#endif

void orc_set_error(const char *fmt, ...);

/* cvRound(float): round half to even; x86 cvtss2si yields INT_MIN when the value does not fit */
static inline int orc_cv_round(float v)
{
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT_MIN;
    return (int)rintf(v);
}
static inline int orc_cv_round_d(double v)
{
    if (!(v >= -2147483648.0 && v < 2147483648.0)) return INT_MIN;
    return (int)rint(v);
}
static inline int16_t orc_sat_s16(int v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
static inline uint8_t orc_sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
/* static_cast<short>(float) as compiled for x86-64: cvttss2si then the low 16 bits */
static inline int16_t orc_trunc_s16(float f)
{
    int t = (f > -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN;
    return (int16_t)(uint16_t)(t & 0xffff);
}

/* cv::borderInterpolate */
static inline int orc_border(int p, int len, int type)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (type == ORC_BORDER_REPLICATE) return p < 0 ? 0 : len - 1;
    if (type == ORC_BORDER_REFLECT || type == ORC_BORDER_REFLECT_101) {
        int delta = type == ORC_BORDER_REFLECT_101;
        if (len == 1) return 0;
        do {
            if (p < 0) p = -p - 1 + delta;
            else p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (type == ORC_BORDER_WRAP) {
        if (p < 0) p -= ((p - len + 1) / len) * len;
        if (p >= len) p %= len;
        return p;
    }
    return -1; /* BORDER_CONSTANT */
}

#endif
