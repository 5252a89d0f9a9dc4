/*
 * ssp_math.h -- deterministic float32 elementary functions for the panorama warp path.
 *
 * Why this exists: OpenCV's projectors (warpers_inl.hpp, reached from
 * stitching_detailed_enhanced.py:1557/:1731 via cv.PyRotationWarper.warp) call the C library's
 * sinf/cosf/atan2f/... on float32 values.  Those results depend on the libm build (glibc's are
 * within ~1 ULP, not correctly rounded), and no GPU math library returns the same bits.  Both the
 * CPU oracle (oracle/ssp_oracle.c) and the HIP kernels therefore evaluate every transcendental
 * through THIS header: each function computes in binary64 with an error far below a float32 half
 * ULP and rounds once to float32.  The result is (a) within 0.5000001 ULP of the true value, hence
 * within 1 ULP of any conforming libm, and (b) bit-identical between gcc/x86-64 and hipcc/gfx950,
 * because only IEEE-754 correctly rounded operations (+ - * / sqrt, conversions) are used and FMA
 * contraction is disabled in both builds (-ffp-contract=off).
 *
 * The oracle can also be built with -DSSP_ORACLE_LIBM to call glibc instead; tests/ use that
 * build to bound the distance between "our spec" and a glibc-based OpenCV build.
 *
 * No lookup tables, no bit tricks except frexp/ldexp-style scaling done by exact multiplications.
 */
#ifndef SSP_MATH_H
#define SSP_MATH_H

#if defined(__HIPCC__)
#define SSP_HD __host__ __device__ inline
#else
#define SSP_HD static inline
#endif

#define SSP_PI_D      3.14159265358979323846
#define SSP_PIO2_D    1.57079632679489661923
#define SSP_PI_F      ((float)SSP_PI_D)          /* static_cast<float>(CV_PI) */

/* ---- double kernels ------------------------------------------------------------------------ */

SSP_HD double ssp_d_abs(double x) { return x < 0 ? -x : x; }

SSP_HD double ssp_d_sqrt(double x) { return __builtin_sqrt(x); }

/* round to nearest integer (ties irrelevant here), |x| < 2^51 */
SSP_HD double ssp_d_rint(double x)
{
    const double big = 6755399441055744.0; /* 1.5 * 2^52 */
    return (x + big) - big;
}

/* sin and cos of a reduced argument |r| <= pi/4 (+slack): Taylor series, error < 1e-18 */
SSP_HD double ssp_d_ksin(double r)
{
    double z = r * r;
    double p = -1.0 / 355687428096000.0;             /* 1/17! */
    p = p * z + 1.0 / 1307674368000.0;                /* 1/15! */
    p = p * z - 1.0 / 6227020800.0;                   /* 1/13! */
    p = p * z + 1.0 / 39916800.0;                     /* 1/11! */
    p = p * z - 1.0 / 362880.0;                       /* 1/9!  */
    p = p * z + 1.0 / 5040.0;                         /* 1/7!  */
    p = p * z - 1.0 / 120.0;                          /* 1/5!  */
    p = p * z + 1.0 / 6.0;                            /* 1/3!  */
    return r - (r * z) * p;
}

SSP_HD double ssp_d_kcos(double r)
{
    double z = r * r;
    double p = 1.0 / 6402373705728000.0;              /* 1/18! */
    p = p * z - 1.0 / 20922789888000.0;               /* 1/16! */
    p = p * z + 1.0 / 87178291200.0;                  /* 1/14! */
    p = p * z - 1.0 / 479001600.0;                    /* 1/12! */
    p = p * z + 1.0 / 3628800.0;                      /* 1/10! */
    p = p * z - 1.0 / 40320.0;                        /* 1/8!  */
    p = p * z + 1.0 / 720.0;                          /* 1/6!  */
    p = p * z - 1.0 / 24.0;                           /* 1/4!  */
    p = p * z + 0.5;                                  /* 1/2!  */
    return 1.0 - z * p;
}

/* Cody-Waite reduction by pi/2 for |x| up to ~2^27 (float32 arguments far beyond any panorama).
 * P1 has 27 significant bits, so n*P1 is exact for |n| < 2^26. */
SSP_HD void ssp_d_sincos(double x, double *s, double *c)
{
    const double INV_PIO2 = 0.6366197723675814;
    const double P1 = 0x1.921fb54000000p+0;   /* pi/2, leading 27 bits          */
    const double P2 = 0x1.10b4610000000p-30;  /* next 27 bits                   */
    const double P3 = 0x1.a62633145c06ep-58;  /* remainder, rounded to binary64 */
    double n = ssp_d_rint(x * INV_PIO2);
    double r = ((x - n * P1) - n * P2) - n * P3;
    double sr = ssp_d_ksin(r), cr = ssp_d_kcos(r);
    /* quadrant = n mod 4 */
    double q = n - 4.0 * ssp_d_rint(n * 0.25);   /* in {-2,-1,0,1,2} */
    if (q == 0.0)                { *s = sr;  *c = cr;  }
    else if (q == 1.0)           { *s = cr;  *c = -sr; }
    else if (q == -1.0)          { *s = -cr; *c = sr;  }
    else                         { *s = -sr; *c = -cr; }
}

/* atan for any finite x: two half-angle steps then Taylor; error < 1e-16 relative */
SSP_HD double ssp_d_atan(double x)
{
    double ax = ssp_d_abs(x);
    int inv = 0;
    if (ax > 1.0) { ax = 1.0 / ax; inv = 1; }
    /* atan(a) = 2 atan(a / (1 + sqrt(1 + a^2))), twice: |a| <= tan(pi/16) = 0.1989 */
    ax = ax / (1.0 + ssp_d_sqrt(1.0 + ax * ax));
    ax = ax / (1.0 + ssp_d_sqrt(1.0 + ax * ax));
    double z = ax * ax;
    double p = 1.0 / 27.0;
    p = -1.0 / 25.0 + z * p;
    p = 1.0 / 23.0 + z * p;
    p = -1.0 / 21.0 + z * p;
    p = 1.0 / 19.0 + z * p;
    p = -1.0 / 17.0 + z * p;
    p = 1.0 / 15.0 + z * p;
    p = -1.0 / 13.0 + z * p;
    p = 1.0 / 11.0 + z * p;
    p = -1.0 / 9.0 + z * p;
    p = 1.0 / 7.0 + z * p;
    p = -1.0 / 5.0 + z * p;
    p = 1.0 / 3.0 + z * p;
    double a = 4.0 * (ax - (ax * z) * p);
    if (inv) a = SSP_PIO2_D - a;
    return x < 0 ? -a : a;
}

/* atan2 with the C99 quadrant conventions for finite arguments (NaN in -> NaN out) */
SSP_HD double ssp_d_atan2(double y, double x)
{
    if (x != x || y != y) return x + y;
    if (y == 0.0) {
        /* sign of zero is not tracked: +0 assumed; the callers never depend on -0 */
        return (x >= 0.0) ? 0.0 : SSP_PI_D;
    }
    if (x == 0.0) return y > 0 ? SSP_PIO2_D : -SSP_PIO2_D;
    double ay = ssp_d_abs(y), ax = ssp_d_abs(x);
    double a;
    if (ay <= ax) a = ssp_d_atan(ay / ax);
    else          a = SSP_PIO2_D - ssp_d_atan(ax / ay);
    if (x < 0) a = SSP_PI_D - a;
    return y < 0 ? -a : a;
}

/* exact 2^k by binary exponentiation of 2.0 / 0.5, |k| <= 1000 */
SSP_HD double ssp_d_pow2i(int k)
{
    double base = 2.0, r = 1.0;
    if (k < 0) { base = 0.5; k = -k; }
    while (k) {
        if (k & 1) r *= base;
        base *= base;
        k >>= 1;
    }
    return r;
}

/* exp for |x| < 700 */
SSP_HD double ssp_d_exp(double x)
{
    const double INV_LN2 = 1.44269504088896338700;
    const double LN2_HI = 0.693147180369123816490;     /* 0x3FE62E42FEE00000 */
    const double LN2_LO = 1.90821492927058770002e-10;  /* ln2 - LN2_HI */
    if (x > 700.0) x = 700.0;
    if (x < -700.0) x = -700.0;
    double n = ssp_d_rint(x * INV_LN2);
    double r = (x - n * LN2_HI) - n * LN2_LO;
    double p = 1.0 / 6227020800.0;   /* 1/13! */
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    return p * ssp_d_pow2i((int)n);
}

/* natural log for finite x > 0 (covers float32 subnormals after widening) */
SSP_HD double ssp_d_log(double x)
{
    const double LN2_HI = 0.693147180369123816490;
    const double LN2_LO = 1.90821492927058770002e-10;
    const double SQRT2 = 1.41421356237309504880;
    const double SQRTH = 0.70710678118654752440;
    int e = 0;
    /* exact scaling into [sqrt(1/2), sqrt(2)) */
    while (x >= 4294967296.0) { x *= (1.0 / 4294967296.0); e += 32; }
    while (x < (1.0 / 4294967296.0)) { x *= 4294967296.0; e -= 32; }
    while (x >= SQRT2) { x *= 0.5; e += 1; }
    while (x < SQRTH) { x *= 2.0; e -= 1; }
    double s = (x - 1.0) / (x + 1.0);
    double z = s * s;
    double p = 1.0 / 27.0;
    p = 1.0 / 25.0 + z * p;
    p = 1.0 / 23.0 + z * p;
    p = 1.0 / 21.0 + z * p;
    p = 1.0 / 19.0 + z * p;
    p = 1.0 / 17.0 + z * p;
    p = 1.0 / 15.0 + z * p;
    p = 1.0 / 13.0 + z * p;
    p = 1.0 / 11.0 + z * p;
    p = 1.0 / 9.0 + z * p;
    p = 1.0 / 7.0 + z * p;
    p = 1.0 / 5.0 + z * p;
    p = 1.0 / 3.0 + z * p;
    double lm = 2.0 * (s + (s * z) * p);
    double de = (double)e;
    return (de * LN2_HI + lm) + de * LN2_LO;
}

SSP_HD double ssp_d_sinh(double x)
{
    double ax = ssp_d_abs(x);
    double r;
    if (ax < 0.5) {
        double z = ax * ax;
        double p = 1.0 / 1307674368000.0;  /* 1/15! */
        p = p * z + 1.0 / 6227020800.0;
        p = p * z + 1.0 / 39916800.0;
        p = p * z + 1.0 / 362880.0;
        p = p * z + 1.0 / 5040.0;
        p = p * z + 1.0 / 120.0;
        p = p * z + 1.0 / 6.0;
        r = ax + (ax * z) * p;
    } else {
        double e = ssp_d_exp(ax);
        r = 0.5 * (e - 1.0 / e);
    }
    return x < 0 ? -r : r;
}

SSP_HD double ssp_d_cosh(double x)
{
    double e = ssp_d_exp(ssp_d_abs(x));
    return 0.5 * (e + 1.0 / e);
}

/* ---- float32 entry points (one rounding each) ------------------------------------------------ */

SSP_HD float ssp_sinf(float x)  { double s, c; ssp_d_sincos((double)x, &s, &c); return (float)s; }
SSP_HD float ssp_cosf(float x)  { double s, c; ssp_d_sincos((double)x, &s, &c); return (float)c; }
SSP_HD float ssp_tanf(float x)  { double s, c; ssp_d_sincos((double)x, &s, &c); return (float)(s / c); }
SSP_HD float ssp_atanf(float x) { return (float)ssp_d_atan((double)x); }
SSP_HD float ssp_atan2f(float y, float x) { return (float)ssp_d_atan2((double)y, (double)x); }
SSP_HD float ssp_asinf(float x)
{
    double d = (double)x;
    if (!(d >= -1.0 && d <= 1.0)) return (float)(d - d) / 0.0f;      /* NaN like libm */
    return (float)ssp_d_atan2(d, ssp_d_sqrt((1.0 - d) * (1.0 + d)));
}
SSP_HD float ssp_acosf(float x)
{
    double d = (double)x;
    if (!(d >= -1.0 && d <= 1.0)) return (float)(d - d) / 0.0f;
    return (float)ssp_d_atan2(ssp_d_sqrt((1.0 - d) * (1.0 + d)), d);
}
SSP_HD float ssp_logf(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return (x - x) / 0.0f;
    if (x == 0.0f) return -1.0f / 0.0f;
    if (x > 3.4028234663852886e38f) return x;
    return (float)ssp_d_log((double)x);
}
SSP_HD float ssp_sinhf(float x) { return (float)ssp_d_sinh((double)x); }
SSP_HD float ssp_coshf(float x) { return (float)ssp_d_cosh((double)x); }

#endif /* SSP_MATH_H */
