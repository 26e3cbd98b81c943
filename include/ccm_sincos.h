/* ccm_sincos.h -- one sine/cosine shared by the CPU oracle and the HIP kernels.
 *
 * The reference steers the BRIEF pattern with
 *     a = (float)cos(angle), b = (float)sin(angle)      (cslam/src/ORBextractor.cpp:104-105)
 * where `angle` is a float in radians and overload resolution picks libm
 * cosf/sinf.  glibc's cosf and ROCm's device cosf are
 * not guaranteed to agree in the last ulp, and one ulp can flip a cvRound() and
 * hence a descriptor bit.  So both sides of this project evaluate THIS function:
 * argument widened to double, quadrant reduction against a two-part pi/2, a
 * fixed-order double polynomial written with plain IEEE +,-,* (compile with
 * -ffp-contract=off on both compilers), one final rounding to float.
 * Result = correctly rounded cosf/sinf except on inputs whose double result
 * falls within ~1e-16 relative of a float rounding boundary; tests/ compares it
 * with libm on 10^6 angles.
 *
 * Valid for 0 <= x < 64 (angles here are < 2*pi).
 */
#ifndef CCM_SINCOS_H
#define CCM_SINCOS_H

#if defined(__HIPCC__)
#define CCM_HD __host__ __device__ static inline
#else
#define CCM_HD static inline
#endif

CCM_HD void ccm_sincosf(float xf, float* s_out, float* c_out)
{
    const double x = (double)xf;
    /* k = nearest integer to x * 2/pi, x >= 0 */
    const double two_over_pi = 0.63661977236758134308;
    const int k = (int)(x * two_over_pi + 0.5);
    const double kd = (double)k;
    /* pi/2 = hi + lo, hi has 33 significant bits so kd*hi is exact for k < 2^20 */
    const double pio2_hi = 1.57079632673412561417e+00;
    const double pio2_lo = 6.07710050650619224932e-11;
    double r = x - kd * pio2_hi;
    r = r - kd * pio2_lo;
    const double z = r * r;
    /* Taylor coefficients; |r| <= pi/4 so the truncation error is < 1e-17 */
    double ps = -1.0 / 355687428096000.0;          /* -1/17! */
    ps = ps * z + 1.0 / 1307674368000.0;           /*  1/15! */
    ps = ps * z - 1.0 / 6227020800.0;              /* -1/13! */
    ps = ps * z + 1.0 / 39916800.0;                /*  1/11! */
    ps = ps * z - 1.0 / 362880.0;                  /* -1/9!  */
    ps = ps * z + 1.0 / 5040.0;                    /*  1/7!  */
    ps = ps * z - 1.0 / 120.0;                     /* -1/5!  */
    ps = ps * z + 1.0 / 6.0;                       /*  1/3!  */
    const double sr = r - (r * z) * ps;
    double pc = 1.0 / 6402373705728000.0;          /*  1/18! */
    pc = pc * z - 1.0 / 20922789888000.0;          /* -1/16! */
    pc = pc * z + 1.0 / 87178291200.0;             /*  1/14! */
    pc = pc * z - 1.0 / 479001600.0;               /* -1/12! */
    pc = pc * z + 1.0 / 3628800.0;                 /*  1/10! */
    pc = pc * z - 1.0 / 40320.0;                   /* -1/8!  */
    pc = pc * z + 1.0 / 720.0;                     /*  1/6!  */
    pc = pc * z - 1.0 / 24.0;                      /* -1/4!  */
    pc = pc * z + 0.5;                             /*  1/2!  */
    const double cr = 1.0 - z * pc;
    double s, c;
    switch (k & 3) {
    case 0:  s = sr;  c = cr;  break;
    case 1:  s = cr;  c = -sr; break;
    case 2:  s = -sr; c = -cr; break;
    default: s = -cr; c = sr;  break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

#endif
