/* slamfe_math.h -- the floating-point routines whose exact operation order is
 * part of the front-end's numerical contract.
 *
 * The descriptor bits of a keypoint depend on the last bit of cos/sin of its
 * orientation, and host libm and the GPU math library do not agree to the last
 * ulp.  So the rotation is defined by the routines below: IEEE-754 +,-,*,/ and
 * round-to-nearest-even only, one rounding per operation (build with
 * -ffp-contract=off so that no multiply-add is fused).  The same source is
 * compiled for the GPU kernels, the host library and the CPU checker; the
 * results are identical bit for bit on every IEEE-754 machine.
 *
 * What the reference computes at these points (inside OpenCV, reached from
 * Init.cc:78): cvRound(), cv::fastAtan2() and (float)cos / (float)sin of
 * angle * (float)(CV_PI/180.f).
 */
#ifndef SLAMFE_MATH_H
#define SLAMFE_MATH_H

#include <math.h>

#if defined(__HIPCC__)
#define SLAMFE_HD __host__ __device__ static inline
#else
#define SLAMFE_HD static inline
#endif

/* cvRound: round half to even, float and double. */
SLAMFE_HD int slamfe_round_f(float x) { return (int)rintf(x); }
SLAMFE_HD int slamfe_round_d(double x) { return (int)rint(x); }

/* cv::fastAtan2(y, x): degrees in [0, 360), degree-7 odd polynomial in the
 * octant-reduced ratio.  All arithmetic is float, in this order. */
SLAMFE_HD float slamfe_fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* sin and cos of an angle given in degrees, as floats.
 *
 * theta = angle_deg * (float)(pi/180) is formed in float (as the reference
 * path does); everything after that is double: Cody-Waite reduction by pi/2,
 * Taylor polynomials on |r| <= pi/4 (truncation error < 3e-14), one final
 * rounding to float.  The result equals the correctly rounded float value
 * except for arguments within ~1e-6 ulp of a rounding boundary.  Valid for
 * |angle_deg| < 1e5. */
SLAMFE_HD void slamfe_sincos_deg(float angle_deg, float* s_out, float* c_out)
{
    const float deg2rad = (float)(3.14159265358979323846 / 180.0);
    float theta = angle_deg * deg2rad;
    double x = (double)theta;
    double kd = rint(x * 0.63661977236758134308);
    double r = (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
    double r2 = r * r;
    double sp = 1.0 / 6227020800.0; /* 1/13! */
    sp = -1.0 / 39916800.0 + r2 * sp;
    sp = 1.0 / 362880.0 + r2 * sp;
    sp = -1.0 / 5040.0 + r2 * sp;
    sp = 1.0 / 120.0 + r2 * sp;
    sp = -1.0 / 6.0 + r2 * sp;
    double sn = r + (r * r2) * sp;
    double cp = -1.0 / 87178291200.0;
    cp = 1.0 / 479001600.0 + r2 * cp;
    cp = -1.0 / 3628800.0 + r2 * cp;
    cp = 1.0 / 40320.0 + r2 * cp;
    cp = -1.0 / 720.0 + r2 * cp;
    cp = 1.0 / 24.0 + r2 * cp;
    cp = -0.5 + r2 * cp;
    double cs = 1.0 + r2 * cp;
    int k = ((int)kd) & 3;
    double s, c;
    if (k == 0)      { s = sn;  c = cs;  }
    else if (k == 1) { s = cs;  c = -sn; }
    else if (k == 2) { s = -sn; c = -cs; }
    else             { s = -cs; c = sn;  }
    *s_out = (float)s;
    *c_out = (float)c;
}

#endif /* SLAMFE_MATH_H */
