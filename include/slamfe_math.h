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

/* Row 3 of Vt of cv::SVD::compute(A, W, U, Vt, MODIFY_A | FULL_UV) for a 4x4
 * float matrix, i.e. the right singular vector of the smallest singular value
 * that Init::triangulate reads (Init.cc:14-15, 28-29).  OpenCV's small-matrix
 * path is a one-sided Jacobi method on the columns of A (kept as rows of At):
 * column norms and rotation parameters in double, matrix entries in float, at
 * most 30 sweeps, threshold 2 * FLT_EPSILON, singular values sorted in
 * descending order.  hypot(p, beta) is evaluated as sqrt(p*p + beta*beta) so
 * that host and device round alike.  A[r*4 + c] row-major. */
SLAMFE_HD void slamfe_svd4_last_row(const float* A, float* vt3)
{
    float At[4][4], Vt[4][4];
    double W[4];
    const double eps = 1.1920928955078125e-07 * 2; /* FLT_EPSILON * 2 */
    for (int i = 0; i < 4; ++i) {
        double sd = 0;
        for (int k = 0; k < 4; ++k) {
            At[i][k] = A[k * 4 + i];
            Vt[i][k] = i == k ? 1.f : 0.f;
            double t = (double)At[i][k];
            sd += t * t;
        }
        W[i] = sd;
    }
    for (int iter = 0; iter < 30; ++iter) {
        int changed = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = i + 1; j < 4; ++j) {
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < 4; ++k) p += (double)At[i][k] * (double)At[j][k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = sqrt(p * p + beta * beta);
                double c, s;
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (int k = 0; k < 4; ++k) {
                    float t0 = (float)(c * (double)At[i][k] + s * (double)At[j][k]);
                    float t1 = (float)(-s * (double)At[i][k] + c * (double)At[j][k]);
                    At[i][k] = t0;
                    At[j][k] = t1;
                    a += (double)t0 * (double)t0;
                    b += (double)t1 * (double)t1;
                }
                W[i] = a;
                W[j] = b;
                changed = 1;
                for (int k = 0; k < 4; ++k) {
                    float t0 = (float)(c * (double)Vt[i][k] + s * (double)Vt[j][k]);
                    float t1 = (float)(-s * (double)Vt[i][k] + c * (double)Vt[j][k]);
                    Vt[i][k] = t0;
                    Vt[j][k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < 4; ++i) {
        double sd = 0;
        for (int k = 0; k < 4; ++k) {
            double t = (double)At[i][k];
            sd += t * t;
        }
        W[i] = sqrt(sd);
    }
    /* descending selection sort of the singular values; the row that ends last */
    int idx[4] = {0, 1, 2, 3};
    for (int i = 0; i < 3; ++i) {
        int j = i;
        for (int k = i + 1; k < 4; ++k)
            if (W[idx[j]] < W[idx[k]]) j = k;
        int t = idx[i];
        idx[i] = idx[j];
        idx[j] = t;
    }
    for (int k = 0; k < 4; ++k) vt3[k] = Vt[idx[3]][k];
}

/* Init::triangulate (Init.cc:5-31): the 4x4 system of one point pair and its
 * solution.  P1, P2: 3x4 projection matrices, row-major float.  Rows of A are
 * x * P.row(2) - P.row(0) and y * P.row(2) - P.row(1) in float; the result is
 * Vt.row(3)[0..2] scaled by the float value of 1.0 / Vt.row(3)[3] (Init.cc:16). */
SLAMFE_HD void slamfe_triangulate_point(const float* P1, const float* P2, float x1, float y1, float x2,
                                        float y2, float* xyz)
{
    float A[16], v[4];
    for (int k = 0; k < 4; ++k) {
        A[k] = x1 * P1[8 + k] - P1[k];
        A[4 + k] = y1 * P1[8 + k] - P1[4 + k];
        A[8 + k] = x2 * P2[8 + k] - P2[k];
        A[12 + k] = y2 * P2[8 + k] - P2[4 + k];
    }
    slamfe_svd4_last_row(A, v);
    const float sc = (float)(1.0 / (double)v[3]);
    xyz[0] = v[0] * sc;
    xyz[1] = v[1] * sc;
    xyz[2] = v[2] * sc;
}

#endif /* SLAMFE_MATH_H */
