/* CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Ray triangulation of matched points: image3.cpp:102-132 (calculateXYZ2) and
 * image3.cpp:148-264 (calculateXYZ, FAST_LINE variant compiled at image3.cpp:147).
 * Strict float32, built with -ffp-contract=off (the 2012 reference had no fma).
 *
 * OpenCV 2.4 arithmetic restated from its published behaviour (library absent here):
 *   Mat::inv() on a 3x3 CV_32F matrix: cofactor formula, determinant and products in double,
 *     each element narrowed to float;
 *   A*b (gemm on CV_32F): dot products accumulated in double, narrowed to float;
 *   -A*b: gemm with alpha = -1 applied in double before narrowing;
 *   (z*a) + t: scaleAdd, float multiply then float add. */
#include <math.h>
#include "oracle.h"

static void inv3(const float* S, float* D)
{
#define Sf(r, c) ((double)S[(r) * 3 + (c)])
    double d = Sf(0,0) * (Sf(1,1) * Sf(2,2) - Sf(1,2) * Sf(2,1))
             - Sf(0,1) * (Sf(1,0) * Sf(2,2) - Sf(1,2) * Sf(2,0))
             + Sf(0,2) * (Sf(1,0) * Sf(2,1) - Sf(1,1) * Sf(2,0));
    if (d == 0.) { for (int i = 0; i < 9; i++) D[i] = 0.f; return; }
    d = 1. / d;
    D[0] = (float)((Sf(1,1) * Sf(2,2) - Sf(1,2) * Sf(2,1)) * d);
    D[1] = (float)((Sf(0,2) * Sf(2,1) - Sf(0,1) * Sf(2,2)) * d);
    D[2] = (float)((Sf(0,1) * Sf(1,2) - Sf(0,2) * Sf(1,1)) * d);
    D[3] = (float)((Sf(1,2) * Sf(2,0) - Sf(1,0) * Sf(2,2)) * d);
    D[4] = (float)((Sf(0,0) * Sf(2,2) - Sf(0,2) * Sf(2,0)) * d);
    D[5] = (float)((Sf(0,2) * Sf(1,0) - Sf(0,0) * Sf(1,2)) * d);
    D[6] = (float)((Sf(1,0) * Sf(2,1) - Sf(1,1) * Sf(2,0)) * d);
    D[7] = (float)((Sf(0,1) * Sf(2,0) - Sf(0,0) * Sf(2,1)) * d);
    D[8] = (float)((Sf(0,0) * Sf(1,1) - Sf(0,1) * Sf(1,0)) * d);
#undef Sf
}

static void matvec(const float* A, const float* b, double alpha, float* out)
{
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)A[i * 3 + k] * (double)b[k];
        out[i] = (float)(s * alpha);
    }
}

/* image3.cpp:102-132 */
static float xyz2(const float* ang1, const float* t1p, const float* ang2, const float* t2p,
                  float z1, float* X, float* Y, float* Z)
{
    float w1[3], w2[3];
    for (int i = 0; i < 3; i++) { float p = z1 * ang1[i]; w1[i] = p + t1p[i]; }
    float z2 = (w1[2] - t2p[2]) / ang2[2];
    for (int i = 0; i < 3; i++) { float p = z2 * ang2[i]; w2[i] = p + t2p[i]; }
    float dx = w1[0] - w2[0], dy = w1[1] - w2[1];
    float a = dx * dx, b = dy * dy;
    float s = a + b;
    float err = sqrtf(s);
    *X = w1[0]; *Y = w1[1]; *Z = w1[2];
    return err;
}

/* image3.cpp:148-264.  The reference takes integer pixel coordinates and converts each to float where
 * it is used (image3.cpp:185-194); the coordinates arrive here already converted, so that the same
 * arithmetic also serves fractional columns (orc_triangulate_f). */
static void xyz(const orc_camera* c1, const orc_camera* c2, float u1, float v1, float u2, float v2,
                float* X, float* Y, float* Z)
{
    float R1p[9], R2p[9], t1p[3], t2p[3];
    inv3(c1->R, R1p); matvec(R1p, c1->T, -1.0, t1p);     /* image3.cpp:169-172 */
    inv3(c2->R, R2p); matvec(R2p, c2->T, -1.0, t2p);
    float fx1 = c1->A[0], fy1 = c1->A[4], cx1 = c1->A[2], cy1 = c1->A[5];
    float fx2 = c2->A[0], fy2 = c2->A[4], cx2 = c2->A[2], cy2 = c2->A[5];
    float a1[3] = { (u1 - cx1) / fx1, -(v1 - cy1) / fy1, -1.0f };  /* image3.cpp:185-194 */
    float a2[3] = { (u2 - cx2) / fx2, -(v2 - cy2) / fy2, -1.0f };
    float ang1[3], ang2[3];
    matvec(R1p, a1, 1.0, ang1);
    matvec(R2p, a2, 1.0, ang2);

    float X1, Y1, Z1;
    float z0 = 0.0f;
    float error0 = xyz2(ang1, t1p, ang2, t2p, z0, X, Y, Z);
    float z1 = 1000.0f;
    float error1 = xyz2(ang1, t1p, ang2, t2p, z1, X, Y, Z);
    {
        float q = (z1 - z0) / (error1 - error0);
        float m = q * (0 - error0);
        z0 = z0 + m;                                   /* image3.cpp:216 */
    }
    float min_error = error0;
    int counter = 0;
    for (z1 = z0; z1 < t1p[2]; z1 += 1.0f) {          /* image3.cpp:220-232 */
        float e = xyz2(ang1, t1p, ang2, t2p, z1, &X1, &Y1, &Z1);
        if (e < min_error) { counter = 0; min_error = e; *X = X1; *Y = Y1; *Z = Z1; }
        else if (counter++ > 100) break;
    }
    counter = 0;
    for (z1 = z0; z1 > 0; z1 -= 1.0f) {               /* image3.cpp:233-246 */
        float e = xyz2(ang1, t1p, ang2, t2p, z1, &X1, &Y1, &Z1);
        if (e < min_error) { counter = 0; min_error = e; *X = X1; *Y = Y1; *Z = Z1; }
        else if (counter++ > 100) break;
    }
}

void orc_triangulate(const orc_camera* c1, const orc_camera* c2, const int32_t* p, int n, float* out)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++)
        xyz(c1, c2, (float)p[4 * i], (float)p[4 * i + 1], (float)p[4 * i + 2], (float)p[4 * i + 3],
            out + 3 * i, out + 3 * i + 1, out + 3 * i + 2);
}

/* The same for float pixel coordinates (a subpixel column in the second image).  No counterpart in the
 * reference, whose calculateXYZ takes ints (image3.cpp:148, called with ints at 812-824): PARITY UNPINNED
 * beyond integer-valued coordinates, where it equals orc_triangulate bit for bit. */
void orc_triangulate_f(const orc_camera* c1, const orc_camera* c2, const float* p, int n, float* out)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++)
        xyz(c1, c2, p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3],
            out + 3 * i, out + 3 * i + 1, out + 3 * i + 2);
}
