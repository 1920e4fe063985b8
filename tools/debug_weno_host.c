#include <math.h>
#include "../include/ocn_weno_coeffs.h"
double oro_weno5_biased(const double S[6], int left);
double oro_newton_div_f32(double a, double b);
static double beta3(double p0, double p1, double p2, double C1, double C2, double C3, double C4, double C5, double C6) {
    double in1 = fma(C3, p2, fma(C2, p1, C1 * p0));
    double in2 = fma(C5, p2, C4 * p1);
    return fma(p2 * p2, C6, fma(p1, in2, p0 * in1));
}
void host_stages(const double *s, int left, double *o) {
    double a0 = left ? s[2] : s[3], a1 = left ? s[3] : s[2], a2 = left ? s[4] : s[1];
    double b0 = left ? s[1] : s[4];
    double c0 = left ? s[0] : s[5];
    double be0 = beta3(a0, a1, a2, 10, -31, 11, 25, -19, 4);
    double be1 = beta3(b0, a0, a1, 4, -13, 5, 13, -13, 4);
    double be2 = beta3(c0, b0, a0, 4, -19, 11, 25, -31, 10);
    double tau = fabs(be0 - be2);
    double r0 = oro_newton_div_f32(tau, be0 + OCN_WENO_EPS);
    float bl = (float)(be0 + OCN_WENO_EPS);
    float inv = 1.0f / bl;
    double al0 = OCN_W3C0 * (1.0 + r0 * r0);
    double sinv = 1.0 / (al0 + 0.37);
    o[0] = be0; o[1] = be1; o[2] = be2; o[3] = r0; o[4] = (double)inv; o[5] = al0; o[6] = sinv;
    o[7] = oro_weno5_biased(s, left);
}
