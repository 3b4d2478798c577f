/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (plain C restatement, scalar loops).
 * Same role and rules as oracle/fep_oracle.py: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline may load it; the product never does.
 *
 * Parity status: PINNED — checked in tests/test_oracle_c.py against the NumPy oracle and the
 * golden vectors recorded from the reference (tests/golden/retmap.npz, hotpath_dp.npz).
 *
 * Reference lines (DP = Plasticity2D_DP/pythonFEM.py):
 *   oracle_return_map       DP:646-757 (+ TSX:1052 e0)      per-point loop
 *   oracle_element_matrices DP:1047-1050, DP:1058           K_e = sum_q w B^T DS B,  f_e = sum_q w B^T s
 * Layouts: per-point arrays are (rows, n) C-order; k = e*n_q + q; ds row m = 3*i + j.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* returns counts in counts[0] (smooth) / counts[1] (apex) */
void oracle_return_map(int64_t n, const double* e, const double* e0, double* ep, const double* shear,
                       const double* bulk, const double* eta, const double* c, int accept,
                       double* s, double* ds, uint8_t* indp, int64_t* counts) {
    const double I3 = 1.0 / 3.0, DD = 1.0 - 1.0 / 3.0, RS2 = sqrt(2.0);
    counts[0] = counts[1] = 0;
    for (int64_t k = 0; k < n; ++k) {
        double E[4] = {e[k], e[n + k], e[2 * n + k], 0.0}, p[4] = {0, 0, 0, 0}, Et[4], dv[4], S[4];
        if (e0) for (int i = 0; i < 4; ++i) E[i] += e0[i];                       /* TSX:1052 */
        if (ep) for (int i = 0; i < 4; ++i) p[i] = ep[i * n + k];
        for (int i = 0; i < 4; ++i) Et[i] = E[i] - p[i];                         /* DP:666-668 */
        const double G = shear[k], K = bulk[k], et = eta[k], cc = c[k];
        const double tr = Et[0] + Et[1] + Et[3];
        dv[0] = DD * Et[0] - I3 * Et[1] - I3 * Et[3];                            /* DP:673 */
        dv[1] = -I3 * Et[0] + DD * Et[1] - I3 * Et[3];
        dv[2] = 0.5 * Et[2];
        dv[3] = -I3 * Et[0] - I3 * Et[1] + DD * Et[3];
        const double Ktr = K * tr;
        S[0] = 2 * G * dv[0] + Ktr; S[1] = 2 * G * dv[1] + Ktr; S[2] = 2 * G * dv[2]; S[3] = 2 * G * dv[3] + Ktr;
        double n2 = Et[0] * dv[0] + Et[1] * dv[1] + Et[2] * dv[2] + Et[3] * dv[3];
        const double nE = sqrt(n2 > 0 ? n2 : 0.0);                               /* DP:676 */
        const double rho = 2 * (G * nE), da = K * (et * et), dS = G + da;
        const double c1 = rho / RS2 + et * Ktr - cc;                             /* DP:689 */
        const double c2 = et * Ktr - da * rho / (G * RS2) - cc;                  /* DP:690 */
        const double Dev[3][3] = {{DD, -I3, 0}, {-I3, DD, 0}, {0, 0, 0.5}};
        const double Vol[3][3] = {{1, 1, 0}, {1, 1, 0}, {0, 0, 0}};
        double D[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) D[i][j] = 2 * Dev[i][j] * G + Vol[i][j] * K;   /* DP:703 */
        int br = 0;
        if (c1 > 0 && c2 <= 0) {                                                 /* smooth, DP:696 */
            br = 1;
            const double lam = c1 / dS;
            double N[4], M[4];
            const double iota[4] = {1, 1, 0, 1};
            for (int i = 0; i < 4; ++i) { N[i] = dv[i] / nE; M[i] = RS2 * G * N[i] + K * et * iota[i]; S[i] -= lam * M[i]; }
            const double cf = 2 * RS2 * (G * G) * lam / rho;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
                D[i][j] = D[i][j] - cf * (Dev[i][j] - N[i] * N[j]) - M[i] * M[j] / dS;               /* DP:727 */
            if (accept && ep) {
                const double mult[4] = {1, 1, 2, 1};
                for (int i = 0; i < 4; ++i) ep[i * n + k] = p[i] + mult[i] * lam * (N[i] / RS2 + iota[i] * et / 3);  /* DP:752 */
            }
        } else if (c1 > 0) {                                                     /* apex, DP:699 */
            br = 2;
            const double iota[4] = {1, 1, 0, 1};
            for (int i = 0; i < 4; ++i) S[i] = iota[i] * (cc / et);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) D[i][j] = 0.0;
            if (accept && ep) for (int i = 0; i < 4; ++i) ep[i * n + k] = Et[i] - iota[i] * (cc / (3 * K * et));   /* DP:755 */
        }
        for (int i = 0; i < 4; ++i) s[i * n + k] = S[i];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) ds[(3 * i + j) * n + k] = D[i][j];
        indp[k] = br != 0;
        if (br) counts[br - 1]++;
    }
}

/* Ke: (n_e, 2*n_p, 2*n_p) C-order; fe: (n_e, 2*n_p).  dphi1/2: (n_p, n_int); local DOF 2*a + comp. */
void oracle_element_matrices(int n_p, int n_q, int64_t n_e, const double* dphi1, const double* dphi2,
                             const double* weight, const double* ds, const double* s, double* Ke, double* fe) {
    const int nd = 2 * n_p;
    const int64_t n_int = n_e * n_q;
    for (int64_t e = 0; e < n_e; ++e) {
        double* K = Ke + e * nd * nd;
        double* f = fe + e * nd;
        for (int i = 0; i < nd * nd; ++i) K[i] = 0.0;
        for (int i = 0; i < nd; ++i) f[i] = 0.0;
        for (int q = 0; q < n_q; ++q) {
            const int64_t k = e * n_q + q;
            const double w = weight[k];
            double D[3][3];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) D[i][j] = w * ds[(3 * i + j) * n_int + k];   /* DP:1047 */
            for (int a = 0; a < n_p; ++a) {
                /* B_a = [[d1,0],[0,d2],[d2,d1]]  (DP:549-554) */
                const double Ba[3][2] = {{dphi1[a * n_int + k], 0}, {0, dphi2[a * n_int + k]},
                                         {dphi2[a * n_int + k], dphi1[a * n_int + k]}};
                for (int ci = 0; ci < 2; ++ci) {
                    double acc = 0;
                    for (int r = 0; r < 3; ++r) acc += Ba[r][ci] * (w * s[r * n_int + k]);               /* DP:1058 */
                    f[2 * a + ci] += acc;
                }
                for (int b = 0; b < n_p; ++b) {
                    const double Bb[3][2] = {{dphi1[b * n_int + k], 0}, {0, dphi2[b * n_int + k]},
                                             {dphi2[b * n_int + k], dphi1[b * n_int + k]}};
                    for (int ci = 0; ci < 2; ++ci) for (int cj = 0; cj < 2; ++cj) {
                        double acc = 0;
                        for (int r = 0; r < 3; ++r) for (int t = 0; t < 3; ++t) acc += Ba[r][ci] * D[r][t] * Bb[t][cj];
                        K[(2 * a + ci) * nd + 2 * b + cj] += acc;                                        /* DP:1050 */
                    }
                }
            }
        }
    }
}
