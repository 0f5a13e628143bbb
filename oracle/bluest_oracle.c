/*
 * oracle/bluest_oracle.c -- CPU restatement of the BLUEST sample-allocation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in bluest_amd/ may import, link or call this file; it exists so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the HIP path against an
 * independent plain-C statement of what the reference computes.  Parity status: PINNED -- tests/test_oracle.py
 * checks every function below against tests/golden/*.npz, which were produced by running the real reference
 * (compiled /root/reference/bluest/cmisc.cpp + imported bluest/misc.py, sap.py, mosap.py, spg.py) in the build
 * container with oracle/gen_golden.py.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference/).
 * Arithmetic: IEEE float64, indices int64, exactly like the reference (cmisc.cpp uses `long int`).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * L0 -- bluest/cmisc.cpp
 * ---------------------------------------------------------------------------------------------- */

/* bluest/cmisc.cpp:10-23  psi[(N*g_j+g_l), i] += invcov_i[j,l]; psi is C-order (N*N, Lk), caller zero-fills */
void orc_assemble_psi(double *psi, int N, int k, int64_t Lk, const int64_t *g, const double *ic)
{
    const int ksq = k * k;
    for (int64_t i = 0; i < Lk; i++)
        for (int j = 0; j < k; j++)
            for (int l = 0; l < k; l++)
                psi[Lk * (N * g[k * i + j] + g[k * i + l]) + i] += ic[ksq * i + k * j + l];
}

/* bluest/cmisc.cpp:25-40 (T=double, overload at :104)  PHI[N*g_j+g_l] += m_i*invcov_i[j,l] */
void orc_objectiveK_f64(double *PHI, int N, int k, int64_t Lk, const double *mk, const int64_t *g,
                        const double *ic)
{
    const int ksq = k * k;
    for (int64_t i = 0; i < Lk; i++)
        for (int j = 0; j < k; j++)
            for (int l = 0; l < k; l++)
                PHI[N * g[k * i + j] + g[k * i + l]] += mk[i] * ic[ksq * i + k * j + l];
}

/* bluest/cmisc.cpp:25-40 (T=long int, overload at :105) */
void orc_objectiveK_i64(double *PHI, int N, int k, int64_t Lk, const int64_t *mk, const int64_t *g,
                        const double *ic)
{
    const int ksq = k * k;
    for (int64_t i = 0; i < Lk; i++)
        for (int j = 0; j < k; j++)
            for (int l = 0; l < k; l++)
                PHI[N * g[k * i + j] + g[k * i + l]] += mk[i] * ic[ksq * i + k * j + l];
}

/* bluest/cmisc.cpp:42-56  NOTE the reference assigns with `=` (line 51), so only the l=k-1 term survives:
 * X[g_j, i] = invcov_i[j,k-1]*v[g_{k-1}].  Restated as written, quirk included. X is C-order (N, Lk). */
void orc_cleanupK(double *X, int k, int64_t Lk, const int64_t *g, const double *ic, const double *v)
{
    const int ksq = k * k;
    for (int64_t i = 0; i < Lk; i++)
        for (int j = 0; j < k; j++)
            for (int l = 0; l < k; l++)
                X[Lk * g[k * i + j] + i] = ic[ksq * i + k * j + l] * v[g[k * i + l]];
}

/* bluest/cmisc.cpp:58-72  grad_i += v[g_j]*invcov_i[j,l]*v[g_l] */
void orc_gradK(double *grad, int k, int64_t Lk, const int64_t *g, const double *ic, const double *v)
{
    const int ksq = k * k;
    for (int64_t i = 0; i < Lk; i++)
        for (int j = 0; j < k; j++)
            for (int l = 0; l < k; l++)
                grad[i] += v[g[k * i + j]] * ic[ksq * i + k * j + l] * v[g[k * i + l]];
}

/* bluest/cmisc.cpp:74-97  hess[ik,iq] += v[gk_lk]*Ck[lk,jk]*invPHI[gk_jk,gq_jq]*Cq[jq,lq]*v[gq_lq],
 * v = row 0 of invPHI (the reference indexes invPHI_x[g] i.e. the first N entries of the flat array). */
void orc_hessKQ(double *hess, int N, int k, int q, int64_t Lk, int64_t Lq, const int64_t *gk, const int64_t *gq,
                const double *ick, const double *icq, const double *invPHI)
{
    const int ksq = k * k, qsq = q * q;
    for (int64_t ik = 0; ik < Lk; ik++)
        for (int64_t iq = 0; iq < Lq; iq++)
            for (int lk = 0; lk < k; lk++)
                for (int jk = 0; jk < k; jk++)
                    for (int jq = 0; jq < q; jq++)
                        for (int lq = 0; lq < q; lq++)
                            hess[ik * Lq + iq] += invPHI[gk[k * ik + lk]] * ick[ksq * ik + k * lk + jk] *
                                                  invPHI[N * gk[k * ik + jk] + gq[q * iq + jq]] *
                                                  icq[qsq * iq + q * jq + lq] * invPHI[gq[q * iq + lq]];
}

/* ------------------------------------------------------------------------------------------------
 * Small dense linear algebra standing in for the numpy.linalg calls on the path
 * (np.linalg.pinv at sap.py:74, misc.py:487,490; np.linalg.solve at misc.py:472).
 * ---------------------------------------------------------------------------------------------- */

/* Cyclic Jacobi eigen-decomposition of a symmetric n x n matrix: A = V diag(w) V^T.  A is overwritten. */
static void sym_jacobi(double *A, int n, double *V, double *w)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; i++) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-60 * (diag + off) || off == 0.0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int r = 0; r < n; r++) { /* columns p,q */
                    const double arp = A[r * n + p], arq = A[r * n + q];
                    A[r * n + p] = c * arp - s * arq;
                    A[r * n + q] = s * arp + c * arq;
                }
                for (int r = 0; r < n; r++) { /* rows p,q */
                    const double apr = A[p * n + r], aqr = A[q * n + r];
                    A[p * n + r] = c * apr - s * aqr;
                    A[q * n + r] = s * apr + c * aqr;
                }
                for (int r = 0; r < n; r++) {
                    const double vrp = V[r * n + p], vrq = V[r * n + q];
                    V[r * n + p] = c * vrp - s * vrq;
                    V[r * n + q] = s * vrp + c * vrq;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
}

/* numpy.linalg.pinv(A) for SYMMETRIC A: singular values are |lambda|, cutoff rcond*max|lambda| (numpy default
 * rcond = 1e-15).  out = sum_{|l_i|>cut} v_i v_i^T / l_i.  Returns the numerical rank. */
int orc_sym_pinv(const double *A, int n, double rcond, double *out)
{
    double *W = (double *)malloc(sizeof(double) * (2 * (size_t)n * n + n));
    double *S = W, *V = W + (size_t)n * n, *w = V + (size_t)n * n;
    for (int i = 0; i < n; i++) /* symmetrise the input the way an SVD would not care about */
        for (int j = 0; j < n; j++) S[i * n + j] = 0.5 * (A[i * n + j] + A[j * n + i]);
    sym_jacobi(S, n, V, w);
    double wmax = 0.0;
    for (int i = 0; i < n; i++) if (fabs(w[i]) > wmax) wmax = fabs(w[i]);
    const double cut = rcond * wmax;
    int rank = 0;
    memset(out, 0, sizeof(double) * (size_t)n * n);
    for (int e = 0; e < n; e++) {
        if (!(fabs(w[e]) > cut)) continue;
        rank++;
        const double inv = 1.0 / w[e];
        for (int i = 0; i < n; i++) {
            const double vi = V[i * n + e] * inv;
            for (int j = 0; j < n; j++) out[i * n + j] += vi * V[j * n + e];
        }
    }
    free(W);
    return rank;
}

/* numpy.linalg.solve(A,b): LU with partial pivoting (LAPACK dgesv).  Returns 0, or 1 if exactly singular. */
int orc_solve(const double *A, const double *b, int n, double *x)
{
    double *M = (double *)malloc(sizeof(double) * ((size_t)n * n + n));
    double *r = M + (size_t)n * n;
    memcpy(M, A, sizeof(double) * (size_t)n * n);
    memcpy(r, b, sizeof(double) * n);
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int i = c + 1; i < n; i++) if (fabs(M[i * n + c]) > fabs(M[piv * n + c])) piv = i;
        if (M[piv * n + c] == 0.0) { free(M); return 1; }
        if (piv != c) {
            for (int j = 0; j < n; j++) { double t = M[c * n + j]; M[c * n + j] = M[piv * n + j]; M[piv * n + j] = t; }
            double t = r[c]; r[c] = r[piv]; r[piv] = t;
        }
        for (int i = c + 1; i < n; i++) {
            const double f = M[i * n + c] / M[c * n + c];
            if (f == 0.0) continue;
            for (int j = c; j < n; j++) M[i * n + j] -= f * M[c * n + j];
            r[i] -= f * r[c];
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = r[i];
        for (int j = i + 1; j < n; j++) s -= M[i * n + j] * x[j];
        x[i] = s / M[i * n + i];
    }
    free(M);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * L2 setup -- bluest/sap.py:66-79  invcovs[k-1] = vstack(pinv(C[g,g]) for g in groups[k-1]).flatten()
 * ---------------------------------------------------------------------------------------------- */
void orc_group_pinv(const double *C, int N, int k, int64_t Lk, const int64_t *g, double *invcov)
{
    double *sub = (double *)malloc(sizeof(double) * (size_t)k * k);
    for (int64_t i = 0; i < Lk; i++) {
        for (int j = 0; j < k; j++)
            for (int l = 0; l < k; l++) sub[j * k + l] = C[N * g[k * i + j] + g[k * i + l]];
        orc_sym_pinv(sub, k, 1e-15, invcov + (size_t)k * k * i);
    }
    free(sub);
}

/* ------------------------------------------------------------------------------------------------
 * L1 -- bluest/misc.py on a flat SAP description:
 *   N models, K = max group size, sizes[0..K] with sizes[0]=0 (sap.py:68), groups = concat_k (L_k*k) int64,
 *   invcovs = concat_k (L_k*k*k) f64, m = (L) f64 with L = sum sizes.
 * ---------------------------------------------------------------------------------------------- */

/* bluest/misc.py:453-457  models touched by a group with |m_i| > 1e-6 */
void orc_nnz_models(const double *m, int N, int K, const int64_t *sizes, const int64_t *groups, uint8_t *mask)
{
    memset(mask, 0, N);
    int64_t mo = 0, go = 0;
    for (int k = 1; k <= K; k++) {
        const int64_t Lk = sizes[k];
        for (int64_t i = 0; i < Lk; i++)
            if (fabs(m[mo + i]) > 1.0e-6)
                for (int j = 0; j < k; j++) mask[groups[go + k * i + j]] = 1;
        mo += Lk; go += Lk * k;
    }
}

/* bluest/misc.py:459-461  PHI = delta*I + (psi@m).reshape(N,N) with DENSE psi (N*N, L) -- baseline "B1" */
void orc_phi_dense(const double *psi, const double *m, int N, int64_t L, double delta, double *PHI)
{
    for (int r = 0; r < N * N; r++) {
        const double *row = psi + (size_t)r * L;
        double s = 0.0;
        for (int64_t c = 0; c < L; c++) s += row[c] * m[c];
        PHI[r] = s;
    }
    for (int i = 0; i < N; i++) PHI[i * N + i] += delta;
}

/* same Phi through the sparse group loop (cmisc.cpp:25-40 per k) -- baseline "B2" */
void orc_phi_sparse(const double *m, int N, int K, const int64_t *sizes, const int64_t *groups,
                    const double *invcovs, double delta, double *PHI)
{
    memset(PHI, 0, sizeof(double) * (size_t)N * N);
    int64_t mo = 0, go = 0, io = 0;
    for (int k = 1; k <= K; k++) {
        const int64_t Lk = sizes[k];
        orc_objectiveK_f64(PHI, N, k, Lk, m + mo, groups + go, invcovs + io);
        mo += Lk; go += Lk * k; io += Lk * k * k;
    }
    for (int i = 0; i < N; i++) PHI[i * N + i] += delta;
}

static double absmax(const double *m, int64_t L)
{
    double a = 0.0;
    for (int64_t i = 0; i < L; i++) if (fabs(m[i]) > a) a = fabs(m[i]);
    return a;
}

static int restrict_phi(const double *PHI, int N, const uint8_t *mask, double *R, int *idx)
{
    int nr = 0;
    for (int i = 0; i < N; i++) if (mask[i]) idx[nr++] = i;
    for (int a = 0; a < nr; a++)
        for (int b = 0; b < nr; b++) R[a * nr + b] = PHI[idx[a] * N + idx[b]];
    return nr;
}

/* status codes shared by orc_variance / orc_variance_GH */
enum { ORC_OK = 0, ORC_INF = 1 /* max|m|<0.05 -> inf (misc.py:464,484) */,
       ORC_ASSERT_MODEL0 = 2 /* misc.py:470 */, ORC_SINGULAR = 3 /* misc.py:473-474 */ };

/* bluest/misc.py:463-477 variance_full.  psi may be NULL -> Phi via the sparse loop (same numbers to rounding). */
int orc_variance(const double *m, const double *psi, int N, int K, const int64_t *sizes, const int64_t *groups,
                 const double *invcovs, double delta, double *var)
{
    int64_t L = 0;
    for (int k = 1; k <= K; k++) L += sizes[k];
    if (absmax(m, L) < 0.05) { *var = INFINITY; return ORC_INF; }
    double *W = (double *)malloc(sizeof(double) * (2 * (size_t)N * N + 2 * N));
    double *PHI = W, *R = W + (size_t)N * N, *b = R + (size_t)N * N, *x = b + N;
    uint8_t mask[256]; int idx[256];
    if (psi) orc_phi_dense(psi, m, N, L, delta, PHI);
    else     orc_phi_sparse(m, N, K, sizes, groups, invcovs, delta, PHI);
    orc_nnz_models(m, N, K, sizes, groups, mask);
    if (!mask[0]) { free(W); *var = NAN; return ORC_ASSERT_MODEL0; }
    const int nr = restrict_phi(PHI, N, mask, R, idx);
    for (int i = 0; i < nr; i++) b[i] = (i == 0) ? 1.0 : 0.0;
    const int rc = orc_solve(R, b, nr, x);
    *var = rc ? NAN : x[0];
    free(W);
    return rc ? ORC_SINGULAR : ORC_OK;
}

/* bluest/misc.py:479-495 variance_GH_full(nohess=True):
 *   invPHI = pinv(PHI) (full N x N, :487); var = pinv(PHI[idx])[0,0] (:490); grad = -concat_k gradK(.., invPHI) (:493)
 * Also returns v = invPHI[0,:] and the full PHI for inspection (either may be NULL). */
int orc_variance_GH(const double *m, const double *psi, int N, int K, const int64_t *sizes,
                    const int64_t *groups, const double *invcovs, double delta, double *var, double *grad,
                    double *v_out, double *PHI_out)
{
    int64_t L = 0;
    for (int k = 1; k <= K; k++) L += sizes[k];
    if (absmax(m, L) < 0.05) {
        *var = INFINITY;
        for (int64_t i = 0; i < L; i++) grad[i] = INFINITY;
        return ORC_INF;
    }
    double *W = (double *)malloc(sizeof(double) * (4 * (size_t)N * N));
    double *PHI = W, *R = W + (size_t)N * N, *P = R + (size_t)N * N, *Pr = P + (size_t)N * N;
    uint8_t mask[256]; int idx[256];
    if (psi) orc_phi_dense(psi, m, N, L, delta, PHI);
    else     orc_phi_sparse(m, N, K, sizes, groups, invcovs, delta, PHI);
    if (PHI_out) memcpy(PHI_out, PHI, sizeof(double) * (size_t)N * N);
    orc_sym_pinv(PHI, N, 1e-15, P);
    orc_nnz_models(m, N, K, sizes, groups, mask);
    const int nr = restrict_phi(PHI, N, mask, R, idx);
    orc_sym_pinv(R, nr, 1e-15, Pr);
    *var = Pr[0]; /* [0,0] of the restricted pseudo-inverse; NOTE: row 0 of PHI[idx] is the smallest sampled model */
    if (v_out) memcpy(v_out, P, sizeof(double) * N);
    int64_t mo = 0, go = 0, io = 0;
    for (int64_t i = 0; i < L; i++) grad[i] = 0.0;
    for (int k = 1; k <= K; k++) {
        const int64_t Lk = sizes[k];
        orc_gradK(grad + mo, k, Lk, groups + go, invcovs + io, P /* row 0 */);
        mo += Lk; go += Lk * k; io += Lk * k * k;
    }
    for (int64_t i = 0; i < L; i++) grad[i] = -grad[i];
    free(W);
    return ORC_OK;
}
