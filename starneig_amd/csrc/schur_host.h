// Host-side sequential kernels of the Schur path (see schur_host.hip).
#pragma once

namespace sn { namespace host {

struct AedResult {
    int deflated;   // converged eigenvalues at the bottom of the window
    int shifts;     // usable shifts returned in sr/si
    int failed;     // the window's Schur reduction did not fully converge
};

void lanv2(double &a, double &b, double &c, double &d,
    double &rt1r, double &rt1i, double &rt2r, double &rt2i, double &cs, double &sn);
void helper_session(bool on, int count = 5);   // `count` helper threads for the window kernels (schur_host_team.h)
int small_schur(int n, double *T, int ldt, double *Z, int ldz, double *wr, double *wi);
int move_block_up(int n, double *T, int ldt, double *Z, int ldz, int from, int to);
int deflate_window(int w, double *T, int ldt, double *Z, int ldz, double *spike, double sub,
    double thres, int carried, int *undeflated);
int reorder_window(int w, double *T, int ldt, double *Z, int ldz, int *sel, int *failed);
void extract_eigenvalues(int n, const double *T, int ldt, double *wr, double *wi);
int extract_shifts(int n, const double *T, int ldt, double *wr, double *wi);
int order_shifts(int n, double *wr, double *wi);
AedResult aed_window(int nw, double *T, int ldt, double *Z, int ldz, double sub,
    double thres, double *spike, double *sr, double *si);

// ---- generalized problem (schur_host_gep.hip) ---------------------------------------------
void gep_extract_eigenvalues(int n, const double *S, int lds, const double *T, int ldt,
    double *ar, double *ai, double *be);
void gep_push_inf_window(int w, double *A, int lda, double *B, int ldb, double *Q, int ldq,
    double *Z, int ldz, int from, int to, int deflate);
void gep_push_inf_down_window(int w, double *A, int lda, double *B, int ldb, double *Q, int ldq,
    double *Z, int ldz, int from, int deflate);
// thres_b: magnitude below which an entry of B is negligible (conf->right_threshold; LAPACK dhgeqz's
// BTOL): > 0 that value, otherwise max(safmin, u * ||B||_F) of the pencil at hand
int gep_small_schur(int n, double *A, int lda, double *B, int ldb, double *Q, int ldq,
    double *Z, int ldz, int nq, double *ar, double *ai, double *be, double thres_b = -1.0);
void gep_ht_reduce(int n, int ilo, int ihi, double *A, int lda, double *B, int ldb,
    double *Q, int ldq, double *Z, int ldz, int nq);
int gep_move_block_up(int nw, double *A, int lda, double *B, int ldb, double *Q, int ldq,
    double *Z, int ldz, int nq, int from, int to);
int gep_window_shifts(int hi, double const *A, int lda, double const *B, int ldb, double *sr, double *si);
int gep_deflate_window(int w, double *A, int lda, double *B, int ldb, double *Q, int ldq, double *Z, int ldz,
    double *spike, double sub, double thres, int carried, int *undeflated);
int gep_reorder_window(int w, double *A, int lda, double *B, int ldb, double *Q, int ldq, double *Z, int ldz,
    int *sel, int *failed);
AedResult gep_aed_window(int nw, double *A, int lda, double *B, int ldb, double *Q, int ldq,
    double *Z, int ldz, double sub, double thres, double *spike, double *sr, double *si,
    double thres_b = -1.0);

}} // namespace sn::host
