// Developer switches of the MI355X path, read from the environment ONCE (first use) and only when
// STARNEIG_AMD_TUNING=1 is set; otherwise every field keeps its default.  The hot paths read plain
// struct members, never the environment.  (scratch/README.md lists what each switch is for.)
#pragma once

namespace sn {

struct Tuning {
    // Hessenberg
    int hess_wgs = 1024;            // SN_HESS_WGS: streaming workgroups of one gemv launch
    int hess_max_split = 32;        // SN_HESS_MAXSPLIT
    int hess_max_panels = 1 << 30;  // SN_HESS_MAX_PANELS: stop after k panels (PMC runs); the result is then partial
    long hess_cache_mb = 256;       // SN_HESS_CACHE_MB: trailing matrices below this size are read with temporal loads
    int hess_side_cus = 0;          // SN_HESS_SIDE_CUS: CU mask of the side stream (0 = none)
    bool hess_noside = false;       // SN_HESS_NOSIDE: delayed updates on the critical stream
    int hess_fold = 0;              // SN_HESS_FOLD: fold of the sharded gemv's partials (0 in the launch, behind a release / acquire ticket; 2 a launch of its own)
    bool team_pooled_stream = false;// SN_TEAM_POOLED_STREAM: the ranks' streams plain (pooled hardware queues) as in round 4 (reproducer)
    bool team_verify = false;       // SN_TEAM_VERIFY: hash every in-process collective's result on every rank and compare
    int team_fail_rank = -1;        // SN_TEAM_FAIL_RANK: this rank of the one-process team reports "no memory" (tests of the error path);
                                    // -2: the rank is read from SN_TEAM_FAIL_RANK_NOW at every reduction (a failure BETWEEN two successes)
    // Schur
    bool schur_nolazyrows = false;  // SN_SCHUR_NOLAZYROWS
    int schur_lazy_batch = 32;      // SN_SCHUR_LAZY_BATCH
    int schur_helpers = -1;         // SN_SCHUR_HELPERS: helper threads of the host window kernels (0 = none, -1 = five if the node has the cores)
    int schur_reuse = 0;            // SN_SCHUR_REUSE: fixed shift multiplicity (0 = adaptive)
    bool schur_nolookahead = false; // SN_SCHUR_NOLOOKAHEAD
    bool schur_profile = false;     // SN_SCHUR_PROFILE: one line of host-side timings per reduction on stderr
    bool aed_profile = false;       // SN_AED_PROFILE
    bool schur_hs_prio = true;      // SN_SCHUR_HS_PRIO=0: lazy H stream at the priority of the lazy Q stream (else one level above)
    int schur_cumask = 0;           // SN_SCHUR_CUMASK: CUs kept free of the lazy update streams (0 = no mask)
    // streams
    int stream_mode = 0;            // SN_STREAM_MODE: bit 0 critical streams, bit 1 lazy streams on hardware queues of their own (util.hip make_stream)
    int stream_lazy_free = 0;       // SN_STREAM_LAZY_FREE: CUs the lazy update streams of the Schur leg may not use (with bit 1 of the mode)
    char const *stream_space = nullptr; // SN_STREAM_SPACE: digit k = dummy queues created before the k-th stream (experiment)
    int stream_pad_prio = 0;        // SN_STREAM_PAD_PRIO: priority level of those dummy streams (0 high, 1 normal, 2 low)
    int stream_pad = 0;             // SN_STREAM_PAD: dummy high-priority streams created first (what a host application may have done)
    // GEMM
    bool gemm_separate_sum = true;  // SN_GEMM_SEPSUM=0: C += A B with the accumulators STARTING as C (every partial sum rounded at |C|)
    bool gemm_nosplit = false;      // SN_GEMM_NOSPLIT: whole tiles in the last round of workgroups too
    int gemm_kchunk = 0;            // SN_GEMM_KCHUNK: longest k of one split-K slice (0 = the built-in policy)
    int ht_two_stage = -1;          // SN_HT_TWOSTAGE: the two-stage Householder path of the Hessenberg-triangular reduction (ht_twostage.hip):
                                    // 1 always, 0 never, unset: from n = ht2_min_n on
    int ht2_min_n = 1100;           // SN_HT2_MIN_N: (the rotation path is the faster one up to n ~ 1000: profiles/r6_ht_crossover.txt; 1500 until the two-stage path's round-6 work)
    // QZ
    bool gep_serial = false;        // SN_GEP_SERIAL
    int gep_reuse = 0;              // SN_GEP_REUSE
    int gep_window = 0;             // SN_GEP_WINDOW: default AED window of the QZ path (0 = the built-in rule)
};

Tuning const &tuning();

} // namespace sn
