/* Expert configuration structures.  Replaces: reference
 * src/include/starneig/expert.h:67-99 (Hessenberg) and :113-370 (Schur).
 * Field order and the -1/-2/-3 sentinels are ABI: a caller compiled against
 * the reference header passes the same bytes. */
#ifndef STARNEIG_AMD_EXPERT_H
#define STARNEIG_AMD_EXPERT_H
#ifdef __cplusplus
extern "C" {
#endif

#define STARNEIG_HESSENBERG_DEFAULT_TILE_SIZE     -1
#define STARNEIG_HESSENBERG_DEFAULT_PANEL_WIDTH   -1

struct starneig_hessenberg_conf {
    int tile_size;    /* accepted and range-checked; the device layout is untiled */
    int panel_width;  /* columns per WY panel; default follows interface.c:74-78 */
};

void starneig_hessenberg_init_conf(struct starneig_hessenberg_conf *conf);

#define STARNEIG_SCHUR_DEFAULT_INTERATION_LIMIT        -1
#define STARNEIG_SCHUR_DEFAULT_TILE_SIZE               -1
#define STARNEIG_SCHUR_DEFAULT_SMALL_LIMIT             -1
#define STARNEIG_SCHUR_DEFAULT_AED_WINDOW_SIZE         -1
#define STARNEIG_SCHUR_DEFAULT_AED_NIBBLE              -1
#define STARNEIG_SCHUR_DEFAULT_AED_PARALLEL_SOFT_LIMIT -1
#define STARNEIG_SCHUR_DEFAULT_AED_PARALLEL_HARD_LIMIT -1
#define STARNEIG_SCHUR_DEFAULT_SHIFT_ORIGIN            -1
#define STARNEIG_SCHUR_DEFAULT_SHIFT_COUNT             -1
#define STARNEIG_SCHUR_DEFAULT_WINDOW_SIZE             -1
#define STARNEIG_SCHUR_ROUNDED_WINDOW_SIZE             -2
#define STARNEIG_SCHUR_DEFAULT_SHIFTS_PER_WINDOW       -1
#define STARNEIG_SCHUR_DEFAULT_UPDATE_WIDTH            -1
#define STARNEIG_SCHUR_DEFAULT_UPDATE_HEIGHT           -1
#define STARNEIG_SCHUR_DEFAULT_THRESHOLD               -1
#define STARNEIG_SCHUR_NORM_STABLE_THRESHOLD           -2
#define STARNEIG_SCHUR_LAPACK_THRESHOLD                -3

struct starneig_schur_conf {
    int iteration_limit;
    int tile_size;
    int small_limit;
    int aed_window_size;
    int aed_nibble;
    int aed_parallel_soft_limit;
    int aed_parallel_hard_limit;
    int shift_origin;
    int shift_count;
    int window_size;
    int shifts_per_window;
    int update_width;
    int update_height;
    double left_threshold;
    double right_threshold;
    double inf_threshold;
};

void starneig_schur_init_conf(struct starneig_schur_conf *conf);

#ifdef __cplusplus
}
#endif
#endif
