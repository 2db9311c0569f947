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

/* Eigenvalue reordering (reference expert.h:372-766).  Plans and blueprints name the reference's
 * task-insertion strategies: valid values are accepted, there is one schedule here. */
typedef enum {
    STARNEIG_REORDER_DEFAULT_PLAN    = 1,
    STARNEIG_REORDER_ONE_PART_PLAN   = 2,
    STARNEIG_REORDER_MULTI_PART_PLAN = 3
} starneig_reorder_plan_t;

typedef enum {
    STARNEIG_REORDER_DEFAULT_BLUEPRINT = 1,
    STARNEIG_REORDER_DUMMY_INSERT_A = 2,
    STARNEIG_REORDER_DUMMY_INSERT_B = 3,
    STARNEIG_REORDER_CHAIN_INSERT_A = 4,
    STARNEIG_REORDER_CHAIN_INSERT_B = 5,
    STARNEIG_REORDER_CHAIN_INSERT_C = 6,
    STARNEIG_REORDER_CHAIN_INSERT_D = 7,
    STARNEIG_REORDER_CHAIN_INSERT_E = 8,
    STARNEIG_REORDER_CHAIN_INSERT_F = 9
} starneig_reorder_blueprint_t;

#define STARNEIG_REORDER_DEFAULT_UPDATE_WIDTH               -1
#define STARNEIG_REORDER_DEFAULT_UPDATE_HEIGHT              -1
#define STARNEIG_REORDER_DEFAULT_TILE_SIZE                  -1
#define STARNEIG_REORDER_DEFAULT_VALUES_PER_CHAIN           -1
#define STARNEIG_REORDER_DEFAULT_WINDOW_SIZE                -1
#define STARNEIG_REORDER_ROUNDED_WINDOW_SIZE                -2
#define STARNEIG_REORDER_DEFAULT_SMALL_WINDOW_SIZE          -1
#define STARNEIG_REORDER_DEFAULT_SMALL_WINDOW_THRESHOLD     -1

struct starneig_reorder_conf {
    starneig_reorder_plan_t plan;
    starneig_reorder_blueprint_t blueprint;
    int tile_size;
    int values_per_chain;   /* rows of selected blocks that travel together (default: half a window) */
    int window_size;        /* rows of a diagonal window, at most 128 here (default 128) */
    int small_window_size;
    int small_window_threshold;
    int update_width;
    int update_height;
};

void starneig_reorder_init_conf(struct starneig_reorder_conf *conf);

#ifdef __cplusplus
}
#endif
#endif
