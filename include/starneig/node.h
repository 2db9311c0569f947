/* Library life cycle.  Replaces: reference src/include/starneig/node.h:72-220
 * (implementation common/node.c:435-650).  There is no StarPU underneath:
 * "cores" = host threads the library may use (staging copies, the helper team of
 * the host window kernels); "gpus" = how many MI355X devices of the node THIS
 * process drives, min(gpus, devices present) like common/node.c:200-216, one host
 * thread per device (csrc/node_team.hip): with more than one,
 * starneig_SEP_SM_Hessenberg / _Schur run sharded over them. */
#ifndef STARNEIG_AMD_NODE_H
#define STARNEIG_AMD_NODE_H
#ifdef __cplusplus
extern "C" {
#endif

#define STARNEIG_USE_ALL           -1

typedef unsigned starneig_flag_t;

#define STARNEIG_DEFAULT           0x0
#define STARNEIG_HINT_SM           0x0
#define STARNEIG_HINT_DM           0x1
#define STARNEIG_FXT_DISABLE       0x2
#define STARNEIG_AWAKE_WORKERS     0x4
#define STARNEIG_AWAKE_MPI_WORKER  0x8
#define STARNEIG_FAST_DM \
    (STARNEIG_HINT_DM | STARNEIG_AWAKE_WORKERS | STARNEIG_AWAKE_MPI_WORKER)
#define STARNEIG_NO_VERBOSE        0x10
#define STARNEIG_NO_MESSAGES       (STARNEIG_NO_VERBOSE | 0x20)

/* node.h:178 -- aborts (like the reference, node.c:442-443) when called twice
 * or when no gfx950 device is usable: there is no CPU fallback. */
void starneig_node_init(int cores, int gpus, starneig_flag_t flags);
int  starneig_node_initialized(void);            /* node.h:186 */
int  starneig_node_get_cores(void);              /* node.h:193 */
void starneig_node_set_cores(int cores);         /* node.h:200 */
int  starneig_node_get_gpus(void);               /* node.h:207 */
void starneig_node_set_gpus(int gpus);           /* node.h:214 */
void starneig_node_finalize(void);               /* node.h:220 */
/* node.h:234-241.  Accepted, no effect: host arrays always travel through the
 * library's own pinned staging slots (csrc/staging.hip). */
void starneig_node_enable_pinning(void);
void starneig_node_disable_pinning(void);

#ifdef __cplusplus
}
#endif
#endif
