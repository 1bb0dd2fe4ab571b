/*
 * swarm_env.h -- C ABI of libswarmenv.so: the MI355X-native batched AssemblySwarm environment step.
 *
 * This is the drop-in boundary for the reference's env-step hot path.  The reference crosses its
 * Python -> native boundary with ctypes into libAssemblyEnv.so
 * (/root/reference/cus_gym/gym/envs/customized_envs/envs_cplus/c_lib.py:11-37), five times per step, one
 * environment at a time, from AssemblySwarmEnv.step()/_get_obs()/_get_reward()
 * (/root/reference/cus_gym/gym/envs/customized_envs/assembly.py:234-255,357-380,460-466,495-504,613-624).
 * Here the same Python host binds ONE handle-based library with ctypes; state lives in HBM and one
 * call advances E independent environments.
 *
 * Two groups of entry points:
 *
 *  (1) the batched ABI (swarm_*): opaque handle, plain pointers and sizes, int status returns.
 *      It replaces, fused into one launch per step, what the reference spreads over
 *        - AssemblySwarmEnv.step numpy glue             assembly.py:487-666 (incl. _get_dist_b2b :442-457)
 *        - _sf_b2b_all                                  AssemblyEnv.h:64-73   / AssemblyEnv.cpp:735-815
 *        - _get_dist_b2w                                AssemblyEnv.h:75-81   / AssemblyEnv.cpp:817-855
 *        - calculateActionPrior / robotPolicy           AssemblyEnv.h:98-109  / AssemblyEnv.cpp:1061-1196
 *        - _get_observation (+_get_focused, _get_target_grid_state, _make_periodic)
 *                                                       AssemblyEnv.h:13-34   / AssemblyEnv.cpp:18-351,628-732,858-908
 *        - _get_reward (+_rho_cos_dec)                  AssemblyEnv.h:35-58   / AssemblyEnv.cpp:354-626,1012-1020
 *
 *  (2) the legacy symbols (_get_observation, _get_reward, _sf_b2b_all, _get_dist_b2w,
 *      calculateActionPrior) with the reference's exact signatures and host-pointer / caller-owned
 *      buffer contract, so an unmodified assembly.py can load this library in place of
 *      libAssemblyEnv.so.  They run the same HIP kernels on a private one-environment handle.
 *
 * There is no CPU fallback anywhere in this library: without a HIP device every call fails.
 *
 * Memory contract of the batched ABI: every pointer passed to swarm_step / swarm_observe /
 * swarm_get_indices is a DEVICE pointer (hipMalloc / torch.Tensor.data_ptr()) valid on the handle's
 * device; the calls are asynchronous on the handle's stream (swarm_set_stream).  swarm_set_cells /
 * swarm_set_state / swarm_get_state accept host or device pointers (hipMemcpyDefault).
 *
 * Layouts (E = n_env, N = n_agents, D = obs_dim = 4*(topo + 1 + with_self) + 2*num_obs_grid_max):
 *   p, dp            double [E][2][N]     component-major per env, exactly the reference's (2, n_a) arrays
 *   cells            double [E][2][n_cells_max]  the reference's grid_center (2, n_g), row stride n_cells_max
 *   action           float|double [E][N][2]      agent-major pairs (u_x, u_y)
 *   obs              float|double [E][N][D]      one contiguous row per agent; row content and order are the
 *                                                reference's obs column (AssemblyEnv.cpp:294-306)
 *   reward           float  [E][N]    in {0,1}
 *   done             uint8  [E][N]    always 0 (assembly.py:480-482)
 *   a_prior          float|double [E][N][2]
 *   neighbor_index   int32  [E][N][topo], in_flags int32 [E][N], sensed_index int32 [E][N][num_obs_grid_max],
 *   occupied_index   int32  [E][N][num_occupied_grid_max]   (the reference's index scratch; debug export)
 * All arithmetic that decides an index, a flag, the state update or the reward is IEEE double in the
 * reference's operation order (no FMA contraction); obs / a_prior are rounded once to obs_dtype at the store.
 */
#ifndef SWARM_ENV_H
#define SWARM_ENV_H

#include <stdint.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define SWARM_ABI_VERSION 4

enum { SWARM_F32 = 0, SWARM_F64 = 1, SWARM_BF16 = 2 };

enum {
    SWARM_OK = 0,
    SWARM_ERR_INVALID = 1,      /* bad argument / unsupported configuration */
    SWARM_ERR_HIP = 2,          /* a HIP runtime call failed (no device, OOM, launch failure ...) */
    SWARM_ERR_STATE = 3         /* call order violated (e.g. step before cells/state were set) */
};

typedef struct swarm_env swarm_env_t;

/* Mirrors the constants AssemblySwarmEnv hard-codes or derives (assembly.py:27-81,99-131,193-199). */
typedef struct swarm_config {
    int32_t n_env;                  /* E >= 1 */
    int32_t n_agents;               /* N in [1, 256] */
    int32_t n_cells_max;            /* capacity of one env's cell list, >= every n_g, <= 32767 */
    int32_t topo_nei_max;           /* assembly.py:34   = 6 (1..6) */
    int32_t num_obs_grid_max;       /* assembly.py:128  = 80 */
    int32_t num_occupied_grid_max;  /* assembly.py:130  = 200 */
    int32_t is_boundary;            /* 1 = walls (assembly.py:99-103), 0 = periodic */
    int32_t with_self_state;        /* is_con_self_state */
    int32_t with_prior;             /* training_method == 'llm_rl': compute a_prior (assembly.py:605-624) */
    int32_t obs_dtype;              /* SWARM_F32 (product), SWARM_F64 (bit-exact parity mode) or SWARM_BF16 (obs and a_prior as
                                     * bfloat16 = the f32 value rounded to nearest-even: half the rollout's bytes, feeds the bf16
                                     * policy kernel of swarm_policy.h directly) */
    int32_t device;                 /* HIP device ordinal, -1 = current */
    int32_t debug_flags;            /* bit 0: force every exact (fp64) fallback path of the fp32 pre-filters; bit 1: disable the lattice path;
                                     * bit 2: a small batch keeps the full workgroup geometry (no half-occupied variant) -- results are
                                     * identical with any of them; bits 8..15: diagnostics (tools/ablate.py) */
    double d_sen;                   /* assembly.py:199  = 0.4 */
    double r_avoid;                 /* assembly.py:124 */
    double size_a;                  /* assembly.py:44   = 0.035 */
    double k_ball, k_wall, c_wall;  /* assembly.py:71-74 = 30, 100, 5 */
    double vel_max;                 /* assembly.py:52   = 0.8 */
    double dt;                      /* assembly.py:79   = 0.1 */
    double boundary[4];             /* [x_min, y_max, x_max, y_min], assembly.py:193-196 */
    double prior_gain[3];           /* prior policy gains: attraction, repulsion, alignment -- AssemblyEnv.cpp:1128-1132 = 2, 3, 2 */
    double llm_repulsion;           /* repulsion gain of the prior's Python twin robot_prior_policy, assembly.py:895 = 1.0 */
    int32_t llm_action;             /* 1: agent_strategy == 'llm' (assembly.py:525-529): every pass also evaluates that twin on
                                     * the new state; swarm_step(action = NULL) then applies it as the action */
    int32_t reserved_;
} swarm_config_t;

int  swarm_abi_version(void);
void swarm_default_config(swarm_config_t *cfg);     /* fills the reference's constants; caller sets sizes */

int  swarm_create(const swarm_config_t *cfg, swarm_env_t **out);
int  swarm_destroy(swarm_env_t *h);
/* Message of the last failing call on `h`; h == NULL: last failing swarm_create on this thread. */
const char *swarm_last_error(const swarm_env_t *h);

int  swarm_set_stream(swarm_env_t *h, void *hip_stream);   /* hipStream_t; NULL = default stream */
int  swarm_synchronize(swarm_env_t *h);
int  swarm_obs_dim(const swarm_env_t *h);

/* Target cells of envs [env_begin, env_begin+count): cells[count][2][n_cells_max] (host or device),
 * n_g[count], l_cell[count] (HOST arrays; l_cell feeds the in-shape threshold sqrt(2)*l_cell/2). */
int  swarm_set_cells(swarm_env_t *h, int env_begin, int count,
                     const double *cells, const int32_t *n_g, const double *l_cell);
int  swarm_set_state(swarm_env_t *h, const double *p, const double *dp);   /* [E][2][N], host or device */

/* Batched device-side reset (what AssemblySwarmEnv.reset() does per environment, assembly.py:156-219): a shape set
 * (HOST arrays: shape_cells[S][2][n_cells_max] in the shape frame = grid_center_origins[s].T, n_g[S], l_cell[S]) is
 * uploaded once; swarm_reset then draws, for every env, shape index, rotation, offset and the agents' positions /
 * velocities from a counter-based generator keyed by (seed, episode, env_offset + env) -- no host round trip, any
 * env range reproducible on any rank -- and runs the observation pass (obs may be NULL). */
int  swarm_set_shapes(swarm_env_t *h, int n_shapes, const double *shape_cells, const int32_t *n_g, const double *l_cell);
int  swarm_reset(swarm_env_t *h, uint64_t seed, uint64_t episode, int64_t env_offset, void *obs);
int  swarm_get_state(swarm_env_t *h, double *p, double *dp);
int  swarm_get_cells(swarm_env_t *h, double *cells, int32_t *n_g);        /* [E][2][n_cells_max], [E]; host or device */
/* Shape index each env drew in the last swarm_reset (assembly.py:160 `rand_shape_index`), [E] host or device; -1 for envs whose
 * cells were set through swarm_set_cells.  The host needs it for l_cell / shape_frequency (assembly.py:161-163). */
int  swarm_get_shape_index(swarm_env_t *h, int32_t *shape_index);

/* Recompute observations and the obs-derived caches (neighbor_index, in_flags, nearest cell) from the
 * current state: what AssemblySwarmEnv.reset() does with its final _get_obs() (assembly.py:221).
 * Must be called after swarm_set_state / swarm_set_cells and before swarm_step.  obs may be NULL. */
int  swarm_observe(swarm_env_t *h, void *obs);

/* One AssemblySwarmEnv.step(a) for every env.  action_dtype: SWARM_F32 / SWARM_F64.
 * reward / done / a_prior may be NULL (not written). */
/* action == NULL (handles created with llm_action only): apply the 'llm' strategy's action of the current state. */
int  swarm_step(swarm_env_t *h, const void *action, int action_dtype,
                void *obs, float *reward, uint8_t *done, void *a_prior);

/* ---- reference-shaped HOST outputs: the numpy API of AssemblySwarmEnv.step / reset (assembly.py:487-666,156-223) ----
 * The reference returns obs (D, n_a) float64, reward (1, n_a) float64, done (1, n_a) bool, a_prior (2, n_a) float64
 * (assembly.py:227-231,353,480-482,612,663-666).  With E environments the agent axis is n_a = E*N, env-major.  The library
 * owns the step outputs on the device, widens / transposes them into that layout on the device and moves them with ONE
 * asynchronous copy into pinned host memory it owns: two slots (ping-pong; the arrays of the previous step stay valid while
 * the next one is taken).  swarm_host_outputs returns the four host arrays of a slot (valid until swarm_destroy);
 * a_prior is meaningful only with with_prior.  Both calls below return after the data has landed in the slot. */
typedef struct swarm_host_out {
    double  *obs;        /* (D, E*N)  row-major: obs[r * E*N + e*N + i] */
    double  *a_prior;    /* (2, E*N) */
    double  *reward;     /* (1, E*N) */
    uint8_t *done;       /* (1, E*N)  0 / 1 */
} swarm_host_out_t;
int  swarm_host_outputs(swarm_env_t *h, int slot, swarm_host_out_t *out);
/* swarm_observe + export of obs into `slot` (the tail of reset(), assembly.py:221-223). */
int  swarm_observe_host(swarm_env_t *h, int slot);
/* One step + export into `slot`.  action: action_on_device == 0: HOST array in the reference's layout (2, E*N)
 * (component-major, envs side by side), float or double; action_on_device == 1: DEVICE [E][N][2] as swarm_step;
 * NULL: the 'llm' strategy's own action (llm_action handles only). */
int  swarm_step_host(swarm_env_t *h, const void *action, int action_dtype, int action_on_device, int slot);

/* The 'llm' strategy's action for the CURRENT state (what swarm_step(action = NULL) would apply): [E][N][2] doubles, host or
 * device pointer.  llm_action handles only. */
int  swarm_get_llm_action(swarm_env_t *h, double *action);

/* Evaluation metrics of the current state, per env: out[E][3] (DEVICE pointer, double) =
 * [coverage_rate, distribution_uniformity, voronoi_based_uniformity] of
 * /root/reference/cus_gym/gym/wrappers/customized_envs/assembly_wrapper.py:48-128 (numpy semantics incl. np.var). */
int  swarm_metrics(swarm_env_t *h, double *out);

/* The reference's rule-based expert controller (agent_strategy == 'rule', assembly.py:530-601) evaluated on the CURRENT
 * state: action[E][N][2] (DEVICE pointer, double), clipped to [-1, 1]; feed it to swarm_step with SWARM_F64 to reproduce
 * a rule-mode step (with is_collected the reference returns it as the fifth element, assembly.py:663-664).  Runs the
 * observation pass with the index export switched on, then a one-thread-per-agent kernel.  fp64 in numpy's operation
 * order; np.cos differs from the device cos by a few ulp, so parity is 1e-12 absolute, not bit-exact. */
int  swarm_rule_action(swarm_env_t *h, double *action);

/* Index scratch of the CURRENT state (device pointers, any may be NULL).  The step keeps neighbor_index / in_flags / the
 * sensed and occupied lists in LDS and writes none of them to HBM; this call re-runs the observation pass on the current
 * state with the export switched on (it recomputes the same caches from the same state: idempotent) and copies them out. */
int  swarm_get_indices(swarm_env_t *h, int32_t *neighbor_index, int32_t *in_flags,
                       int32_t *sensed_index, int32_t *occupied_index);

/* How many environments currently have target cells that are a row-major subset of a square lattice (the reference's
 * tiled shapes always are, its seven fig PNGs included).  When ALL do (and n_agents <= 64), the sensed / occupied bit sets are built by a row walk
 * over the lattice instead of the all-cells scan; results are identical.  debug_flags bit 1 disables that path. */
int  swarm_lattice_envs(const swarm_env_t *h);

/* Roofline helper: bytes one swarm_step moves by SURVEY.md section 8d's accounting IN THIS BUILD'S DTYPES (fp64 state and
 * cells); bench.py's `roofline` uses section 8d's own fp32 figure (821 B per agent-step + 8 B per cell) and reports this one
 * beside it. */
double swarm_step_algorithmic_bytes(const swarm_env_t *h);
/* Time the last N launches?  No: timing lives in the caller (HIP events on the handle's stream).
 * These two record / read HIP events on that stream so a ctypes host needs no HIP binding. */
int  swarm_timer_start(swarm_env_t *h);
int  swarm_timer_stop(swarm_env_t *h, float *elapsed_ms);   /* synchronizes on the stop event */

/* ---- legacy symbols: exact reference signatures (AssemblyEnv.h:13-34,35-58,64-73,75-81,98-109) ---- */
void _get_observation(double *p_input, double *dp_input, double *heading_input, double *obs_input,
                      double *boundary_pos_input, double *grid_center_input, int *neighbor_index_input,
                      int *in_flags_input, int *sensed_index_input, int *occupied_index_input,
                      double d_sen, double r_avoid, double l_cell, double Vel_max, int topo_nei_max,
                      int num_obs_grid_max, int num_occupied_grid_max, int n_a, int n_g,
                      int obs_dim_agent, int dim, bool *condition);
void _get_reward(double *p_input, double *dp_input, double *heading_input, double *act_input,
                 double *reward_input, double *boundary_pos_input, double *grid_center_input,
                 int *neighbor_index_input, int *in_flags_input, int *sensed_index_input,
                 int *occupied_index_input, double d_sen, double r_avoid, double l_cell,
                 int topo_nei_max, int num_obs_grid_max, int num_occupied_grid_max, int n_a, int n_g,
                 int dim, bool *condition, bool *is_collide_b2b_input, bool *is_collide_b2w_input,
                 double *coefficients);
void _sf_b2b_all(double *p_input, double *sf_b2b_input, double *d_b2b_edge_input,
                 bool *is_collide_b2b_input, double *boundary_pos_input, double *d_b2b_center_input,
                 int n_a, int dim, double k_ball, bool is_periodic);
void _get_dist_b2w(double *p_input, double *r_input, double *d_b2w_input, bool *isCollision_input,
                   int dim, int n_a, double *boundary_pos);
void calculateActionPrior(double *p_input, double *dp_input, double *a_prior_input,
                          double *grid_center_input, int *neighbor_index_input, double d_sen,
                          double r_avoid, double l_cell, int topo_nei_max, int n_a, int n_g, int dim);

/* The legacy entry points return void like the reference's; after a call, its status on the calling thread: 0 = ok,
 * 1 = failed (outputs were filled with NaN, a line was written to stderr) and the reason (e.g. n_a > 256, no HIP device). */
int  swarm_legacy_status(void);
const char *swarm_legacy_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SWARM_ENV_H */
