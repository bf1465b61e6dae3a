/*
 * dibs_hip.h -- C ABI of the MI355X-native DiBS SVGD engine (libdibs_hip.so).
 *
 * The reference (larslorch/dibs) has no FFI: its boundary is the Python class API
 *   MarginalDiBS(...).sample(...)   dibs/inference/svgd.py:60-77, 274-331
 *   JointDiBS(...).sample(...)      dibs/inference/svgd.py:425-442, 730-795
 * Each entry point below names the reference interface it replaces.  The Python facade
 * (dibs_amd/inference/svgd.py) binds these with ctypes; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions: every function returns 0 on success, non-zero on error; dibs_last_error() gives
 * the thread-local message.  Host buffers are caller-owned, row-major, float32 / int32 / uint32.
 * The engine owns all device memory.  A handle is not thread-safe; distinct handles are independent.
 */
#ifndef DIBS_HIP_H
#define DIBS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIBS_ABI_VERSION 2
#define DIBS_MAX_HIDDEN_LAYERS 8

enum { DIBS_LIK_BGE = 0, DIBS_LIK_LINGAUSS = 1, DIBS_LIK_DENSENN = 2 };
enum { DIBS_PRIOR_ER = 0, DIBS_PRIOR_SF = 1, DIBS_PRIOR_UNIFORM = 2 };
enum { DIBS_EST_SCORE = 0, DIBS_EST_REPARAM = 1 };
enum { DIBS_OPT_GD = 0, DIBS_OPT_RMSPROP = 1 };
enum { DIBS_RNG_LEGACY = 0, DIBS_RNG_PARTITIONABLE = 1 };
enum { DIBS_ACT_RELU = 0, DIBS_ACT_TANH = 1, DIBS_ACT_SIGMOID = 2, DIBS_ACT_LEAKYRELU = 3 };

/* POD mirror of the constructor kwargs (svgd.py:60-77 / 425-442) + the sizes sample() fixes
 * (svgd.py:274) + the hyper-parameters of the recognised model classes. */
typedef struct dibs_config {
  int32_t abi_version;      /* DIBS_ABI_VERSION */
  int32_t n_vars;           /* d   = x.shape[-1]                                   dibs.py:67        */
  int32_t n_dim;            /* k   = n_dim_particles (default d)                   svgd.py:138-139   */
  int32_t n_particles;      /* M (global, all ranks)                               svgd.py:274       */
  int32_t n_observations;   /* N   = x.shape[0]                                                      */
  int32_t n_grad_mc_samples;        /* S                                            dibs.py:60        */
  int32_t n_acyclicity_mc_samples;  /* Sa                                           dibs.py:61        */
  int32_t joint;            /* 0 = MarginalDiBS, 1 = JointDiBS                                       */
  int32_t likelihood;       /* DIBS_LIK_*                                                            */
  int32_t graph_prior;      /* DIBS_PRIOR_*                                                          */
  int32_t grad_estimator_z; /* DIBS_EST_*                                           dibs.py:309-318   */
  int32_t optimizer;        /* DIBS_OPT_*                                           svgd.py:117-122   */
  int32_t rng_layout;       /* DIBS_RNG_*  (jax_threefry_partitionable)                              */
  int32_t logistic_minval_tiny; /* 0: uniform minval = finfo.eps (jax default), 1: finfo.tiny        */
  int32_t has_interventions;/* advisory: 0 = interv_mask all zero (the engine derives everything from the mask passed to
                               dibs_engine_set_data; the field only records the caller's view)          svgd.py:86-87     */
  int32_t nn_n_hidden;      /* DenseNonlinearGaussian: len(hidden_layers)           nonlinearGaussian.py:105 */
  int32_t nn_hidden[DIBS_MAX_HIDDEN_LAYERS];
  int32_t nn_activation;    /* DIBS_ACT_*                                                            */
  int32_t nn_bias;
  int32_t rank;             /* particle shard: this engine owns particles                            */
  int32_t n_ranks;          /*   [rank*M/n_ranks, (rank+1)*M/n_ranks)                                */
  int32_t device_id;
  int32_t reserved_i[5];

  double alpha_linear;      /* dibs.py:70 */
  double beta_linear;       /* dibs.py:71 */
  double tau;               /* dibs.py:72 */
  double h_latent;          /* kernel.py:16 / :46  (kernel_param "h" / "h_latent") */
  double h_theta;           /* kernel.py:46 */
  double scale_latent;      /* kernel.py:16 / :46 */
  double scale_theta;
  double stepsize;          /* optimizer_param["stepsize"]   svgd.py:83 */
  double score_function_baseline; /* dibs.py:76 */
  double latent_prior_std;  /* <= 0: default 1/sqrt(k)       svgd.py:142, 301-302 */
  double graph_prior_edges_per_node; /* graph.py:27-30 */
  double bge_alpha_mu;      /* linearGaussian.py:44 */
  double bge_alpha_lambd;   /* <= 0: default d + 2           linearGaussian.py:45 */
  double lin_obs_noise;     /* linearGaussian.py:190 */
  double lin_mean_edge;
  double lin_sig_edge;
  double lin_min_edge;
  double nn_obs_noise;      /* nonlinearGaussian.py:105 */
  double nn_sig_param;
  double reserved_d[6];
} dibs_config;

typedef struct dibs_engine dibs_engine;

/* names of device buffers readable through dibs_engine_read_buffer (parity tests / callbacks) */
enum {
  DIBS_BUF_Z = 0,          /* f32 [Mloc, d, k, 2]                                 */
  DIBS_BUF_V_Z = 1,        /* f32 [Mloc, d, k, 2]   rmsprop avg_sq_grad           */
  DIBS_BUF_THETA = 2,      /* f32 [Mloc, P]         theta leaves concatenated     */
  DIBS_BUF_V_THETA = 3,
  DIBS_BUF_SCORES = 4,     /* f32 [Mloc, d, d]      U V^T                          */
  DIBS_BUF_LOGPROBS_Z = 5, /* f32 [Mloc, S]         l_s of the Z estimator         */
  DIBS_BUF_W_LIK = 6,      /* f32 [Mloc, d, d]      dl/dscores of the likelihood   */
  DIBS_BUF_W_ACYC = 7,     /* f32 [Mloc, d, d]      E[dh/dscores]                  */
  DIBS_BUF_GRAD_Z = 8,     /* f32 [Mloc, d, k, 2]   d/dz log p(z, D)               */
  DIBS_BUF_GRAD_THETA = 9, /* f32 [Mloc, P]                                        */
  DIBS_BUF_KXX = 10,       /* f32 [Mloc, M]         kxx[a_local, b_global]         */
  DIBS_BUF_PHI_Z = 11,     /* f32 [Mloc, d, k, 2]                                  */
  DIBS_BUF_BASELINE = 12,  /* f32 [Mloc]                                           */
  DIBS_BUF_NODE_SCORES = 13,/* f64 [Mloc, d, S]     BGe per-node scores            */
  DIBS_BUF_PARENT_MASKS = 14,/* u64 [Mloc, d, S, W] sampled parent sets of node j (bit i of word i/64 = g[i, j]) */
  DIBS_BUF_LOGPROBS_THETA = 15, /* f32 [Mloc, S]                                   */
  DIBS_BUF_PHI_THETA = 16,
  DIBS_BUF_GATHER = 17,    /* f32 packed all-gather payload        */
  DIBS_BUF_COUNT
};

/* kernel ids for dibs_engine_get_timers */
enum {
  DIBS_K_EDGE = 0, DIBS_K_BGE_NODES = 1, DIBS_K_LIK_WEIGHTS = 2, DIBS_K_ACYC = 3, DIBS_K_ZGRAD = 4,
  DIBS_K_KMAT = 5, DIBS_K_PHI_UPDATE = 6, DIBS_K_LIN_THETA = 7, DIBS_K_LIN_Z = 8, DIBS_K_NN_THETA = 9,
  DIBS_K_NN_Z = 10, DIBS_K_PACK = 11, DIBS_K_BGE_BIG = 12 /* the three queue launches */, DIBS_K_ACYC_REDUCE = 13 /* k_acyc_reduce */, DIBS_K_TAIL = 14 /* k_particle_grad */,
  DIBS_K_COUNT = 16
};

const char* dibs_last_error(void);
int dibs_abi_version(void);

/* replaces MarginalDiBS.__init__ / JointDiBS.__init__ (svgd.py:60-122, 425-487): validates the config,
 * selects the device, allocates all device state.  `stream` is a hipStream_t to launch on (NULL: the
 * engine creates its own). */
int dibs_engine_create(const dibs_config* cfg, void* stream, dibs_engine** out);
int dibs_engine_destroy(dibs_engine* e);

/* x: f32 [N, d]; interv_mask: i32 [N, d] or NULL; bge_mean_obs: f32 [d] or NULL  (svgd.py:61-64, 86-87) */
int dibs_engine_set_data(dibs_engine* e, const float* x, const int32_t* interv_mask, const float* bge_mean_obs);

/* replaces the prologue of sample() (svgd.py:293-307 / 750-766): key,subk = split(key);
 * _sample_initial_random_particles(subk); zero optimizer state and baselines; keeps the carry key. */
int dibs_engine_init_particles(dibs_engine* e, const uint32_t key[2]);

/* checkpoint / resume of the loop carry (svgd.py:315: (opt_state_z[, opt_state_theta], key, sf_baseline));
 * any pointer may be NULL (left untouched / not returned).  z, v_z: f32 [Mloc, d, k, 2]; theta, v_theta:
 * f32 [Mloc, P]; key: u32 [2]; baseline: f32 [Mloc]. */
int dibs_engine_set_state(dibs_engine* e, const float* z, const float* v_z, const float* theta,
                          const float* v_theta, const uint32_t* key, const float* baseline);
int dibs_engine_get_state(dibs_engine* e, float* z, float* v_z, float* theta, float* v_theta,
                          uint32_t* key, float* baseline);

/* replaces _svgd_loop(start, n_steps, carry) (svgd.py:269-272 / 724-727): runs steps t_start ..
 * t_start+n_steps-1 entirely on the device, blocking until done.  Single-rank engines only. */
int dibs_engine_run(dibs_engine* e, int32_t t_start, int32_t n_steps);

/* Inside a chunk the engine synchronises its two streams through flag words polled by kernels (cheaper than events by ~10 us per step).  The
 * waits are bounded; should one run into its bound (a device masked down to a few CUs, another process filling the GPU, a serialising tool),
 * dibs_engine_run / dibs_engine_run_sharded restore the loop carry they copied at the start of the chunk, switch this engine to events for
 * good and run the same steps again -- callers see only the delay.  (In a sharded run the ranks agree on this with one tiny all-gather per
 * chunk and repeat together.)  Returns how many chunks were repeated so far (-1: null handle). */
int dibs_engine_flag_fallbacks(const dibs_engine* e);
/* fault injection for the tests of that path: the next step that would publish the second stream's completion flag does not, so the polling
 * kernel runs into its bound (0.2 s) exactly once.  No effect on an engine that is not using the flags. */
int dibs_engine_debug_drop_next_flag(dibs_engine* e);

/* multi-rank split of one _svgd_step (svgd.py:226-267): phase A = per-particle estimators for the local
 * shard + packing of [z | grad_z (| theta | grad_theta)] into the send buffer; the caller all-gathers
 * (RCCL, torch.distributed) send -> recv; phase B = kernel matrix slab, phi and optimizer step for the
 * local shard.  Buffers are device pointers owned by the caller (torch tensors). Asynchronous on the
 * engine stream. */
int dibs_engine_step_local(dibs_engine* e, int32_t t, void* send_dev);
int dibs_engine_step_update(dibs_engine* e, int32_t t, const void* recv_dev);
int64_t dibs_engine_gather_elems_per_rank(const dibs_engine* e);
int dibs_engine_sync(dibs_engine* e);

/* The same split with the exchange OVERLAPPED (no reference counterpart: the reference has no multi-device code; the arithmetic is
 * svgd.py:226-267 / 673-721 unchanged).  The kernel matrix and the repulsive term need the VALUES [z | theta] of all particles, which
 * are final as soon as the previous optimizer step is done; only the GRADIENTS depend on phase A.  So each rank
 *   dibs_engine_step_update_planes(e, t, planes, vals_send)      ... ends phase B of step t with its new values in `vals_send`, rows
 *       [Mloc, Ev] of [z | theta] (dibs_engine_export_values writes them for the initial state);
 *   the caller all-gathers them on a SIDE stream into plane 0 of `planes` ([2][M][Ev] floats) and, behind the gather on that stream,
 *   dibs_engine_kmat_values(e, plane0, side_stream)              launches the kernel-matrix slab of step t + 1 -- both beside phase A;
 *   dibs_engine_step_local_grads(e, t + 1, grads_send)           phase A, writing only [grad_z | grad_theta] rows [Mloc, Ev];
 *   the caller all-gathers the gradient rows into plane 1 (the only exchange between the phases: half the bytes of the packed rows),
 *   makes the engine stream wait for the side stream's work, and calls dibs_engine_step_update_planes for step t + 1: phi + optimizer.
 * Results are bit-identical to step_local / step_update and to the single-rank engine.  Ev = dibs_engine_plane_elems_per_rank / Mloc. */
int64_t dibs_engine_plane_elems_per_rank(const dibs_engine* e);
int dibs_engine_export_values(dibs_engine* e, void* vals_send_dev);
int dibs_engine_step_local_grads(dibs_engine* e, int32_t t, void* grads_send_dev);
int dibs_engine_kmat_values(dibs_engine* e, const void* vals_all_dev, void* stream);
int dibs_engine_step_update_planes(dibs_engine* e, int32_t t, const void* planes_dev, void* vals_send_dev);

/* The exchange INSIDE the engine (north_star: "a single RCCL all-gather over xGMI per step"; no reference counterpart -- the reference has
 * no multi-device code): the step loop of a particle-sharded run in C, the collective issued by the engine on its own
 * stream through RCCL (bound at run time: librccl.so.1, in a torch process torch's copy).  One process per GPU:
 *   rank 0:      dibs_comm_unique_id(id)            once per communicator (DIBS_COMM_ID_BYTES each); the bytes go to every rank by any
 *                                                   means (MPI, a file, torch.distributed.broadcast_object_list ...)
 *   every rank:  dibs_engine_comm_init(e, ids, n)   ncclCommInitRank with rank = cfg.rank of cfg.n_ranks on the engine's device;
 *                                                   n = 1: one communicator; n = 2: a second one for the side stream of the overlapped exchange
 *                dibs_engine_run_sharded(e, t_start, n_steps, overlapped)   replaces _svgd_loop (svgd.py:269-272 / 724-727), blocking:
 *                   overlapped = 0:  phase A -> ncclAllGather of the packed rows [z | grad_z | theta | grad_theta] -> phase B   (per step)
 *                   overlapped = 1:  values [z | theta] all-gathered on a side stream beside phase A (kernel-matrix slab behind the gather),
 *                                    only the gradient rows between the phases (see the step_update_planes protocol above)
 *                dibs_engine_gather_particles(e, z_all, theta_all)          all ranks' particles on every rank (sample()'s return value)
 * Results are bit-identical to dibs_engine_run of a single-rank engine, for any number of ranks. */
#define DIBS_COMM_ID_BYTES 128
int dibs_comm_unique_id(void* id_out);
int dibs_engine_comm_init(dibs_engine* e, const void* ids, int32_t n_ids);
int dibs_engine_comm_destroy(dibs_engine* e);
int dibs_engine_run_sharded(dibs_engine* e, int32_t t_start, int32_t n_steps, int32_t overlapped);
int dibs_engine_gather_particles(dibs_engine* e, float* z_all, float* theta_all);

/* The same loop with the exchange through MAPPED PEER MEMORY instead of RCCL (no reference counterpart; dibs_amd/csrc/exchange_ipc.h): ranks that
 * share one device -- where RCCL refuses a communicator ("duplicate GPU") -- or devices with peer access.  One PROCESS per rank:
 *   every rank:  dibs_engine_ipc_export(e, blob)          allocates the rank's exchange arena and writes DIBS_IPC_HANDLE_BYTES describing it
 *                                                         (hipIpcMemHandle + sizes); the bytes of ALL ranks go to every rank, in rank order,
 *                                                         by any means (a file, a pipe, torch.distributed.all_gather_object ...)
 *                dibs_engine_comm_init_ipc(e, blobs_all)  maps the peers' arenas (n_ranks * DIBS_IPC_HANDLE_BYTES bytes)
 *                dibs_engine_run_sharded / dibs_engine_gather_particles as above (both protocols); dibs_engine_comm_destroy unmaps.
 * The all-gather of a step is every rank storing its rows into every peer's arena plus one sequence word per peer and exchange; buffers
 * alternate between two copies, so no acknowledgement travels back.  A rank waits at most DIBS_IPC_TIMEOUT_MS (environment, read by
 * dibs_engine_comm_init_ipc; default 10 000) for its peers' rows, then the run returns an error.  Results are bit-identical to
 * dibs_engine_run of a single-rank engine and to the RCCL path. */
#define DIBS_IPC_HANDLE_BYTES 128
int dibs_engine_ipc_export(dibs_engine* e, void* blob_out);
int dibs_engine_comm_init_ipc(dibs_engine* e, const void* blobs_all);

/* debugging / parity: copy a device buffer to the host (nbytes must match); theta size query */
int dibs_engine_read_buffer(dibs_engine* e, int32_t which, void* host, int64_t nbytes);
int64_t dibs_engine_buffer_bytes(const dibs_engine* e, int32_t which);
int64_t dibs_engine_theta_size(const dibs_engine* e);

/* per-kernel HIP-event timers.  enable=1 brackets every launch with events on the stream it is launched on and serialises the step
 * (the main stream waits for the acyclicity kernel before it continues: each duration is the kernel alone on the GPU); enable=2 keeps the production schedule -- the acyclicity kernel on the engine's second
 * stream beside the likelihood kernels -- and times it with events on that stream (durations of overlapping kernels include the sharing). */
int dibs_engine_set_profiling(dibs_engine* e, int32_t enable);
/* Gradient estimators of ONE step for the engine's current particles (dibs_engine_set_state) with EXPLICIT per-particle PRNG keys --
 * what the reference exposes as DiBS.eltwise_grad_z_likelihood(zs, thetas, baselines, t, subkeys) -> (grads, baselines)
 * (dibs/inference/dibs.py:295-321), DiBS.eltwise_grad_theta_likelihood(zs, thetas, t, subkeys) (:467-485) and
 * DiBS.eltwise_grad_latent_prior(zs, subkeys, t) (:626-658).  Same kernels as a step of dibs_engine_run, only the key of particle m
 * is keys_*[m] instead of row 1 + m of split(loop-carry key, M + 1); the loop-carry key does not advance.
 *   keys_theta / keys_lik / keys_prior : uint32 [Mloc][2] host arrays or NULL (NULL, NULL -> no likelihood pass; keys_prior NULL -> no
 *                                        prior pass).  Joint models run the theta and the Z estimator in one pass: pass both or neither.
 *   grad_z_lik  [Mloc][d][k][2] : estimator of grad_Z log p(theta, D | Z);  baseline_out [Mloc]: the updated score-function baselines
 *                                 (input baselines = the engine's state);  grad_theta [Mloc][P]: estimator of grad_theta (joint only)
 *   grad_z_prior [Mloc][d][k][2]: -beta(t) E[grad h] - Z / sigma_z^2 + grad log p(G_alpha(Z))
 * Any output may be NULL.  Blocking.  The loop state -- particles, optimizer moments, loop-carry key, score-function baselines -- is left
 * untouched (the updated baselines are only returned); the per-step scratch of the engine (packed rows, W_lik, log-probabilities, queues)
 * is overwritten, as by any step. */
int dibs_engine_eval_gradients(dibs_engine* e, int32_t t, const uint32_t* keys_theta, const uint32_t* keys_lik, const uint32_t* keys_prior,
                               float* grad_z_lik, float* baseline_out, float* grad_theta, float* grad_z_prior);
int dibs_engine_get_timers(dibs_engine* e, double* total_ms, int64_t* launches, int32_t n); /* arrays of DIBS_K_COUNT */
int dibs_engine_reset_timers(dibs_engine* e);
/* effective (executed) flop / problem-size counters of the BGe node kernel for the roofline */
int dibs_engine_get_counters(dibs_engine* e, double* out, int32_t n);

/* replaces likelihood_model.interventional_log_marginal_prob / interventional_log_joint_prob evaluated on a
 * batch of hard graphs (svgd.py:110-113, 370-372, 475-478, 838-841): g: i32 [n, d, d]; theta: f32 [n, P] or
 * NULL; x_ho: f32 [n_ho, d]; mask_ho: i32 [n_ho, d] or NULL; out: f32 [n]. */
int dibs_score_graphs(dibs_engine* e, const int32_t* g, const float* theta, int32_t n, const float* x_ho,
                      const int32_t* mask_ho, int32_t n_ho, float* out);

#ifdef __cplusplus
}
#endif
#endif /* DIBS_HIP_H */
