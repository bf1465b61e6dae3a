// libdibs_hip.so -- engine + C ABI (include/dibs_hip.h).  gfx950 only.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <string>
#include <vector>
#include <utility>

#define DIBS_TU_ENGINE
#include "../../include/dibs_hip.h"
#include "launch.h"
#include <map>
#include <mutex>
#include <dlfcn.h>
#include <rccl/rccl.h>  // declarations only: librccl is bound at run time (dibs_rccl below), libdibs_hip.so does not link it
#include "kernels_marginal.h"
#include "kernels_tail.h"
#include "kernels_joint.h"
#include "kernels_nn.h"
#include "kernels_bge_soft.h"
#include "exchange_ipc.h"
#include <unistd.h>

#define LDS_LIMIT ((size_t)160 * 1024)
// profiling counters (dibs_engine_get_counters): [0] executed Cholesky flops, [1..4] phases of k_particle_grad (100 MHz ticks of block 0),
// [8..12] phases of k_edge_scores, [16..21] phases of k_phi_update, [24..] k_bge_chol
#define DIBS_N_COUNTERS 8192  // ([64 ..]: per-block (start, end) clock stamps of the kernel under investigation)
static thread_local std::string g_err;
static int fail(const std::string& m) {
  g_err = m;
  return 1;
}
#define HIP_OK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(_e));              \
  } while (0)

// BGe statistics that do not depend on the graph (linearGaussian.py:78-94): R_j, N_j, the (j, l) table of log_gamma_term,
// and for the complement form of kernels_bge.h R_j^-1 and logdet R_j.  Computed once per data set on the host in double,
// uploaded as f32 / f64.  Owns its device buffers.
struct BgeStats {
  float *R = nullptr, *Rp = nullptr, *Qp = nullptr;  // Qp = Rp + n_mats (d+1)^2: ONE allocation (see bge_upload)
  double *gam = nullptr, *Nj = nullptr, *ldR = nullptr;
  int n_mats = 1;
  double alpha_lambd = 0, alpha_mu = 0, log_t = 0;
  void release() {
    void* ptrs[] = {R, Rp, gam, Nj, ldR};
    for (void* p_ : ptrs)
      if (p_) hipFree(p_);
    R = Rp = Qp = nullptr;
    gam = Nj = ldR = nullptr;
  }
  ~BgeStats() { release(); }
  BgeStats() = default;
  BgeStats(const BgeStats&) = delete;
  BgeStats& operator=(const BgeStats&) = delete;
  BgeParams params() const { return BgeParams{Rp, Qp, gam, Nj, ldR, alpha_lambd, n_mats}; }
};

struct dibs_engine {
  dibs_config cfg;
  DibsTuning tune;  // the environment switches (tuning.h), latched at creation
  int d, k, M, Mloc, m0, N, S, Sa, W;
  int64_t D, P, E, Ev;  // z elems / theta elems per particle, packed row stride [z | grad_z | theta | grad_theta], plane row stride [z | theta] (floats)
  int dpad, ldk, edge_kc, acyc_nt, acyc_cpb, acyc_nblk, acyc_units;
  float sigz;
  hipStream_t stream;
  bool own_stream;
  // state
  float *z, *vz, *theta, *vtheta, *baseline, *baseline2;
  Key2 key;
  // data
  float* x;
  int32_t* mask;
  BgeStats bge;
  bool kmat_fused;  // this step's latent kernel matrix was computed inside the k_bge_sample launch
  float* soft_ds;  // [Mloc, S, d, d]  BGe reparam estimator: per-sample score-space gradients
  float* soft_tri = nullptr;  // ... beyond 128 variables: the waves' packed triangles (factor | inverse columns) in global scratch
  int soft_blocks = 0;        //     of this many persistent blocks (kernels_bge_soft.h, GLOB)
  bool has_data;
  // work
  float* w_tot;     // [Mloc][d][d] total score-space gradient when a particle's W, U, V do not fit in one block's LDS (kernels_tail.h)
  float* acyc_big;  // n_vars > 112: buffers of the global-memory matrix powers (kernels_acyc_big.h)
  float* eas;       // [Mloc][d][d] exp(-alpha s) of this step (k_edge_scores -> k_acyc_hf / k_acyc_hfw); n_vars <= 112 only
  float *scores, *probs, *w_lik, *acyc_part, *w_acyc, *logprobs_z, *logprobs_th, *pack, *kz, *kt, *phi_z, *phi_th;
  unsigned int* fork_flag = nullptr;  // [0] sequence number published by k_edge_scores_p's last block, [1] its block counter (flag fork)
  unsigned int fork_seq = 0;
  double* kpart = nullptr;  // tiled kernel matrix (kernels_kmat.h): partial squared distances [nsplit][Mloc][M]
  int kmat_ns_max = 0;     // 0: the direct kernel k_kmat; otherwise the largest nsplit kpart has room for
  unsigned int* kmat_ctr = nullptr;  // one counter per tile (units riding in k_particle_grad: the last unit of a tile writes the entries)
  float* ksum = nullptr;  // joint models: kz + kt, formed by the k_kmat launch of kt (the weight matrix of the SVGD transform as ONE scalar-loadable array)
  uint32_t* thr;
  uint64_t* masks;
  BgeQueues bq;
  double* node_scores;
  unsigned long long* counters;
  JointWork jw;
  // profiling
  bool profiling;
  bool profiling_concurrent;  // set_profiling(2): keep the second stream while timing (the acyclicity kernel is timed on its own stream)
  hipEvent_t ev0, ev1;
  hipStream_t stream2;      // the acyclicity kernel (needs only the edge scores) runs beside sampling -> factorisation -> weights: its bf16 MFMAs
                            // overlap with their vector work.  Same arithmetic, same results; DIBS_NO_ACYC_STREAM2 keeps one stream.
  hipEvent_t ev_fork, ev_join, ev_k0, ev_k1;
  // round 5: the fork of a step without a record packet on the main stream -- the event IS the edge kernel's completion signal
  // (hipExtLaunchKernel stop event; scripts/probe/stream_hop.hip: 5.7 -> 2.2 us between k_edge_scores and k_bge_sample) -- and, optionally,
  // the join as a flag polled inside k_particle_grad instead of an event wait in front of it (DIBS_FLAG_JOIN=1)
  unsigned int* join_flag = nullptr;   // device word: sequence number stored by the second stream's last kernel of a step (k_join_flag)
  unsigned int* join_err = nullptr;    // pinned host word: raised by tail_join_wait when the flag did not arrive (checked after every chunk)
  unsigned int join_seq = 0;
  bool streams_concurrent = false;     // kernels of the two streams run side by side (probed at creation): the in-kernel join is safe
  // The in-kernel flags (fork: k_wait_flag, join: tail_join_wait) need the two streams to make progress side by side.  That is probed at
  // creation and holds for an engine alone on its GPU; a masked-down device, a second process that fills the machine or a serialising tool
  // can still starve the polled kernel.  The waits are bounded; a chunk that saw a time-out is REPEATED on events from a copy of its
  // loop carry taken at the chunk's start, and the engine stays on events from then on (run_chunk_guarded).
  bool flags_now = false;              // this chunk / call uses the flags (latch_flags)
  bool flags_off = false;              // a wait timed out once: events for the rest of the engine's life
  int flag_fallbacks = 0;              // chunks repeated on events (dibs_engine_flag_fallbacks)
  bool debug_drop_flag = false;        // tests: the next step that would publish the join flag does not (dibs_engine_debug_drop_next_flag)
  float* carry_bak = nullptr;          // [Mloc (2 D + 2 P + 1)] z | v_z | theta | v_theta | baseline at the start of the chunk
  Key2 key_bak;
  bool kmat_early;  // this step's kernel matrices were launched on the second stream (behind the acyclicity kernel)
  bool kmat_ext;    // ... or by dibs_engine_kmat_values on a stream of the caller (overlapped exchange)
  double t_ms[DIBS_K_COUNT];
  int64_t t_n[DIBS_K_COUNT];
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending;
  bool has_mean_obs;
  std::vector<float> mean_obs;
  // dibs_score_graphs: statistics / device copies of the last (x_ho, mask_ho) scored against (held-out evaluators and mixture weights
  // call it repeatedly with the same data: svgd.py:110-113, 370-372)
  // in-engine exchange (dibs_engine_comm_init / dibs_engine_run_sharded): RCCL communicators of this rank -- comm[0] on the engine stream,
  // comm[1] on the side stream of the overlapped protocol -- and the buffers of that protocol
  ncclComm_t comm[2] = {nullptr, nullptr};
  int n_comms = 0;
  float *planes = nullptr, *vsend = nullptr;  // [2][M][Ev] values | gradients of all particles, [Mloc][Ev] this rank's new values
  hipStream_t side = nullptr;
  hipEvent_t ev_exported = nullptr, ev_vals = nullptr;
  bool vals_fresh = false;  // plane 0 (and the kernel slab computed from it) belongs to the engine's current particles
  bool loopback = false;    // comm_init(NULL): collectives skipped (per-rank timing on one GPU)
  IpcComm ipc;              // the exchange through mapped peer memory instead of RCCL (exchange_ipc.h; dibs_engine_comm_init_ipc)
  uint32_t* agree_dev = nullptr;  // [4 + 4 n_ranks] this rank's error word of a chunk (16 bytes) | all ranks' (run_sharded's agreement)
  uint32_t* agree_host = nullptr; // pinned mirror
  struct ScoreCache {
    std::vector<float> x;
    std::vector<int32_t> mask;
    bool has_mask = false, valid = false;
    BgeStats st;
    JointWork jw;
    ScoreCache() { memset(&jw, 0, sizeof jw); }
    ~ScoreCache() { joint_free(&jw); }
    bool matches(const float* x_, const int32_t* m_, size_t n) const {
      return valid && x.size() == n && has_mask == (m_ != nullptr) && memcmp(x.data(), x_, n * 4) == 0 && (!m_ || memcmp(mask.data(), m_, n * 4) == 0);
    }
    void remember(const float* x_, const int32_t* m_, size_t n) {
      x.assign(x_, x_ + n);
      has_mask = m_ != nullptr;
      if (m_) mask.assign(m_, m_ + n);
      valid = true;
    }
  } score_cache;
};

extern "C" const char* dibs_last_error(void) { return g_err.c_str(); }
extern "C" int dibs_abi_version(void) { return DIBS_ABI_VERSION; }

static NNParams nn_params(const dibs_config& c) {
  NNParams p{c.nn_hidden[0], c.nn_activation, c.nn_bias, (float)c.nn_obs_noise, (float)c.nn_sig_param, c.nn_n_hidden, {}};
  for (int l = 0; l < c.nn_n_hidden && l < DIBS_MAX_HIDDEN_LAYERS; ++l) p.hidden[l] = c.nn_hidden[l];
  return p;
}

static int64_t theta_size(const dibs_config& c) {
  const int d = c.n_vars;
  if (!c.joint) return 0;
  if (c.likelihood == DIBS_LIK_LINGAUSS) return (int64_t)d * d;
  if (c.likelihood == DIBS_LIK_DENSENN) {
    int64_t p = 0;
    int in = d;
    for (int l = 0; l <= c.nn_n_hidden; ++l) {
      const int out = l < c.nn_n_hidden ? c.nn_hidden[l] : 1;
      p += (int64_t)d * in * out + (c.nn_bias ? (int64_t)d * out : 0);
      in = out;
    }
    return p;
  }
  return 0;
}

template <typename T>
static hipError_t dalloc(T** p, size_t n) {
  *p = nullptr;
  if (n == 0) return hipSuccess;
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e == hipSuccess) e = hipMemset(*p, 0, n * sizeof(T));
  return e;
}

// Engines alive in this process.  The in-kernel flags (fork: k_wait_flag, join: tail_join_wait) are used by an engine that is ALONE in its
// process -- the production layout, one process per GPU: its two or three streams have a hardware queue each.  Several engines in one process
// (the single-GPU emulation of a sharded run, tests with rank engines) share hardware queues, and a polling kernel at the head of a shared
// queue holds up the kernels behind it, possibly the one it waits for, until its bound: those engines use events.  DIBS_FLAGS_MULTI=1 lifts
// the rule (scripts/gpu_shard_scaling.py: what a rank of a real run would do).
static std::atomic<int> g_live_engines{0};
extern "C" int dibs_engine_destroy(dibs_engine* e);
extern "C" int dibs_engine_comm_destroy(dibs_engine* e);
// sizes, stream, events and every device buffer of a new engine; on failure the caller destroys the half-built engine
static int engine_alloc(dibs_engine* e, const dibs_config& c, void* stream) {
  e->cfg = c;
  e->tune = dibs_tuning_from_env();
  e->d = c.n_vars;
  e->k = c.n_dim;
  e->M = c.n_particles;
  e->Mloc = c.n_particles / c.n_ranks;
  e->m0 = c.rank * e->Mloc;
  e->N = c.n_observations;
  e->S = c.n_grad_mc_samples;
  e->Sa = c.n_acyclicity_mc_samples;
  e->W = (e->d + 63) / 64;
  e->D = (int64_t)e->d * e->k * 2;
  e->P = theta_size(c);
  e->E = ((2 * e->D + 2 * e->P) + 3) & ~(int64_t)3;
  e->Ev = ((e->D + e->P) + 3) & ~(int64_t)3;
  e->dpad = (e->d + 15) & ~15;
  {
    // k_edge_scores walks the latent dimension in chunks of edge_kc columns: one chunk when U and V fit in LDS (n_vars <= 112 with k = d)
    e->edge_kc = e->k;
    for (;;) {
      const int kp = (e->edge_kc + 3) & ~3;
      e->ldk = kp + ((2 - kp) % 32 + 32) % 32;  // ldk == 2 (mod 32): conflict-free MFMA operand reads
      if ((size_t)2 * e->dpad * e->ldk * 4 <= LDS_LIMIT - 8192 || e->edge_kc <= 16) break;
      e->edge_kc = e->edge_kc > 64 ? 64 : e->edge_kc / 2;
    }
  }
  e->acyc_nt = e->dpad / 16;
  {  // chains per block: fill the 256 CUs in whole rounds (resident blocks per CU limited by the 3 LDS matrices)
    const size_t lds = (size_t)(3 * e->dpad + 1) * (e->dpad + 4) * 4;
    const int per_cu = (int)(LDS_LIMIT / lds) < 1 ? 1 : (int)(LDS_LIMIT / lds);
    const int slots = 256 * (per_cu > 8 ? 8 : per_cu);
    // a work unit is a PAIR of chains when the PRNG layout lets one Threefry call serve both (see k_acyc), else one chain
    const bool paired = c.rng_layout == DIBS_RNG_LEGACY && (e->Sa & 1) == 0 && (uint64_t)e->Sa * e->d * e->d < 0xFFFFFFFFull;
    const int n_units = paired ? e->Sa / 2 : e->Sa;
    e->acyc_units = n_units;
    int best = 1;
    long best_cost = -1;
    for (int cpb = 1; cpb <= n_units; ++cpb) {
      // (sized for the GLOBAL particle count: the grouping of the Sa chains into partial sums must not depend on how the
      //  particles are sharded, or results would differ between rank counts in the last float bit)
      const long nblk = (long)((n_units + cpb - 1) / cpb) * e->M;
      const long cost = ((nblk + slots - 1) / slots) * cpb;
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cpb; }
    }
    e->acyc_cpb = best;
  }
  e->acyc_nblk = (e->acyc_units + e->acyc_cpb - 1) / e->acyc_cpb;
  e->sigz = c.latent_prior_std > 0 ? (float)c.latent_prior_std : 1.0f / sqrtf((float)e->k);
  if (stream) {
    e->stream = (hipStream_t)stream;
    e->own_stream = false;
  } else {
    // the engine's own main stream at the greatest priority (its chain -- sampling, factorisation, tail -- is the later one of a step):
    // bench.py, same box, alternating: 5 336 / 5 349 steps/s against 5 328 / 5 321 at the normal priority
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, hi) != hipSuccess) {
      (void)hipGetLastError();
      HIP_OK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    }
    e->own_stream = true;
  }
  HIP_OK(hipEventCreate(&e->ev0));
  HIP_OK(hipEventCreate(&e->ev1));
  if (!e->tune.no_stream2) {
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);  // (lo = least, hi = greatest priority)
    // round 3, headline size: serial 3 906 steps/s; second stream at least / normal / greatest priority 3 906 / 3 964 / 4 001.
    // round 5 (fork by flag, both chains start together -- see flag_fork in step_local): least priority; with the event fork the three
    // priorities measure the same now (5 193-5 217), configs 3 / 5 gain 1-2 % at the least priority, config 4 is unchanged.
    const int prio = lo;
    // (an optimisation only: without it every kernel goes to the engine stream)
    if (hipStreamCreateWithPriority(&e->stream2, hipStreamNonBlocking, prio) != hipSuccess &&
        hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking) != hipSuccess) {
      e->stream2 = nullptr;
      (void)hipGetLastError();
    }
  }
  if (e->stream2) {
    HIP_OK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&e->ev_k0, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&e->ev_k1, hipEventDisableTiming));
    HIP_OK(dalloc(&e->join_flag, (size_t)4));
    HIP_OK(dalloc(&e->fork_flag, (size_t)4));
    HIP_OK(hipHostMalloc((void**)&e->join_err, 4, hipHostMallocDefault));
    *e->join_err = 0u;
    // the in-kernel join needs the two streams to make progress side by side (see k_probe_wait): asked once per process and device
    static std::mutex mu;
    static std::map<int, bool> concurrent;
    std::lock_guard<std::mutex> lock(mu);
    auto it = concurrent.find(c.device_id);
    if (it == concurrent.end()) {
      unsigned int* pr = nullptr;
      HIP_OK(dalloc(&pr, (size_t)2));
      hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, e->stream, pr, pr + 1);
      hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, e->stream2, pr);
      HIP_OK(hipStreamSynchronize(e->stream));
      HIP_OK(hipStreamSynchronize(e->stream2));
      unsigned int seen = 0;
      HIP_OK(hipMemcpy(&seen, pr + 1, 4, hipMemcpyDeviceToHost));
      hipFree(pr);
      it = concurrent.emplace(c.device_id, seen != 0u).first;
    }
    e->streams_concurrent = it->second;
  }
  const size_t Ml = e->Mloc, dd = (size_t)e->d * e->d;
  HIP_OK(dalloc(&e->z, Ml * e->D));
  HIP_OK(dalloc(&e->vz, Ml * e->D));
  HIP_OK(dalloc(&e->theta, Ml * e->P));
  HIP_OK(dalloc(&e->vtheta, Ml * e->P));
  HIP_OK(dalloc(&e->baseline, Ml));
  HIP_OK(dalloc(&e->baseline2, Ml));
  HIP_OK(dalloc(&e->scores, Ml * dd));
  HIP_OK(dalloc(&e->probs, Ml * dd));
  if (e->d <= 112) HIP_OK(dalloc(&e->eas, Ml * dd));
  HIP_OK(dalloc(&e->thr, Ml * dd));
  HIP_OK(dalloc(&e->w_lik, Ml * dd));
  if (e->d > 112) {
    if (hipMalloc((void**)&e->acyc_big, acyc_big_elems(e->Mloc, e->d, e->Sa) * 4) != hipSuccess) {
      e->acyc_big = nullptr;
      (void)hipGetLastError();
      return fail("n_vars > 112: the matrix powers of the acyclicity term go through global memory: hipMalloc of " +
                  std::to_string(acyc_big_elems(e->Mloc, e->d, e->Sa) * 4 >> 20) + " MiB failed");
    }
  } else {
    HIP_OK(dalloc(&e->acyc_part, Ml * e->acyc_nblk * dd));
  }
  HIP_OK(dalloc(&e->w_acyc, Ml * dd));
  {
    const bool score_lik = c.likelihood == DIBS_LIK_BGE && c.grad_estimator_z == DIBS_EST_SCORE;
    const int ldz = tail_ldz(e->d, e->k, e->S, score_lik, LDS_LIMIT - 2048);
    if (tail_lds_bytes(e->d, ldz, e->S, e->W, score_lik, 0) > LDS_LIMIT - 2048) HIP_OK(dalloc(&e->w_tot, Ml * dd));
  }
  HIP_OK(dalloc(&e->logprobs_z, Ml * e->S));
  HIP_OK(dalloc(&e->logprobs_th, Ml * e->S));
  HIP_OK(dalloc(&e->pack, (size_t)e->M * e->E));
  // (k_phi_gemm reads whole 128-row x 32-column tiles without bounds checks: rows padded to a multiple of 128, one more tile row of slack)
  const size_t kpad = (((Ml + 127) / 128) * 128 - Ml) * e->M + 64;
  HIP_OK(dalloc(&e->kz, Ml * e->M + kpad));
  if (c.joint) HIP_OK(dalloc(&e->kt, Ml * e->M + kpad));
  if (c.joint) HIP_OK(dalloc(&e->ksum, Ml * e->M + kpad));
  if (e->M >= e->tune.kmat_tiled_min) {  // (the same rule on every rank: it depends on the global particle count only)
    // room for up to 32 pieces per pair, less for many particles (<= 512 MiB); 1 = no buffer, every unit holds whole distances
    size_t ns = ((size_t)512 << 20) / (Ml * e->M * 8);
    ns = ns > 32 ? 32 : (ns < 1 ? 1 : ns);
    if (!kmat_tile_addressable((size_t)2 * e->M, e->E > e->Ev ? e->E : e->Ev, 0, 0)) ns = 0;  // (32-bit row offsets in the tile kernel)
    if (ns > 1 && hipMalloc((void**)&e->kpart, ns * Ml * e->M * 8) != hipSuccess) {
      e->kpart = nullptr;
      (void)hipGetLastError();
      ns = 1;
    }
    e->kmat_ns_max = (int)ns;
    if (ns > 1 && e->Mloc == e->M) {
      const size_t nta = (e->M + KT_T - 1) / KT_T;
      HIP_OK(dalloc(&e->kmat_ctr, nta * (nta + 1) / 2));
    }
  }
  HIP_OK(dalloc(&e->phi_z, Ml * e->D));
  HIP_OK(dalloc(&e->phi_th, Ml * e->P));
  HIP_OK(dalloc(&e->counters, (size_t)DIBS_N_COUNTERS));
  if (c.likelihood == DIBS_LIK_BGE) {
    HIP_OK(dalloc(&e->masks, Ml * e->S * e->d * e->W));
    HIP_OK(dalloc(&e->node_scores, Ml * e->S * e->d));
    e->bq.cap = (uint32_t)(Ml * e->S * e->d);
    HIP_OK(dalloc(&e->bq.list, (size_t)BGE_NQ * e->bq.cap * bge_entry_u4(e->W)));  // (one list per size tier, each sized for every problem)
    HIP_OK(dalloc(&e->bq.counts, (size_t)16));
    if (c.grad_estimator_z == DIBS_EST_REPARAM) {
      HIP_OK(dalloc(&e->soft_ds, Ml * e->S * dd));
      if (e->d > 128) {
        const size_t prob = Ml * e->S;
        e->soft_blocks = (int)(prob < 1024 ? prob : 1024);
        HIP_OK(dalloc(&e->soft_tri, (size_t)e->soft_blocks * 4 * 2 * bge_soft_tri(e->d)));
      }
    }
  }
  if (c.joint) {
    if (joint_alloc(&e->jw, e->Mloc, e->d, e->N, e->S) != 0) return fail("joint work buffers: hipMalloc failed");
  }
  hipDeviceSynchronize();  // the zero fills above ran on the null stream; the engine's own stream does not wait for it
  return 0;
}


extern "C" int dibs_engine_create(const dibs_config* cfg, void* stream, dibs_engine** out) {
  if (!cfg || !out) return fail("null argument");
  *out = nullptr;
  const dibs_config& c = *cfg;
  if (c.abi_version != DIBS_ABI_VERSION) return fail("dibs_config.abi_version mismatch");
  if (c.n_vars < 2 || c.n_vars > 256) return fail("n_vars must be in [2, 256]");
  // 113 .. 256 variables: the LDS-resident kernels give way to the global-memory paths (kernels_acyc_big.h, k_backproject_big, chunked
  // k_edge_scores, k_bge_chol_wide)
  if (c.n_vars > 112) {
    // the joint models run on their general paths there: LinearGaussian on the Gram-matrix kernels, DenseNonlinearGaussian on
    // kernels_nn_generic.h; beyond the LDS capacity (two n_vars x n_vars float operands: 141, one: 198) the blocks keep them in global
    // scratch (round 5: the limits of 141 / 198 variables are gone); soft-graph BGe has its own limit below
  }
  if (c.n_dim < 1) return fail("n_dim must be >= 1");
  if (c.n_particles < 1 || c.n_grad_mc_samples < 1 || c.n_acyclicity_mc_samples < 1) return fail("sizes must be >= 1");
  if (c.n_ranks < 1 || c.rank < 0 || c.rank >= c.n_ranks) return fail("bad rank / n_ranks");
  if (c.n_particles % c.n_ranks) return fail("n_particles must be divisible by n_ranks");
  if (c.grad_estimator_z != DIBS_EST_SCORE && c.grad_estimator_z != DIBS_EST_REPARAM)
    return fail("Unknown gradient estimator");  // dibs.py:318 (ValueError)
  if (c.optimizer != DIBS_OPT_GD && c.optimizer != DIBS_OPT_RMSPROP) return fail("unknown optimizer");  // svgd.py:122
  if (c.likelihood < 0 || c.likelihood > 2) return fail("unknown likelihood model");
  if (c.graph_prior < 0 || c.graph_prior > 2) return fail("unknown graph prior");
  if (!c.joint && c.likelihood != DIBS_LIK_BGE)
    return fail("MarginalDiBS needs a marginal likelihood (BGe)");
  if (c.joint && c.likelihood == DIBS_LIK_BGE)
    return fail("JointDiBS + BGe is not constructible (BGe has no parameters; linearGaussian.py:53-54)");
  if (c.likelihood == DIBS_LIK_BGE && c.grad_estimator_z == DIBS_EST_REPARAM) {
    // soft-graph BGe (kernels_bge_soft.h): one or two matrix rows per lane, packed factor + inverse columns per wave in LDS; beyond 128
    // variables four rows per lane and the triangles in global scratch
    if (c.n_vars <= 128 && bge_soft_waves(c.n_vars, false) < 1)
      return fail("BGe + reparam estimator: n_vars too large for the device kernel");
  }
  if (c.likelihood == DIBS_LIK_BGE && c.grad_estimator_z == DIBS_EST_SCORE &&
      bge_sample_lds_bytes(c.n_vars, c.n_grad_mc_samples, (c.n_vars + 63) / 64) > LDS_LIMIT - 2048)
    return fail("BGe: n_grad_mc_samples too large (the parent sets of one node's samples are staged in LDS)");
  if (c.likelihood == DIBS_LIK_DENSENN) {
    // one hidden layer of <= 64 units with <= 128 observations runs on the MFMA kernels of kernels_nn.h, every other stack on the
    // general path of kernels_nn_generic.h
    if (c.nn_n_hidden < 1 || c.nn_n_hidden > DIBS_MAX_HIDDEN_LAYERS) return fail("DenseNonlinearGaussian: 1 to " + std::to_string(DIBS_MAX_HIDDEN_LAYERS) + " hidden layers");
    for (int l = 0; l < c.nn_n_hidden; ++l)
      if (c.nn_hidden[l] < 1) return fail("DenseNonlinearGaussian: hidden widths must be >= 1");
    if (c.nn_activation < 0 || c.nn_activation > 3) return fail("Invalid activation function");  // nonlinearGaussian.py:61 (KeyError)
  }
  if (c.graph_prior == DIBS_PRIOR_ER) {
    const double p = c.graph_prior_edges_per_node * c.n_vars / ((c.n_vars * (c.n_vars - 1)) / 2.0);
    if (!(p > 0.0 && p < 1.0)) return fail("Erdos-Renyi prior: edge probability must be in (0, 1)");
  }
  {
    if ((size_t)2 * 4 * c.n_particles * 4 + 4096 > LDS_LIMIT) return fail("n_particles too large (kernel rows must fit in LDS)");
    // k_particle_grad keeps a particle's score-space gradient, its Z and (score estimator) the per-sample weights in LDS; beyond that
    // size W goes through global memory (k_backproject_big) and only the per-sample weights have to fit
    const bool score_lik = c.likelihood == DIBS_LIK_BGE && c.grad_estimator_z == DIBS_EST_SCORE;
    if (tail_lds_bytes(c.n_vars, 0, c.n_grad_mc_samples, (c.n_vars + 63) / 64, score_lik, 0) > LDS_LIMIT - 2048 ||
        backproject_big_lds(c.n_vars) > LDS_LIMIT - 2048)
      return fail("n_grad_mc_samples (or n_vars) too large: a particle's sample weights do not fit in LDS");
  }
  // (LinearGaussian: x beyond the LDS capacity takes the Gram-matrix path; DenseNonlinearGaussian the general path)
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail("no HIP device");
  HIP_OK(hipSetDevice(c.device_id));
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, c.device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(std::string("libdibs_hip is built for gfx950 only; device is ") + prop.gcnArchName);

  dibs_engine* e = new dibs_engine();  // value-initialised: every POD member starts at zero
  g_live_engines.fetch_add(1);
  if (engine_alloc(e, c, stream)) {
    const std::string msg = g_err;
    dibs_engine_destroy(e);  // frees whatever had been allocated (stream, events, buffers)
    g_err = msg;
    return 1;
  }
  *out = e;
  return 0;
}

extern "C" int dibs_engine_destroy(dibs_engine* e) {
  if (!e) return 0;
  g_live_engines.fetch_sub(1);
  hipSetDevice(e->cfg.device_id);
  if (e->stream) hipStreamSynchronize(e->stream);
  if (e->stream2) hipStreamSynchronize(e->stream2);
  void* ptrs[] = {e->z, e->vz, e->theta, e->vtheta, e->baseline, e->baseline2, e->scores, e->probs, e->eas, e->thr, e->w_lik, e->acyc_part, e->w_acyc,
                  e->logprobs_z, e->logprobs_th, e->pack, e->kz, e->kt, e->phi_z, e->phi_th, e->counters, e->masks,
                  e->node_scores, e->x, e->mask, e->bq.list, e->bq.counts, e->soft_ds, e->acyc_big, e->w_tot, e->join_flag, e->fork_flag, e->carry_bak, e->soft_tri, e->ksum, e->kpart, e->kmat_ctr};
  for (void* p : ptrs)
    if (p) hipFree(p);
  joint_free(&e->jw);
  dibs_engine_comm_destroy(e);
  if (e->stream2) hipStreamDestroy(e->stream2);
  if (e->ev_fork) hipEventDestroy(e->ev_fork);
  if (e->ev_join) hipEventDestroy(e->ev_join);
  if (e->ev_k0) hipEventDestroy(e->ev_k0);
  if (e->ev_k1) hipEventDestroy(e->ev_k1);
  if (e->join_err) hipHostFree(e->join_err);
  if (e->ev0) hipEventDestroy(e->ev0);
  if (e->ev1) hipEventDestroy(e->ev1);
  for (auto& pe : e->pending) {
    hipEventDestroy(pe.second.first);
    hipEventDestroy(pe.second.second);
  }
  if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
  delete e;
  return 0;
}

// in-place inverse and log-determinant of an SPD matrix (Cholesky, double)
static bool spd_inverse_logdet(std::vector<double>& a, int n, double* logdet) {
  std::vector<double> L((size_t)n * n, 0.0), Li((size_t)n * n, 0.0);
  double ld = 0;
  for (int j = 0; j < n; ++j) {
    double s = a[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) s -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    if (!(s > 0.0)) return false;
    const double dj = sqrt(s);
    ld += 2.0 * log(dj);
    L[(size_t)j * n + j] = dj;
    for (int i = j + 1; i < n; ++i) {
      double t = a[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) t -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      L[(size_t)i * n + j] = t / dj;
    }
  }
  for (int c = 0; c < n; ++c) {  // Li = L^-1 (lower), column by column
    Li[(size_t)c * n + c] = 1.0 / L[(size_t)c * n + c];
    for (int i = c + 1; i < n; ++i) {
      double t = 0;
      for (int k = c; k < i; ++k) t -= L[(size_t)i * n + k] * Li[(size_t)k * n + c];
      Li[(size_t)i * n + c] = t / L[(size_t)i * n + i];
    }
  }
  for (int i = 0; i < n; ++i)  // A^-1 = Li^T Li
    for (int j = 0; j <= i; ++j) {
      double t = 0;
      for (int k = i; k < n; ++k) t += Li[(size_t)k * n + i] * Li[(size_t)k * n + j];
      a[(size_t)i * n + j] = a[(size_t)j * n + i] = t;
    }
  *logdet = ld;
  return true;
}

static int bge_prepare(BgeStats* st, const dibs_config& cfg, int d, int N, const float* x, const int32_t* mask, const float* mean_obs) {
  const double amu = cfg.bge_alpha_mu;
  st->release();
  st->alpha_lambd = cfg.bge_alpha_lambd > 0 ? cfg.bge_alpha_lambd : d + 2.0;
  if (!(st->alpha_lambd > d + 1)) return fail("BGe: alpha_lambd must be > n_vars + 1");  // linearGaussian.py:47
  const double small_t = amu * (st->alpha_lambd - d - 1) / (amu + 1);
  st->alpha_mu = amu;
  st->log_t = log(small_t);
  bool any = false;
  if (mask)
    for (int64_t i = 0; i < (int64_t)N * d; ++i) any |= mask[i] != 0;
  st->n_mats = any ? d : 1;
  const int n_mats = st->n_mats, dp = d + 1;
  std::vector<float> R((size_t)n_mats * d * d), Rp((size_t)n_mats * dp * dp, 0.f), Qp((size_t)n_mats * dp * dp, 0.f);
  std::vector<double> Nj(d), gam((size_t)d * (d + 1)), xb(d), ldR(n_mats), Rd((size_t)d * d);
  for (int jm = 0; jm < n_mats; ++jm) {
    double Nn = 0;
    for (int n = 0; n < N; ++n) Nn += (any && mask[(int64_t)n * d + jm]) ? 0.0 : 1.0;
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int n = 0; n < N; ++n)
        if (!(any && mask[(int64_t)n * d + jm])) s += (double)x[(int64_t)n * d + a];
      xb[a] = Nn > 0 ? s / Nn : 0.0;
    }
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) {
        double s = 0;
        for (int n = 0; n < N; ++n)
          if (!(any && mask[(int64_t)n * d + jm]))
            s += ((double)x[(int64_t)n * d + a] - xb[a]) * ((double)x[(int64_t)n * d + b] - xb[b]);
        const double ma = mean_obs ? (double)mean_obs[a] : 0.0, mb = mean_obs ? (double)mean_obs[b] : 0.0;
        const double v = (a == b ? small_t : 0.0) + s + (Nn * amu / (Nn + amu)) * (xb[a] - ma) * (xb[b] - mb);
        Rd[(size_t)a * d + b] = v;
        R[(size_t)jm * d * d + a * d + b] = (float)v;
        Rp[(size_t)jm * dp * dp + (size_t)a * dp + b] = (float)v;
      }
    if (!spd_inverse_logdet(Rd, d, &ldR[jm])) return fail("BGe: R is not positive definite");
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) Qp[(size_t)jm * dp * dp + (size_t)a * dp + b] = (float)Rd[(size_t)a * d + b];
    if (any) Nj[jm] = Nn;
    else
      for (int j = 0; j < d; ++j) Nj[j] = Nn;
  }
  for (int j = 0; j < d; ++j)
    for (int l = 0; l <= d; ++l) {
      const double Nn = Nj[j], al = st->alpha_lambd;
      gam[(size_t)j * (d + 1) + l] = 0.5 * (log(amu) - log(Nn + amu)) + lgamma(0.5 * (Nn + al - d + l + 1)) -
                                     lgamma(0.5 * (al - d + l + 1)) - 0.5 * Nn * log(M_PI) +
                                     0.5 * (al - d + 2 * l + 1) * log(small_t);
    }
  HIP_OK(dalloc(&st->R, R.size()));
  // R and Q = R^-1 in one allocation: the factorisation kernel addresses a problem's matrix as a 32-bit float offset from Rp, and two
  // separate hipMalloc blocks can lie more than 2^31 floats apart on a 288 GB device (intermittent memory faults with interventions or
  // d > 80, where the matrices are not LDS-resident; found by tests/tools/gpu_fuzz.py)
  HIP_OK(dalloc(&st->Rp, Rp.size() + Qp.size()));
  st->Qp = st->Rp + Rp.size();
  HIP_OK(dalloc(&st->gam, gam.size()));
  HIP_OK(dalloc(&st->Nj, Nj.size()));
  HIP_OK(dalloc(&st->ldR, ldR.size()));
  HIP_OK(hipMemcpy(st->R, R.data(), R.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(st->Rp, Rp.data(), Rp.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(st->Qp, Qp.data(), Qp.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(st->gam, gam.data(), gam.size() * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(st->Nj, Nj.data(), Nj.size() * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(st->ldR, ldR.data(), ldR.size() * 8, hipMemcpyHostToDevice));
  return 0;
}

extern "C" int dibs_engine_set_data(dibs_engine* e, const float* x, const int32_t* interv_mask, const float* bge_mean_obs) {
  if (!e || !x) return fail("null argument");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  HIP_OK(hipStreamSynchronize(e->stream));
  e->score_cache.valid = false;  // (the BGe prior mean travels with the data)
  e->has_data = false;  // (a failure below leaves the engine without data: the next step reports it instead of reading freed statistics)
  const size_t n = (size_t)e->N * e->d;
  if (e->x) hipFree(e->x);
  if (e->mask) hipFree(e->mask);
  e->x = nullptr;
  e->mask = nullptr;
  HIP_OK(dalloc(&e->x, n));
  HIP_OK(dalloc(&e->mask, n));
  HIP_OK(hipMemcpy(e->x, x, n * 4, hipMemcpyHostToDevice));
  if (interv_mask) HIP_OK(hipMemcpy(e->mask, interv_mask, n * 4, hipMemcpyHostToDevice));
  if (e->cfg.likelihood == DIBS_LIK_BGE) {
    e->has_mean_obs = bge_mean_obs != nullptr;
    if (bge_mean_obs) e->mean_obs.assign(bge_mean_obs, bge_mean_obs + e->d);
    if (bge_prepare(&e->bge, e->cfg, e->d, e->N, x, interv_mask, bge_mean_obs)) return 1;
  } else {
    if (joint_set_data(&e->jw, x, interv_mask, e->N, e->d)) return fail("joint_set_data failed");
    if (e->cfg.likelihood == DIBS_LIK_LINGAUSS && !joint_lin_fast_path(e->d, e->N, e->tune.lin_gram) && joint_lin_set_gram(&e->jw, x, interv_mask, e->N, e->d))
      return fail("LinearGaussian: Gram matrices: hipMalloc failed");
  }
  e->has_data = true;
  return 0;
}

extern "C" int dibs_engine_init_particles(dibs_engine* e, const uint32_t key[2]) {
  if (!e || !key) return fail("null argument");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  const int L = e->cfg.rng_layout;
  const Key2 k0{key[0], key[1]};
  e->key = rng_split_row(k0, 2, 0, L);                     // key, subk = split(key)            svgd.py:294
  const Key2 subk = rng_split_row(k0, 2, 1, L);
  const Key2 ikey = rng_split_row(subk, 2, 0, L);          // key, subk = split(key)            svgd.py:145 / :509
  const Key2 isub = rng_split_row(subk, 2, 1, L);
  const uint64_t ntot = (uint64_t)e->M * e->D, nloc = (uint64_t)e->Mloc * e->D;
  hipLaunchKernelGGL(k_init_z, dim3((unsigned)((nloc + 255) / 256)), dim3(256), 0, e->stream, e->z, isub, ntot,
                     (uint64_t)e->m0 * e->D, nloc, e->sigz, L);
  if (e->cfg.joint) {
    const Key2 tsub = rng_split_row(ikey, 2, 1, L);        // key, subk = split(key); sample_parameters(key=subk)  svgd.py:512-513
    if (e->cfg.likelihood == DIBS_LIK_LINGAUSS) {
      const uint64_t tt = (uint64_t)e->M * e->P, tl = (uint64_t)e->Mloc * e->P;
      hipLaunchKernelGGL(k_init_theta_lin, dim3((unsigned)((tl + 255) / 256)), dim3(256), 0, e->stream, e->theta, tsub, tt,
                         (uint64_t)e->m0 * e->P, tl, (float)e->cfg.lin_mean_edge, (float)e->cfg.lin_sig_edge,
                         (float)e->cfg.lin_min_edge, L);
    } else if (e->cfg.likelihood == DIBS_LIK_DENSENN) {
      const NNParams np_ = nn_params(e->cfg);
      joint_nn_init_theta(e->theta, (size_t)e->P, tsub, e->m0, e->Mloc, e->M, e->d, np_, L, e->stream);
    } else {
      return fail("sample_parameters not implemented for this likelihood");
    }
  }
  e->kmat_ext = false;  // (a kernel slab computed by dibs_engine_kmat_values belonged to the particles that were just replaced)
  e->vals_fresh = false;
  HIP_OK(hipMemsetAsync(e->vz, 0, (size_t)e->Mloc * e->D * 4, e->stream));
  if (e->P) HIP_OK(hipMemsetAsync(e->vtheta, 0, (size_t)e->Mloc * e->P * 4, e->stream));
  HIP_OK(hipMemsetAsync(e->baseline, 0, (size_t)e->Mloc * 4, e->stream));
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" int dibs_engine_set_state(dibs_engine* e, const float* z, const float* v_z, const float* theta,
                                     const float* v_theta, const uint32_t* key, const float* baseline) {
  if (!e) return fail("null engine");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  HIP_OK(hipStreamSynchronize(e->stream));
  const size_t nz = (size_t)e->Mloc * e->D * 4, nt = (size_t)e->Mloc * e->P * 4;
  if (z || theta) e->vals_fresh = false;
  if (z || theta) e->kmat_ext = false;  // (an externally computed kernel slab belonged to the old values: phase B computes its own unless
                                        //  dibs_engine_kmat_values is called again for the new ones)
  if (z) HIP_OK(hipMemcpy(e->z, z, nz, hipMemcpyHostToDevice));
  if (v_z) HIP_OK(hipMemcpy(e->vz, v_z, nz, hipMemcpyHostToDevice));
  if (theta && nt) HIP_OK(hipMemcpy(e->theta, theta, nt, hipMemcpyHostToDevice));
  if (v_theta && nt) HIP_OK(hipMemcpy(e->vtheta, v_theta, nt, hipMemcpyHostToDevice));
  if (key) e->key = Key2{key[0], key[1]};
  if (baseline) HIP_OK(hipMemcpy(e->baseline, baseline, (size_t)e->Mloc * 4, hipMemcpyHostToDevice));
  return 0;
}

extern "C" int dibs_engine_get_state(dibs_engine* e, float* z, float* v_z, float* theta, float* v_theta, uint32_t* key,
                                     float* baseline) {
  if (!e) return fail("null engine");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  HIP_OK(hipStreamSynchronize(e->stream));
  const size_t nz = (size_t)e->Mloc * e->D * 4, nt = (size_t)e->Mloc * e->P * 4;
  if (z) HIP_OK(hipMemcpy(z, e->z, nz, hipMemcpyDeviceToHost));
  if (v_z) HIP_OK(hipMemcpy(v_z, e->vz, nz, hipMemcpyDeviceToHost));
  if (theta && nt) HIP_OK(hipMemcpy(theta, e->theta, nt, hipMemcpyDeviceToHost));
  if (v_theta && nt) HIP_OK(hipMemcpy(v_theta, e->vtheta, nt, hipMemcpyDeviceToHost));
  if (key) {
    key[0] = e->key.a;
    key[1] = e->key.b;
  }
  if (baseline) HIP_OK(hipMemcpy(baseline, e->baseline, (size_t)e->Mloc * 4, hipMemcpyDeviceToHost));
  return 0;
}

// kernels that may need more than the default 64 KiB of dynamic LDS (see launch.h).  One attribute call per (device, kernel) and size
// increase, not one per launch; the table is shared by every engine of the process, so it is keyed by device and guarded by a mutex
// (ctypes releases the GIL: two engines may be stepped from two host threads).
int dibs_cu_count() {
  static std::mutex mu;
  static std::map<int, int> cus;
  int dev = 0;
  hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  int& n = cus[dev];
  if (n == 0) {
    hipDeviceProp_t prop;
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n;
}
void dibs_allow_lds(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> granted;
  if (bytes <= 48 * 1024) return;
  int dev = 0;
  hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  size_t& g = granted[{dev, kernel}];
  if (bytes > g) {
    hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    g = bytes;
  }
}
template <typename K>
static void allow_lds(K kernel, size_t bytes) { dibs_allow_lds((const void*)kernel, bytes); }

// ---- profiling helpers -----------------------------------------------------------------------
struct KTimer {
  dibs_engine* e;
  int id;
  hipEvent_t a, b;
  hipStream_t st;
  KTimer(dibs_engine* e_, int id_, hipStream_t st_ = nullptr) : e(e_), id(id_), a(nullptr), b(nullptr), st(st_ ? st_ : e_->stream) {
    if (e->profiling) {
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipEventRecord(a, st);
    }
  }
  ~KTimer() {
    if (e->profiling) {
      hipEventRecord(b, st);
      e->pending.push_back({id, {a, b}});
    }
  }
};

// the matrix-power launch of the acyclicity term; while profiling, a single-kernel launch is stamped by the launch itself (kernel start /
// end, what rocprofv3 reports) instead of an event pair around it, which on the second stream also times ~7 us of dispatch latency
static void acyc_power_timed(dibs_engine* e, AcycLaunch al, hipStream_t st) {
  if (e->profiling && acyc_power_takes_events(al)) {
    hipEventCreate(&al.ev_start);
    hipEventCreate(&al.ev_stop);
    acyc_launch_power(al);
    e->pending.push_back({DIBS_K_ACYC, {al.ev_start, al.ev_stop}});
    return;
  }
  KTimer tm(e, DIBS_K_ACYC, st);
  acyc_launch_power(al);
}

static void drain_timers(dibs_engine* e) {
  for (auto& pe : e->pending) {
    hipEventSynchronize(pe.second.second);
    float ms = 0.f;
    hipEventElapsedTime(&ms, pe.second.first, pe.second.second);
    e->t_ms[pe.first] += ms;
    e->t_n[pe.first] += 1;
    hipEventDestroy(pe.second.first);
    hipEventDestroy(pe.second.second);
  }
  e->pending.clear();
}

// device buffer that frees itself (error paths)
template <typename T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() {
    if (p) hipFree(p);
  }
  hipError_t alloc(size_t n) { return dalloc(&p, n); }
};

// One kernel-matrix algorithm per global particle count, on every rank and at every launch site: from e->tune.kmat_tiled_min particles the tiled
// kernel (whose entries do not depend on how the work was cut, kernels_kmat.h), below it the direct one.
static bool kmat_tiled_on(const dibs_engine* e) { return e->kmat_ns_max > 0 && e->M >= e->tune.kmat_tiled_min; }
// rows of all M particles at x + m * stride + off (len floats); this engine's slab [Mloc][M] (symmetric when it holds every particle)
static void kmat_launch_tiled(dibs_engine* e, hipStream_t st, const float* x, size_t stride, size_t off, size_t len, float* kout, float scale, float h,
                              const float* kadd, float* ksum) {
  const int sym = e->Mloc == e->M;
  // many particles: 64 x 64 tiles (half the bytes per pair; entries bit-identical to the 32 x 32 kernel's) -- from 512 particles, where
  // there are enough of them for every CU (DibsTuning::kmat_t64_min)
  if (e->M >= e->tune.kmat_t64_min && kmat_tile64_ok(x, stride, off, len)) {
    const int nta = (e->Mloc + KT2_T - 1) / KT2_T, ntb = (e->M + KT2_T - 1) / KT2_T, tiles = kmat_tile_count(nta, ntb, sym);
    const int nchunk = kmat_nchunk64((int)len), ns = kmat_pick_nsplit(tiles, nchunk, e->kmat_ns_max), cps = (nchunk + ns - 1) / ns;
    const KmatTile kt{x, stride, off, (int)len, e->kpart, e->m0, e->Mloc, e->M, nchunk, nta, ntb, sym, ns, cps, scale, h, kout, kadd, ksum, nullptr};
    dibs_allow_lds((const void*)k_kmat_tile64, kmat_tile64_lds_bytes());
    const int units = tiles * ns;
    hipLaunchKernelGGL(k_kmat_tile64, dim3((unsigned)(units < 256 ? units : 256)), dim3(KT2_NT), kmat_tile64_lds_bytes(), st, kt);
    if (ns > 1)
      hipLaunchKernelGGL(k_kmat_finish, dim3(e->Mloc), dim3(256), 0, st, (const double*)e->kpart, ns, e->Mloc, e->M, sym, scale, h, kout, kadd, ksum, KT2_T);
    return;
  }
  const int nta = (e->Mloc + KT_T - 1) / KT_T, ntb = (e->M + KT_T - 1) / KT_T, tiles = kmat_tile_count(nta, ntb, sym);
  const int nchunk = kmat_nchunk((int)len), ns = kmat_pick_nsplit(tiles, nchunk, e->kmat_ns_max), cps = (nchunk + ns - 1) / ns;
  const KmatTile kt{x, stride, off, (int)len, e->kpart, e->m0, e->Mloc, e->M, nchunk, nta, ntb, sym, ns, cps, scale, h, kout, kadd, ksum, nullptr};
  dibs_allow_lds((const void*)k_kmat_tile, kmat_tile_lds_bytes());
  // (persistent blocks, one per CU by their registers, looping over the units with the next step's rows prefetched)
  const int units = tiles * ns;
  hipLaunchKernelGGL(k_kmat_tile, dim3((unsigned)(units < 256 ? units : 256)), dim3(KT_NT), kmat_tile_lds_bytes(), st, kt);
  if (ns > 1) hipLaunchKernelGGL(k_kmat_finish, dim3(e->Mloc), dim3(256), 0, st, (const double*)e->kpart, ns, e->Mloc, e->M, sym, scale, h, kout, kadd, ksum, KT_T);
}

// ---- one SVGD step, split at the exchange point ----------------------------------------------
// carry keys: the loop-carry key advances by one split(key, M+1) per estimator batch (svgd.py:245, 251 / 695, 699, 703);
// the host walks the chain (row 0), kernels derive row 1 + m.
static Key2 next_carry(const dibs_engine* e, Key2 k) { return rng_split_row(k, (uint32_t)e->M + 1u, 0u, e->cfg.rng_layout); }

// where phase A writes its per-particle rows (indexed by GLOBAL particle id): packed rows [z | grad_z | theta | grad_theta] (stride E, the
// single-rank buffer and the one-collective protocol) or gradient rows [grad_z | grad_theta] (stride Ev, the overlapped protocol, where
// the values travel separately).
struct RowTarget {
  float* base;
  size_t stride, gz_off, th_off, gth_off;
  int copy_vals;
};
static RowTarget packed_rows(const dibs_engine* e, float* pack) {
  return RowTarget{pack, (size_t)e->E, (size_t)e->D, (size_t)(2 * e->D), (size_t)(2 * e->D + e->P), 1};
}

// explicit per-particle keys of one evaluation (dibs_engine_eval_gradients): device arrays Key2[Mloc], one per estimator family
struct StepKeys {
  const Key2 *theta, *lik, *prior;
};
enum { TERMS_LIK = 1, TERMS_PRIOR = 2, TERMS_ALL = 3 };
// (rng.h: rng_explicit_row) the carry slot carries the address of the key of GLOBAL particle 0
static Key2 key_array_as_carry(const Key2* local, int m0) {
  const uint64_t p = (uint64_t)(uintptr_t)(local - m0);
  return Key2{(uint32_t)p, (uint32_t)(p >> 32)};
}

// xk == null: the step of the SVGD loop (keys from the loop-carry key, which advances).  xk != null: the same kernels with the caller's
// per-particle keys, the loop-carry key untouched; `terms` selects the likelihood part (estimators + their share of grad_z), the prior
// part (acyclicity, Gaussian and graph prior), or both.
static int step_local(dibs_engine* e, int t, const RowTarget& rt, const StepKeys* xk = nullptr, int terms = TERMS_ALL, const float* zero_w = nullptr) {
  float* const pack = rt.base;
  const dibs_config& c = e->cfg;
  const float alpha = (float)(c.alpha_linear * t), beta = (float)(c.beta_linear * t);
  const int L = c.rng_layout;
  Key2 carry_theta{0, 0}, carry_lik, carry_prior;
  int Mg = e->M;  // particle count of the key derivation (row 1 + m of split(carry, M + 1)); -1: explicit keys
  if (xk) {
    Mg = -1;
    carry_theta = key_array_as_carry(xk->theta, e->m0);
    carry_lik = key_array_as_carry(xk->lik, e->m0);
    carry_prior = key_array_as_carry(xk->prior, e->m0);
  } else {
    Key2 carry = e->key;
    if (c.joint) {
      carry_theta = carry;
      carry = next_carry(e, carry);
    }
    carry_lik = carry;
    carry = next_carry(e, carry);
    carry_prior = carry;
    carry = next_carry(e, carry);
    e->key = carry;
  }
  const bool do_lik = (terms & TERMS_LIK) != 0, do_prior = (terms & TERMS_PRIOR) != 0;

  e->kmat_early = false;
  e->kmat_fused = false;
  // While per-kernel timing is on (set_profiling(1)) the main stream joins right away, so that every duration is a kernel alone on the
  // GPU -- but the launch still goes to the second stream: with that (high-priority) queue in existence the same kernel takes 104 us
  // on the main stream and 96 us on its own.
  // (Until round 4 a small acyclicity launch -- <= 512 blocks: config 2, or a rank of a sharded headline run -- stayed on the main stream: the
  //  fork / join events cost 6 + 6 us of the critical path, more than such a launch could hide.  With the fork as the edge kernel's completion
  //  signal and the join polled inside k_particle_grad the second stream pays at every size: config 2 18 460 -> 20 440 steps/s, a rank of
  //  a 4- / 8-way headline run 101.0 -> 91.4 / 85.8 -> 78.0 us per step.)
  const bool fork = do_prior && do_lik && e->stream2 != nullptr, join_now = e->profiling && !e->profiling_concurrent;
  // the join inside k_particle_grad (tail_join_wait, agent-scope loads of a flag word the second stream's last kernel stores) instead of an
  // event wait in front of it: -7 us per step.  The polling blocks hold their CUs while the second stream still has kernels to place, so
  // the flag is used only while they cannot fill the machine (<= 128 particles: one block each on half of the CUs) and the engine's flags
  // are on for this chunk (flags_now: latch_flags).  The wait is bounded (join_err; a chunk that saw a time-out is run again on events:
  // run_chunk_guarded).  Per-kernel timing always uses the event.
  const bool flag_join = fork && !join_now && e->flags_now && e->Mloc <= 128;
  // where this step's kernel matrices come from (single rank): the joint models and many particles put them on the second stream behind the
  // acyclicity chain (kmat_on_s2, see below); otherwise the latent matrix rides inside k_bge_sample
  const bool kmat_on_s2 = c.joint || (long)e->M * e->D > 4L * e->S * e->d * e->d;
  const bool kmat_early_now = fork && !xk && kmat_on_s2 && e->Mloc == e->M && !e->kmat_ext;
  // marginal models, single rank, 128+ particles: the latent matrix as tile units riding in the k_particle_grad launch (TailArgs::kt)
  const bool tile_in_grad = !c.joint && !xk && !kmat_early_now && !e->kmat_ext && e->Mloc == e->M && e->kmat_ns_max > 1 && e->kmat_ctr != nullptr &&
                            e->M >= e->tune.kmat_tiled_min && e->Mloc < 256 && e->w_tot == nullptr && !e->tune.no_kmat_fuse && !e->tune.no_kmat_grad;
  // fork without an event (marginal models): k_edge_scores_p stores what
  // the second stream reads (scores, exp(-alpha s)) at agent scope, every block counts itself and the last one publishes a sequence number;
  // one polling wave (k_wait_flag) heads the second stream's chain.  The completion signal cost the NEXT kernel of the main stream 4.7 us
  // (edge -> sample gap; 1.0 us between plain launches).  With the two chains starting together the acyclicity stream must not have
  // priority over the sampling kernel (it took the machine: sampling 130 us, the factorisation then alone for 33): the stream is created
  // with the LOWEST priority.  bench.py, same box: event fork 5 193-5 217 steps/s; flag fork with greatest / normal / lowest priority
  // 5 218-5 226 / 5 296 / 5 341; config 2 20 560 -> 22 200.  Joint models keep the event (config 3: 2 345 vs 2 311 with the flag).
  const bool edge_p = e->d <= 64 && e->k <= 64 && e->edge_kc >= e->k && e->ldk <= 128;  // one 16-wave block per particle (k_edge_scores_p)
  const bool flag_fork = flag_join && !c.joint && !e->profiling && e->fork_flag != nullptr && edge_p;
  // BGe with the score estimator: the flag is published by the FIRST BLOCK OF k_bge_sample instead (it starts when the edge kernel has ended and
  // released its plain stores): no agent-scope stores and no counting in the edge kernel
  const bool fork_pub_in_sample = flag_fork && do_lik && c.likelihood == DIBS_LIK_BGE && c.grad_estimator_z == DIBS_EST_SCORE;
  auto launch_edge = [&](hipStream_t st, hipEvent_t stop_ev) {
    KTimer tm(e, DIBS_K_EDGE, st);
    const size_t lds = (size_t)2 * e->dpad * e->ldk * 4;
    if (edge_p) {
      allow_lds(k_edge_scores_p, lds);
      unsigned int* const none = nullptr;
      if (flag_fork && fork_pub_in_sample) {
        ++e->fork_seq;
        hipLaunchKernelGGL(k_edge_scores_p, dim3(e->Mloc), dim3(1024), lds, st, e->z, e->scores, e->thr, e->probs, e->eas, alpha, e->d,
                           e->k, e->dpad, e->ldk, none, none, 0u);
      } else if (flag_fork)  // (the last block publishes fork_seq: k_wait_flag on the second stream)
        hipLaunchKernelGGL(k_edge_scores_p, dim3(e->Mloc), dim3(1024), lds, st, e->z, e->scores, e->thr, e->probs, e->eas, alpha, e->d,
                           e->k, e->dpad, e->ldk, e->fork_flag + 1, e->fork_flag, ++e->fork_seq);
      else if (stop_ev)
        hipExtLaunchKernelGGL(k_edge_scores_p, dim3(e->Mloc), dim3(1024), lds, st, nullptr, stop_ev, 0, e->z, e->scores, e->thr, e->probs,
                              e->eas, alpha, e->d, e->k, e->dpad, e->ldk, none, none, 0u);
      else
        hipLaunchKernelGGL(k_edge_scores_p, dim3(e->Mloc), dim3(1024), lds, st, e->z, e->scores, e->thr, e->probs, e->eas, alpha, e->d,
                           e->k, e->dpad, e->ldk, none, none, 0u);
      return;
    }

    const int ntile = (e->dpad / 16) * (e->dpad / 16);
    int nby = ntile >= 16 ? 4 : (ntile >= 8 ? 2 : 1);
    while (4 * nby * EDGE_MAXT < ntile) nby *= 2;  // (a wave keeps the accumulators of at most EDGE_MAXT tiles)
    const int per_wave = (ntile + 4 * nby - 1) / (4 * nby);
#define EDGE_LAUNCH(MAXT_)                                                                                                             \
    {                                                                                                                                    \
      allow_lds(k_edge_scores<MAXT_>, lds);                                                                                              \
      if (stop_ev)                                                                                                                       \
        hipExtLaunchKernelGGL(k_edge_scores<MAXT_>, dim3(e->Mloc, nby), dim3(256), lds, st, nullptr, stop_ev, 0, e->z, e->scores, e->thr, \
                              e->probs, e->eas, alpha, e->d, e->k, e->dpad, e->ldk, e->edge_kc, (unsigned long long*)nullptr);                                          \
      else                                                                                                                               \
        hipLaunchKernelGGL(k_edge_scores<MAXT_>, dim3(e->Mloc, nby), dim3(256), lds, st, e->z, e->scores, e->thr, e->probs, e->eas,      \
                           alpha, e->d, e->k, e->dpad, e->ldk, e->edge_kc, e->profiling ? e->counters + 8 : (unsigned long long*)nullptr);                                                              \
    }
    if (per_wave <= 1) EDGE_LAUNCH(1) else if (per_wave <= 4) EDGE_LAUNCH(4) else EDGE_LAUNCH(EDGE_MAXT)
#undef EDGE_LAUNCH
  };
  // fork without a record packet on the main stream: the event is the edge kernel's own completion signal (hipExtLaunchKernel stop event)
  const bool ext_fork = fork && !e->profiling;
  launch_edge(e->stream, ext_fork ? e->ev_fork : nullptr);
  bool score_lik = false;
  if (fork) {
    if (flag_fork) {
      hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(64), 0, e->stream2, (const unsigned int*)e->fork_flag, e->fork_seq, 0u, e->join_err);
    } else {
      if (!ext_fork) hipEventRecord(e->ev_fork, e->stream);
      hipStreamWaitEvent(e->stream2, e->ev_fork, 0);
    }
    const AcycLaunch al{e->stream2, e->scores, e->acyc_part, e->w_acyc, e->acyc_big, carry_prior, e->m0, Mg, e->Mloc, e->d, e->Sa,
                        e->acyc_cpb, e->acyc_units, e->acyc_nblk, alpha, (float)c.tau, c.rng_layout, c.logistic_minval_tiny, nullptr, nullptr,
                        e->eas, e->tune.acyc_pipe, e->tune.acyc_hfw_max};
    acyc_power_timed(e, al, e->stream2);
    {
      KTimer tm(e, DIBS_K_ACYC_REDUCE, e->stream2);
      acyc_launch_reduce(al);
    }
  }
  // Single rank: the kernel matrices need only z (and theta), which are final when the step starts.  For the joint models, and for the
  // marginal model once the matrix is large against the sampling work (M D > 4 S d^2), they follow the acyclicity kernel on the second
  // stream, which otherwise idles until the likelihood chain on the main stream is done; the join before k_wtotal covers them.
  // Measured: config 3 (joint, 128 particles) 1 400 -> 1 453 steps/s, config 4 (1 024 particles) 329 -> 395.  At the headline size the
  // latent matrix stays inside the k_bge_sample launch (KmatFuse: 8 us of that kernel's 70; on the second stream 3 999 -> 3 902 steps/s,
  // and ahead of the acyclicity kernel it delays that kernel).
  if (kmat_early_now) {
    if (join_now) {  // per-kernel timing: one kernel at a time
      hipEventRecord(e->ev_k1, e->stream2);
      hipStreamWaitEvent(e->stream, e->ev_k1, 0);
      hipEventRecord(e->ev_k0, e->stream);
      hipStreamWaitEvent(e->stream2, e->ev_k0, 0);
    }
    KTimer tm(e, DIBS_K_KMAT, e->stream2);
    // tiled (kernels_kmat.h: partial sums per 32 x 32 tile and chunk, then one finishing block per row) from 128 particles: config 4 597 ->
    // 645 steps/s, config 5 108.5 -> 115, config 3 2290 -> 2328 on the same box (each row is read once per tile instead of once per pair)
    if (kmat_tiled_on(e)) {
      kmat_launch_tiled(e, e->stream2, e->z, (size_t)e->D, 0, (size_t)e->D, e->kz, (float)c.scale_latent, (float)c.h_latent, nullptr, nullptr);
      if (c.joint) kmat_launch_tiled(e, e->stream2, e->theta, (size_t)e->P, 0, (size_t)e->P, e->kt, (float)c.scale_theta, (float)c.h_theta, e->kz, e->ksum);
    } else {
      auto kmat_lds = [](size_t len) { return (size_t)(((len < KMAT_CH ? len : (size_t)KMAT_CH) + 3) & ~(size_t)3) * 4; };
      allow_lds(k_kmat, kmat_lds(e->D > e->P ? e->D : e->P));
      const dim3 kg(e->Mloc, (e->M + KMAT_BT - 1) / KMAT_BT);
      hipLaunchKernelGGL(k_kmat, kg, dim3(256), kmat_lds(e->D), e->stream2, e->z, (size_t)e->D, (size_t)0, (int)e->D, e->kz, 0, e->M,
                         (float)c.scale_latent, (float)c.h_latent, 1, (const float*)nullptr, (float*)nullptr);
      if (c.joint)
        hipLaunchKernelGGL(k_kmat, kg, dim3(256), kmat_lds(e->P), e->stream2, e->theta, (size_t)e->P, (size_t)0, (int)e->P, e->kt, 0, e->M,
                           (float)c.scale_theta, (float)c.h_theta, 1, (const float*)e->kz, e->ksum);
    }
    e->kmat_early = true;
  }
  if (fork) {
    if (flag_join) {
      ++e->join_seq;
      if (e->debug_drop_flag) e->debug_drop_flag = false;  // (dibs_engine_debug_drop_next_flag: this step's flag is never stored)
      else hipLaunchKernelGGL(k_join_flag, dim3(1), dim3(1), 0, e->stream2, e->join_flag, e->join_seq);
    }
    else hipEventRecord(e->ev_join, e->stream2);
  }
  if (fork && join_now) hipStreamWaitEvent(e->stream, e->ev_join, 0);
  if (!do_lik) {
    // (prior terms only: no estimator runs, the tail takes a zero likelihood gradient)
  } else if (c.likelihood == DIBS_LIK_BGE && c.grad_estimator_z == DIBS_EST_REPARAM) {
    const BgeSoftParams sp{e->bge.R, e->bge.Nj, e->bge.alpha_lambd, e->bge.alpha_mu, e->bge.log_t, e->bge.n_mats};
    KTimer tm(e, DIBS_K_BGE_NODES);
    bge_soft_launch(sp, e->scores, carry_lik, e->m0, Mg, e->Mloc, e->d, e->S, alpha, (float)c.tau, L, c.logistic_minval_tiny,
                    e->soft_ds, e->logprobs_z, e->w_lik, e->stream, e->soft_tri, e->soft_blocks);
  } else if (c.likelihood == DIBS_LIK_BGE) {
    const BgeParams bp = e->bge.params();
    {  // (queue counters: zero at creation, reset by k_particle_grad at the end of every step)
      KTimer tm(e, DIBS_K_BGE_NODES);
      KmatFuse kf{nullptr, nullptr, 0, 0, 0, 0.f, 0.f, nullptr, 0u};
      e->kmat_fused = false;
      // single rank, vector fits one LDS chunk: the latent kernel matrix rides along (see KmatFuse)
      if (!tile_in_grad && !kmat_tiled_on(e) && !xk && !e->kmat_early && !e->kmat_ext && e->Mloc == e->M && e->D <= KMAT_CH && (size_t)e->D * 4 + 64 <= 80 * 1024 && !e->tune.no_kmat_fuse) {
        kf = KmatFuse{e->z, e->kz, (int)e->D, e->M, (e->d + 3) / 4, (float)c.scale_latent, (float)c.h_latent, nullptr, 0u};
        e->kmat_fused = true;
      }
      if (fork_pub_in_sample) {  // (the edge kernel stored plainly and published nothing: this launch's first block does, see KmatFuse)
        kf.pub_flag = e->fork_flag;
        kf.pub_seq = e->fork_seq;
      }
      bge_launch_sample(true, e->stream, e->thr, e->masks, e->node_scores, bp, carry_lik, e->m0, Mg, e->Mloc, e->d, e->S, e->W, L,
                        e->bq, kf);
    }
    {
      KTimer tm(e, DIBS_K_BGE_BIG);
      bge_launch_chol(e->stream, e->node_scores, bp, e->bq, e->d, e->S, e->profiling ? e->counters : nullptr);
    }
    score_lik = true;  // softmax weights, W_lik and the baseline are part of k_particle_grad below
  } else if (c.likelihood == DIBS_LIK_LINGAUSS) {
    JointLaunch jl{e->stream, e->z, e->theta, e->scores, e->thr, e->w_lik, e->logprobs_z, e->logprobs_th, e->baseline,
                   e->baseline2, pack, rt.stride, rt.th_off, rt.gth_off, rt.copy_vals, e->m0, Mg, e->Mloc, e->d,
                   e->N, e->S, alpha, (float)c.tau, L, c.logistic_minval_tiny, c.grad_estimator_z, c.score_function_baseline,
                   (float)c.lin_obs_noise, (float)c.lin_mean_edge, (float)c.lin_sig_edge, e->tune.lin_f32, e->tune.nn_f32};
    {
      KTimer tm(e, DIBS_K_LIN_THETA);  // ("lin_logprobs": both log-prob launches)
      joint_lin_all_logprobs(&e->jw, jl, carry_theta, carry_lik);
    }
    {
      KTimer tm(e, DIBS_K_LIN_Z);      // ("lin_grad": the theta and the Z estimator in one launch)
      joint_lin_all_grads(&e->jw, jl, carry_theta, carry_lik);
      if (!xk) std::swap(e->baseline, e->baseline2);  // (explicit-key evaluation: the loop's baselines stay, the updated ones are read from baseline2)
    }
  } else if (c.likelihood == DIBS_LIK_DENSENN) {
    JointLaunch jl{e->stream, e->z, e->theta, e->scores, e->thr, e->w_lik, e->logprobs_z, e->logprobs_th, e->baseline,
                   e->baseline2, pack, rt.stride, rt.th_off, rt.gth_off, rt.copy_vals, e->m0, Mg, e->Mloc, e->d,
                   e->N, e->S, alpha, (float)c.tau, L, c.logistic_minval_tiny, c.grad_estimator_z, c.score_function_baseline,
                   0.f, 0.f, 0.f, e->tune.lin_f32, e->tune.nn_f32};
    const NNParams np_ = nn_params(c);
    {
      KTimer tm(e, DIBS_K_NN_THETA);
      if (joint_nn_dispatch(&e->jw, jl, carry_theta, LIN_MODE_THETA, np_, (size_t)e->P)) return fail("DenseNonlinearGaussian: scratch area: hipMalloc failed");
    }
    {
      KTimer tm(e, DIBS_K_NN_Z);
      if (joint_nn_dispatch(&e->jw, jl, carry_lik, c.grad_estimator_z == 0 ? LIN_MODE_Z_SCORE : LIN_MODE_Z_REPARAM, np_, (size_t)e->P))
        return fail("DenseNonlinearGaussian: scratch area: hipMalloc failed");
      if (!xk) std::swap(e->baseline, e->baseline2);  // (explicit-key evaluation: the loop's baselines stay, the updated ones are read from baseline2)
    }
  }
  if (fork) {
    // (covers the kernel matrices: they precede the end of the second stream's chain.  flag_join: k_particle_grad polls the flag itself)
    if (!flag_join) hipStreamWaitEvent(e->stream, e->ev_join, 0);
  } else if (do_prior) {
    const AcycLaunch al{e->stream, e->scores, e->acyc_part, e->w_acyc, e->acyc_big, carry_prior, e->m0, Mg, e->Mloc, e->d, e->Sa, e->acyc_cpb, e->acyc_units,
                        e->acyc_nblk, alpha, (float)c.tau, c.rng_layout, c.logistic_minval_tiny, nullptr, nullptr, e->eas, e->tune.acyc_pipe,
                        e->tune.acyc_hfw_max};
    acyc_power_timed(e, al, e->stream);
    {
      // (folding this reduction into k_particle_grad for small grids -- one dependent launch less -- was measured and dropped: the tail
      //  kernel grows by more than the launch it saves: config 2 54.3 -> 55.3 us/step, a rank of an 8-way headline run 91.4 -> 99.2)
      KTimer tm(e, DIBS_K_ACYC_REDUCE);
      acyc_launch_reduce(al);
    }
  }
  {
    // one block per particle: (score estimator: softmax weights -> W_lik,) total score-space gradient, back-projection, packed row
    KTimer tm(e, DIBS_K_TAIL);
    float er_c = 0.f;
    if (c.graph_prior == DIBS_PRIOR_ER) {
      const double p = c.graph_prior_edges_per_node * e->d / ((e->d * (e->d - 1)) / 2.0);
      er_c = (float)(log(p) - log(1 - p));
    }
    // (w_tot != null: W, U, V of a particle do not fit in one block's LDS -- phases A, B here, the back-projection in k_backproject_big)
    // (terms: without the prior part beta = 0, no graph prior, no Gaussian term; without the likelihood part a zero W_lik is the input)
    const float inv_sig2 = do_prior ? 1.0f / (e->sigz * e->sigz) : 0.f;
    const int ldz = e->w_tot ? 0 : tail_ldz(e->d, e->k, e->S, score_lik, LDS_LIMIT - 2048);
    const int cap = score_lik ? tail_stage_cap(e->d, ldz, e->S, e->W, LDS_LIMIT - 2048) : 0;
    const size_t lds = tail_lds_bytes(e->d, ldz, e->S, e->W, score_lik, cap);
    TailArgs ta{score_lik ? e->node_scores : nullptr, e->masks, e->logprobs_z, e->baseline, e->baseline2, c.score_function_baseline,
                      score_lik ? e->bq.counts : nullptr, e->S, e->W, cap, e->probs, do_lik ? e->w_lik : const_cast<float*>(zero_w), e->w_acyc, alpha,
                      do_prior ? beta : 0.f, do_prior ? c.graph_prior : (int)DIBS_PRIOR_UNIFORM, er_c,
                      e->z, pack, rt.stride, rt.copy_vals, e->m0, e->d, e->k, ldz, inv_sig2, e->profiling ? e->counters : nullptr,
                      e->w_tot, flag_join ? e->join_flag : nullptr, e->join_seq, e->join_err, e->Mloc,
                      KmatTile{nullptr, 0, 0, 0, nullptr, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr}};
    size_t lds_g = lds;
    int nrider = 0;
    if (tile_in_grad) {
      const int nta = (e->M + KT_T - 1) / KT_T, tiles = kmat_tile_count(nta, nta, 1), nchunk = kmat_nchunk((int)e->D);
      // (pieces: enough units for the CUs the particles leave free, one round of them -- measured at the headline size, launch time on the
      //  event timer: no units 20.7 us; 100 units of 2 chunks 21.2; 70 of 3 chunks 25.2; 200 of 1 chunk on 128 blocks 25.8)
      int ns = (256 - e->Mloc + tiles - 1) / tiles;
      ns = ns > nchunk ? nchunk : ns;
      ns = ns > e->kmat_ns_max ? e->kmat_ns_max : ns;
      const int cps = (nchunk + ns - 1) / ns;
      ns = (nchunk + cps - 1) / cps;
      if (ns > 1) {
        ta.kt = KmatTile{e->z, (size_t)e->D, 0, (int)e->D, e->kpart, 0, e->Mloc, e->M, nchunk, nta, nta, 1, ns, cps, (float)c.scale_latent,
                         (float)c.h_latent, e->kz, nullptr, nullptr, e->kmat_ctr};
        nrider = tiles * ns < 256 - e->Mloc ? tiles * ns : 256 - e->Mloc;
        lds_g = lds > kmat_tile_lds_bytes() ? lds : kmat_tile_lds_bytes();
        e->kmat_fused = true;
      }
    }
    allow_lds(k_particle_grad, lds_g);
    hipLaunchKernelGGL(k_particle_grad, dim3(e->Mloc + nrider), dim3(TAIL_NT), lds_g, e->stream, ta);
    if (e->w_tot) {
      const size_t lb = backproject_big_lds(e->d);
      allow_lds(k_backproject_big, lb);
      hipLaunchKernelGGL(k_backproject_big, dim3(e->Mloc, (e->d + 15) / 16, (e->k + 31) / 32), dim3(256), lb, e->stream, e->w_tot, e->z, pack, rt.stride,
                         rt.copy_vals, e->m0, e->d, e->k, inv_sig2);
    }
    if (score_lik && !xk) std::swap(e->baseline, e->baseline2);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(std::string("kernel launch failed: ") + hipGetErrorString(err));
  return 0;
}

// where phase B reads the rows of ALL particles: packed rows (stride E) or the two planes [values | gradients] of the overlapped protocol
// (one allocation, [2][M][Ev]: both planes share the row stride, the gradient plane starts M * Ev floats later)
struct RowSource {
  const float* base;
  size_t stride, z_off, gz_off, th_off, gth_off;
};
static RowSource packed_source(const dibs_engine* e, const float* pack) {
  return RowSource{pack, (size_t)e->E, 0, (size_t)e->D, (size_t)(2 * e->D), (size_t)(2 * e->D + e->P)};
}
static RowSource plane_source(const dibs_engine* e, const float* planes) {
  const size_t g = (size_t)e->M * e->Ev;
  return RowSource{planes, (size_t)e->Ev, 0, g, (size_t)e->D, g + (size_t)e->D};
}

static int step_update(dibs_engine* e, int t, const RowSource& rs, float* vals_send = nullptr) {
  (void)t;
  const float* const pack = rs.base;
  const dibs_config& c = e->cfg;
  const bool kmat_ext = e->kmat_ext;  // computed from the gathered values on the caller's side stream (dibs_engine_kmat_values)
  e->kmat_ext = false;
  // the particles move: plane 0 of the in-engine overlapped exchange no longer holds them (run_sharded's overlapped branch gathers the new
  // values right behind this call and sets the flag again; every other caller -- packed protocol, dibs_engine_run, step_update -- leaves
  // it cleared, so that the next overlapped chunk / gather_particles re-gathers instead of using the stale plane)
  e->vals_fresh = false;
  if (!kmat_ext && !e->kmat_early && (!e->kmat_fused || c.joint) && kmat_tiled_on(e)) {
    KTimer tm(e, DIBS_K_KMAT);
    if (!e->kmat_fused)
      kmat_launch_tiled(e, e->stream, pack, rs.stride, rs.z_off, (size_t)e->D, e->kz, (float)c.scale_latent, (float)c.h_latent, nullptr, nullptr);
    if (c.joint)
      kmat_launch_tiled(e, e->stream, pack, rs.stride, rs.th_off, (size_t)e->P, e->kt, (float)c.scale_theta, (float)c.h_theta, e->kz, e->ksum);
  } else if (!kmat_ext && !e->kmat_early && (!e->kmat_fused || c.joint)) {
    KTimer tm(e, DIBS_K_KMAT);
    const int ksym = e->Mloc == e->M;  // single rank: the slab is the whole (symmetric) matrix
    auto kmat_lds = [](size_t len) { return (size_t)(((len < KMAT_CH ? len : (size_t)KMAT_CH) + 3) & ~(size_t)3) * 4; };
    allow_lds(k_kmat, kmat_lds(e->D > e->P ? e->D : e->P));
    const dim3 kg(e->Mloc, (e->M + KMAT_BT - 1) / KMAT_BT);
    if (!e->kmat_fused)
      hipLaunchKernelGGL(k_kmat, kg, dim3(256), kmat_lds(e->D), e->stream, pack, rs.stride, rs.z_off,
                         (int)e->D, e->kz, e->m0, e->M, (float)c.scale_latent, (float)c.h_latent, ksym, (const float*)nullptr, (float*)nullptr);
    if (c.joint)
      hipLaunchKernelGGL(k_kmat, kg, dim3(256), kmat_lds(e->P), e->stream, pack, rs.stride,
                         rs.th_off, (int)e->P, e->kt, e->m0, e->M, (float)c.scale_theta, (float)c.h_theta, ksym, (const float*)e->kz, e->ksum);
  }
  {
    KTimer tm(e, DIBS_K_PHI_UPDATE);
    auto phi = [&](size_t val_off, size_t grad_off, size_t len, int is_theta, float* x, float* v, float* phi_out, float h) {
      // particles per block: as many as keep >= 1024 blocks in flight and the tables within the LDS budget
      // (headline size: TA = 16 / 8 / 4 measured 20.5 / 18.9 / 26.0 us)
      const long cols = (long)((len + 63) / 64);
      if (e->M >= 256) {  // many particles: the transform as one GEMM on the matrix pipe (a function of the GLOBAL count only)
        const int nrb = (e->Mloc + PG_BM - 1) / PG_BM;
#define PHI_GEMM(J_)                                                                                                                        \
        hipLaunchKernelGGL(k_phi_gemm<J_>, dim3((unsigned)(8 * nrb * ((cols + 7) / 8))), dim3(256), 0, e->stream, pack, rs.stride, val_off,   \
                           grad_off, (int)len, e->kz, e->kt, is_theta, x, v, phi_out, e->m0, e->Mloc, e->M, h, (float)c.stepsize,              \
                           c.optimizer == DIBS_OPT_RMSPROP, (int)cols, nrb, vals_send, (size_t)e->Ev, is_theta ? (size_t)e->D : (size_t)0);
        if (e->kt) { PHI_GEMM(true) } else { PHI_GEMM(false) }
#undef PHI_GEMM
        return;
      }
      int ta = 16;
      while (ta > 4 && (cols * ((e->Mloc + ta - 1) / ta) < 1024 || phi_update_lds_bytes(ta, e->M) > 56 * 1024)) ta >>= 1;
      const size_t lds = phi_update_lds_bytes(ta, e->M);
      const int ngroups = (e->Mloc + ta - 1) / ta;
      const dim3 g((unsigned)(8 * ngroups * ((cols + 7) / 8)));
      // joint models: the weights are kz + kt (e->ksum, formed by the k_kmat launch of kt), the repulsion uses the segment's own matrix
      const float* const kw = e->kt ? e->ksum : e->kz;
      const float* const kseg = e->kt ? (is_theta ? e->kt : e->kz) : nullptr;
      // FULL: whole 8-pair batches per wave and whole particle groups (no clamps inside the kernel)
      const bool full = e->M % 64 == 0 && e->Mloc % ta == 0 && (size_t)e->M * rs.stride * 4 < ((size_t)1 << 32);  // (32-bit buffer offsets)
#define PHI_LAUNCH(TA_, F_, J_)                                                                                                \
      {                                                                                                                          \
        allow_lds(k_phi_update<TA_, F_, J_>, lds);                                                                               \
        hipLaunchKernelGGL((k_phi_update<TA_, F_, J_>), g, dim3(256), lds, e->stream, pack, rs.stride, val_off, grad_off, (int)len, kw,     \
                           kseg, is_theta, x, v, phi_out, e->m0, e->Mloc, e->M, h, (float)c.stepsize, c.optimizer == DIBS_OPT_RMSPROP,   \
                           (int)cols, ngroups, vals_send, (size_t)e->Ev, is_theta ? (size_t)e->D : (size_t)0);               \
      }
#define PHI_PICK(TA_)                                                                                                          \
      if (e->kt) { if (full) PHI_LAUNCH(TA_, true, true) else PHI_LAUNCH(TA_, false, true) }                                     \
      else { if (full) PHI_LAUNCH(TA_, true, false) else PHI_LAUNCH(TA_, false, false) }
      if (ta == 16) { PHI_PICK(16) } else if (ta == 8) { PHI_PICK(8) } else { PHI_PICK(4) }
#undef PHI_PICK
#undef PHI_LAUNCH
    };
    phi(rs.z_off, rs.gz_off, (size_t)e->D, 0, e->z, e->vz, e->phi_z, (float)c.h_latent);
    if (c.joint) phi(rs.th_off, rs.gth_off, (size_t)e->P, 1, e->theta, e->vtheta, e->phi_th, (float)c.h_theta);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(std::string("kernel launch failed: ") + hipGetErrorString(err));
  return 0;
}

// does this chunk / call use the in-kernel flags?  Decided here, once per chunk (not per step): the tuning switch, an earlier time-out,
// the creation-time probe, and the engine being alone in its process (several engines share hardware queues: a polling kernel at the head
// of a shared queue holds up the kernels behind it, possibly the one it waits for)
static void latch_flags(dibs_engine* e) {
  e->flags_now = !e->tune.no_flags && !e->flags_off && e->join_flag != nullptr && e->streams_concurrent &&
                 (e->tune.flags_multi || g_live_engines.load() == 1);
}

// after a chunk has been synchronised: did a kernel give up waiting for a flag (tail_join_wait / k_wait_flag)?  Clears the word.
static unsigned int take_join_err(dibs_engine* e) {
  if (!e->join_err || !*e->join_err) return 0u;
  const unsigned int code = *e->join_err;
  *e->join_err = 0u;
  return code;
}
static int join_failure(unsigned int code, const char* what) {
  return fail(std::string(code == 2u ? "internal: the edge kernel's completion flag did not arrive (k_wait_flag on the second stream timed out)"
                                     : "internal: the acyclicity stream's completion flag did not arrive (k_particle_grad timed out waiting)") + what);
}

// the loop carry of this rank (svgd.py:315: optimizer states, key, baselines) copied aside / back in ONE launch
struct CopySegs {
  const float* src[5];
  float* dst[5];
  size_t n[5];
};
__global__ __launch_bounds__(256) void k_copy_segs(CopySegs c) {
  const int sg = (int)blockIdx.y;
  const float* __restrict__ a = c.src[sg];
  float* __restrict__ b = c.dst[sg];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < c.n[sg]; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
static int carry_copy(dibs_engine* e, bool restore) {
  const size_t nz = (size_t)e->Mloc * e->D, nt = (size_t)e->Mloc * e->P, nb = (size_t)e->Mloc;
  if (!e->carry_bak) HIP_OK(dalloc(&e->carry_bak, 2 * nz + 2 * nt + nb));
  float* const b = e->carry_bak;
  float* live[5] = {e->z, e->vz, e->theta, e->vtheta, e->baseline};
  float* bak[5] = {b, b + nz, b + 2 * nz, b + 2 * nz + nt, b + 2 * nz + 2 * nt};
  const size_t n[5] = {nz, nz, nt, nt, nb};
  CopySegs c;
  for (int i = 0; i < 5; ++i) {
    c.src[i] = restore ? bak[i] : live[i];
    c.dst[i] = restore ? live[i] : bak[i];
    c.n[i] = n[i];
  }
  hipLaunchKernelGGL(k_copy_segs, dim3(256, 5), dim3(256), 0, e->stream, c);
  if (restore) e->key = e->key_bak;
  else e->key_bak = e->key;
  return 0;
}

static int run_steps(dibs_engine* e, int t_start, int n_steps) {
  for (int t = t_start; t < t_start + n_steps; ++t) {
    if (step_local(e, t, packed_rows(e, e->pack))) return 1;
    if (step_update(e, t, packed_source(e, e->pack))) return 1;
    if (e->profiling && e->pending.size() > 4096) drain_timers(e);
  }
  HIP_OK(hipStreamSynchronize(e->stream));
  if (e->stream2) HIP_OK(hipStreamSynchronize(e->stream2));
  if (e->profiling) drain_timers(e);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int dibs_engine_run(dibs_engine* e, int32_t t_start, int32_t n_steps) {
  if (!e) return fail("null engine");
  if (!e->has_data) return fail("dibs_engine_set_data has not been called");
  if (e->cfg.n_ranks != 1) return fail("dibs_engine_run is single-rank; use step_local / step_update");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  latch_flags(e);
  const bool guarded = e->flags_now && n_steps > 0;
  if (guarded && carry_copy(e, false)) return 1;  // (one launch per chunk: 10 MB at the headline size, ~4 us)
  if (run_steps(e, t_start, n_steps)) return 1;
  unsigned int code = take_join_err(e);
  if (code && guarded) {
    // a polling kernel ran into its bound: its step, and every step behind it, used operands that were not complete.  Back to the
    // chunk's start, flags off for good, the same steps again on events.
    e->flags_off = true;
    ++e->flag_fallbacks;
    latch_flags(e);
    if (carry_copy(e, true)) return 1;
    if (run_steps(e, t_start, n_steps)) return 1;
    code = take_join_err(e);
  }
  if (code) return join_failure(code, "; the results of this chunk are invalid");
  return 0;
}

extern "C" int dibs_engine_flag_fallbacks(const dibs_engine* e) { return e ? e->flag_fallbacks : -1; }
extern "C" int dibs_engine_debug_drop_next_flag(dibs_engine* e) {
  if (!e) return fail("null engine");
  e->debug_drop_flag = true;
  return 0;
}

extern "C" int dibs_engine_step_local(dibs_engine* e, int32_t t, void* send_dev) {
  if (!e || !send_dev) return fail("null argument");
  if (!e->has_data) return fail("dibs_engine_set_data has not been called");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  latch_flags(e);
  // send_dev holds only this rank's rows: [Mloc, E]; kernels index rows by global particle id
  float* base = (float*)send_dev - (size_t)e->m0 * e->E;
  return step_local(e, t, packed_rows(e, base));
}

// ---- overlapped exchange (see include/dibs_hip.h): values travel right after the optimizer step, gradients between the phases ----
extern "C" int64_t dibs_engine_plane_elems_per_rank(const dibs_engine* e) { return e ? (int64_t)e->Mloc * e->Ev : 0; }

extern "C" int dibs_engine_export_values(dibs_engine* e, void* vals_send_dev) {
  if (!e || !vals_send_dev) return fail("null argument");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  float* dst = (float*)vals_send_dev;
  HIP_OK(hipMemcpy2DAsync(dst, (size_t)e->Ev * 4, e->z, (size_t)e->D * 4, (size_t)e->D * 4, (size_t)e->Mloc, hipMemcpyDeviceToDevice, e->stream));
  if (e->P)
    HIP_OK(hipMemcpy2DAsync(dst + e->D, (size_t)e->Ev * 4, e->theta, (size_t)e->P * 4, (size_t)e->P * 4, (size_t)e->Mloc, hipMemcpyDeviceToDevice,
                            e->stream));
  return 0;
}

extern "C" int dibs_engine_step_local_grads(dibs_engine* e, int32_t t, void* grads_send_dev) {
  if (!e || !grads_send_dev) return fail("null argument");
  if (!e->has_data) return fail("dibs_engine_set_data has not been called");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  latch_flags(e);
  // rows [grad_z | grad_theta] of this rank's particles, stride Ev; kernels index rows by global particle id
  float* base = (float*)grads_send_dev - (size_t)e->m0 * e->Ev;
  return step_local(e, t, RowTarget{base, (size_t)e->Ev, 0, 0, (size_t)e->D, 0});
}

// kernel-matrix slab(s) of the NEXT phase B from the values of all particles (plane 0), launched on `stream` -- the caller's side stream,
// behind its all-gather of the values, i.e. beside phase A and without any synchronisation of its own.  The caller orders phase B behind it
// (one event it needs anyway: phase B reads plane 0 as well).
extern "C" int dibs_engine_kmat_values(dibs_engine* e, const void* vals_all_dev, void* stream) {
  if (!e || !vals_all_dev || !stream) return fail("null argument");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  const dibs_config& c = e->cfg;
  hipStream_t st = (hipStream_t)stream;
  if (kmat_tiled_on(e)) {
    const float* v = (const float*)vals_all_dev;
    kmat_launch_tiled(e, st, v, (size_t)e->Ev, 0, (size_t)e->D, e->kz, (float)c.scale_latent, (float)c.h_latent, nullptr, nullptr);
    if (c.joint) kmat_launch_tiled(e, st, v, (size_t)e->Ev, (size_t)e->D, (size_t)e->P, e->kt, (float)c.scale_theta, (float)c.h_theta, e->kz, e->ksum);
    HIP_OK(hipGetLastError());
    e->kmat_ext = true;
    return 0;
  }
  auto kmat_lds = [](size_t len) { return (size_t)(((len < KMAT_CH ? len : (size_t)KMAT_CH) + 3) & ~(size_t)3) * 4; };
  dibs_allow_lds((const void*)k_kmat, kmat_lds(e->D > e->P ? e->D : e->P));
  const dim3 kg(e->Mloc, (e->M + KMAT_BT - 1) / KMAT_BT);
  const int ksym = e->Mloc == e->M;
  const float* vals = (const float*)vals_all_dev;
  hipLaunchKernelGGL(k_kmat, kg, dim3(256), kmat_lds(e->D), st, vals, (size_t)e->Ev, (size_t)0, (int)e->D, e->kz, e->m0, e->M, (float)c.scale_latent,
                     (float)c.h_latent, ksym, (const float*)nullptr, (float*)nullptr);
  if (c.joint)
    hipLaunchKernelGGL(k_kmat, kg, dim3(256), kmat_lds(e->P), st, vals, (size_t)e->Ev, (size_t)e->D, (int)e->P, e->kt, e->m0, e->M, (float)c.scale_theta,
                       (float)c.h_theta, ksym, (const float*)e->kz, e->ksum);
  HIP_OK(hipGetLastError());
  e->kmat_ext = true;
  return 0;
}

extern "C" int dibs_engine_step_update_planes(dibs_engine* e, int32_t t, const void* planes_dev, void* vals_send_dev) {
  if (!e || !planes_dev) return fail("null argument");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  return step_update(e, t, plane_source(e, (const float*)planes_dev), (float*)vals_send_dev);
}

extern "C" int dibs_engine_step_update(dibs_engine* e, int32_t t, const void* recv_dev) {
  if (!e || !recv_dev) return fail("null argument");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  return step_update(e, t, packed_source(e, (const float*)recv_dev));
}

// ---- gradient estimators for explicit per-particle keys (include/dibs_hip.h) ------------------------------------------------
// reference: DiBS.eltwise_grad_z_likelihood (dibs.py:295-321), eltwise_grad_theta_likelihood (:467-485), eltwise_grad_latent_prior (:626-658)
extern "C" int dibs_engine_eval_gradients(dibs_engine* e, int32_t t, const uint32_t* keys_theta, const uint32_t* keys_lik, const uint32_t* keys_prior,
                                          float* grad_z_lik, float* baseline_out, float* grad_theta, float* grad_z_prior) {
  if (!e) return fail("null engine");
  if (!e->has_data) return fail("dibs_engine_set_data has not been called");
  const dibs_config& c = e->cfg;
  const bool want_lik = keys_lik != nullptr || keys_theta != nullptr, want_prior = keys_prior != nullptr;
  if (c.joint && want_lik && (!keys_lik || !keys_theta)) return fail("joint model: pass the keys of the theta AND the Z estimator (both run in one pass)");
  if (!c.joint && keys_theta) return fail("keys_theta given for a marginal model");
  if (want_lik && !c.joint && !keys_lik) return fail("keys_lik missing");
  HIP_OK(hipSetDevice(c.device_id));
  HIP_OK(hipStreamSynchronize(e->stream));
  const size_t nk = (size_t)e->Mloc * 2;
  DevBuf<uint32_t> dk;   // [3][Mloc][2]
  DevBuf<float> zero_w;  // [Mloc][d][d] zeros: the likelihood gradient of the prior-only pass
  HIP_OK(dk.alloc(3 * nk));
  const uint32_t* src[3] = {keys_theta, keys_lik, keys_prior};
  for (int i = 0; i < 3; ++i)
    if (src[i]) HIP_OK(hipMemcpy(dk.p + i * nk, src[i], nk * 4, hipMemcpyHostToDevice));
  const StepKeys xk{reinterpret_cast<const Key2*>(dk.p), reinterpret_cast<const Key2*>(dk.p + nk), reinterpret_cast<const Key2*>(dk.p + 2 * nk)};
  const size_t wz = (size_t)e->D * 4, wt = (size_t)e->P * 4;
  const float* rows = e->pack + (size_t)e->m0 * e->E;
  if (want_lik) {
    if (step_local(e, t, packed_rows(e, e->pack), &xk, TERMS_LIK)) return 1;
    HIP_OK(hipStreamSynchronize(e->stream));
    if (grad_z_lik) HIP_OK(hipMemcpy2D(grad_z_lik, wz, rows + e->D, (size_t)e->E * 4, wz, e->Mloc, hipMemcpyDeviceToHost));
    if (grad_theta && e->P) HIP_OK(hipMemcpy2D(grad_theta, wt, rows + 2 * e->D + e->P, (size_t)e->E * 4, wt, e->Mloc, hipMemcpyDeviceToHost));
    if (baseline_out) HIP_OK(hipMemcpy(baseline_out, e->baseline2, (size_t)e->Mloc * 4, hipMemcpyDeviceToHost));  // (not swapped in: see step_local)
  }
  if (want_prior) {
    HIP_OK(zero_w.alloc((size_t)e->Mloc * e->d * e->d));
    if (step_local(e, t, packed_rows(e, e->pack), &xk, TERMS_PRIOR, zero_w.p)) return 1;
    HIP_OK(hipStreamSynchronize(e->stream));
    if (grad_z_prior) HIP_OK(hipMemcpy2D(grad_z_prior, wz, rows + e->D, (size_t)e->E * 4, wz, e->Mloc, hipMemcpyDeviceToHost));
  }
  if (e->profiling) drain_timers(e);
  HIP_OK(hipGetLastError());
  return 0;
}

// ---- in-engine exchange: RCCL bound at run time, the step loop of a sharded run in C (include/dibs_hip.h) ---------------------------------
// librccl.so.1 is dlopen'ed on first use: in a process that has imported torch this is torch's bundled copy (same SONAME, already
// mapped), otherwise ROCm's -- one RCCL per process either way, and libdibs_hip.so loads on machines without it.
struct dibs_rccl {
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  bool ok = false;
  std::string why;
};
static const dibs_rccl& rccl() {
  static const dibs_rccl r = [] {
    dibs_rccl q;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      q.why = std::string("librccl not found: ") + dlerror();
      return q;
    }
    q.get_unique_id = (decltype(q.get_unique_id))dlsym(h, "ncclGetUniqueId");
    q.comm_init_rank = (decltype(q.comm_init_rank))dlsym(h, "ncclCommInitRank");
    q.comm_destroy = (decltype(q.comm_destroy))dlsym(h, "ncclCommDestroy");
    q.all_gather = (decltype(q.all_gather))dlsym(h, "ncclAllGather");
    q.error_string = (decltype(q.error_string))dlsym(h, "ncclGetErrorString");
    q.ok = q.get_unique_id && q.comm_init_rank && q.comm_destroy && q.all_gather && q.error_string;
    if (!q.ok) q.why = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclGetErrorString";
    return q;
  }();
  return r;
}
#define RCCL_OK(expr)                                                                                    \
  do {                                                                                                   \
    ncclResult_t _r = (expr);                                                                            \
    if (_r != ncclSuccess) return fail(std::string(#expr) + ": " + rccl().error_string(_r));             \
  } while (0)

static_assert(DIBS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "include/dibs_hip.h: DIBS_COMM_ID_BYTES");

extern "C" int dibs_comm_unique_id(void* id_out) {
  if (!id_out) return fail("null argument");
  if (!rccl().ok) return fail(rccl().why);
  ncclUniqueId id;
  RCCL_OK(rccl().get_unique_id(&id));
  memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

extern "C" int dibs_engine_comm_destroy(dibs_engine* e) {
  if (!e) return 0;
  for (int i = 0; i < 2; ++i)
    if (e->comm[i]) {
      rccl().comm_destroy(e->comm[i]);
      e->comm[i] = nullptr;
    }
  e->n_comms = 0;
  if (e->agree_dev) hipFree(e->agree_dev);
  if (e->agree_host) hipHostFree(e->agree_host);
  e->agree_dev = e->agree_host = nullptr;
  if (e->ipc.arena || e->ipc.err) {
    // (the peers must have left their last exchange: every rank returns from dibs_engine_run_sharded / gather_particles only after it has seen
    //  all of its peers' rows, and nobody writes into an arena outside an exchange)
    for (int r = 0; r < IPC_MAX_RANKS; ++r)
      if (e->ipc.opened[r]) hipIpcCloseMemHandle(e->ipc.peers.base[r]);
    if (e->ipc.arena) hipFree(e->ipc.arena);
    if (e->ipc.err) hipHostFree(e->ipc.err);
    e->ipc = IpcComm{};
  }
  if (e->planes) hipFree(e->planes);
  if (e->vsend) hipFree(e->vsend);
  e->planes = e->vsend = nullptr;
  if (e->side) hipStreamDestroy(e->side);
  if (e->ev_exported) hipEventDestroy(e->ev_exported);
  if (e->ev_vals) hipEventDestroy(e->ev_vals);
  e->side = nullptr;
  e->ev_exported = e->ev_vals = nullptr;
  return 0;
}

extern "C" int dibs_engine_comm_init(dibs_engine* e, const void* ids, int32_t n_ids) {
  if (!e) return fail("null argument");
  if (n_ids < 1 || n_ids > 2) return fail("n_ids must be 1 (one all-gather per step) or 2 (overlapped exchange as well)");
  // ids == NULL: LOOPBACK -- no communicator, the all-gathers are skipped and the rows of the other ranks keep whatever the buffers hold.
  // A measuring device (scripts/gpu_shard_scaling.py: what ONE rank of an N-way run costs per step in this loop, on one GPU), not a
  // way to run a sharded job.
  if (ids && !rccl().ok) return fail(rccl().why);
  HIP_OK(hipSetDevice(e->cfg.device_id));
  dibs_engine_comm_destroy(e);
  e->loopback = ids == nullptr;
  for (int i = 0; ids && i < n_ids; ++i) {
    ncclUniqueId id;
    memcpy(id.internal, (const char*)ids + (size_t)i * NCCL_UNIQUE_ID_BYTES, NCCL_UNIQUE_ID_BYTES);
    RCCL_OK(rccl().comm_init_rank(&e->comm[i], e->cfg.n_ranks, id, e->cfg.rank));
  }
  e->n_comms = n_ids;
  HIP_OK(dalloc(&e->agree_dev, (size_t)4 + 4 * e->cfg.n_ranks));
  HIP_OK(hipHostMalloc((void**)&e->agree_host, ((size_t)4 + 4 * e->cfg.n_ranks) * 4, hipHostMallocDefault));
  if (n_ids == 2) {
    HIP_OK(dalloc(&e->planes, (size_t)2 * e->M * e->Ev));
    HIP_OK(dalloc(&e->vsend, (size_t)e->Mloc * e->Ev));
    HIP_OK(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&e->ev_exported, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&e->ev_vals, hipEventDisableTiming));
    HIP_OK(hipDeviceSynchronize());
  }
  e->vals_fresh = false;
  return 0;
}

// ---- the exchange through mapped peer memory (exchange_ipc.h): ranks that share a device, or devices with peer access ------------------
static_assert(DIBS_IPC_HANDLE_BYTES == sizeof(IpcBlob), "include/dibs_hip.h: DIBS_IPC_HANDLE_BYTES");

// allocates this rank's exchange arena (zeroed: no exchange has arrived) and writes the blob its peers need to map it
extern "C" int dibs_engine_ipc_export(dibs_engine* e, void* blob_out) {
  if (!e || !blob_out) return fail("null argument");
  if (e->cfg.n_ranks > IPC_MAX_RANKS) return fail("the mapped-memory exchange supports at most " + std::to_string(IPC_MAX_RANKS) + " ranks");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  dibs_engine_comm_destroy(e);
  IpcComm& c = e->ipc;
  c.n_ranks = e->cfg.n_ranks;
  c.rank = e->cfg.rank;
  c.pack_elems = (size_t)e->M * e->E;
  c.set_elems = (size_t)2 * e->M * e->Ev;
  c.arena_bytes = IPC_FLAG_BYTES + (2 * c.pack_elems + 2 * c.set_elems) * 4;
  HIP_OK(hipMalloc((void**)&c.arena, c.arena_bytes));
  HIP_OK(hipMemset(c.arena, 0, c.arena_bytes));
  HIP_OK(hipDeviceSynchronize());
  IpcBlob b;
  memset(&b, 0, sizeof b);
  b.magic = IPC_MAGIC;
  b.abi = DIBS_ABI_VERSION;
  b.rank = (uint32_t)c.rank;
  b.n_ranks = (uint32_t)c.n_ranks;
  b.arena_bytes = c.arena_bytes;
  b.pack_elems = c.pack_elems;
  b.set_elems = c.set_elems;
  b.device_id = e->cfg.device_id;
  b.pid = (int32_t)getpid();
  HIP_OK(hipIpcGetMemHandle(&b.handle, c.arena));
  memcpy(blob_out, &b, sizeof b);
  return 0;
}

// blobs_all: the n_ranks blobs of dibs_engine_ipc_export in rank order (every rank passes the same bytes)
extern "C" int dibs_engine_comm_init_ipc(dibs_engine* e, const void* blobs_all) {
  if (!e || !blobs_all) return fail("null argument");
  IpcComm& c = e->ipc;
  if (!c.arena) return fail("dibs_engine_ipc_export has not been called on this engine");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  const IpcBlob* B = reinterpret_cast<const IpcBlob*>(blobs_all);
  for (int r = 0; r < c.n_ranks; ++r) {
    IpcBlob b;
    memcpy(&b, B + r, sizeof b);
    if (b.magic != IPC_MAGIC || b.abi != DIBS_ABI_VERSION) return fail("blob of rank " + std::to_string(r) + ": not a dibs_engine_ipc_export blob of this ABI version");
    if ((int)b.rank != r || (int)b.n_ranks != c.n_ranks) return fail("blob " + std::to_string(r) + " belongs to rank " + std::to_string(b.rank) + " of " + std::to_string(b.n_ranks));
    if (b.arena_bytes != c.arena_bytes || b.pack_elems != c.pack_elems || b.set_elems != c.set_elems)
      return fail("rank " + std::to_string(r) + " was created with a different configuration (exchange arena sizes differ)");
    if (r == c.rank) {
      c.peers.base[r] = c.arena;
      continue;
    }
    if (b.pid == (int32_t)getpid()) return fail("the mapped-memory exchange needs one PROCESS per rank (rank " + std::to_string(r) + " lives in this process)");
    void* p = nullptr;
    HIP_OK(hipIpcOpenMemHandle(&p, b.handle, hipIpcMemLazyEnablePeerAccess));
    c.peers.base[r] = (char*)p;
    c.opened[r] = true;
  }
  HIP_OK(hipHostMalloc((void**)&c.err, 4, hipHostMallocDefault));
  *c.err = 0u;
  if (e->tune.ipc_timeout_ms > 0) c.wait_ticks = (unsigned long long)e->tune.ipc_timeout_ms * 100000ull;  // (100 MHz ticks; tuning.h)
  HIP_OK(dalloc(&e->agree_dev, (size_t)4 + 4 * e->cfg.n_ranks));
  HIP_OK(hipHostMalloc((void**)&e->agree_host, ((size_t)4 + 4 * e->cfg.n_ranks) * 4, hipHostMallocDefault));
  HIP_OK(dalloc(&e->vsend, (size_t)e->Mloc * e->Ev));
  HIP_OK(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
  HIP_OK(hipEventCreateWithFlags(&e->ev_exported, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&e->ev_vals, hipEventDisableTiming));
  HIP_OK(hipDeviceSynchronize());
  c.on = true;
  e->loopback = false;
  e->n_comms = 2;  // (both protocols: the arena holds the packed rows and the planes)
  e->vals_fresh = false;
  return 0;
}

// one all-gather through the arenas on stream `st`: `n` floats at `src` (this rank's rows) -> byte offset dst_off of every peer's arena
// (include_self: and of the own one), then the announcement + wait of this exchange on `channel`
static int ipc_all_gather(dibs_engine* e, int channel, hipStream_t st, const float* src, size_t dst_off, size_t n, bool include_self) {
  IpcComm& c = e->ipc;
  if ((n & 3) || (dst_off & 15) || (reinterpret_cast<uintptr_t>(src) & 15)) return fail("internal: exchange rows are not 16-byte aligned");
  const size_t n4 = n / 4;
  const int ndst = include_self ? c.n_ranks : c.n_ranks - 1;
  if (ndst > 0 && n4 > 0) {
    const unsigned bx = (unsigned)((n4 + 255) / 256 < 256 ? (n4 + 255) / 256 : 256);
    hipLaunchKernelGGL(k_ipc_push, dim3(bx, (unsigned)ndst), dim3(256), 0, st, c.peers, c.rank, c.n_ranks, include_self ? 1 : 0,
                       reinterpret_cast<const float4*>(src), dst_off, n4);
  }
  const unsigned int seq = ++c.seq[channel];
  if (c.n_ranks > 1) hipLaunchKernelGGL(k_ipc_signal_wait, dim3(1), dim3(64), 0, st, c.peers, c.rank, c.n_ranks, channel, seq, c.wait_ticks, c.err);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int dibs_engine_kmat_values(dibs_engine* e, const void* vals_all_dev, void* stream);
// Loopback stand-in for the all-gather of the values (per-rank timing on one GPU).  A plain copy KERNEL: hipMemcpyAsync(DeviceToDevice) on the
// side stream made the un-profiled loop of a 4-way rank take 380 us per step instead of 103 (and 107 under rocprofv3, which turns the copy
// into a blit kernel): the runtime's copy path resolves the cross-stream dependency on the host.  RCCL's all-gather is a kernel as well.
__global__ void k_copy_rows(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) dst[i] = src[i];
}

// values of this rank (already in vsend unless `exported`) -> plane 0 of every rank on the side stream, kernel slab behind the gather
static int exchange_values(dibs_engine* e, bool exported) {
  if (!exported) {
    HIP_OK(hipMemcpy2DAsync(e->vsend, (size_t)e->Ev * 4, e->z, (size_t)e->D * 4, (size_t)e->D * 4, (size_t)e->Mloc, hipMemcpyDeviceToDevice, e->stream));
    if (e->P)
      HIP_OK(hipMemcpy2DAsync(e->vsend + e->D, (size_t)e->Ev * 4, e->theta, (size_t)e->P * 4, (size_t)e->P * 4, (size_t)e->Mloc,
                              hipMemcpyDeviceToDevice, e->stream));
  }
  HIP_OK(hipEventRecord(e->ev_exported, e->stream));
  HIP_OK(hipStreamWaitEvent(e->side, e->ev_exported, 0));
  const float* plane0 = e->planes;
  if (e->ipc.on) {  // value exchange n goes to plane set n & 1 of every arena (the own one included); the gradient rows of that step follow it there
    e->ipc.vset = (int)((e->ipc.seq[1] + 1u) & 1u);
    if (ipc_all_gather(e, 1, e->side, e->vsend, e->ipc.set_off(e->ipc.vset) + (size_t)e->m0 * e->Ev * 4, (size_t)e->Mloc * e->Ev, true)) return 1;
    plane0 = e->ipc.set(e->ipc.vset);
  } else if (e->loopback) {  // (own rows only; a kernel of our own, not hipMemcpyAsync: see k_copy_rows)
    const size_t n4 = (size_t)e->Mloc * e->Ev / 4;  // (Ev is a multiple of 4)
    hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, e->side, reinterpret_cast<const float4*>(e->vsend),
                       reinterpret_cast<float4*>(e->planes + (size_t)e->m0 * e->Ev), n4);
  }
  else
    RCCL_OK(rccl().all_gather(e->vsend, e->planes, (size_t)e->Mloc * e->Ev, ncclFloat, e->comm[1], e->side));
  if (dibs_engine_kmat_values(e, plane0, e->side)) return 1;
  HIP_OK(hipEventRecord(e->ev_vals, e->side));
  e->vals_fresh = true;
  return 0;
}

// after the streams have been synchronised: did a rank give up waiting for its peers' rows?
static int ipc_check(dibs_engine* e) {
  if (e->ipc.err && *e->ipc.err) {
    *e->ipc.err = 0u;
    return fail("mapped-memory exchange: the rows of a peer rank did not arrive within the time-out (a rank that died, or ranks that did not "
                "call the same sequence of runs); the results of this chunk are invalid");
  }
  return 0;
}

// the steps of one sharded chunk, enqueued and synchronised
static int run_sharded_steps(dibs_engine* e, int t_start, int n_steps, int overlapped) {
  const size_t grad_plane = (size_t)e->M * e->Ev;
  if (overlapped && !e->vals_fresh && exchange_values(e, false)) return 1;
  for (int t = t_start; t < t_start + n_steps; ++t) {
    if (!overlapped) {
      const int pp = (int)(e->ipc.pack_seq & 1u);
      float* const pk = e->ipc.on ? e->ipc.pack(pp) : e->pack;
      if (step_local(e, t, packed_rows(e, pk))) return 1;
      if (e->ipc.on) {
        if (ipc_all_gather(e, 0, e->stream, pk + (size_t)e->m0 * e->E, e->ipc.pack_off(pp) + (size_t)e->m0 * e->E * 4, (size_t)e->Mloc * e->E, false)) return 1;
        ++e->ipc.pack_seq;
      } else if (!e->loopback)
        RCCL_OK(rccl().all_gather(pk + (size_t)e->m0 * e->E, pk, (size_t)e->Mloc * e->E, ncclFloat, e->comm[0], e->stream));
      if (step_update(e, t, packed_source(e, pk))) return 1;
    } else {
      float* const planes = e->ipc.on ? e->ipc.set(e->ipc.vset) : e->planes;  // (the set the values of this step were gathered into)
      float* const gplane = planes + grad_plane;  // rows [grad_z | grad_theta], indexed by global particle id
      if (step_local(e, t, RowTarget{gplane, (size_t)e->Ev, 0, 0, (size_t)e->D, 0})) return 1;
      if (e->ipc.on) {
        if (ipc_all_gather(e, 0, e->stream, gplane + (size_t)e->m0 * e->Ev, e->ipc.set_off(e->ipc.vset) + (grad_plane + (size_t)e->m0 * e->Ev) * 4,
                           (size_t)e->Mloc * e->Ev, false))
          return 1;
      } else if (!e->loopback)
        RCCL_OK(rccl().all_gather(gplane + (size_t)e->m0 * e->Ev, gplane, (size_t)e->Mloc * e->Ev, ncclFloat, e->comm[0], e->stream));
      HIP_OK(hipStreamWaitEvent(e->stream, e->ev_vals, 0));  // values + kernel slab of this step (gathered during the step before)
      if (step_update(e, t, plane_source(e, planes), e->vsend)) return 1;
      if (exchange_values(e, true)) return 1;  // values of step t + 1, beside its phase A
    }
    if (e->profiling && e->pending.size() > 4096) drain_timers(e);
  }
  HIP_OK(hipStreamSynchronize(e->stream));
  if (e->stream2) HIP_OK(hipStreamSynchronize(e->stream2));
  if (overlapped) HIP_OK(hipStreamSynchronize(e->side));
  if (e->profiling) drain_timers(e);
  HIP_OK(hipGetLastError());
  return ipc_check(e);
}

// Every rank learns whether ANY rank's chunk saw a flag time-out (the rows such a rank exchanged were computed from incomplete operands, so
// all ranks' results are invalid together): one all-gather of the ranks' error words, through the same backend as the rows.  *any = the
// largest word.  Costs one tiny collective + a host round trip per CHUNK.
static int agree_on_error(dibs_engine* e, unsigned int mine, unsigned int* any) {
  *any = mine;
  if (e->loopback || e->cfg.n_ranks == 1) return 0;
  const int R = e->cfg.n_ranks;
  for (int i = 0; i < 4; ++i) e->agree_host[i] = mine;
  HIP_OK(hipMemcpyAsync(e->agree_dev, e->agree_host, 16, hipMemcpyHostToDevice, e->stream));
  if (e->ipc.on) {
    const size_t off = IPC_AGREE_OFF + (size_t)(e->ipc.agree_seq & 1u) * IPC_MAX_RANKS * 16;
    ++e->ipc.agree_seq;
    if (ipc_all_gather(e, 0, e->stream, reinterpret_cast<const float*>(e->agree_dev), off + (size_t)e->cfg.rank * 16, 4, true)) return 1;
    HIP_OK(hipMemcpyAsync(e->agree_host + 4, e->ipc.arena + off, (size_t)R * 16, hipMemcpyDeviceToHost, e->stream));
  } else {
    RCCL_OK(rccl().all_gather(e->agree_dev, e->agree_dev + 4, 4, ncclUint32, e->comm[0], e->stream));
    HIP_OK(hipMemcpyAsync(e->agree_host + 4, e->agree_dev + 4, (size_t)R * 16, hipMemcpyDeviceToHost, e->stream));
  }
  HIP_OK(hipStreamSynchronize(e->stream));
  if (ipc_check(e)) return 1;
  for (int r = 0; r < R; ++r)
    if (e->agree_host[4 + 4 * r] > *any) *any = e->agree_host[4 + 4 * r];
  return 0;
}

// replaces _svgd_loop for a particle-sharded run: every rank calls it with the same (t_start, n_steps).  overlapped = 0: phase A -> ONE
// all-gather of the packed rows [z | grad_z | theta | grad_theta] (in place in the row buffer, on the engine stream) -> phase B.
// overlapped = 1: values gathered on the side stream beside phase A, only the gradient rows between the phases.
// A flag time-out on ANY rank (see latch_flags) makes ALL ranks repeat the chunk on events from their chunk-start copies of the carry.
extern "C" int dibs_engine_run_sharded(dibs_engine* e, int32_t t_start, int32_t n_steps, int32_t overlapped) {
  if (!e) return fail("null engine");
  if (!e->has_data) return fail("dibs_engine_set_data has not been called");
  if (e->n_comms < 1) return fail("dibs_engine_comm_init has not been called");
  if (overlapped && e->n_comms < 2) return fail("the overlapped exchange needs two communicators (dibs_engine_comm_init with n_ids = 2)");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  latch_flags(e);
  if (n_steps > 0 && carry_copy(e, false)) return 1;
  if (run_sharded_steps(e, t_start, n_steps, overlapped)) return 1;
  if (n_steps <= 0) return 0;
  unsigned int any = 0u;
  if (agree_on_error(e, take_join_err(e), &any)) return 1;
  if (any) {
    e->flags_off = true;
    ++e->flag_fallbacks;
    latch_flags(e);
    if (carry_copy(e, true)) return 1;
    e->vals_fresh = false;  // (the overlapped protocol gathers the restored values again: every rank does)
    if (run_sharded_steps(e, t_start, n_steps, overlapped)) return 1;
    if (agree_on_error(e, take_join_err(e), &any)) return 1;
    if (any) return join_failure(any, " on a rank of this run, twice; the results of this chunk are invalid");
  }
  return 0;
}

// z (and theta) of ALL ranks' particles after a sharded run, on every rank: [M][d][k][2] and [M][P] host buffers (either may be NULL).
// overlapped runs already hold them in plane 0; otherwise one all-gather of the values.
extern "C" int dibs_engine_gather_particles(dibs_engine* e, float* z_all, float* theta_all) {
  if (!e) return fail("null engine");
  if (e->n_comms < 1) return fail("dibs_engine_comm_init has not been called");
  if (e->loopback) return fail("loopback communicator (timing only): there are no other ranks to gather from");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  DevBuf<float> tmp_all, tmp_send;
  const float* vals = nullptr;
  if (e->n_comms == 2) {
    if (!e->vals_fresh && exchange_values(e, false)) return 1;
    HIP_OK(hipStreamSynchronize(e->stream));
    HIP_OK(hipStreamSynchronize(e->side));
    if (ipc_check(e)) return 1;
    vals = e->ipc.on ? e->ipc.set(e->ipc.vset) : e->planes;
  } else {
    HIP_OK(tmp_all.alloc((size_t)e->M * e->Ev));
    HIP_OK(tmp_send.alloc((size_t)e->Mloc * e->Ev));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy2DAsync(tmp_send.p, (size_t)e->Ev * 4, e->z, (size_t)e->D * 4, (size_t)e->D * 4, (size_t)e->Mloc, hipMemcpyDeviceToDevice, e->stream));
    if (e->P)
      HIP_OK(hipMemcpy2DAsync(tmp_send.p + e->D, (size_t)e->Ev * 4, e->theta, (size_t)e->P * 4, (size_t)e->P * 4, (size_t)e->Mloc,
                              hipMemcpyDeviceToDevice, e->stream));
    RCCL_OK(rccl().all_gather(tmp_send.p, tmp_all.p, (size_t)e->Mloc * e->Ev, ncclFloat, e->comm[0], e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    vals = tmp_all.p;
  }
  if (z_all) HIP_OK(hipMemcpy2D(z_all, (size_t)e->D * 4, vals, (size_t)e->Ev * 4, (size_t)e->D * 4, (size_t)e->M, hipMemcpyDeviceToHost));
  if (theta_all && e->P)
    HIP_OK(hipMemcpy2D(theta_all, (size_t)e->P * 4, vals + e->D, (size_t)e->Ev * 4, (size_t)e->P * 4, (size_t)e->M, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int64_t dibs_engine_gather_elems_per_rank(const dibs_engine* e) { return e ? (int64_t)e->Mloc * e->E : 0; }

extern "C" int dibs_engine_sync(dibs_engine* e) {
  if (!e) return fail("null engine");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  HIP_OK(hipStreamSynchronize(e->stream));
  if (e->profiling) drain_timers(e);
  if (const unsigned int code = take_join_err(e)) {
    e->flags_off = true;  // (a loop driven step by step from outside cannot be repeated here: the caller's steps since the last sync are lost)
    return join_failure(code, "; the steps since the last dibs_engine_sync are invalid (the engine uses events from now on)");
  }
  return 0;
}

extern "C" int64_t dibs_engine_theta_size(const dibs_engine* e) { return e ? e->P : 0; }

struct BufInfo {
  const void* p;
  int64_t bytes;
};
static BufInfo buf_info(const dibs_engine* e, int which) {
  const int64_t Ml = e->Mloc, dd = (int64_t)e->d * e->d;
  switch (which) {
    case DIBS_BUF_Z: return {e->z, Ml * e->D * 4};
    case DIBS_BUF_V_Z: return {e->vz, Ml * e->D * 4};
    case DIBS_BUF_THETA: return {e->theta, Ml * e->P * 4};
    case DIBS_BUF_V_THETA: return {e->vtheta, Ml * e->P * 4};
    case DIBS_BUF_SCORES: return {e->scores, Ml * dd * 4};
    case DIBS_BUF_LOGPROBS_Z: return {e->logprobs_z, Ml * e->S * 4};
    case DIBS_BUF_LOGPROBS_THETA: return {e->logprobs_th, Ml * e->S * 4};
    case DIBS_BUF_W_LIK: return {e->w_lik, Ml * dd * 4};
    case DIBS_BUF_W_ACYC: return {e->w_acyc, Ml * dd * 4};
    case DIBS_BUF_KXX: return {e->kz, Ml * e->M * 4};
    case DIBS_BUF_PHI_Z: return {e->phi_z, Ml * e->D * 4};
    case DIBS_BUF_PHI_THETA: return {e->phi_th, Ml * e->P * 4};
    case DIBS_BUF_BASELINE: return {e->baseline, Ml * 4};
    case DIBS_BUF_NODE_SCORES: return {e->node_scores, e->node_scores ? Ml * e->S * e->d * 8 : 0};
    case DIBS_BUF_PARENT_MASKS: return {e->masks, e->masks ? Ml * e->S * e->d * e->W * 8 : 0};
    case DIBS_BUF_GATHER: return {e->pack, (int64_t)e->M * e->E * 4};
    case DIBS_BUF_GRAD_Z: return {nullptr, Ml * e->D * 4};
    case DIBS_BUF_GRAD_THETA: return {nullptr, Ml * e->P * 4};
    default: return {nullptr, -1};
  }
}

extern "C" int64_t dibs_engine_buffer_bytes(const dibs_engine* e, int32_t which) { return e ? buf_info(e, which).bytes : -1; }

extern "C" int dibs_engine_read_buffer(dibs_engine* e, int32_t which, void* host, int64_t nbytes) {
  if (!e || !host) return fail("null argument");
  HIP_OK(hipSetDevice(e->cfg.device_id));
  HIP_OK(hipStreamSynchronize(e->stream));
  const BufInfo bi = buf_info(e, which);
  if (bi.bytes < 0) return fail("unknown buffer id");
  if (bi.bytes != nbytes) return fail("buffer size mismatch: expected " + std::to_string(bi.bytes) + " bytes");
  if (nbytes == 0) return 0;
  if (which == DIBS_BUF_GRAD_Z || which == DIBS_BUF_GRAD_THETA) {  // strided rows of the packed buffer (single-rank engine buffer)
    const size_t off = which == DIBS_BUF_GRAD_Z ? (size_t)e->D : (size_t)(2 * e->D + e->P);
    const size_t w = which == DIBS_BUF_GRAD_Z ? (size_t)e->D * 4 : (size_t)e->P * 4;
    HIP_OK(hipMemcpy2D(host, w, e->pack + (size_t)e->m0 * e->E + off, (size_t)e->E * 4, w, e->Mloc, hipMemcpyDeviceToHost));
    return 0;
  }
  if (which == DIBS_BUF_KXX && e->kt) {  // kxx = k_z + k_theta
    std::vector<float> a((size_t)e->Mloc * e->M), b(a.size());
    HIP_OK(hipMemcpy(a.data(), e->kz, a.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(b.data(), e->kt, a.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < a.size(); ++i) ((float*)host)[i] = a[i] + b[i];
    return 0;
  }
  HIP_OK(hipMemcpy(host, bi.p, nbytes, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int dibs_engine_set_profiling(dibs_engine* e, int32_t enable) {
  if (!e) return fail("null engine");
  hipStreamSynchronize(e->stream);
  drain_timers(e);
  e->profiling = enable != 0;
  e->profiling_concurrent = enable == 2;
  return 0;
}

extern "C" int dibs_engine_reset_timers(dibs_engine* e) {
  if (!e) return fail("null engine");
  hipStreamSynchronize(e->stream);
  drain_timers(e);
  for (int i = 0; i < DIBS_K_COUNT; ++i) {
    e->t_ms[i] = 0;
    e->t_n[i] = 0;
  }
  hipMemset(e->counters, 0, DIBS_N_COUNTERS * sizeof(unsigned long long));
  return 0;
}

extern "C" int dibs_engine_get_timers(dibs_engine* e, double* total_ms, int64_t* launches, int32_t n) {
  if (!e) return fail("null engine");
  hipStreamSynchronize(e->stream);
  drain_timers(e);
  for (int i = 0; i < n && i < DIBS_K_COUNT; ++i) {
    if (total_ms) total_ms[i] = e->t_ms[i];
    if (launches) launches[i] = e->t_n[i];
  }
  return 0;
}

extern "C" int dibs_engine_get_counters(dibs_engine* e, double* out, int32_t n) {
  if (!e || !out) return fail("null argument");
  HIP_OK(hipStreamSynchronize(e->stream));
  unsigned long long h[DIBS_N_COUNTERS];
  HIP_OK(hipMemcpy(h, e->counters, sizeof h, hipMemcpyDeviceToHost));
  for (int i = 0; i < n && i < DIBS_N_COUNTERS; ++i) out[i] = (double)h[i];
  return 0;
}


struct JointWorkGuard {
  JointWork jw;
  JointWorkGuard() { memset(&jw, 0, sizeof jw); }
  ~JointWorkGuard() { joint_free(&jw); }
};

extern "C" int dibs_score_graphs(dibs_engine* e, const int32_t* g, const float* theta, int32_t n, const float* x_ho,
                                 const int32_t* mask_ho, int32_t n_ho, float* out) {
  if (!e || !g || !x_ho || !out) return fail("null argument");
  if (n <= 0) return 0;
  HIP_OK(hipSetDevice(e->cfg.device_id));
  HIP_OK(hipStreamSynchronize(e->stream));
  const dibs_config& c = e->cfg;
  const int d = e->d;
  const size_t dd = (size_t)d * d;
  DevBuf<float> d_out;
  HIP_OK(d_out.alloc((size_t)n));
  dibs_engine::ScoreCache& sc = e->score_cache;
  const size_t n_x = (size_t)n_ho * d;
  const bool cached = sc.matches(x_ho, mask_ho, n_x);
  if (!cached) sc.valid = false;
  if (c.likelihood == DIBS_LIK_BGE) {
    BgeStats& st = sc.st;  // statistics of (x_ho, mask_ho)
    if (!cached) {
      if (bge_prepare(&st, c, d, n_ho, x_ho, mask_ho, e->has_mean_obs ? e->mean_obs.data() : nullptr)) return 1;
      sc.remember(x_ho, mask_ho, n_x);
    }
    const int W = e->W, CH = 512;
    DevBuf<uint64_t> d_masks;
    DevBuf<double> d_ns;
    DevBuf<uint4> q_list;
    DevBuf<unsigned int> q_counts;
    HIP_OK(d_masks.alloc((size_t)d * CH * W));
    HIP_OK(d_ns.alloc((size_t)d * CH));
    HIP_OK(q_list.alloc((size_t)BGE_NQ * d * CH * bge_entry_u4(W)));
    HIP_OK(q_counts.alloc((size_t)BGE_NQ));
    const BgeQueues sq{q_list.p, q_counts.p, (uint32_t)(d * CH)};  // scratch queues for this call
    const BgeParams bp = st.params();
    std::vector<uint64_t> hm((size_t)d * CH * W);
    for (int q0 = 0; q0 < n; q0 += CH) {
      const int S = n - q0 < CH ? n - q0 : CH;
      std::fill(hm.begin(), hm.end(), 0ull);
      for (int s = 0; s < S; ++s)
        for (int i = 0; i < d; ++i)
          for (int j = 0; j < d; ++j)
            if (i != j && g[(size_t)(q0 + s) * dd + (size_t)i * d + j] != 0) hm[((size_t)j * S + s) * W + (i >> 6)] |= 1ull << (i & 63);
      HIP_OK(hipMemcpy(d_masks.p, hm.data(), (size_t)d * S * W * 8, hipMemcpyHostToDevice));
      HIP_OK(hipMemsetAsync(sq.counts, 0, BGE_NQ * sizeof(unsigned int), e->stream));
      bge_launch_sample(false, e->stream, nullptr, d_masks.p, d_ns.p, bp, Key2{0, 0}, 0, 1, 1, d, S, W, 0, sq,
                        KmatFuse{nullptr, nullptr, 0, 0, 0, 0.f, 0.f, nullptr, 0u});
      bge_launch_chol(e->stream, d_ns.p, bp, sq, d, S, nullptr);
      bge_launch_sum_nodes(e->stream, d_ns.p, d_out.p + q0, d, S);
      HIP_OK(hipStreamSynchronize(e->stream));
    }
  } else if (c.likelihood == DIBS_LIK_LINGAUSS || c.likelihood == DIBS_LIK_DENSENN) {
    if (!theta) return fail("theta required");
    const bool nn = c.likelihood == DIBS_LIK_DENSENN;
    struct { JointWork& jw; } jg{sc.jw};
    if (!cached) {
      if (joint_set_data(&jg.jw, x_ho, mask_ho, n_ho, d)) return fail("joint_set_data failed");
      if (!nn && !joint_lin_fast_path(d, n_ho, e->tune.lin_gram) && joint_lin_set_gram(&jg.jw, x_ho, mask_ho, n_ho, d)) return fail("LinearGaussian: Gram matrices: hipMalloc failed");
      sc.remember(x_ho, mask_ho, n_x);
    }
    const size_t P = nn ? (size_t)e->P : dd;
    DevBuf<float> d_th;
    DevBuf<int32_t> d_g;
    HIP_OK(d_th.alloc((size_t)n * P));
    HIP_OK(d_g.alloc((size_t)n * dd));
    HIP_OK(hipMemcpy(d_th.p, theta, (size_t)n * P * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_g.p, g, (size_t)n * dd * 4, hipMemcpyHostToDevice));
    if (nn) {
      const NNParams np_ = nn_params(c);
      if (joint_nn_score_given(jg.jw, d_th.p, d_g.p, d_out.p, n, d, n_ho, np_, P, e->stream)) return fail("DenseNonlinearGaussian: scratch area: hipMalloc failed");
    } else {
      joint_lin_score_given(jg.jw, d_th.p, d_g.p, d_out.p, n, d, n_ho, (float)c.lin_obs_noise, (float)c.lin_mean_edge,
                            (float)c.lin_sig_edge, e->stream);
    }
    HIP_OK(hipStreamSynchronize(e->stream));
  } else {
    return fail("dibs_score_graphs: likelihood not supported yet");
  }
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpy(out, d_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return 0;
}
