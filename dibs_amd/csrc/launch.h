// Host-side launchers of the kernel families that live in their own translation units (tu_*.hip): plain arguments, no templates,
// so engine.hip does not instantiate those kernels.
#pragma once
#include "common.h"
#include "kernels_kmat.h"
#include "kernels_bge.h"
#include "tuning.h"

// kernels that may need more than the default 64 KiB of dynamic LDS: raises hipFuncAttributeMaxDynamicSharedMemorySize once per
// (device, kernel) and size increase (engine.hip; thread-safe -- engines on several devices / host threads share the table)
void dibs_allow_lds(const void* kernel, size_t bytes);

// ---- tu_bge.hip --------------------------------------------------------------------------------------
// sampling + queueing (sample = false: parent sets given in `masks`); kf.z != null appends the kernel-matrix blocks
void bge_launch_sample(bool sample, hipStream_t stream, const uint32_t* thr, uint64_t* masks, double* node_scores, const BgeParams& bp,
                       Key2 carry, int m0, int M, int Mloc, int d, int S, int W, int layout, const BgeQueues& qs, const KmatFuse& kf);
size_t bge_sample_lds_bytes(int d, int S, int W);
void bge_launch_chol(hipStream_t stream, double* node_scores, const BgeParams& bp, const BgeQueues& qs, int d, int S,
                     unsigned long long* counters);
void bge_launch_sum_nodes(hipStream_t stream, const double* node_scores, float* out, int d, int S);

// ---- tu_acyc.hip -------------------------------------------------------------------------------------
struct AcycLaunch {
  hipStream_t stream;
  const float* scores;
  float* part;            // [Mloc][nblk][d*d] partial sums of the blocks
  float* w_acyc;          // [Mloc][d*d] mean over the chains (k_acyc_reduce, same stream)
  float* big;             // n_vars > 112: acyc_big_elems() floats (matrix powers through global memory); null otherwise
  Key2 carry;
  int m0, M, Mloc, d, Sa, cpb, units, nblk;  // units != Sa: chains are taken in Threefry pairs
  float alpha, tau;
  int layout, tiny;
  hipEvent_t ev_start, ev_stop;  // profiling: kernel-level start / stop time stamps of the matrix-power kernel (see acyc_power_takes_events); else null
  const float* eas;       // [Mloc][d*d] exp(-alpha scores) (k_edge_scores), read by k_acyc_hf when tau == 1; may be null (then it is evaluated in place)
  int pipe = DIBS_PIPE_DEFAULT, hfw_max = 112;  // the engine's DibsTuning::acyc_pipe / acyc_hfw_max
};
// true: acyc_launch_power is ONE kernel and stamps a.ev_start / a.ev_stop around it (hipExtLaunchKernelGGL: the kernel's own start and
// end as the profiler sees them, without the dispatch latency an event pair recorded around the launch includes)
bool acyc_power_takes_events(const AcycLaunch& a);
void acyc_launch_power(const AcycLaunch& a);   // matrix powers -> per-block partial sums (n_vars > 112: everything, straight into w_acyc)
void acyc_launch_reduce(const AcycLaunch& a);  // partial sums -> w_acyc (same stream)
size_t acyc_big_elems(int Mloc, int d, int Sa);
