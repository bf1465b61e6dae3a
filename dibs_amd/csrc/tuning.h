// Every environment switch of libdibs_hip.so.  They are read ONCE per engine, by dibs_engine_create, into the engine's DibsTuning; the
// launchers of the kernel families get it through their launch structs.  Nothing on the per-step path calls getenv, and an engine keeps the
// settings it was created with (tests create engines under different settings in one process).
//   precision A/B (the bench line records the trade: profiles/round6_bench*.json)
//     DIBS_ACYC_F32=1       acyclicity matrix powers on the f32 MFMA (24-bit operands) instead of the two-piece f16 scheme (22 bits), 33 <= n_vars <= 112
//     DIBS_ACYC_BF16=1      ... on the three-piece bf16 scheme (24 bits)
//     DIBS_ACYC_HFW_MAX=n   65 <= n_vars <= 112: the f16 kernel up to n variables, the bf16 kernel beyond (default 112)
//     DIBS_LIN_F32=1        LinearGaussian log-probabilities on the f32 MFMA instead of the two-piece f16 scheme (33 <= n_vars <= 64)
//     DIBS_NN_F32=1         DenseNonlinearGaussian log-probabilities on the f32 MFMA
//   path selection (tests force the alternative paths of one size through these)
//     DIBS_LIN_GRAM=1       LinearGaussian on the Gram-matrix kernels at every size
//     DIBS_KMAT_TILED_MIN=n tiled kernel matrix from n particles (default 128; must agree on all ranks)
//     DIBS_KMAT_T64_MIN=n   64 x 64 tiles from n particles (default 512)
//     DIBS_NO_KMAT_FUSE=1   the latent kernel matrix as a launch of its own (not inside k_bge_sample / k_particle_grad)
//     DIBS_NO_KMAT_GRAD=1   ... not as tile units inside k_particle_grad (falls back to the other fused forms)
//     DIBS_NO_ACYC_STREAM2=1  everything on the engine stream (no second stream)
//     DIBS_NO_FLAGS=1       fork / join of the second stream by events (not by flags polled inside kernels)
//     DIBS_FLAGS_MULTI=1    flags also while several engines live in the process (what a rank of a real run does; scripts/gpu_shard_scaling.py)
//   DIBS_IPC_TIMEOUT_MS=n   mapped-memory exchange: how long a rank waits for its peers' rows (default 10 000)
#pragma once
#include <stdlib.h>

enum { DIBS_PIPE_DEFAULT = 0, DIBS_PIPE_F32 = 1, DIBS_PIPE_BF16 = 2 };

struct DibsTuning {
  int acyc_pipe = DIBS_PIPE_DEFAULT;
  int acyc_hfw_max = 112;
  bool lin_f32 = false, lin_gram = false, nn_f32 = false;
  int kmat_tiled_min = 128, kmat_t64_min = 512;
  bool no_kmat_fuse = false, no_kmat_grad = false, no_stream2 = false, no_flags = false, flags_multi = false;
  long ipc_timeout_ms = 10000;
};

inline DibsTuning dibs_tuning_from_env() {
  DibsTuning t;
  auto on = [](const char* n) { return getenv(n) != nullptr; };
  auto num = [](const char* n, long dflt) {
    const char* v = getenv(n);
    return v ? atol(v) : dflt;
  };
  t.acyc_pipe = on("DIBS_ACYC_F32") ? DIBS_PIPE_F32 : (on("DIBS_ACYC_BF16") ? DIBS_PIPE_BF16 : DIBS_PIPE_DEFAULT);
  t.acyc_hfw_max = (int)num("DIBS_ACYC_HFW_MAX", 112);
  t.lin_f32 = on("DIBS_LIN_F32");
  t.lin_gram = on("DIBS_LIN_GRAM");
  t.nn_f32 = on("DIBS_NN_F32");
  t.kmat_tiled_min = (int)num("DIBS_KMAT_TILED_MIN", 128);
  t.kmat_t64_min = (int)num("DIBS_KMAT_T64_MIN", 512);
  t.no_kmat_fuse = on("DIBS_NO_KMAT_FUSE");
  t.no_kmat_grad = on("DIBS_NO_KMAT_GRAD");
  t.no_stream2 = on("DIBS_NO_ACYC_STREAM2");
  t.no_flags = on("DIBS_NO_FLAGS");
  t.flags_multi = on("DIBS_FLAGS_MULTI");
  t.ipc_timeout_ms = num("DIBS_IPC_TIMEOUT_MS", 10000);
  return t;
}
