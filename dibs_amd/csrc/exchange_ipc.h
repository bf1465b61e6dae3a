// The exchange of a particle-sharded run WITHOUT a collective library: every rank maps its peers' exchange arenas (hipIpcMemHandle) and
// the all-gather is each rank WRITING its rows into every peer's buffer, with per-source sequence words for completion.
// No reference counterpart (the reference has no multi-device code); the arithmetic of a step is svgd.py:226-267 / 673-721.
//
// Why beside RCCL: ranks that share ONE device (RCCL refuses a communicator with a duplicate GPU) -- several ranks per GPU, and the
// single-GPU execution of the N > 1 loop of dibs_engine_run_sharded (tests/test_gpu_ipc.py, bench.py with DIBS_COMM=ipc).  Devices with
// peer access work the same way (hipIpcOpenMemHandle maps the peer's memory; the stores travel over xGMI).
//
// Arena of a rank (one hipMalloc, one handle):
//   [ flags: u32 [2 channels][IPC_MAX_RANKS] | error words of the chunk agreement: 2 x [IPC_MAX_RANKS] x 16 B; 1 KB in all ]
//   [ pack[0] | pack[1] : M x E floats each ] [ set[0] | set[1] : 2 x M x Ev floats each ]
//   pack[p]   packed rows [z | grad_z | theta | grad_theta] of ALL particles (one exchange per step)
//   set[p]    planes of the overlapped protocol: plane 0 = values [z | theta], plane 1 = gradients, of ALL particles
// Every buffer exists twice.  Exchange number n of a channel writes into copy n & 1: a rank that is already in exchange n has passed its wait
// of exchange n - 1, i.e. every peer had pushed n - 1, which each of them did -- in stream order -- after it finished reading copy n & 1 in
// exchange n - 2.  So no acknowledgement travels back: one push and one flag per peer and exchange.
//   flags[c][s]  the number of the last exchange of channel c whose rows from rank s are complete in THIS arena (written by rank s).
//                Channel 0: the engine stream (packed rows, or gradient rows); channel 1: the side stream (values).
// Per exchange a rank launches, on the channel's stream:
//   k_ipc_push         copies its rows into every destination arena (grid.y = destination); every block ends with a system-scope release
//   k_ipc_signal_wait  one wave, lane s: stores n into peer s's flags[c][me], then polls flags[c][s] of its own arena until it reads >= n
//                      (bounded: IPC_WAIT_TICKS, then *err = 3 and the host fails the chunk)
// The consumer kernels behind it start with the usual kernel-start acquire.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IPC_MAX_RANKS 16
#define IPC_FLAG_BYTES 1024
#define IPC_AGREE_OFF 256  /* [2 copies][IPC_MAX_RANKS] x 16 bytes: the ranks' error words of a chunk (run_sharded's agreement) */
#define IPC_MAGIC 0x43504944u /* "DIPC" */

struct IpcPeers {
  char* base[IPC_MAX_RANKS];  // arena of rank r as mapped in THIS process (own rank: the allocation itself)
};

// blob a rank hands to its peers (DIBS_IPC_HANDLE_BYTES = 128): header + the arena's memory handle
struct IpcBlob {
  uint32_t magic, abi, rank, n_ranks;
  uint64_t arena_bytes, pack_elems, set_elems;
  int32_t device_id, pid;
  uint8_t pad[64 - 48];
  hipIpcMemHandle_t handle;  // 64 bytes
};
static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t");
static_assert(sizeof(IpcBlob) == 128, "IpcBlob must be DIBS_IPC_HANDLE_BYTES");

struct IpcComm {
  bool on = false;
  int n_ranks = 0, rank = 0;
  char* arena = nullptr;  // own allocation
  size_t arena_bytes = 0, pack_elems = 0, set_elems = 0;
  IpcPeers peers{};
  bool opened[IPC_MAX_RANKS] = {};
  uint32_t seq[2] = {0u, 0u};    // exchanges issued so far per channel
  uint32_t pack_seq = 0;          // packed-row exchanges (selects pack[0 / 1])
  uint32_t agree_seq = 0;         // chunk agreements (selects the copy of the error words)
  int vset = 0;                   // plane set of the latest value exchange (the gradient rows of that step go to the same set)
  unsigned int* err = nullptr;    // pinned host word raised by k_ipc_signal_wait on a time-out
  unsigned long long wait_ticks = 1000000000ull;  // 10 s of the 100 MHz clock
  float* pack(int p) const { return reinterpret_cast<float*>(arena + IPC_FLAG_BYTES) + (size_t)p * pack_elems; }
  float* set(int p) const { return reinterpret_cast<float*>(arena + IPC_FLAG_BYTES) + 2 * pack_elems + (size_t)p * set_elems; }
  size_t pack_off(int p) const { return IPC_FLAG_BYTES + (size_t)p * pack_elems * 4; }
  size_t set_off(int p) const { return IPC_FLAG_BYTES + (2 * pack_elems + (size_t)p * set_elems) * 4; }
};

// rows src[0 .. n4) (float4) -> byte offset dst_off of every destination arena; grid = (blocks, n_dst), block = 256.
// dst_first = 1: every rank but `me` (the rows are already in place in the own arena); 0: all ranks, own arena included (values from vsend)
__global__ __launch_bounds__(256) void k_ipc_push(IpcPeers P, int me, int n_ranks, int include_self, const float4* __restrict__ src, size_t dst_off,
                                                  size_t n4) {
  int r = (int)blockIdx.y;
  if (!include_self && r >= me) ++r;  // (n_ranks - 1 destinations, own rank skipped)
  if (r < n_ranks) {
    float4* __restrict__ dst = reinterpret_cast<float4*>(P.base[r] + dst_off);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
  }
  // the stores are complete and written back past the L2 before the kernel ends (the signal is the NEXT kernel of the stream): every wave
  // waits for its own stores, then ONE system-scope release per block (a release per wave would walk the L2 once per wave)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __threadfence_system();
}

// one wave: lane s < n_ranks, s != me announces exchange `seq` of `channel` to rank s and waits for rank s's announcement
__global__ void k_ipc_signal_wait(IpcPeers P, int me, int n_ranks, int channel, unsigned int seq, unsigned long long max_ticks, unsigned int* err) {
  const int s = (int)threadIdx.x;
  if (s >= n_ranks || s == me) return;
  unsigned int* theirs = reinterpret_cast<unsigned int*>(P.base[s]) + channel * IPC_MAX_RANKS + me;
  const unsigned int* mine = reinterpret_cast<const unsigned int*>(P.base[me]) + channel * IPC_MAX_RANKS + s;
  // (RELAXED: k_ipc_push in front of this kernel has released the rows; the consumers behind it begin with the kernel-start acquire.  A
  //  release here would write the L2 back once more, an acquire inside the loop would invalidate it under the other kernels' feet)
  __hip_atomic_store(theirs, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long t0 = wall_clock64();
  while ((int)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > max_ticks) {
      if (err) *err = 3u;
      break;
    }
  }
}
