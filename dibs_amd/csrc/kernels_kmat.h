// latent / parameter kernel matrix slab (device function shared by k_kmat and the ride-along blocks of k_bge_sample)
#pragma once
#include "common.h"

// The latent kernel matrix only needs z, which is final when a step starts.  On a single rank its blocks ride along in the
// k_bge_sample launch (extra blockIdx.x range): that kernel is bound by VALU issue, k_kmat by the latency of the far cache
// levels, so the two overlap almost for free.  (Several ranks: the rows of the other ranks arrive with the all-gather, the
// kernel matrix stays in phase B.)
#define KMAT_BT 16
struct KmatFuse {
  const float* z;   // [M, len] (null: nothing fused)
  float* kout;      // [M, M]
  int len, M, nbx;  // nbx: first blockIdx.x of the kernel-matrix range
  float scale, h;
  // (round 5) fork flag published by this launch: its first block can only start when the edge kernel in front of it has ended -- stores
  // released -- so block (0, 0) stores pub_seq to *pub_flag at agent scope and the second stream's waiter (k_wait_flag) lets the acyclicity
  // chain go.  null: nothing to publish
  unsigned int* pub_flag;
  unsigned int pub_seq;
};

// ------------------------------------------------------------------------------------------------
// K8a' (round 5)  the same matrix, TILED: a block takes a 32 x 32 tile of (a, b) pairs and a range of KT_CH-element chunks, stages the
//     32 + 32 row pieces of a chunk in LDS (each row read once per tile instead of once per pair: 1/8 of the L2 traffic of kmat_block, which
//     is what bounded it -- 164 MB per launch at 128 particles) and every thread accumulates a 4 x 4 block of squared distances on packed
//     FMAs (direct differences as before).  Per chunk the 16 waves' partial sums (floats, all terms >= 0) meet in LDS and are added in
//     double into the thread that owns the pair; with the whole chunk range in one unit (many tiles) that thread applies exp(-./h), mirrors
//     the upper triangle (symmetric = one rank holds all particles: only tiles tb >= ta are computed) and forms kz + kt for the joint
//     models; with few tiles the range is cut into nsplit pieces (kmat_pick_nsplit) whose sums k_kmat_finish adds.
//     Grouping: a wave's 16-element sum is a float; everything above it is added in double, where sums of a few thousand floats of similar
//     size are exact -- so the entry does not depend on how the chunk range was cut into pieces, on which unit finished last, or on
//     whether the slab is the symmetric whole (one rank) or a rank's rows (sharded): rank engines and the single-rank engine agree bit
//     for bit, as they did with the direct kernel.
//     Bound: the per-CU fetch rate (64 KB per chunk at ~11 B/clk/CU when every CU fetches) -- 3.5 us per chunk against 1.7 us of VALU work.
//     Measured (profiles/round5_cfg{3,4,5}_kernel_stats.csv): config 4 (1 024 particles) tile + finish 227 us (direct kernel, round 3: 434), 597 -> 664 steps/s;
//     config 5 108.5 -> 115; config 3 2 290 -> 2 330.
//     reference: kernel.py:20-30 / 52-71, svgd.py:165-176 / 537-551
// ------------------------------------------------------------------------------------------------
#define KT_CH 256
#define KT_T 32
#define KT_LD (KT_CH + 4)  // row stride (floats): 16-byte aligned, rows 4 banks apart (conflict-free 16-byte reads of 8 different rows)
struct KmatTile {
  const float* x;        // rows [M][stride], the segment starts at `off`; all rows within 4 GiB of x (32-bit byte offsets)
  size_t stride, off;
  int len;
  double* part;          // [nsplit][Mloc][M] partial squared distances (nsplit > 1), in double: see the note on grouping below
  int m0, Mloc, M, nchunk, nta, ntb, symmetric;
  // the chunk range of a tile is cut into nsplit pieces of cps chunks, one unit each.  nsplit == 1: the unit holds the whole distance and
  // writes the matrix entries itself (exp, mirror, kadd + k); otherwise k_kmat_finish adds the pieces.
  int nsplit, cps;
  float scale, h;
  float* kout;
  const float* kadd;
  float* ksum;
  // nsplit > 1 and tile_ctr != null: no second pass -- every unit stores its piece at agent scope and counts itself on its tile's counter, the LAST one of
  // a tile adds the pieces (fixed order: the result does not depend on which unit came last) and writes the entries.  Lets the units ride
  // in another kernel's launch (k_particle_grad).  Counters are zero between launches.
  unsigned int* tile_ctr;
};
// host: how many pieces.  Cost model in chunk times: the blocks (one per CU, 256 of them) take ceil(units / 256) rounds of cps chunks plus
// the exposed first fetch; the pieces cost a second pass over nsplit matrices.
__host__ inline int kmat_pick_nsplit(int tiles, int nchunk, int max_ns) {
  int best = 1;
  double bc = 1e30;
  for (int ns = 1; ns <= nchunk && ns <= max_ns; ++ns) {
    const int cps = (nchunk + ns - 1) / ns, ns_eff = (nchunk + cps - 1) / cps;
    if (ns_eff != ns) continue;
    const double c = (double)((tiles * ns + 255) / 256) * (cps + 0.6) + (ns > 1 ? 0.3 + 0.05 * ns : 0.0);
    if (c < bc - 1e-9) bc = c, best = ns;
  }
  return best;
}
// (host) the tile kernel addresses the rows with 32-bit byte offsets
__host__ inline bool kmat_tile_addressable(size_t rows, size_t stride, size_t off, size_t len) {
  return (rows * stride + off + len) * 4 < ((size_t)1 << 32);
}
__host__ __device__ inline size_t kmat_tile_lds_bytes() { return (size_t)2 * KT_T * KT_LD * 4; }  // (>= 16 x 1024 floats of partial sums)
__host__ __device__ inline int kmat_tile_count(int nta, int ntb, int symmetric) { return symmetric ? nta * (nta + 1) / 2 : nta * ntb; }
__host__ __device__ inline int kmat_nchunk(int len) { return (len + KT_CH - 1) / KT_CH; }

// (tile, chunk) units by a block of KT_NT = 1024 threads.  Staging: 64 row pieces as float4 (16-byte aligned rows) or scalars, 16 floats per
// thread, fetched for the NEXT unit of the block while the current one is computed.  Wave w takes the elements 16 w .. 16 w + 15 of the chunk;
// lane (ag, bg) = (lane / 8, lane % 8) the 4 x 4 pairs (ag + 8 i, bg + 8 j): 8 LDS reads of 16 bytes feed 64 packed instructions (rows 1
// apart are 4 banks apart: conflict-free), so a unit is VALU-bound, not LDS-bound.  The 16 waves' partial sums meet in LDS and are added in
// double.
#define KT_NT 1024
#define KT_RV (2 * KT_T * KT_CH / KT_NT)  // staged floats per thread
struct KtUnit {
  int a0, b0, c0, clen, chunk;
};
__device__ __forceinline__ KtUnit kt_unit(const KmatTile& K, int tile, int chunk) {
  int ta, tb;
  if (K.symmetric) {
    ta = 0;
    int t = tile;
    while (t >= K.nta - ta) { t -= K.nta - ta; ++ta; }
    tb = ta + t;
  } else {
    ta = tile / K.ntb;
    tb = tile - ta * K.ntb;
  }
  const int c0 = chunk * KT_CH;
  return KtUnit{ta * KT_T, tb * KT_T, c0, K.len - c0 < KT_CH ? K.len - c0 : KT_CH, chunk};
}
template <bool VEC>
__device__ __forceinline__ void kt_fetch(const KmatTile& K, const KtUnit& U, int tid, float (&v)[KT_RV]) {
  auto grow = [&](int row) {
    return row < KT_T ? K.m0 + (U.a0 + row < K.Mloc ? U.a0 + row : K.Mloc - 1) : (U.b0 + row - KT_T < K.M ? U.b0 + row - KT_T : K.M - 1);
  };
  // uniform base + 32-bit byte offsets (the SGPR-base form of global_load: no 64-bit address per load; this hipcc lowers the b64 / b128
  // buffer-load builtins to a single dword).  Clamped offsets here + zeroing at staging time keep the batch free of branches and waits,
  // i.e. in flight together.
  const char* base = reinterpret_cast<const char*>(K.x + K.off + (size_t)U.c0);
  if (VEC) {  // 16-byte aligned rows, segment length a multiple of 4: a float4 is all inside or all outside
#pragma unroll
    for (int u = 0; u < KT_RV / 4; ++u) {
      const int i = u * KT_NT + tid, row = i / (KT_CH / 4), e = (i - row * (KT_CH / 4)) * 4;
      const uint32_t bo = ((uint32_t)grow(row) * (uint32_t)K.stride + (uint32_t)(e < U.clen ? e : 0)) * 4u;
      const float4 t = *reinterpret_cast<const float4*>(base + bo);
      v[4 * u] = t.x, v[4 * u + 1] = t.y, v[4 * u + 2] = t.z, v[4 * u + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int u = 0; u < KT_RV; ++u) {
      const int i = u * KT_NT + tid, row = i / KT_CH, e = i - row * KT_CH;
      const uint32_t bo = ((uint32_t)grow(row) * (uint32_t)K.stride + (uint32_t)(e < U.clen ? e : 0)) * 4u;
      v[u] = *reinterpret_cast<const float*>(base + bo);
    }
  }
}
template <bool VEC>
__device__ __forceinline__ void kt_stage(float* __restrict__ smem, int clen, int tid, const float (&v)[KT_RV]) {
  if (VEC) {
#pragma unroll
    for (int u = 0; u < KT_RV / 4; ++u) {
      const int i = u * KT_NT + tid, row = i / (KT_CH / 4), e = (i - row * (KT_CH / 4)) * 4;
      const bool in = e < clen;
      *reinterpret_cast<float4*>(smem + row * KT_LD + e) =
          make_float4(in ? v[4 * u] : 0.f, in ? v[4 * u + 1] : 0.f, in ? v[4 * u + 2] : 0.f, in ? v[4 * u + 3] : 0.f);
    }
  } else {
#pragma unroll
    for (int u = 0; u < KT_RV; ++u) {
      const int i = u * KT_NT + tid, row = i / KT_CH, e = i - row * KT_CH;
      smem[row * KT_LD + e] = e < clen ? v[u] : 0.f;
    }
  }
}
// units first, first + step, ... of nta (x ntb) tiles x nsplit pieces; one flat sequence of (unit, chunk) steps with the next step's rows
// in flight
template <bool VEC>
__device__ __forceinline__ void kmat_tile_loop(float* __restrict__ smem, const KmatTile& K, int first, int step, int units, int tid) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int lane = tid & 63, wave = tid >> 6, ag = lane >> 3, bg = lane & 7;
  const float4* pa = reinterpret_cast<const float4*>(smem + ag * KT_LD) + wave * (KT_CH / 4 / 16);
  const float4* pb = reinterpret_cast<const float4*>(smem + (KT_T + bg) * KT_LD) + wave * (KT_CH / 4 / 16);
  float v[KT_RV];
  int u = first, c = (u % K.nsplit) * K.cps;
  KtUnit U = kt_unit(K, u / K.nsplit, c);
  kt_fetch<VEC>(K, U, tid, v);
  double tot = 0.0;
  for (;;) {
    kt_stage<VEC>(smem, U.clen, tid, v);
    __syncthreads();
    const KtUnit Ucur = U;
    const int sp = u % K.nsplit, c_hi = (sp + 1) * K.cps < K.nchunk ? (sp + 1) * K.cps : K.nchunk;
    int un = u, cn = c + 1;
    if (cn >= c_hi) {
      un = u + step;
      cn = (un % K.nsplit) * K.cps;
    }
    const bool more = un < units;  // (block-uniform)
    if (more) {
      U = kt_unit(K, un / K.nsplit, cn);
      kt_fetch<VEC>(K, U, tid, v);
    }
    f32x2 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = f32x2{0.f, 0.f};
#pragma unroll
    for (int it = 0; it < KT_CH / 4 / 16; ++it) {
      float4 x[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = pa[i * 8 * (KT_LD / 4) + it];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 y = pb[j * 8 * (KT_LD / 4) + it];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x2 t = f32x2{x[i].x, x[i].y} - f32x2{y.x, y.y};
          acc[i * 4 + j] = __builtin_elementwise_fma(t, t, acc[i * 4 + j]);
          t = f32x2{x[i].z, x[i].w} - f32x2{y.z, y.w};
          acc[i * 4 + j] = __builtin_elementwise_fma(t, t, acc[i * 4 + j]);
        }
      }
    }
    __syncthreads();  // (the staged rows are dead: their space takes the 16 x 1024 partial sums)
#pragma unroll
    for (int q = 0; q < 16; ++q) smem[wave * 1024 + q * 64 + lane] = acc[q].x + acc[q].y;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += (double)smem[w * 1024 + tid];
    if (un != u) {  // (block-uniform) last chunk of the unit: thread tid holds the pair (a, b)
      const int q = tid >> 6, a = Ucur.a0 + ag + 8 * (q >> 2), b = Ucur.b0 + bg + 8 * (q & 3);
      if (a < K.Mloc && b < K.M) {
        if (K.nsplit > 1) {
          // (tile_ctr: agent-scope store -- written through, complete at agent scope once its wave has waited for vmcnt(0) below --
          //  instead of a plain store + release fence: the fence writes the XCD's whole L2 back, +2 us on the launch the units ride in)
          if (K.tile_ctr)
            __hip_atomic_store(K.part + ((size_t)sp * K.Mloc + a) * K.M + b, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else
            K.part[((size_t)sp * K.Mloc + a) * K.M + b] = tot;
        } else {
          const float kv = (float)((double)K.scale * exp(-tot / (double)K.h));
          const bool mirror = K.symmetric && Ucur.b0 > Ucur.a0;  // (diagonal tiles hold both orders of their pairs: the same sums)
          K.kout[(size_t)a * K.M + b] = kv;
          if (mirror) K.kout[(size_t)b * K.M + a] = kv;
          if (K.ksum) {
            K.ksum[(size_t)a * K.M + b] = K.kadd[(size_t)a * K.M + b] + kv;
            if (mirror) K.ksum[(size_t)b * K.M + a] = K.kadd[(size_t)b * K.M + a] + kv;
          }
        }
      }
      tot = 0.0;
      if (K.nsplit > 1 && K.tile_ctr) {  // (block-uniform)
        unsigned int* const ctr = K.tile_ctr + u / K.nsplit;
        unsigned int* const last = reinterpret_cast<unsigned int*>(smem + 16 * 1024);  // (first word behind the partial sums)
        // every wave drains its OWN piece stores before the barrier: a workgroup-scope barrier does not wait for vmcnt on gfx950 (the
        // compiler emits store sc1 -> s_barrier -> atomic with no wait in between), and the counter must not run ahead of the pieces
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          const unsigned int done = atomicAdd(ctr, 1u) + 1u;
          if (done == (unsigned int)K.nsplit) atomicExch(ctr, 0u);
          *last = done == (unsigned int)K.nsplit;
        }
        __syncthreads();
        if (*last && a < K.Mloc && b < K.M) {
          // (agent-scope loads: this XCD's L2 may hold last step's lines of `part`; the other units released theirs to memory)
          double t2 = 0.0;
          const double* pp = K.part + (size_t)a * K.M + b;
          const size_t cs = (size_t)K.Mloc * K.M;
          // (all pieces of a pair requested together, 16 at a time: each is a trip past the L2)
          for (int s16 = 0; s16 < K.nsplit; s16 += 16) {
            double pv[16];
#pragma unroll
            for (int w = 0; w < 16; ++w) {
              const int sq = s16 + w < K.nsplit ? s16 + w : K.nsplit - 1;
              pv[w] = __hip_atomic_load(pp + (size_t)sq * cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int w = 0; w < 16; ++w) t2 += s16 + w < K.nsplit ? pv[w] : 0.0;
          }
          const float kv = (float)((double)K.scale * exp(-t2 / (double)K.h));
          const bool mirror = K.symmetric && Ucur.b0 > Ucur.a0;
          K.kout[(size_t)a * K.M + b] = kv;
          if (mirror) K.kout[(size_t)b * K.M + a] = kv;
          if (K.ksum) {
            K.ksum[(size_t)a * K.M + b] = K.kadd[(size_t)a * K.M + b] + kv;
            if (mirror) K.ksum[(size_t)b * K.M + a] = K.kadd[(size_t)b * K.M + a] + kv;
          }
        }
      }
    }
    if (!more) break;
    u = un;
    c = cn;
    __syncthreads();
  }
}
__device__ __forceinline__ void kmat_tile_block(float* __restrict__ smem, const KmatTile& K, int first, int step, int tid) {
  const int units = kmat_tile_count(K.nta, K.ntb, K.symmetric) * K.nsplit;
  if (first >= units) return;
  // (two copies of the loop: merged, the compiler shares the tails of the two fetch batches and waits on every load)
  const bool vec = (((K.stride | K.off | (size_t)K.len) & 3) == 0) && ((reinterpret_cast<uintptr_t>(K.x) & 15) == 0);
  if (vec)
    kmat_tile_loop<true>(smem, K, first, step, units, tid);
  else
    kmat_tile_loop<false>(smem, K, first, step, units, tid);
}

// ------------------------------------------------------------------------------------------------
// K8a''  the same units on 64 x 64 tiles (many particles): half the bytes fetched per pair -- the 32 x 32 kernel is bound by the per-CU
//     fetch rate.  512 threads; chunks of 128 elements (LDS 128 rows x 132 floats = 66 KB + 64 KB of partial sums); wave w takes the elements 16 w .. 16 w + 15
//     of the chunk, lane (ag, bg) the 8 x 8 pairs (ag + 8 i, bg + 8 j): 16 LDS reads of 16 bytes feed 256 packed instructions.  A wave's
//     16-element float sum covers the SAME 16 aligned elements in the same order as in the 32 x 32 kernel and everything above it is added in
//     double, so the entries are bit-identical to that kernel's.  Needs float4-addressable rows (else the 32 x 32 kernel runs).
// ------------------------------------------------------------------------------------------------
#define KT2_T 64
#define KT2_CH 128
#define KT2_LD (KT2_CH + 4)
#define KT2_NT 512
#define KT2_RV (2 * KT2_T * KT2_CH / KT2_NT)  // staged floats per thread (32)
__host__ __device__ inline size_t kmat_tile64_lds_bytes() { return (size_t)2 * KT2_T * KT2_LD * 4 + 8 * 2048 * 4; }  // (staged rows + partial sums)
__host__ __device__ inline int kmat_nchunk64(int len) { return (len + KT2_CH - 1) / KT2_CH; }
__host__ inline bool kmat_tile64_ok(const float* x, size_t stride, size_t off, size_t len) {
  return ((stride | off | len) & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
}
__device__ __forceinline__ void kt2_fetch(const KmatTile& K, int a0, int b0, int c0, int clen, int tid, float (&v)[KT2_RV]) {
  const char* base = reinterpret_cast<const char*>(K.x + K.off + (size_t)c0);
#pragma unroll
  for (int u = 0; u < KT2_RV / 4; ++u) {
    const int i = u * KT2_NT + tid, row = i / (KT2_CH / 4), e = (i - row * (KT2_CH / 4)) * 4;
    const int gr = row < KT2_T ? K.m0 + (a0 + row < K.Mloc ? a0 + row : K.Mloc - 1) : (b0 + row - KT2_T < K.M ? b0 + row - KT2_T : K.M - 1);
    const uint32_t bo = ((uint32_t)gr * (uint32_t)K.stride + (uint32_t)(e < clen ? e : 0)) * 4u;
    const float4 t = *reinterpret_cast<const float4*>(base + bo);
    v[4 * u] = t.x, v[4 * u + 1] = t.y, v[4 * u + 2] = t.z, v[4 * u + 3] = t.w;
  }
}
__device__ __forceinline__ void kmat_tile64_block(float* __restrict__ smem, const KmatTile& K, int first, int step, int tid) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int units = kmat_tile_count(K.nta, K.ntb, K.symmetric) * K.nsplit;
  if (first >= units) return;
  const int lane = tid & 63, wave = tid >> 6, ag = lane >> 3, bg = lane & 7;
  const float4* pa = reinterpret_cast<const float4*>(smem + ag * KT2_LD) + wave * (KT2_CH / 4 / 8);
  const float4* pb = reinterpret_cast<const float4*>(smem + (KT2_T + bg) * KT2_LD) + wave * (KT2_CH / 4 / 8);
  auto tile_of = [&](int tile, int& a0, int& b0) {
    int ta, tb;
    if (K.symmetric) {
      ta = 0;
      int t = tile;
      while (t >= K.nta - ta) { t -= K.nta - ta; ++ta; }
      tb = ta + t;
    } else {
      ta = tile / K.ntb;
      tb = tile - ta * K.ntb;
    }
    a0 = ta * KT2_T;
    b0 = tb * KT2_T;
  };
  float v[KT2_RV];
  int u = first, c = (u % K.nsplit) * K.cps, a0, b0;
  tile_of(u / K.nsplit, a0, b0);
  int clen = K.len - c * KT2_CH < KT2_CH ? K.len - c * KT2_CH : KT2_CH;
  kt2_fetch(K, a0, b0, c * KT2_CH, clen, tid, v);
  double tot[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) tot[r] = 0.0;
  for (;;) {
#pragma unroll
    for (int w = 0; w < KT2_RV / 4; ++w) {
      const int i = w * KT2_NT + tid, row = i / (KT2_CH / 4), e = (i - row * (KT2_CH / 4)) * 4;
      const bool in = e < clen;
      *reinterpret_cast<float4*>(smem + row * KT2_LD + e) =
          make_float4(in ? v[4 * w] : 0.f, in ? v[4 * w + 1] : 0.f, in ? v[4 * w + 2] : 0.f, in ? v[4 * w + 3] : 0.f);
    }
    __syncthreads();
    const int ca0 = a0, cb0 = b0;
    const int sp = u % K.nsplit, c_hi = (sp + 1) * K.cps < K.nchunk ? (sp + 1) * K.cps : K.nchunk;
    int un = u, cn = c + 1;
    if (cn >= c_hi) {
      un = u + step;
      cn = (un % K.nsplit) * K.cps;
    }
    const bool more = un < units;  // (block-uniform)
    if (more) {
      if (un != u) tile_of(un / K.nsplit, a0, b0);
      clen = K.len - cn * KT2_CH < KT2_CH ? K.len - cn * KT2_CH : KT2_CH;
      kt2_fetch(K, a0, b0, cn * KT2_CH, clen, tid, v);
    }
    // two halves of the a-rows (i < 4, i >= 4: 32 accumulators each); the 8 waves' partial sums of a half meet in their own LDS area
    // (8 x 2048 floats behind the staged rows); thread tid owns the pairs tid + 512 r of each half
    float* const red = smem + 2 * KT2_T * KT2_LD;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      f32x2 acc[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) acc[q] = f32x2{0.f, 0.f};
#pragma unroll
      for (int it = 0; it < KT2_CH / 4 / 8; ++it) {
        float4 x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = pa[(hh * 4 + i) * 8 * (KT2_LD / 4) + it];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 y = pb[j * 8 * (KT2_LD / 4) + it];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f32x2 t = f32x2{x[i].x, x[i].y} - f32x2{y.x, y.y};
            acc[i * 8 + j] = __builtin_elementwise_fma(t, t, acc[i * 8 + j]);
            t = f32x2{x[i].z, x[i].w} - f32x2{y.z, y.w};
            acc[i * 8 + j] = __builtin_elementwise_fma(t, t, acc[i * 8 + j]);
          }
        }
      }
      if (hh) __syncthreads();  // (the first half's sums have been read)
#pragma unroll
      for (int q = 0; q < 32; ++q) red[wave * 2048 + q * 64 + lane] = acc[q].x + acc[q].y;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += (double)red[w * 2048 + r * 512 + tid];
        tot[hh * 4 + r] += t;
      }
    }
    if (un != u) {  // (block-uniform) last chunk of the unit
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const int hh = r8 >> 2, idx = (r8 & 3) * 512 + tid, q = idx >> 6, ln = idx & 63;
        const int a = ca0 + (ln >> 3) + 8 * (hh * 4 + (q >> 3)), b = cb0 + (ln & 7) + 8 * (q & 7);
        if (a < K.Mloc && b < K.M) {
          if (K.nsplit > 1) {
            K.part[((size_t)sp * K.Mloc + a) * K.M + b] = tot[r8];
          } else {
            const float kv = (float)((double)K.scale * exp(-tot[r8] / (double)K.h));
            const bool mirror = K.symmetric && cb0 > ca0;
            K.kout[(size_t)a * K.M + b] = kv;
            if (mirror) K.kout[(size_t)b * K.M + a] = kv;
            if (K.ksum) {
              K.ksum[(size_t)a * K.M + b] = K.kadd[(size_t)a * K.M + b] + kv;
              if (mirror) K.ksum[(size_t)b * K.M + a] = K.kadd[(size_t)b * K.M + a] + kv;
            }
          }
        }
        tot[r8] = 0.0;
      }
    }
    if (!more) break;
    u = un;
    c = cn;
    __syncthreads();
  }
}

// row a of the matrix from the pieces: threads tid, tid + nthr, ... take the columns b
__device__ __forceinline__ void kmat_finish_row(const double* __restrict__ part, int nsplit, int Mloc, int M, int symmetric, float scale, float h,
                                                float* __restrict__ kout, const float* __restrict__ kadd, float* __restrict__ ksum, int a, int tid,
                                                int nthr, int tile = KT_T) {
  for (int b = tid; b < M; b += nthr) {
    // (symmetric: only tiles tb >= ta exist; below them the transposed entry is the same sum)
    const bool up = !symmetric || (b / tile) >= (a / tile);
    const double* p = part + (up ? (size_t)a * M + b : (size_t)b * M + a);
    const size_t cs = (size_t)Mloc * M;
    double tot = 0.0;
    int c = 0;
    for (; c + 8 <= nsplit; c += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(c + u) * cs];
#pragma unroll
      for (int u = 0; u < 8; ++u) tot += v[u];
    }
    for (; c < nsplit; ++c) tot += p[(size_t)c * cs];
    const float kv = (float)((double)scale * exp(-tot / (double)h));
    kout[(size_t)a * M + b] = kv;
    if (ksum) ksum[(size_t)a * M + b] = kadd[(size_t)a * M + b] + kv;
  }
}

// ------------------------------------------------------------------------------------------------
// K8a kernel matrix slab: kz[a, b] = scale * exp(-||z_a - z_b||^2 / h) for local a, all b (direct differences:
//     the entries are ~e^-40 at d = 50 and must not be flushed or computed by cancellation).
//     reference: kernel.py:20-30 / 52-71, svgd.py:165-176 / 537-551
// grid = Mloc, block = 256; dynamic LDS = len * 4
// ------------------------------------------------------------------------------------------------
#define KMAT_CH 32768  // floats of z_a staged in LDS at a time (128 KiB); longer vectors (DenseNN theta at d = 100) go in chunks
__device__ __forceinline__ void kmat_block(float* __restrict__ smem, const float* __restrict__ pack, size_t pack_stride,
                                           size_t seg_off, int len, float* __restrict__ kout, int m0, int M, float scale, float h,
                                           int symmetric, int a, int bt, const float* __restrict__ kadd = nullptr,
                                           float* __restrict__ ksum = nullptr) {
  // (ksum != null: also writes kadd[a][b] + k[a][b] -- the joint models' weight matrix kz + kt, formed once here instead of per use in the
  //  SVGD transform, where it can then be a scalar operand; kadd is the matrix of an EARLIER launch)
  // block (a, bt): particle a (local) against b = bt * KMAT_BT .. +KMAT_BT-1; wave w takes b = b0 + w, b0 + w + 4, ...
  // symmetric (one rank holds all particles): tiles below the diagonal are skipped and k[a][b] is mirrored into k[b][a]
  // -- the sum of squared differences is the same number either way, so the slab is bit-identical to the full computation.
  const int b0 = bt * KMAT_BT, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (symmetric && b0 + KMAT_BT - 1 < a) return;
  const float* za = pack + (size_t)(m0 + a) * pack_stride + seg_off;
  double acc[KMAT_BT / 4];
#pragma unroll
  for (int q = 0; q < KMAT_BT / 4; ++q) acc[q] = 0.0;
  for (int c0 = 0; c0 < len; c0 += KMAT_CH) {
    const int clen = len - c0 < KMAT_CH ? len - c0 : KMAT_CH;
    const int len4 = clen >> 2;
    if (c0) __syncthreads();
    for (int e0 = 0; e0 < len4; e0 += 8 * 256) {  // (loads requested together: a rolled load -> LDS store loop makes one trip per element)
      float4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * 256 + tid;
        t[u] = reinterpret_cast<const float4*>(za + c0)[e < len4 ? e : len4 - 1];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * 256 + tid;
        if (e < len4) reinterpret_cast<float4*>(smem)[e] = t[u];
      }
    }
    for (int e = (len4 << 2) + tid; e < clen; e += 256) smem[e] = za[c0 + e];
    __syncthreads();
    // the wave's four b rows advance together: 8 independent 16-byte loads in flight per lane and pass (the kernel is
    // bound by the latency of the far cache levels, not by their bandwidth)
    const float4* zb4[KMAT_BT / 4];
    const float* zbs[KMAT_BT / 4];
#pragma unroll
    for (int q = 0; q < KMAT_BT / 4; ++q) {
      const int b = b0 + wave + 4 * q;
      zbs[q] = pack + (size_t)(b < M ? b : M - 1) * pack_stride + seg_off + c0;  // rows past the end repeat the last one
      zb4[q] = reinterpret_cast<const float4*>(zbs[q]);
    }
    const float4* za4 = reinterpret_cast<const float4*>(smem);
    float s0[KMAT_BT / 4], s1[KMAT_BT / 4];
#pragma unroll
    for (int q = 0; q < KMAT_BT / 4; ++q) s0[q] = s1[q] = 0.f;
    int e = lane;
    for (; e + 64 < len4; e += 128) {
      float4 qa[KMAT_BT / 4], qb[KMAT_BT / 4];
#pragma unroll
      for (int q = 0; q < KMAT_BT / 4; ++q) {
        qa[q] = zb4[q][e];
        qb[q] = zb4[q][e + 64];
      }
      const float4 pa = za4[e], pb = za4[e + 64];
#pragma unroll
      for (int q = 0; q < KMAT_BT / 4; ++q) {
        float t;
        t = pa.x - qa[q].x; s0[q] = fmaf(t, t, s0[q]); t = pa.y - qa[q].y; s0[q] = fmaf(t, t, s0[q]);
        t = pa.z - qa[q].z; s0[q] = fmaf(t, t, s0[q]); t = pa.w - qa[q].w; s0[q] = fmaf(t, t, s0[q]);
        t = pb.x - qb[q].x; s1[q] = fmaf(t, t, s1[q]); t = pb.y - qb[q].y; s1[q] = fmaf(t, t, s1[q]);
        t = pb.z - qb[q].z; s1[q] = fmaf(t, t, s1[q]); t = pb.w - qb[q].w; s1[q] = fmaf(t, t, s1[q]);
      }
    }
    for (; e < len4; e += 64) {
      const float4 pa = za4[e];
#pragma unroll
      for (int q = 0; q < KMAT_BT / 4; ++q) {
        const float4 qa = zb4[q][e];
        float t;
        t = pa.x - qa.x; s0[q] = fmaf(t, t, s0[q]); t = pa.y - qa.y; s0[q] = fmaf(t, t, s0[q]);
        t = pa.z - qa.z; s0[q] = fmaf(t, t, s0[q]); t = pa.w - qa.w; s0[q] = fmaf(t, t, s0[q]);
      }
    }
    for (int e1 = (len4 << 2) + lane; e1 < clen; e1 += 64) {
#pragma unroll
      for (int q = 0; q < KMAT_BT / 4; ++q) {
        const float t = smem[e1] - zbs[q][e1];
        s1[q] = fmaf(t, t, s1[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < KMAT_BT / 4; ++q) acc[q] += (double)s0[q] + (double)s1[q];
  }
#pragma unroll
  for (int q = 0; q < KMAT_BT / 4; ++q) {
    const int b = b0 + wave + 4 * q;
    if (b >= M) continue;
    const double tot = wave_sum_d(acc[q]);
    if (lane == 0) {
      const float kv = (float)((double)scale * exp(-tot / (double)h));
      kout[(size_t)a * M + b] = kv;
      if (symmetric && b > a) kout[(size_t)b * M + a] = kv;
      if (ksum) {
        const float sv = kadd[(size_t)a * M + b] + kv;
        ksum[(size_t)a * M + b] = sv;
        if (symmetric && b > a) ksum[(size_t)b * M + a] = sv;
      }
    }
  }
}

