// bge_chol_quad<NB> / bge_chol_lane<NMAX> of kernels_bge.h against a double Cholesky on the host (random SPD matrix, random index sets)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "../../dibs_amd/csrc/kernels_bge.h"

template <int NB>
__global__ void k_quad(const float* Rp, int d, const uint64_t* w0s, const int* js, const int* lis, float* ld2, float* last) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ldr = d + 1, msz = ldr * ldr;
  float* Rs = reinterpret_cast<float*>(smem);
  for (int e = threadIdx.x; e < msz; e += blockDim.x) Rs[e] = Rp[e];
  __syncthreads();
  int* qidx = reinterpret_cast<int*>(smem + ((msz * 4 + 15) & ~15)) + (threadIdx.x >> 2) * BGE_QS;
  const int pr = blockIdx.x * 16 + (threadIdx.x >> 2);
  float a, b;
  bge_chol_quad<NB, false>(Rs, 0, ldr, d, qidx, w0s[pr], 0ull, js[pr], lis[pr], a, b);
  if ((threadIdx.x & 3) == 0) { ld2[pr] = a; last[pr] = b; }
}
template <int NMAX>
__global__ void k_lane(const float* Rp, int d, const uint64_t* w0s, const int* js, const int* lis, float* ld2, float* last) {
  const int pr = blockIdx.x * 64 + threadIdx.x;
  float a[1], b[1];
  uint64_t w0[1] = {w0s[pr]}, w1[1] = {0ull};
  const int mat[1] = {0}, jj[1] = {js[pr]}, li[1] = {lis[pr]};
  bge_chol_lane<NMAX, false, 1>(Rp, mat, d + 1, d, w0, w1, jj, li, a, b);
  ld2[pr] = a[0]; last[pr] = b[0];
}

int main() {
  const int d = 50, ldr = d + 1, NP = 64;
  std::mt19937_64 rng(1);
  std::normal_distribution<double> nd;
  std::vector<double> X(100 * d), R(d * d);
  for (auto& v : X) v = nd(rng);
  for (int a = 0; a < d; ++a) for (int b = 0; b < d; ++b) { double s = a == b ? 0.5 : 0; for (int n = 0; n < 100; ++n) s += X[n * d + a] * X[n * d + b]; R[a * d + b] = s; }
  std::vector<float> Rp(ldr * ldr, 0.f);
  for (int a = 0; a < d; ++a) for (int b = 0; b < d; ++b) Rp[a * ldr + b] = (float)R[a * d + b];
  for (int NBt : {5, 8, 0}) {
    const int nmax = NBt ? 4 * NBt : 16;
    std::vector<uint64_t> w0(NP); std::vector<int> js(NP), lis(NP);
    std::vector<double> ref_ld(NP), ref_last(NP);
    for (int p = 0; p < NP; ++p) {
      const int n = nmax - ((p / 2) % 4), li = n - 1;
      std::vector<int> perm(d); for (int i = 0; i < d; ++i) perm[i] = i;
      std::shuffle(perm.begin(), perm.end(), rng);
      js[p] = perm[li]; lis[p] = li; w0[p] = 0;
      for (int i = 0; i < li; ++i) w0[p] |= 1ull << perm[i];
      std::vector<int> idx; for (int i = 0; i < d; ++i) if ((w0[p] >> i) & 1) idx.push_back(i); idx.push_back(js[p]);
      std::vector<double> L(n * n, 0.0); double ld = 0, lastp = 0;
      for (int k = 0; k < n; ++k) {
        double s = R[idx[k] * d + idx[k]]; for (int q = 0; q < k; ++q) s -= L[k * n + q] * L[k * n + q];
        if (k < li) ld += std::log2(s); else lastp = s;
        const double dk = std::sqrt(s); L[k * n + k] = dk;
        for (int i = k + 1; i < n; ++i) { double t = R[idx[i] * d + idx[k]]; for (int q = 0; q < k; ++q) t -= L[i * n + q] * L[k * n + q]; L[i * n + k] = t / dk; }
      }
      ref_ld[p] = ld; ref_last[p] = lastp;
    }
    float *dR, *dld, *dlast; uint64_t* dw; int *dj, *dl;
    hipMalloc(&dR, Rp.size() * 4); hipMalloc(&dld, NP * 4); hipMalloc(&dlast, NP * 4); hipMalloc(&dw, NP * 8); hipMalloc(&dj, NP * 4); hipMalloc(&dl, NP * 4);
    hipMemcpy(dR, Rp.data(), Rp.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, w0.data(), NP * 8, hipMemcpyHostToDevice);
    hipMemcpy(dj, js.data(), NP * 4, hipMemcpyHostToDevice); hipMemcpy(dl, lis.data(), NP * 4, hipMemcpyHostToDevice);
    const size_t lds = ((ldr * ldr * 4 + 15) & ~15) + 16 * BGE_QS * 4;
    if (NBt == 5) hipLaunchKernelGGL(k_quad<5>, dim3(NP / 16), dim3(64), lds, 0, dR, d, dw, dj, dl, dld, dlast);
    else if (NBt == 8) hipLaunchKernelGGL(k_quad<8>, dim3(NP / 16), dim3(64), lds, 0, dR, d, dw, dj, dl, dld, dlast);
    else hipLaunchKernelGGL(k_lane<16>, dim3(NP / 64), dim3(64), 0, 0, dR, d, dw, dj, dl, dld, dlast);
    std::vector<float> ld(NP), last(NP);
    hipMemcpy(ld.data(), dld, NP * 4, hipMemcpyDeviceToHost); hipMemcpy(last.data(), dlast, NP * 4, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0; int bad = 0;
    for (int p = 0; p < NP; ++p) {
      const double a = std::fabs(ld[p] - ref_ld[p]), b = std::fabs(last[p] - ref_last[p]) / ref_last[p];
      if (!(a < 1e-2) || !(b < 1e-3)) { if (bad < 40) printf("  p=%d quad=%d li=%d: ld2 %.5f ref %.5f  last %.5f ref %.5f\n", p, p % 16, lis[p], ld[p], ref_ld[p], last[p], ref_last[p]); ++bad; }
      if (a == a) e1 = std::fmax(e1, a); if (b == b) e2 = std::fmax(e2, b);
    }
    printf("%s nmax=%d: bad %d of %d, max |ld2 err| %.3e, max rel last err %.3e\n", NBt ? "quad" : "lane", nmax, bad, NP, e1, e2);
  }
  return 0;
}
