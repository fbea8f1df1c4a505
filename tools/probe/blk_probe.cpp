// Stand-alone correctness + timing probe for the fused LN+MLP kernels (csrc/block_kernels.hip), no torch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/probe/blk_probe.cpp revisiting-at_amd/csrc/block_kernels.hip -o tools/probe/blk_probe
//   ./blk_probe [iters]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "apgd_hip.h"
#include "convnext_hip.h"

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                \
    }                                                                         \
  } while (0)

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = static_cast<uint32_t>(h) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static float rbf(float f) { return bf2f(f2bf(f)); }

template <typename T>
T* dev(const std::vector<T>& v) {
  T* p;
  CK(hipMalloc(&p, v.size() * sizeof(T)));
  CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  const int only_c = argc > 2 ? atoi(argv[2]) : 0;
  const int B = 256;
  int bad_total = 0;
  struct Shape { int C, HW; } shapes[] = {{96, 56}, {192, 28}, {384, 14}};
  for (auto sh : shapes) {
    const int C = sh.C;
    if (only_c && only_c != C) continue;
    if (!cnx_block_mlp_supported(C)) continue;
    const long M = static_cast<long>(B) * sh.HW * sh.HW - 8;      // ragged tail on purpose (a multiple of 8: the emit path stores 8 rows per lane)
    std::mt19937 rng(C);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<uint16_t> u(M * C);
    std::vector<float> x(M * C), lnw(C), lnb(C), b1(4 * C), b2(C), gm(C), W1(4 * C * C), W2(4 * C * C);
    for (auto& v : u) v = f2bf(nd(rng) * 1.5f + 0.3f);
    for (auto& v : x) v = nd(rng);
    for (auto& v : lnw) v = 1.f + 0.2f * nd(rng);
    for (auto& v : lnb) v = 0.2f * nd(rng);
    for (auto& v : b1) v = 0.3f * nd(rng);
    for (auto& v : b2) v = 0.3f * nd(rng);
    for (auto& v : gm) v = 0.5f * nd(rng);
    const float s1 = 1.0f / std::sqrt(static_cast<float>(C)), s2 = 1.0f / std::sqrt(4.0f * C);
    for (auto& v : W1) v = s1 * nd(rng);      // [4C][C]
    for (auto& v : W2) v = s2 * nd(rng);      // [C][4C]
    uint16_t* du = dev(u);
    float *dx = dev(x), *dlnw = dev(lnw), *dlnb = dev(lnb), *db1 = dev(b1), *db2 = dev(b2), *dgm = dev(gm), *dW1 = dev(W1),
          *dW2 = dev(W2);
    float *dout, *dmean, *drstd;
    uint16_t *dWf, *dy2;
    CK(hipMalloc(&dout, M * C * 4));
    CK(hipMalloc(&dmean, M * 4));
    CK(hipMalloc(&drstd, M * 4));
    CK(hipMalloc(&dy2, M * C * 2));
    CK(hipMalloc(&dWf, cnx_mlp_packed_elems(C) * 2));
    CK(hipMemset(dout, 0xff, M * C * 4));
    int rc = cnx_mlp_pack_weights(dW1, dW2, APGD_F32, dWf, C, nullptr);
    if (rc) { printf("pack rc=%d\n", rc); return 2; }
    rc = cnx_block_mlp_fwd(du, dlnw, dlnb, 1e-6f, dmean, drstd, dWf, db1, db2, dgm, dx, APGD_F32, dout, APGD_F32, dy2, M, C,
                           nullptr);
    if (rc) { printf("fwd rc=%d\n", rc); return 2; }
    CK(hipDeviceSynchronize());
    std::vector<float> out(M * C), mean(M), rstd(M);
    CK(hipMemcpy(out.data(), dout, M * C * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(mean.data(), dmean, M * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(rstd.data(), drstd, M * 4, hipMemcpyDeviceToHost));
    // ---- CPU reference on sampled rows (bf16 roundings where the kernel has them)
    std::vector<long> rows = {0, 1, 31, 32, 33, 63, 64, 127, 128, 129, 255, 1000, 4097, M / 2, M - 130, M - 129, M - 33, M - 2, M - 1};
    for (int i = 0; i < 40; ++i) rows.push_back(static_cast<long>(rng() % M));
    double max_err = 0, max_ref = 0, max_stat = 0;
    std::vector<float> a(C), h(4 * C);
    for (long m : rows) {
      double s = 0;
      for (int c = 0; c < C; ++c) s += bf2f(u[m * C + c]);
      const double mu = s / C;
      double ss = 0;
      for (int c = 0; c < C; ++c) { const double d = bf2f(u[m * C + c]) - mu; ss += d * d; }
      const double rs = 1.0 / std::sqrt(ss / C + 1e-6);
      max_stat = std::fmax(max_stat, std::fabs(mean[m] - mu) + std::fabs(rstd[m] - rs) / rs);
      for (int c = 0; c < C; ++c) a[c] = rbf(static_cast<float>((bf2f(u[m * C + c]) - mu) * rs * lnw[c] + lnb[c]));
      for (int j = 0; j < 4 * C; ++j) {
        double acc = b1[j];
        for (int c = 0; c < C; ++c) acc += static_cast<double>(rbf(W1[static_cast<long>(j) * C + c])) * a[c];
        h[j] = rbf(static_cast<float>(0.5 * acc * (1.0 + std::erf(acc * 0.7071067811865476))));
      }
      for (int c = 0; c < C; ++c) {
        double acc = b2[c];
        for (int j = 0; j < 4 * C; ++j) acc += static_cast<double>(rbf(W2[static_cast<long>(c) * 4 * C + j])) * h[j];
        const double ref = x[m * C + c] + gm[c] * acc;
        max_err = std::fmax(max_err, std::fabs(ref - out[m * C + c]));
        max_ref = std::fmax(max_ref, std::fabs(ref));
      }
    }
    const bool ok = max_err < 2e-2 * std::fmax(1.0, max_ref) * 0.25 && max_stat < 1e-4;
    printf("C=%3d M=%7ld  max|err| %.3e (max|ref| %.2f)  LN-stat err %.2e  %s\n", C, M, max_err, max_ref, max_stat, ok ? "OK" : "MISMATCH");
    bad_total += !ok;
    // ---- timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w)
      cnx_block_mlp_fwd(du, dlnw, dlnb, 1e-6f, dmean, drstd, dWf, db1, db2, dgm, dx, APGD_F32, dout, APGD_F32, nullptr, M, C, nullptr);
    std::vector<float> ts;
    for (int it = 0; it < iters; ++it) {
      CK(hipEventRecord(e0));
      cnx_block_mlp_fwd(du, dlnw, dlnb, 1e-6f, dmean, drstd, dWf, db1, db2, dgm, dx, APGD_F32, dout, APGD_F32, nullptr, M, C, nullptr);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double med = ts[ts.size() / 2], flops = 16.0 * M * C * C, bytes = static_cast<double>(M) * C * (2 + 4 + 4);
    printf("      fwd median %.1f us (min %.1f)   %.0f TFLOP/s   %.0f GB/s algorithmic\n", med * 1e3, ts[0] * 1e3, flops / med / 1e9,
           bytes / med / 1e6);
    // ================= backward (input gradient) =================
    if (cnx_block_mlp_bwd_supported(C)) {
      std::vector<float> g(M * C);
      for (auto& v : g) v = nd(rng);
      float* dg = dev(g);
      uint16_t *dWb, *dda, *da_o, *ddo_o, *dht, *ddhpt;
      CK(hipMalloc(&dWb, cnx_mlp_packed_bwd_elems(C) * 2));
      CK(hipMalloc(&dda, M * C * 2));
      CK(hipMalloc(&da_o, M * C * 2));
      CK(hipMalloc(&ddo_o, M * C * 2));
      CK(hipMalloc(&dht, M * 4 * C * 2));
      CK(hipMalloc(&ddhpt, M * 4 * C * 2));
      rc = cnx_mlp_pack_weights_bwd(dW1, dW2, APGD_F32, dWb, C, nullptr);
      if (rc) { printf("pack bwd rc=%d\n", rc); return 2; }
      rc = cnx_block_mlp_bwd(du, dlnw, dlnb, dmean, drstd, dg, APGD_F32, dgm, dWb, db1, dda, da_o, 0, ddo_o, dht, ddhpt, M, C, nullptr);
      if (rc) { printf("bwd rc=%d\n", rc); return 2; }
      CK(hipDeviceSynchronize());
      std::vector<uint16_t> da(M * C), ht(M * 4 * C), dhpt(M * 4 * C), ao(M * C), doo(M * C);
      CK(hipMemcpy(da.data(), dda, M * C * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ao.data(), da_o, M * C * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(doo.data(), ddo_o, M * C * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ht.data(), dht, M * 4 * C * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(dhpt.data(), ddhpt, M * 4 * C * 2, hipMemcpyDeviceToHost));
      double e_da = 0, r_da = 0, e_h = 0, r_h = 0, e_dh = 0, r_dh = 0, e_a = 0, e_do = 0;
      std::vector<float> dO(C), dhp(4 * C);
      for (long m : rows) {
        const double mu = mean[m], rs = rstd[m];
        for (int c = 0; c < C; ++c) {
          a[c] = rbf(static_cast<float>((bf2f(u[m * C + c]) - mu) * rs * lnw[c] + lnb[c]));
          dO[c] = rbf(g[m * C + c] * gm[c]);
          e_a = std::fmax(e_a, std::fabs(a[c] - bf2f(ao[m * C + c])));
          e_do = std::fmax(e_do, std::fabs(dO[c] - bf2f(doo[m * C + c])));
        }
        for (int j = 0; j < 4 * C; ++j) {
          double hp = b1[j], dh = 0;
          for (int c = 0; c < C; ++c) {
            hp += static_cast<double>(rbf(W1[static_cast<long>(j) * C + c])) * a[c];
            dh += static_cast<double>(rbf(W2[static_cast<long>(c) * 4 * C + j])) * dO[c];
          }
          const double Phi = 0.5 * (1.0 + std::erf(hp * 0.7071067811865476));
          const double gp = Phi + hp * std::exp(-0.5 * hp * hp) * 0.3989422804014327;
          dhp[j] = rbf(static_cast<float>(dh * gp));
          const double hh = hp * Phi;
          e_h = std::fmax(e_h, std::fabs(hh - bf2f(ht[static_cast<long>(j) * M + m])));
          r_h = std::fmax(r_h, std::fabs(hh));
          e_dh = std::fmax(e_dh, std::fabs(dh * gp - bf2f(dhpt[static_cast<long>(j) * M + m])));
          r_dh = std::fmax(r_dh, std::fabs(dh * gp));
        }
        for (int c = 0; c < C; ++c) {
          double acc = 0;
          for (int j = 0; j < 4 * C; ++j) acc += static_cast<double>(rbf(W1[static_cast<long>(j) * C + c])) * dhp[j];
          e_da = std::fmax(e_da, std::fabs(acc - bf2f(da[m * C + c])));
          r_da = std::fmax(r_da, std::fabs(acc));
        }
      }
      const bool okb = e_da < 1.5e-2 * r_da && e_h < 1e-2 * r_h && e_dh < 1e-2 * r_dh && e_a < 1e-6 && e_do < 1e-6;
      printf("      bwd: da err %.3e (max %.2f)  H^T err %.3e (max %.2f)  dHpre^T err %.3e (max %.2f)  a/dO err %.1e/%.1e  %s\n",
             e_da, r_da, e_h, r_h, e_dh, r_dh, e_a, e_do, okb ? "OK" : "MISMATCH");
      bad_total += !okb;
      for (int em = 0; em < 2; ++em) {
        std::vector<float> tb;
        for (int it = 0; it < iters + 3; ++it) {
          CK(hipEventRecord(e0));
          cnx_block_mlp_bwd(du, dlnw, dlnb, dmean, drstd, dg, APGD_F32, dgm, dWb, db1, dda, em ? da_o : nullptr, 0, em ? ddo_o : nullptr,
                            em ? dht : nullptr, em ? ddhpt : nullptr, M, C, nullptr);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (it >= 3) tb.push_back(ms);
        }
        std::sort(tb.begin(), tb.end());
        const double mb = tb[tb.size() / 2];
        printf("      bwd%s median %.1f us (min %.1f)   %.0f TFLOP/s\n", em ? " +emit" : "      ", mb * 1e3, tb[0] * 1e3, 1.5 * flops / mb / 1e9);
      }
      hipFree(dg); hipFree(dWb); hipFree(dda); hipFree(da_o); hipFree(ddo_o); hipFree(dht); hipFree(ddhpt);
    }
    hipFree(du); hipFree(dx); hipFree(dout); hipFree(dmean); hipFree(drstd); hipFree(dy2); hipFree(dWf); hipFree(dW1); hipFree(dW2);
  }
  return bad_total ? 1 : 0;
}
