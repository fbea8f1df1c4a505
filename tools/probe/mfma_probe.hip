// Verifies the operand / result lane maps of the gfx950 bf16 MFMAs used by the fused MLP kernel.
// hipcc --offload-arch=gfx950 -O2 mfma_probe.hip -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __bf16 tobf(float f) { return (__bf16)f; }

// 32x32x16: A[i][k] i<32,k<16 ; B[k][j] ; D[i][j]
__global__ void probe32(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = tobf(A[(l & 31) * 16 + (l >> 5) * 8 + e]);        // lane: row i = l&31, k = (l>>5)*8 + e
    b[e] = tobf(B[((l >> 5) * 8 + e) * 32 + (l & 31)]);      // lane: col j = l&31, k = (l>>5)*8 + e
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    D[row * 32 + col] = c[r];
  }
}
// 16x16x32: A[i][k] i<16,k<32 ; B[k][j] ; D[i][j]
__global__ void probe16(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = tobf(A[(l & 15) * 32 + (l >> 4) * 8 + e]);
    b[e] = tobf(B[((l >> 4) * 8 + e) * 16 + (l & 15)]);
  }
  f32x4 c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

int main() {
  int bad = 0;
  {
    std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), R(32 * 32, 0.f);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (i * 3 + k * 5) % 17 - 8;
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (k * 7 + j * 2) % 13 - 6;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += A[i * 16 + k] * B[k * 32 + j];
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    probe32<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int nb = 0; for (size_t i = 0; i < D.size(); ++i) nb += D[i] != R[i];
    printf("32x32x16 layout mismatches: %d\n", nb); bad += nb;
  }
  {
    std::vector<float> A(16 * 32), B(32 * 16), D(16 * 16), R(16 * 16, 0.f);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (i * 3 + k * 5) % 17 - 8;
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (k * 7 + j * 2) % 13 - 6;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += A[i * 32 + k] * B[k * 16 + j];
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    probe16<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int nb = 0; for (size_t i = 0; i < D.size(); ++i) nb += D[i] != R[i];
    printf("16x16x32 layout mismatches: %d\n", nb); bad += nb;
  }
  return bad != 0;
}
