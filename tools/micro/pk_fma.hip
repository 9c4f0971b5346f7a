// Micro-benchmark (GPU box): issue cost of v_fma_f32 against v_pk_fma_f32 for ONE wavefront per SIMD — the regime of k_forward.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_fma tools/micro/pk_fma.hip && /tmp/pk_fma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int N> __global__ void k_scalar(float* out, long long* cyc, float a, float b, int iters) {
  float x[N];
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = threadIdx.x * 0.001f + i;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
  }
  long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int N> __global__ void k_packed(float* out, long long* cyc, float a, float b, int iters) {
  f2 x[N];
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  const f2 av = {a, a}, bv = {b, b};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
  }
  long long t1 = clock64();
  f2 s = {0, 0};
#pragma unroll
  for (int i = 0; i < N; ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s.x + s.y;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <class F> void run(const char* name, F launch, int nops, int flops_per_op) {
  float* out; long long* cyc; const int nb = 1024, iters = 2000;
  hipMalloc(&out, nb * 64 * 4); hipMalloc(&cyc, nb * 8);
  launch(out, cyc, iters); launch(out, cyc, iters); hipDeviceSynchronize();
  long long h[1024]; hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < nb; ++i) m += h[i]; m /= nb;
  printf("%-28s %8.2f clock64 ticks per instruction (%d per iteration)\n", name, m / iters / nops, nops);
  hipFree(out); hipFree(cyc);
}
int main() {
  run("v_fma_f32 x32 independent", [](float* o, long long* c, int it) { k_scalar<32><<<1024, 64>>>(o, c, 1.0001f, 0.5f, it); }, 32, 2);
  run("v_fma_f32 x1 dependent", [](float* o, long long* c, int it) { k_scalar<1><<<1024, 64>>>(o, c, 1.0001f, 0.5f, it); }, 1, 2);
  run("v_fma_f32 x2", [](float* o, long long* c, int it) { k_scalar<2><<<1024, 64>>>(o, c, 1.0001f, 0.5f, it); }, 2, 2);
  run("v_pk_fma_f32 x16 independent", [](float* o, long long* c, int it) { k_packed<16><<<1024, 64>>>(o, c, 1.0001f, 0.5f, it); }, 16, 4);
  run("v_pk_fma_f32 x1 dependent", [](float* o, long long* c, int it) { k_packed<1><<<1024, 64>>>(o, c, 1.0001f, 0.5f, it); }, 1, 4);
  run("v_pk_fma_f32 x2", [](float* o, long long* c, int it) { k_packed<2><<<1024, 64>>>(o, c, 1.0001f, 0.5f, it); }, 2, 4);
  return 0;
}
