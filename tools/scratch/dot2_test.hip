#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(float* o, const unsigned* a, const unsigned* b) {
  float c = 0.25f;
  c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[threadIdx.x]), __builtin_bit_cast(bf16x2, b[threadIdx.x]), c, false);
  c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[threadIdx.x + 64]), __builtin_bit_cast(bf16x2, b[threadIdx.x + 64]), c, false);
  o[threadIdx.x] = c;
}
static unsigned short f2b(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static float b2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  unsigned ha[128], hb[128]; float ref[64];
  srand(1);
  for (int i = 0; i < 128; ++i) {
    float a0 = (rand() % 2000 - 1000) / 500.f, a1 = (rand() % 2000 - 1000) / 500.f, b0 = (rand() % 2000 - 1000) / 500.f, b1 = (rand() % 2000 - 1000) / 500.f;
    ha[i] = f2b(a0) | ((unsigned)f2b(a1) << 16); hb[i] = f2b(b0) | ((unsigned)f2b(b1) << 16);
  }
  for (int i = 0; i < 64; ++i) {
    double c = 0.25;
    for (int h = 0; h < 2; ++h) { unsigned a = ha[i + 64 * h], b = hb[i + 64 * h]; c += (double)b2f(a & 0xffff) * b2f(b & 0xffff) + (double)b2f(a >> 16) * b2f(b >> 16); }
    ref[i] = (float)c;
  }
  unsigned *da, *db; float* dout; (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dout, 256);
  (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, da, db);
  float out[64]; (void)hipMemcpy(out, dout, 256, hipMemcpyDeviceToHost);
  double me = 0; for (int i = 0; i < 64; ++i) me = fmax(me, fabs(out[i] - ref[i]));
  printf("max |dot2 - ref| = %g   (out[0] %g ref[0] %g, out[1] %g ref[1] %g)\n", me, out[0], ref[0], out[1], ref[1]);
  return 0;
}
