// Round 6: where do the workgroups of a decode-attention-shaped launch land?  Grid (n_split, heads), 256 threads, 52 KB of LDS
// (3 workgroups per CU), every workgroup resident for ~20 us: records XCC_ID / HW_ID and the entry time per workgroup, prints the
// workgroup -> physical CU map statistics (is it `linear id % 256` in some fixed order?  how many workgroups share a CU?).
//   hipcc --offload-arch=gfx950 -O2 tools/scratch/wg_census_r6.hip -o tools/scratch/wg_census_r6 && tools/scratch/wg_census_r6 [gx gy lds_kb]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_census(unsigned* out, long long* t_entry, int spin_us) {
  extern __shared__ unsigned char lds[];
  const int id = blockIdx.y * gridDim.x + blockIdx.x;
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * id] = hw;
    out[2 * id + 1] = xcc;
    t_entry[id] = wall_clock64();
    lds[0] = 1;
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_us * 100) __builtin_amdgcn_s_sleep(8);
}
int main(int argc, char** argv) {
  const int gx = argc > 1 ? atoi(argv[1]) : 21, gy = argc > 2 ? atoi(argv[2]) : 32, lds_kb = argc > 3 ? atoi(argv[3]) : 52;
  const int n = gx * gy;
  unsigned* d; long long* dt;
  CK(hipMalloc(&d, n * 8)); CK(hipMalloc(&dt, n * 8));
  CK(hipFuncSetAttribute((const void*)k_census, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
  std::vector<unsigned> h(2 * n); std::vector<long long> ht(n);
  std::vector<int> first_cu;
  for (int rep = 0; rep < 4; ++rep) {
    hipLaunchKernelGGL(k_census, dim3(gx, gy), dim3(256), lds_kb * 1024, 0, d, dt, 20);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ht.data(), dt, n * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, std::vector<int>> cu;      // key: xcc | se | sh | cu
    std::vector<int> key(n);
    for (int i = 0; i < n; ++i) {
      const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
      const unsigned cu_id = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      key[i] = (int)((xcc << 8) | (se << 5) | (sh << 4) | cu_id);
      cu[key[i]].push_back(i);
    }
    int mx = 0, mn = 1 << 30; std::map<int, int> hist;
    for (auto& kv : cu) { mx = std::max(mx, (int)kv.second.size()); mn = std::min(mn, (int)kv.second.size()); hist[(int)kv.second.size()]++; }
    long long tmin = *std::min_element(ht.begin(), ht.end()), tmax = *std::max_element(ht.begin(), ht.end());
    printf("rep %d: %d workgroups on %zu distinct CUs; per CU min %d max %d; histogram:", rep, n, cu.size(), mn, mx);
    for (auto& kv : hist) printf(" %dx%d", kv.second, kv.first);
    printf("; entry spread %.2f us\n", (tmax - tmin) / 100.0);
    int xcd_ok = 0, same256 = 0, tot256 = 0;
    for (int i = 0; i < n; ++i) xcd_ok += (int)(h[2 * i + 1] & 15) == i % 8;
    for (int i = 0; i + 256 < n; ++i) { same256 += key[i] == key[i + 256]; ++tot256; }
    printf("   xcc == id %% 8 for %d of %d; key[i] == key[i + 256] for %d of %d\n", xcd_ok, n, same256, tot256);
    if (rep == 0) { first_cu = key; }
    else { int same = 0; for (int i = 0; i < n; ++i) same += key[i] == first_cu[i]; printf("   same CU as in rep 0: %d of %d\n", same, n); }
    if (rep == 0) {
      printf("   first 40 workgroups (id: xcc se sh cu | entry us):");
      for (int i = 0; i < 40; ++i) printf(" %d:%u.%u.%u.%u|%.2f", i, h[2 * i + 1] & 15, (h[2 * i] >> 13) & 7, (h[2 * i] >> 12) & 1, (h[2 * i] >> 8) & 15, (ht[i] - tmin) / 100.0);
      printf("\n   workgroups sharing the CU of workgroup 20 (a merger at 21 splits):");
      for (int j : cu[key[20]]) printf(" %d", j);
      printf("\n   the third workgroups of their CU (by entry time), ids:");
      int cnt = 0;
      for (auto& kv : cu) if (kv.second.size() >= 3) { std::vector<int> v = kv.second; std::sort(v.begin(), v.end(), [&](int a, int b) { return ht[a] < ht[b]; }); if (cnt++ < 24) printf(" %d", v[2]); }
      printf("\n");
    }
  }
  return 0;
}
