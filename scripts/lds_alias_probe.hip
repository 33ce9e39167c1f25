// Do two co-resident workgroups with ~80 KB of LDS each really get disjoint LDS on an MI355X CU?  (debug aid for glowk_co.h)
// Every workgroup fills its LDS with a pattern derived from its id, then re-reads it many times while its neighbour does the same, and
// reports mismatches, the CU it ran on (HW_ID) and the LDS base / size the hardware gave it (LDS_ALLOC).
//   hipcc --offload-arch=gfx950 -O2 scripts/lds_alias_probe.hip -o /tmp/lds_probe && /tmp/lds_probe [bytes=80640]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

struct Rec { unsigned bad, hwid, ldsalloc, xcc; unsigned long long t0, t1; };

extern __shared__ unsigned dynlds[];

__global__ __launch_bounds__(256, 2) void probe(Rec* out, int words, int rounds) {
  const unsigned tag = (blockIdx.x + 1) * 2654435761u;
  for (int i = threadIdx.x; i < words; i += 256) dynlds[i] = tag ^ (unsigned)i;
  __syncthreads();
  unsigned bad = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
    for (int i = threadIdx.x; i < words; i += 256) bad += dynlds[i] != (tag ^ (unsigned)i);
    __syncthreads();
    for (int i = threadIdx.x; i < words; i += 256) dynlds[i] = tag ^ (unsigned)i;      // keep writing too
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  unsigned hwid, lds, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(lds));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  __shared__ unsigned tot;
  if (threadIdx.x == 0) tot = 0;
  __syncthreads();
  atomicAdd(&tot, bad);
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = Rec{tot, hwid, lds, xcc, t0, t1};
}

int main(int argc, char** argv) {
  const int bytes = argc > 1 ? atoi(argv[1]) : 80640, wgs = 2048, rounds = 40;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("device %s: sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, CUs %d\n", p.name, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.multiProcessorCount);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  Rec* d;
  hipMalloc(&d, sizeof(Rec) * wgs);
  hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), bytes, 0, d, bytes / 4, rounds);
  hipError_t e = hipDeviceSynchronize();
  printf("launch: %s\n", hipGetErrorString(e));
  std::vector<Rec> h(wgs);
  hipMemcpy(h.data(), d, sizeof(Rec) * wgs, hipMemcpyDeviceToHost);
  unsigned long long bad = 0;
  std::map<unsigned, int> bases;
  int overl = 0;
  for (int i = 0; i < wgs; ++i) {
    bad += h[i].bad;
    bases[h[i].ldsalloc]++;
    // co-residence: another workgroup on the same (xcc, se, cu) whose interval overlaps
  }
  for (int i = 0; i < wgs; ++i)
    for (int j = i + 1; j < wgs; ++j)
      if (h[i].xcc == h[j].xcc && (h[i].hwid & 0xFFF00u) == (h[j].hwid & 0xFFF00u) && h[i].t0 < h[j].t1 && h[j].t0 < h[i].t1) { ++overl; break; }
  printf("LDS bytes per workgroup %d: mismatching words %llu; workgroups that overlapped in time with another one on their CU: %d of %d\n", bytes, bad, overl, wgs);
  for (auto& kv : bases) printf("  LDS_ALLOC 0x%08x (base field %u, size field %u): %d workgroups\n", kv.first, kv.first & 0xFF, (kv.first >> 12) & 0x1FF, kv.second);
  return bad != 0;
}
