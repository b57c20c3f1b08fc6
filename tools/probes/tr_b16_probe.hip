// Microprobe for gfx950's LDS transpose read (ds_read_b64_tr_b16 via __builtin_amdgcn_ds_read_tr16_b64_v4bf16).
// The ISA reference the programming guide cites (cdna4_isa.md) is not present in this image, so the lane <-> address <-> data
// mapping is MEASURED here: LDS holds lds[i] = i (u16), every lane issues one read at a chosen byte address, and the four 16-bit
// values each lane receives are printed.  Patterns: (1) dense 4x16 block per 16-lane group, lane i -> row i>>2, cols (i&3)*4..+3,
// row stride 16 elements; (2) same with row stride 32 elements; (3) one uniform address for all lanes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(uint16_t* out, int pattern) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  int elem;
  if (pattern == 1) elem = g * 64 + (i >> 2) * 16 + (i & 3) * 4;
  else if (pattern == 2) elem = g * 128 + (i >> 2) * 32 + (i & 3) * 4;
  else elem = 8;
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(lds + elem));
  uint16_t r[4];
  __builtin_memcpy(r, &v, 8);
  for (int j = 0; j < 4; j++) out[l * 4 + j] = r[j];
}

int main() {
  uint16_t* d; uint16_t h[256];
  hipMalloc(&d, sizeof(h));
  for (int p = 1; p <= 3; p++) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    printf("pattern %d (lane: addr-elem -> 4 values)\n", p);
    for (int l = 0; l < 64; l++) {
      int g = l >> 4, i = l & 15;
      int elem = p == 1 ? g * 64 + (i >> 2) * 16 + (i & 3) * 4 : (p == 2 ? g * 128 + (i >> 2) * 32 + (i & 3) * 4 : 8);
      printf("  lane %2d: %4d -> %4d %4d %4d %4d\n", l, elem, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
  }
  return 0;
}
