// Hardware probe (MI355X): (1) buffer_load_dwordx4 ... lds: destination = M0 base + lane * 16, out-of-range lanes write ZEROS,
// exec-masked lanes leave LDS untouched; (2) v_mfma_f32_32x32x16_bf16 operand / result lane maps.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_dma(const unsigned* g, int nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned s[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) s[i] = 0xABABABABu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  const int lane = threadIdx.x;
  // lanes 0..47 in range (reversed source order), lanes 48..55 out of range, lanes 56..63 masked off
  unsigned off = lane < 48 ? (unsigned)(47 - lane) * 16u : 0x7fffff00u;
  if (lane < 56) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(s + 64), 16, off, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = s[i];
}
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
__global__ void k_mfma(const unsigned short* A /*32x16*/, const unsigned short* B /*16x32*/, float* D /*32x32*/) {
  const int l = threadIdx.x;
  s16x8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = (short)A[(l & 31) * 16 + 8 * (l >> 5) + e]; b[e] = (short)B[(8 * (l >> 5) + e) * 32 + (l & 31)]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int v = 0; v < 16; v++) D[((v & 3) + 8 * (v >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[v];
}
int main() {
  unsigned *g, *out, hg[48 * 4], ho[1024];
  for (int i = 0; i < 48 * 4; i++) hg[i] = 0x1000u + i;
  hipMalloc(&g, sizeof(hg)); hipMalloc(&out, sizeof(ho));
  hipMemcpy(g, hg, sizeof(hg), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_dma, dim3(1), dim3(64), 0, 0, g, (int)sizeof(hg), out);
  hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int i = 0; i < 64; i++) if (ho[i] != 0xABABABABu) ok = 0;                      // below the base: untouched
  for (int l = 0; l < 48; l++) for (int e = 0; e < 4; e++) if (ho[64 + l * 4 + e] != 0x1000u + (47 - l) * 4 + e) ok = 0;   // lane-linear destination
  int zeros = 1, masked = 1;
  for (int l = 48; l < 56; l++) for (int e = 0; e < 4; e++) if (ho[64 + l * 4 + e] != 0) zeros = 0;
  for (int l = 56; l < 64; l++) for (int e = 0; e < 4; e++) if (ho[64 + l * 4 + e] != 0xABABABABu) masked = 0;
  printf("dma lds: lane-linear destination %s; out-of-range lanes write zeros: %s (word %08x); masked lanes untouched: %s\n", ok ? "OK" : "MISMATCH",
         zeros ? "YES" : "NO", ho[64 + 48 * 4], masked ? "YES" : "NO");
  // MFMA: asymmetric integer operands
  unsigned short hA[32 * 16], hB[16 * 32]; float hD[1024], ref[1024];
  for (int i = 0; i < 32; i++) for (int k = 0; k < 16; k++) hA[i * 16 + k] = f2bf((float)((i * 3 + k * 5) % 7 - 3));
  for (int k = 0; k < 16; k++) for (int j = 0; j < 32; j++) hB[k * 32 + j] = f2bf((float)((k * 2 + j * 7) % 5 - 2));
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { float s = 0; for (int k = 0; k < 16; k++) s += (float)((i * 3 + k * 5) % 7 - 3) * (float)((k * 2 + j * 7) % 5 - 2); ref[i * 32 + j] = s; }
  unsigned short *dA, *dB; float* dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 1024; i++) if (hD[i] != ref[i]) bad++;
  printf("mfma_f32_32x32x16_bf16: A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31], D[(v&3)+8*(v>>2)+4*(l>>5)][l&31]: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
  return 0;
}
