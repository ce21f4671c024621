#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
  const int lane = threadIdx.x;
  for (int j = 0; j < 4; ++j) lds[lane * 4 + j] = 0xDEADBEEF;
  __syncthreads();
  // buffer resource: base = src, num_records = nbytes, dword3 = gfx9 raw format flags
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  // odd lanes out of range
  unsigned voff = (lane & 1) ? 0x7fffff00u : lane * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = lds[lane * 4 + j];
}
int main() {
  unsigned *src, *out; unsigned h[256], ho[256];
  for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
  hipMalloc(&src, 1024); hipMalloc(&out, 1024);
  hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, 1024u, out);
  hipError_t e = hipDeviceSynchronize();
  printf("err=%d\n", (int)e);
  hipMemcpy(ho, out, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 6; ++l) printf("lane %d: %08x %08x %08x %08x\n", l, ho[l*4], ho[l*4+1], ho[l*4+2], ho[l*4+3]);
  return 0;
}
