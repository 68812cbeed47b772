// Probe: does the SGPR offset of a raw buffer load take part in the range check on gfx950?
// One descriptor (base = start of a tensor, num_records = one row) + soffset = byte offset of the row: if the check
// only covers voffset (+ inst offset), every row of the tensor can be addressed through ONE descriptor while taps left
// of column 0 / right of column W-1 still read as 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __amdgpu_buffer_rsrc_t Rsrc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void probe(const float* base, int W, int row, float* out) {
  Rsrc r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, W * 4, 0x00020000);
  const int soff = row * W * 4;
  const int lane = threadIdx.x;
  // lane i reads column (i - 4): lanes 0..3 left of the row (negative voffset), lanes >= W+4 right of it
  const int col = lane - 4;
  out[lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, col * 4, soff, 0));
  // store path
}
__global__ void probe_store(float* base, int W, int row) {
  Rsrc r = __builtin_amdgcn_make_buffer_rsrc(base, 0, W * 4, 0x00020000);
  const int col = (int)threadIdx.x - 4;
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 1000.0f + col), r, col * 4, row * W * 4, 0);
}

int main() {
  const int W = 40, H = 6;
  float h[W * H], *d, *o, ho[64];
  for (int i = 0; i < W * H; ++i) h[i] = (float)(i / W) * 100.0f + (float)(i % W) + 1.0f;  // row*100 + col + 1
  CK(hipMalloc(&d, sizeof(h))); CK(hipMalloc(&o, 64 * 4));
  CK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
  for (int row : {0, 3, 5}) {
    probe<<<1, 64>>>(d, W, row, o);
    CK(hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost));
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
      const int col = l - 4;
      const float want = (col >= 0 && col < W) ? row * 100.0f + col + 1.0f : 0.0f;
      if (ho[l] != want) { ok = 0; printf("  row %d lane %d col %d: got %g want %g\n", row, l, col, ho[l], want); }
    }
    printf("load  row %d via soffset: %s\n", row, ok ? "OK (soffset outside the range check)" : "MISMATCH");
  }
  probe_store<<<1, 64>>>(d, W, 2);
  CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  int ok = 1;
  for (int i = 0; i < W * H; ++i) {
    const int rr = i / W, c = i % W;
    const float want = (rr == 2) ? 1000.0f + c : rr * 100.0f + c + 1.0f;
    if (h[i] != want) { ok = 0; printf("  store: [%d,%d] got %g want %g\n", rr, c, h[i], want); }
  }
  printf("store row 2 via soffset: %s\n", ok ? "OK" : "MISMATCH");
  return 0;
}
