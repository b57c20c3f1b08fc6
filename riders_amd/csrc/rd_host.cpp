// Host-only helpers of the C ABI (no HIP): compiled into libriders_hip.so AND, by plain g++, into riders_amd/libriders_host.so, which the
// data-loading code binds with ctypes WITHOUT touching the GPU runtime -- DataLoader workers (forked or spawned) decode depth maps through it
// and must neither initialise a device context nor depend on the parent having loaded the HIP library.
#include <stdint.h>
#include <stdlib.h>
#include <stddef.h>

extern "C" {

// PNG scanline un-filtering (data/data_utils.py:94-125 reads the 16-bit depth maps through PIL; PIL's writer picks Sub / Up / Average / Paeth
// per scanline, and the serial Average / Paeth recurrences cost ~0.3 s per 256x512 map as an interpreter loop).
// returns 0, 1 (bad arguments) or 2 (a filter type above 4: corrupt data)
int rd_png_unfilter_host(const uint8_t* raw, int32_t h, int32_t row_bytes, int32_t bpp, uint8_t* out) {
  if (!raw || !out || h < 0 || row_bytes <= 0 || bpp <= 0) return 1;
  for (int y = 0; y < h; y++) {
    const uint8_t* line = raw + (size_t)y * (row_bytes + 1);
    uint8_t* cur = out + (size_t)y * row_bytes;
    const uint8_t* prev = y ? cur - row_bytes : nullptr;
    const int ft = line[0];
    line++;
    if (ft > 4) return 2;
    for (int x = 0; x < row_bytes; x++) {
      const int a = x >= bpp ? cur[x - bpp] : 0, b = prev ? prev[x] : 0, c = (prev && x >= bpp) ? prev[x - bpp] : 0;
      int pred = 0;
      if (ft == 1) pred = a;
      else if (ft == 2) pred = b;
      else if (ft == 3) pred = (a + b) >> 1;
      else if (ft == 4) {
        const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
        pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
      }
      cur[x] = (uint8_t)(line[x] + pred);
    }
  }
  return 0;
}

}  // extern "C"
