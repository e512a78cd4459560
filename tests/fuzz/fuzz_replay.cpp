// tests/fuzz/fuzz_replay.cpp — TEST-ONLY mutation fuzzer of the file readers of csrc/vio_replay.cpp (PNG decoder, IMU stream),
// built with -fsanitize=address,undefined by tests/test_replay_cpu.py: recordings are files from outside, a malformed one
// must come back as an error code, never as a memory error. Seeds are PNGs written by the library itself; mutations flip
// bytes / bits, truncate and overwrite header fields, and three times out of four the chunk CRCs are recomputed so that
// the mutated data reaches inflate, the filters and the colour conversion.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <zlib.h>
#include "vio_amd.h"
static void fix_crcs(std::vector<uint8_t>&b){ size_t pos=8; while(pos+12<=b.size()){ unsigned len=(b[pos]<<24)|(b[pos+1]<<16)|(b[pos+2]<<8)|b[pos+3]; if(pos+12+(size_t)len>b.size()) break; unsigned long c=crc32(crc32(0L,Z_NULL,0),&b[pos+4],len+4); b[pos+8+len]=c>>24; b[pos+9+len]=c>>16; b[pos+10+len]=c>>8; b[pos+11+len]=c; pos+=12+len; } }
static unsigned long long st = 88172645463325252ULL;
static unsigned rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (unsigned)(st >> 11); }
int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 100000;
  // seeds: a few valid PNGs written by the library itself
  std::vector<std::vector<uint8_t>> seeds;
  for (int ch : {1, 3, 4}) {
    int rows = 23, cols = 31;
    std::vector<uint8_t> px((size_t)rows * cols * ch);
    for (auto &p : px) p = rnd();
    if (system("rm -rf /tmp/vio_fuzz_d && mkdir -p /tmp/vio_fuzz_d") != 0) return 1;
    vio_replay_write_image("/tmp/vio_fuzz_d", 0, px.data(), rows, cols, ch);
    FILE *f = fopen("/tmp/vio_fuzz_d/0", "rb");
    std::vector<uint8_t> b(1 << 16);
    size_t n = fread(b.data(), 1, b.size(), f);
    fclose(f);
    b.resize(n);
    seeds.push_back(b);
  }
  std::vector<uint8_t> gray(1 << 20);
  long ok = 0, bad = 0;
  for (int it = 0; it < iters; it++) {
    std::vector<uint8_t> b = seeds[rnd() % seeds.size()];
    int nm = 1 + rnd() % 6;
    for (int k = 0; k < nm; k++) {
      unsigned m = rnd() % 4;
      if (m == 0) b[rnd() % b.size()] = rnd();
      else if (m == 1) b[rnd() % b.size()] ^= 1u << (rnd() % 8);
      else if (m == 2 && b.size() > 40) b.resize(b.size() - rnd() % 20);
      else { size_t p = 8 + rnd() % 30; if (p < b.size()) b[p] = rnd() % 3 ? 0xff : 0; }  // header fields (sizes, depth, type)
    }
    if (rnd() % 4) fix_crcs(b);
    int32_t r = 0, c = 0;
    int rc = vio_replay_decode_png_gray(b.data(), (int64_t)b.size(), gray.data(), (int64_t)gray.size(), &r, &c);
    if (rc == VIO_OK) ok++; else bad++;
  }
  fflush(stdout); printf("fuzz done: %ld decoded, %ld rejected\n", ok, bad);
  // IMU reader on garbage
  for (int it = 0; it < 2000; it++) {
    std::vector<uint8_t> b(1 + rnd() % 400);
    for (auto &p : b) p = rnd();
    FILE *f = fopen("/tmp/vio_fuzz_imu", "wb"); fwrite(b.data(), 1, b.size(), f); fclose(f);
    VioImuMsg m[16]; int32_t n = 0;
    vio_replay_read_imu("/tmp/vio_fuzz_imu", m, 16, &n);
  }
  printf("imu fuzz done\n");
  return 0;
}
