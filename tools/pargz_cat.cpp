// pargz_cat: gzip file -> stdout through scalce_amd/csrc/pargz.hpp (host-only test driver; tests/test_host_cpu.py).
// usage: pargz_cat FILE [threads]   ; stderr: "windows <parallel> serial_bytes <n>"; exit 1 on a damaged stream
#include <cstdio>
#include <cstdlib>
#include "../scalce_amd/csrc/pargz.hpp"
int main(int argc, char **argv) {
  if (argc < 2) return 2;
  scalce_host::ParGz z;
  if (!z.open(argv[1], argc > 2 ? atoi(argv[2]) : 8)) return 2;
  std::vector<uint8_t> buf(8u << 20);
  for (;;) {
    const int64_t k = z.read(buf.data(), buf.size());
    if (k < 0) { fprintf(stderr, "damaged\n"); return 1; }
    if (k == 0) break;
    fwrite(buf.data(), 1, (size_t)k, stdout);
  }
  fprintf(stderr, "windows %llu serial_bytes %llu\n", (unsigned long long)z.parallel_windows, (unsigned long long)z.serial_bytes);
  return 0;
}
