// tmpfs / page-cache write throughput: one write(2) stream, T threads with pwrite, T threads copying into a shared mapping.
//   g++ -O2 -pthread -o write_bench write_bench.cpp;  ./write_bench MiB T write|pwrite|mmap PATH
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const size_t n = (size_t)atoll(argv[1]) << 20;
  const int T = atoi(argv[2]);
  const char *mode = argv[3];
  const char *path = argv[4];
  std::vector<char> src(64u << 20);
  memset(src.data(), 7, src.size());
  unlink(path);
  int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
  double t0 = now();
  if (!strcmp(mode, "write")) {
    for (size_t off = 0; off < n; off += src.size()) if (write(fd, src.data(), src.size()) < 0) return 1;
  } else if (!strcmp(mode, "pwrite")) {
    std::vector<std::thread> ts;
    for (int t = 0; t < T; t++) ts.emplace_back([&, t]() { for (size_t off = (size_t)t * src.size(); off < n; off += (size_t)T * src.size()) if (pwrite(fd, src.data(), src.size(), off) < 0) exit(1); });
    for (auto &t : ts) t.join();
  } else {
    if (ftruncate(fd, n)) return 1;
    char *m = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) return 2;
    std::vector<std::thread> ts;
    for (int t = 0; t < T; t++) ts.emplace_back([&, t]() { for (size_t off = (size_t)t * src.size(); off < n; off += (size_t)T * src.size()) memcpy(m + off, src.data(), src.size()); });
    for (auto &t : ts) t.join();
    munmap(m, n);
  }
  close(fd);
  double dt = now() - t0;
  printf("%s T=%d: %.2f s, %.2f GB/s\n", mode, T, dt, n / dt / 1e9);
  unlink(path);
  return 0;
}
