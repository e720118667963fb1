// tools/pcie_probe.hip — what the host link of the box gives, so that the host path's figure (kas_solve_host with
// every scenario's own tables: 288 MB up + 288 MB down in 29 ms = 9.8 GB/s per direction, the same with pinned caller
// buffers) can be read against it.  Copies of the sizes the host path issues (a scenario range = 36 MB up, 36 MB down):
// pageable and pinned, one direction alone, both directions at once on two streams, and the issue pattern of
// kas_solve_host_locked (upload i on stream i % 4, download i - 1 behind it).  No kernel involved.
//   hipcc --offload-arch=gfx950 -O2 -o tools/pcie_probe tools/pcie_probe.hip && tools/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const size_t chunk = (argc > 1 ? (size_t)atol(argv[1]) : 36) << 20;      // bytes per copy
  const int n = argc > 2 ? atoi(argv[2]) : 8;                              // copies per direction per measurement
  const size_t total = chunk * (size_t)n;
  char *d_up = nullptr, *d_down = nullptr, *h_pin_up = nullptr, *h_pin_down = nullptr;
  CHECK(hipMalloc(&d_up, total)); CHECK(hipMalloc(&d_down, total));
  CHECK(hipHostMalloc(&h_pin_up, total, hipHostMallocDefault)); CHECK(hipHostMalloc(&h_pin_down, total, hipHostMallocDefault));
  char* h_page_up = (char*)malloc(total); char* h_page_down = (char*)malloc(total);
  memset(h_pin_up, 1, total); memset(h_page_up, 1, total); memset(h_pin_down, 0, total); memset(h_page_down, 0, total);
  CHECK(hipMemset(d_down, 2, total));
  hipStream_t st[4];
  for (auto& s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto sync_all = [&]() { for (auto& s : st) CHECK(hipStreamSynchronize(s)); };
  auto report = [&](const char* what, double ms, double bytes) { printf("%-78s %8.2f ms  %6.1f GB/s\n", what, ms, bytes / ms / 1e6); };
  for (int pinned = 0; pinned < 2; ++pinned) {
    char* hu = pinned ? h_pin_up : h_page_up; char* hd = pinned ? h_pin_down : h_page_down;
    const char* kind = pinned ? "pinned  " : "pageable";
    char label[160];
    for (int rep = 0; rep < 2; ++rep) {                                     // (the first round touches the pages)
      sync_all(); double t0 = now_ms();
      for (int i = 0; i < n; ++i) CHECK(hipMemcpyAsync(d_up + i * chunk, hu + i * chunk, chunk, hipMemcpyHostToDevice, st[0]));
      sync_all(); double t1 = now_ms();
      for (int i = 0; i < n; ++i) CHECK(hipMemcpyAsync(hd + i * chunk, d_down + i * chunk, chunk, hipMemcpyDeviceToHost, st[1]));
      sync_all(); double t2 = now_ms();
      for (int i = 0; i < n; ++i) {
        CHECK(hipMemcpyAsync(d_up + i * chunk, hu + i * chunk, chunk, hipMemcpyHostToDevice, st[0]));
        CHECK(hipMemcpyAsync(hd + i * chunk, d_down + i * chunk, chunk, hipMemcpyDeviceToHost, st[1]));
      }
      sync_all(); double t3 = now_ms();
      // the host path's order: upload i on stream i % 4, then the download of range i - 1 on ITS stream
      for (int i = 0; i < n; ++i) {
        CHECK(hipMemcpyAsync(d_up + i * chunk, hu + i * chunk, chunk, hipMemcpyHostToDevice, st[i % 4]));
        if (i > 0) CHECK(hipMemcpyAsync(hd + (i - 1) * chunk, d_down + (i - 1) * chunk, chunk, hipMemcpyDeviceToHost, st[(i - 1) % 4]));
      }
      CHECK(hipMemcpyAsync(hd + (n - 1) * chunk, d_down + (n - 1) * chunk, chunk, hipMemcpyDeviceToHost, st[(n - 1) % 4]));
      sync_all(); double t4 = now_ms();
      if (rep == 0) continue;
      snprintf(label, sizeof label, "%s host -> device alone, %d x %zu MB on one stream", kind, n, chunk >> 20); report(label, t1 - t0, (double)total);
      snprintf(label, sizeof label, "%s device -> host alone", kind); report(label, t2 - t1, (double)total);
      snprintf(label, sizeof label, "%s both directions at once, two streams (GB/s = sum of both)", kind); report(label, t3 - t2, 2.0 * total);
      snprintf(label, sizeof label, "%s the host path's issue order over four streams (sum of both)", kind); report(label, t4 - t3, 2.0 * total);
    }
  }
  if (h_page_down[total - 1] != 2 || h_pin_down[0] != 2) { fprintf(stderr, "downloaded bytes are wrong\n"); return 1; }
  return 0;
}
