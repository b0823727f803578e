// Launch throughput of T host threads, each enqueuing tiny kernels on its own stream: does hipLaunchKernel scale across
// threads on this runtime?   build: hipcc -O2 --offload-arch=gfx950 -o tools/bin/launch_mt tools/launch_mt.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <atomic>
struct Args { const float* a; const float* b; float* c; int n, m, k, l; float s, t; const int* p; const int* q; };
__global__ void tiny(Args a) { if (a.n < 0) a.c[threadIdx.x] = a.s; }
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 20000;
  for (int T : {1, 2, 4}) {
    std::vector<hipStream_t> st(T);
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    std::atomic<int> go{0};
    std::vector<double> us(T);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
      hipSetDevice(0);
      Args a{}; a.n = 1;
      for (int i = 0; i < 200; ++i) tiny<<<1, 64, 0, st[t]>>>(a);
      hipStreamSynchronize(st[t]);
      go.fetch_add(1);
      while (go.load() < T) {}
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) tiny<<<1, 64, 0, st[t]>>>(a);
      auto t1 = std::chrono::steady_clock::now();
      hipStreamSynchronize(st[t]);
      auto t2 = std::chrono::steady_clock::now();
      us[t] = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
      (void)t2;
    });
    for (auto& x : th) x.join();
    printf("threads %d: host us per launch per thread:", T);
    for (double u : us) printf(" %.2f", u);
    printf("\n");
  }
  return 0;
}
