// host_window_copy.hip — what the host pays to move audio through the session's windows (nam_hip_api.cpp: host_windows):
// rows of `frames` floats copied into fine-grained device memory through the PCIe BAR (write-combining stores) and out of
// host-mapped memory, for row strides of 1x and 4x the row length, glibc memcpy vs non-temporal 32-byte stores; and the cost of
// a hipStreamQuery on an idle stream.  hipcc -O2 -mavx2 --offload-arch=gfx950 host_window_copy.hip -o host_window_copy
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

static double now_us()
{
  return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now().time_since_epoch()).count();
}

static void nt_copy(float* dst, const float* src, size_t n_floats)
{
  size_t i = 0;
  for (; i + 8 <= n_floats; i += 8)
    _mm256_stream_ps(dst + i, _mm256_loadu_ps(src + i));
  for (; i < n_floats; i++)
    dst[i] = src[i];
}

int main()
{
  const int rows = 256;
  for (int frames : {64, 256, 1024})
    for (int mult : {1, 4})
    {
      const size_t stride = (size_t)frames * mult, total = (size_t)rows * stride;
      float *bar = nullptr, *hmap = nullptr;
      if (hipExtMallocWithFlags((void**)&bar, total * 4, hipDeviceMallocFinegrained) != hipSuccess)
        return 1;
      if (hipHostMalloc((void**)&hmap, total * 4, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
        return 1;
      std::vector<float> src((size_t)rows * frames, 0.25f), dst((size_t)rows * frames);
      std::memset(hmap, 0, total * 4);
      const int reps = frames == 64 ? 2000 : frames == 256 ? 800 : 200;
      for (int form = 0; form < 2; form++)
      {
        double t0 = now_us();
        for (int k = 0; k < reps; k++)
        {
          for (int r = 0; r < rows; r++)
            form ? nt_copy(bar + r * stride, src.data() + (size_t)r * frames, frames) : (void)std::memcpy(bar + r * stride, src.data() + (size_t)r * frames, (size_t)frames * 4);
          _mm_sfence();
        }
        const double t_in = (now_us() - t0) / reps;
        t0 = now_us();
        for (int k = 0; k < reps; k++)
          for (int r = 0; r < rows; r++)
            std::memcpy(dst.data() + (size_t)r * frames, hmap + r * stride, (size_t)frames * 4);
        const double t_out = (now_us() - t0) / reps;
        const double mb = rows * frames * 4 / 1e6;
        std::printf("%4d frames x %d rows, stride %dx, %s: in %.1f us (%.1f GB/s), out %.1f us (%.1f GB/s)\n", frames, rows, mult,
                    form ? "nt stores" : "memcpy   ", t_in, mb / t_in * 1e3, t_out, mb / t_out * 1e3);
      }
      (void)hipFree(bar);
      (void)hipHostFree(hmap);
    }
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  double t0 = now_us();
  int ok = 0;
  for (int k = 0; k < 20000; k++)
    ok += hipStreamQuery(s) == hipSuccess;
  std::printf("hipStreamQuery on an idle stream: %.2f us (%d ok)\n", (now_us() - t0) / 20000, ok);
  return 0;
}
