// NAM/multi_device.h — spreading one batch of independent signals over the GPUs of a node from ONE C++ process.
//
// The reference has no notion of devices (tools/render.cpp:146-197 renders one file on the calling thread). Streams of
// a batch never interact, so the multi-GPU form needs no collective: the signals are dealt to the devices, every device
// gets its own nam_hip_batch driven by its own host thread (a batch handle belongs to one thread at a time,
// include/nam_hip.h), and the outputs land in the caller's buffers. (The torchrun / RCCL route for one process per GPU
// is neuralampmodelercore_amd/sharding.py + bench.py.)
#pragma once

#include <algorithm>
#include <cstdint>
#include <exception>
#include <numeric>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "dsp.h"

namespace nam
{

// "all", "3", "0-7", "0,2,5", "0-3,6" — HIP_VISIBLE_DEVICES-style ordinals of the devices the process can see.
// Duplicates are allowed ("0,0": two batches, two host threads, one GPU). Throws std::invalid_argument.
inline std::vector<int> parse_device_list(const std::string& spec, int device_count)
{
  std::vector<int> out;
  if (spec.empty() || spec == "all")
  {
    for (int d = 0; d < device_count; d++)
      out.push_back(d);
    if (out.empty())
      throw std::invalid_argument("device list: no HIP device is visible");
    return out;
  }
  size_t pos = 0;
  auto number = [&](size_t& p) -> int {
    if (p >= spec.size() || spec[p] < '0' || spec[p] > '9')
      throw std::invalid_argument("device list: expected a device ordinal in \"" + spec + "\"");
    long v = 0;
    while (p < spec.size() && spec[p] >= '0' && spec[p] <= '9')
    {
      v = v * 10 + (spec[p++] - '0');
      if (v > 1000000)
        throw std::invalid_argument("device list: ordinal out of range in \"" + spec + "\"");
    }
    return (int)v;
  };
  while (true)
  {
    const int a = number(pos);
    int b = a;
    if (pos < spec.size() && spec[pos] == '-')
    {
      pos++;
      b = number(pos);
      if (b < a)
        throw std::invalid_argument("device list: descending range in \"" + spec + "\"");
    }
    for (int d = a; d <= b; d++)
    {
      if (d >= device_count)
        throw std::invalid_argument("device list: device " + std::to_string(d) + " does not exist (" + std::to_string(device_count)
                                    + " visible)");
      out.push_back(d);
    }
    if (pos == spec.size())
      break;
    if (spec[pos] != ',')
      throw std::invalid_argument("device list: unexpected character in \"" + spec + "\"");
    pos++;
  }
  return out;
}

// Deal signals to `n_devices` batches: longest first, round-robin in snake order (0..n-1, n-1..0, ...). A batch
// renders as ONE launch walking its longest signal, so every device should get the same count and a similar longest
// member; snake order evens out the sums as well. Returns, per device, the indices of its signals (ascending).
inline std::vector<std::vector<int>> deal_by_length(const std::vector<int64_t>& lengths, int n_devices)
{
  if (n_devices <= 0)
    throw std::invalid_argument("deal_by_length: no devices");
  std::vector<int> order(lengths.size());
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lengths[(size_t)a] > lengths[(size_t)b]; });
  std::vector<std::vector<int>> out((size_t)n_devices);
  for (size_t k = 0; k < order.size(); k++)
  {
    const size_t round = k / (size_t)n_devices, col = k % (size_t)n_devices;
    const size_t dev = (round & 1) ? (size_t)n_devices - 1 - col : col;
    out[dev].push_back(order[k]);
  }
  for (auto& v : out)
    std::sort(v.begin(), v.end());
  return out;
}

// N signals through one model on several devices at once. `slim`: SetSlimmableSize for every stream (< 0: leave).
// in[i] / out[i]: planar float32 host buffers of n_frames[i] frames ([channels][n_frames[i]]). Reset(sample_rate, 64)
// with the model's prewarm runs per device, as tools/render.cpp:146-147 does per file.
inline void render_on_devices(const std::shared_ptr<nam_hip_model>& model, const std::vector<int>& devices, const float* const* in,
                              float* const* out, const int64_t* n_frames, int n_signals, double sample_rate, double slim = -1.0)
{
  std::vector<int64_t> lengths(n_frames, n_frames + n_signals);
  const std::vector<std::vector<int>> deal = deal_by_length(lengths, (int)devices.size());
  std::vector<std::exception_ptr> errors(devices.size());
  std::vector<std::thread> threads;
  for (size_t d = 0; d < devices.size(); d++)
  {
    if (deal[d].empty())
      continue;
    threads.emplace_back([&, d]() {
      try
      {
        const std::vector<int>& mine = deal[d];
        const int n = (int)mine.size();
        BatchDSP dsp(model, n, devices[d]);
        dsp.Reset(sample_rate, 64);
        if (slim >= 0.0)
          dsp.SetSlimmableSize(nullptr, 0, slim);
        std::vector<const float*> ip((size_t)n);
        std::vector<float*> op((size_t)n);
        std::vector<int64_t> nf((size_t)n);
        for (int i = 0; i < n; i++)
        {
          ip[(size_t)i] = in[mine[(size_t)i]];
          op[(size_t)i] = out[mine[(size_t)i]];
          nf[(size_t)i] = n_frames[mine[(size_t)i]];
        }
        dsp.render(ip.data(), op.data(), nf.data());
      }
      catch (...)
      {
        errors[d] = std::current_exception();
      }
    });
  }
  for (auto& t : threads)
    t.join();
  for (auto& e : errors)
    if (e)
      std::rethrow_exception(e);
}

} // namespace nam
