// wav_io.h — minimal RIFF/WAVE reader and float32 writer for the render tool.
// Reads what the reference's render accepts through AudioDSPTools' dsp::wav::Load (tools/render.cpp:129-136):
// mono PCM 16 / 24 / 32-bit and IEEE float 32-bit (also inside WAVE_FORMAT_EXTENSIBLE), little endian; unknown
// chunks (LIST, fact, ...) are skipped. Writes what SaveWavFloat32 writes (tools/render.cpp:20-60): mono,
// format 3 (IEEE float), 32 bits, 44-byte header.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace wavio
{

struct Audio
{
  std::vector<float> samples; // mono
  double sample_rate = 0.0;
};

namespace detail
{
inline uint32_t le32(const unsigned char* p)
{
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
inline uint16_t le16(const unsigned char* p)
{
  return (uint16_t)(p[0] | (p[1] << 8));
}
} // namespace detail

inline Audio load(const std::string& path)
{
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f)
    throw std::runtime_error("wav: cannot open " + path);
  std::vector<unsigned char> buf;
  {
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (got != buf.size())
      throw std::runtime_error("wav: short read on " + path);
  }
  if (buf.size() < 12 || std::memcmp(buf.data(), "RIFF", 4) != 0 || std::memcmp(buf.data() + 8, "WAVE", 4) != 0)
    throw std::runtime_error("wav: not a RIFF/WAVE file: " + path);
  int format = 0, channels = 0, bits = 0;
  uint32_t rate = 0;
  bool have_fmt = false;
  size_t pos = 12;
  while (pos + 8 <= buf.size())
  {
    const unsigned char* ck = buf.data() + pos;
    const uint32_t size = detail::le32(ck + 4);
    const unsigned char* body = ck + 8;
    const size_t avail = buf.size() - (pos + 8);
    if (std::memcmp(ck, "fmt ", 4) == 0)
    {
      if (size < 16 || avail < 16)
        throw std::runtime_error("wav: truncated fmt chunk");
      format = detail::le16(body);
      channels = detail::le16(body + 2);
      rate = detail::le32(body + 4);
      bits = detail::le16(body + 14);
      if (format == 0xFFFE && size >= 40 && avail >= 40) // WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
        format = detail::le16(body + 24);
      have_fmt = true;
    }
    else if (std::memcmp(ck, "data", 4) == 0)
    {
      if (!have_fmt)
        throw std::runtime_error("wav: data chunk before fmt chunk");
      if (channels != 1)
        throw std::runtime_error("wav: only mono files are supported (file has " + std::to_string(channels) + " channels)");
      const size_t bytes = size <= avail ? size : avail; // tolerate a wrong length field at end of file
      Audio a;
      a.sample_rate = (double)rate;
      if (format == 1 && (bits == 16 || bits == 24 || bits == 32))
      {
        const int bps = bits / 8;
        const size_t n = bytes / bps;
        a.samples.resize(n);
        for (size_t i = 0; i < n; i++)
        {
          const unsigned char* p = body + i * bps;
          int32_t v;
          float scale;
          if (bps == 2)
            v = (int16_t)detail::le16(p), scale = 1.0f / 32768.0f;
          else if (bps == 3)
            v = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24) >> 8, scale = 1.0f / 8388608.0f;
          else
            v = (int32_t)detail::le32(p), scale = 1.0f / 2147483648.0f;
          a.samples[i] = (float)v * scale;
        }
      }
      else if (format == 3 && bits == 32)
      {
        const size_t n = bytes / 4;
        a.samples.resize(n);
        std::memcpy(a.samples.data(), body, n * 4);
      }
      else
        throw std::runtime_error("wav: unsupported sample format (tag " + std::to_string(format) + ", " + std::to_string(bits)
                                 + " bits)");
      return a;
    }
    pos += 8 + (size_t)size + (size & 1); // chunks are word aligned
  }
  throw std::runtime_error("wav: no data chunk in " + path);
}

inline void save_float32(const std::string& path, const float* samples, size_t n, double sample_rate)
{
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f)
    throw std::runtime_error("wav: cannot open " + path + " for writing");
  const uint32_t data_size = (uint32_t)(n * sizeof(float));
  unsigned char h[44];
  auto p32 = [&](int off, uint32_t v) { h[off] = v & 255, h[off + 1] = (v >> 8) & 255, h[off + 2] = (v >> 16) & 255, h[off + 3] = v >> 24; };
  auto p16 = [&](int off, uint16_t v) { h[off] = v & 255, h[off + 1] = v >> 8; };
  std::memcpy(h, "RIFF", 4);
  p32(4, 36 + data_size);
  std::memcpy(h + 8, "WAVEfmt ", 8);
  p32(16, 16);
  p16(20, 3); // IEEE float
  p16(22, 1); // mono
  p32(24, (uint32_t)sample_rate);
  p32(28, (uint32_t)sample_rate * 4);
  p16(32, 4);
  p16(34, 32);
  std::memcpy(h + 36, "data", 4);
  p32(40, data_size);
  const bool ok = std::fwrite(h, 1, 44, f) == 44 && (n == 0 || std::fwrite(samples, sizeof(float), n, f) == n);
  if (std::fclose(f) != 0 || !ok)
    throw std::runtime_error("wav: write failed on " + path);
}

} // namespace wavio
