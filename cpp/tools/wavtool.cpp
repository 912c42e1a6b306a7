// wavtool — `wavtool copy in.wav out.wav`: reads with wav_io.h and writes float32 (test helper for the WAV
// reader / writer; no GPU involved).
#include <iostream>

#include "wav_io.h"

int main(int argc, char* argv[])
{
  if (argc != 4 || std::string(argv[1]) != "copy")
  {
    std::cerr << "Usage: wavtool copy <in.wav> <out.wav>\n";
    return 1;
  }
  try
  {
    const wavio::Audio a = wavio::load(argv[2]);
    wavio::save_float32(argv[3], a.samples.data(), a.samples.size(), a.sample_rate);
    std::cout << a.samples.size() << " frames @ " << a.sample_rate << " Hz\n";
  }
  catch (const std::exception& e)
  {
    std::cerr << "Error: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
