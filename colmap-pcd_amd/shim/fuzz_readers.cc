// Host-only robustness fuzz of the PLY / pose readers: corrupt, truncate and mislabel valid files; the readers must return
// true or false without touching memory they do not own.  Build and run by hand (from colmap-pcd_amd/):
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -I. shim/fuzz_readers.cc -o /tmp/fuzz_readers && /tmp/fuzz_readers 30000
// (2026-10-04: 16 518 files parsed, 13 482 rejected, no sanitizer report)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include <map>
#include <array>
#include <cstdint>
#include "shim/ply_reader.h"
#include "shim/pose_reader.h"

static std::string make_ply(std::mt19937& g, bool binary, int n) {
  std::string s = "ply\nformat ";
  s += binary ? "binary_little_endian" : "ascii";
  s += " 1.0\ncomment x\nelement vertex " + std::to_string(n) +
       "\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n";
  if (g() % 2) s += "property uchar red\nproperty uchar green\nproperty uchar blue\n";
  const bool rgb = s.find("red") != std::string::npos;
  s += "end_header\n";
  std::uniform_real_distribution<float> u(-50, 50);
  for (int i = 0; i < n; ++i) {
    float v[6]; for (float& f : v) f = u(g);
    if (binary) { s.append((const char*)v, 24); if (rgb) s.append("\x01\x02\x03", 3); }
    else { char b[256]; snprintf(b, sizeof b, "%g %g %g %g %g %g%s\n", v[0], v[1], v[2], v[3], v[4], v[5], rgb ? " 1 2 3" : ""); s += b; }
  }
  return s;
}
static std::string make_pose(std::mt19937& g, int n) {
  std::string s = "ply\nformat ascii 1.0\nelement vertex " + std::to_string(n) +
                  "\nproperty float x\nproperty float y\nproperty float z\nproperty float roll\nproperty float pitch\nproperty float yaw\nend_header\n";
  std::uniform_real_distribution<float> u(-3, 3);
  for (int i = 0; i < n; ++i) { char b[256]; snprintf(b, sizeof b, "%g %g %g %g %g %g\n", u(g), u(g), u(g), u(g), u(g), u(g)); s += b; }
  return s;
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  std::mt19937 g(12345);
  const std::string path = "/tmp/pcdhip_fuzz_readers.ply";
  int ok = 0, bad = 0;
  for (int it = 0; it < iters; ++it) {
    const int kind = g() % 3;
    std::string s = kind == 2 ? make_pose(g, g() % 40) : make_ply(g, kind == 1, g() % 200);
    const int mode = g() % 5;
    if (mode == 1 && !s.empty()) s.resize(g() % s.size());                       // truncate
    if (mode == 2) for (int k = 0; k < 1 + (int)(g() % 8) && !s.empty(); ++k) s[g() % s.size()] = (char)(g() % 256);   // flip bytes
    if (mode == 3) { size_t p = s.find("vertex "); if (p != std::string::npos) s.replace(p + 7, 1, std::to_string(g() % 4000000000u)); }   // lie about the count
    if (mode == 4) { size_t p = s.find("float"); if (p != std::string::npos) s.replace(p, 5, (g() % 2) ? "double" : "list uchar int"); }
    { std::ofstream f(path, std::ios::binary); f.write(s.data(), (std::streamsize)s.size()); }
    std::vector<float> xyz, nrm;
    std::map<uint32_t, std::array<double, 7>> poses;
    bool r = kind == 2 ? colmap_hip::LoadPosePly(path, &poses) : colmap_hip::ReadPlyXYZNormal(path, &xyz, &nrm);
    if (r) { ++ok; if (kind != 2 && xyz.size() != nrm.size()) { printf("size mismatch\n"); return 1; } } else ++bad;
  }
  printf("OK: %d files parsed, %d rejected, no crash\n", ok, bad);
  return 0;
}
