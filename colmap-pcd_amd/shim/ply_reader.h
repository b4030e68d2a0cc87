// ply_reader.h -- minimal PLY ingest for the LiDAR map.
//
// The reference loads the map with pcl::io::loadPLYFile<lidarpt::Point>(path, cloud) (lidar/ply.cc:14):
// per vertex the float fields x, y, z, normal_x, normal_y, normal_z (lidar/pt_type.h:22-30); everything
// else in the file is ignored.  This reader restates that contract without PCL: `ascii 1.0` and
// `binary_little_endian 1.0`, the vertex element with scalar properties of any PLY type (converted to
// float), normals named normal_x/normal_y/normal_z or nx/ny/nz, missing normals left at 0 (such points
// are later rejected by lidar/ply.cc:101, ||n|| < 1e-6).  Elements after `vertex` (faces ...) are not
// read; list properties inside the vertex element are not supported (returns false, like a failed load).
#pragma once
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

namespace colmap_hip {

inline bool ReadPlyXYZNormal(const std::string& path, std::vector<float>* xyz, std::vector<float>* nrm) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  struct Prop { std::string type, name; };
  std::vector<Prop> props;
  bool in_vertex = false, seen_vertex = false, binary = false, ok_magic = false, bad = false;
  size_t nverts = 0;
  char line[1024];
  bool header_done = false;
  int lineno = 0;
  while (std::fgets(line, sizeof line, f)) {
    std::string s(line);
    while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
    if (lineno++ == 0) { ok_magic = (s == "ply"); if (!ok_magic) break; continue; }
    std::istringstream is(s);
    std::string tok;
    is >> tok;
    if (tok == "format") {
      std::string fmt;
      is >> fmt;
      if (fmt == "binary_little_endian") binary = true;
      else if (fmt != "ascii") bad = true;   // big endian: not supported
    } else if (tok == "element") {
      std::string name;
      size_t cnt = 0;
      is >> name >> cnt;
      if (name == "vertex" && !seen_vertex) { in_vertex = true; seen_vertex = true; nverts = cnt; }
      else { if (!seen_vertex && cnt > 0) bad = true; in_vertex = false; }   // data before the vertices: unsupported
    } else if (tok == "property" && in_vertex) {
      Prop p;
      is >> p.type;
      if (p.type == "list") { bad = true; continue; }
      is >> p.name;
      props.push_back(p);
    } else if (tok == "end_header") {
      header_done = true;
      break;
    }
  }
  if (!ok_magic || !header_done || bad || !seen_vertex) { std::fclose(f); return false; }
  auto size_of = [](const std::string& t) -> int {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return -1;
  };
  int slot[6] = {-1, -1, -1, -1, -1, -1};   // property index of x y z nx ny nz
  std::vector<int> psize(props.size()), poff(props.size());
  int stride = 0;
  for (size_t i = 0; i < props.size(); ++i) {
    const int sz = size_of(props[i].type);
    if (sz < 0) { std::fclose(f); return false; }
    psize[i] = sz; poff[i] = stride; stride += sz;
    const std::string& n = props[i].name;
    if (n == "x") slot[0] = (int)i; else if (n == "y") slot[1] = (int)i; else if (n == "z") slot[2] = (int)i;
    else if (n == "normal_x" || n == "nx") slot[3] = (int)i;
    else if (n == "normal_y" || n == "ny") slot[4] = (int)i;
    else if (n == "normal_z" || n == "nz") slot[5] = (int)i;
  }
  if (slot[0] < 0 || slot[1] < 0 || slot[2] < 0) { std::fclose(f); return false; }
  // a corrupt header must not drive the allocation: the vertex data cannot be larger than what is left of the
  // file (binary: stride bytes per vertex; ascii: at least "0 " per property)
  {
    const long body = std::ftell(f);
    if (body < 0 || std::fseek(f, 0, SEEK_END) != 0) { std::fclose(f); return false; }
    const long end = std::ftell(f);
    if (end < body || std::fseek(f, body, SEEK_SET) != 0) { std::fclose(f); return false; }
    const size_t left = (size_t)(end - body);
    const size_t min_per_vertex = binary ? (size_t)stride : 2 * props.size();
    if (min_per_vertex == 0 || nverts > left / min_per_vertex) { std::fclose(f); return false; }
  }
  xyz->assign(3 * nverts, 0.f);
  nrm->assign(3 * nverts, 0.f);
  auto as_float = [](const std::string& t, const unsigned char* p) -> float {
    if (t == "float" || t == "float32") { float v; std::memcpy(&v, p, 4); return v; }
    if (t == "double" || t == "float64") { double v; std::memcpy(&v, p, 8); return (float)v; }
    if (t == "char" || t == "int8") return (float)*(const int8_t*)p;
    if (t == "uchar" || t == "uint8") return (float)*p;
    if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, p, 2); return (float)v; }
    if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, p, 2); return (float)v; }
    if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, p, 4); return (float)v; }
    uint32_t v; std::memcpy(&v, p, 4); return (float)v;
  };
  bool ok = true;
  if (binary) {
    std::vector<unsigned char> buf((size_t)stride * 65536);
    size_t done = 0;
    while (done < nverts && ok) {
      const size_t n = std::min<size_t>(65536, nverts - done);
      if (std::fread(buf.data(), stride, n, f) != n) { ok = false; break; }
      for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 6; ++k)
          if (slot[k] >= 0) {
            const float v = as_float(props[slot[k]].type, buf.data() + i * stride + poff[slot[k]]);
            (k < 3 ? (*xyz)[3 * (done + i) + k] : (*nrm)[3 * (done + i) + k - 3]) = v;
          }
      done += n;
    }
  } else {
    for (size_t i = 0; i < nverts && ok; ++i) {
      if (!std::fgets(line, sizeof line, f)) { ok = false; break; }
      // every token goes through strtod: it reads "nan", "-inf", "-nan" with their sign, which operator>>
      // does not (it fails on them after consuming the '-')
      const char* cur = line;
      for (size_t p = 0; p < props.size(); ++p) {
        char* endp = nullptr;
        const double v = std::strtod(cur, &endp);
        if (endp == cur) { ok = false; break; }   // too few / unparsable tokens on the line
        cur = endp;
        for (int k = 0; k < 6; ++k)
          if (slot[k] == (int)p) (k < 3 ? (*xyz)[3 * i + k] : (*nrm)[3 * i + k - 3]) = (float)v;
      }
    }
  }
  std::fclose(f);
  return ok;
}

}  // namespace colmap_hip
