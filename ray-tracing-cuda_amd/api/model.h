#pragma once
// Model<HasTexCoord>(path, transform): host-side mesh import feeding BVH<Face<...>, AABB>
// (reference: model.h:49-113, which goes through assimp).  assimp is not available in this
// build; Wavefront OBJ is parsed directly: `v`, `vt`, `f` (polygons fan-triangulated, negative
// indices allowed), `usemtl`/`mtllib` with `map_Kd` diffuse textures.  As in the reference,
// every vertex is transformed by `transform` and divided by w, one Mesh is produced per
// material (an OBJ without materials yields meshes[0]), and a mesh's diffuse image is loaded
// with stbi_load from <parent of the model's directory>/textures/<basename> (model.h:81-82).
#include <stb_image.h>

#include <cstdio>
#include <cstring>
#include <glm/glm.hpp>
#include <map>
#include <string>
#include <vector>

#include "bvh.cuh"
#include "utils.cuh"

template <bool HasTexCoord>
struct Mesh {
 public:
  std::vector<Face<HasTexCoord>> faces;
  int32_t texture_id = -1;
};

struct Image {
 public:
  int height = 0, width = 0;
  std::string data;
};

namespace rt_model {
template <bool B>
inline void set_vertex(Face<B> *f, int k, glm::vec3 p, glm::vec2 uv) {
  f->position(k) = p;
  f->tex_coord(k) = uv;
}
template <>
inline void set_vertex<false>(Face<false> *f, int k, glm::vec3 p, glm::vec2) {
  f->position(k) = p;
}
inline int resolve(int idx, int count) { return idx > 0 ? idx - 1 : count + idx; }
}  // namespace rt_model

template <bool HasTexCoord>
struct Model {
 public:
  std::vector<Mesh<HasTexCoord>> meshes;
  std::vector<Image> textures;

  Model(const std::string &path, glm::mat4 transform) {
    FILE *f = std::fopen(path.c_str(), "r");
    CHECK(f != nullptr) << "cannot open model " << path;
    std::vector<glm::vec3> pos;
    std::vector<glm::vec2> uv;
    std::map<std::string, int> material_index;
    std::map<std::string, std::string> material_texture;
    int current = 0;
    meshes.resize(1);
    textures.resize(1);
    char line[4096];
    while (std::fgets(line, sizeof(line), f)) {
      if (line[0] == 'v' && line[1] == ' ') {
        float x, y, z;
        if (std::sscanf(line + 2, "%f %f %f", &x, &y, &z) == 3) {
          glm::vec4 v = transform * glm::vec4(x, y, z, 1);
          pos.push_back(glm::vec3(v) / v.w);
        }
      } else if (line[0] == 'v' && line[1] == 't') {
        float u = 0, v = 0;
        std::sscanf(line + 3, "%f %f", &u, &v);
        uv.push_back(glm::vec2(u, v));
      } else if (line[0] == 'f' && line[1] == ' ') {
        int vi[64], ti[64], n = 0;
        char *p = line + 2;
        while (*p && n < 64) {
          while (*p == ' ' || *p == '\t') p++;
          if (*p == '\n' || *p == '\r' || !*p) break;
          int a = 0, b = 0;
          a = (int)std::strtol(p, &p, 10);
          if (*p == '/') {
            p++;
            if (*p != '/') b = (int)std::strtol(p, &p, 10);
            if (*p == '/') {
              p++;
              (void)std::strtol(p, &p, 10);
            }
          }
          vi[n] = rt_model::resolve(a, (int)pos.size());
          ti[n] = b ? rt_model::resolve(b, (int)uv.size()) : -1;
          n++;
        }
        for (int k = 1; k + 1 < n; k++) {
          Face<HasTexCoord> face;
          const int idx[3] = {0, k, k + 1};
          for (int c = 0; c < 3; c++) {
            glm::vec2 t(0);
            if (HasTexCoord && ti[idx[c]] >= 0 && ti[idx[c]] < (int)uv.size()) t = uv[ti[idx[c]]];
            rt_model::set_vertex<HasTexCoord>(&face, c, pos[vi[idx[c]]], t);
          }
          meshes[current].faces.emplace_back(face);
        }
      } else if (!std::strncmp(line, "usemtl ", 7)) {
        std::string name = trim(line + 7);
        auto it = material_index.find(name);
        if (it == material_index.end()) {
          // the first material reuses mesh 0 while it is still empty
          int id = (material_index.empty() && meshes[0].faces.empty()) ? 0 : (int)meshes.size();
          if (id == (int)meshes.size()) {
            meshes.emplace_back();
            textures.emplace_back();
          }
          it = material_index.emplace(name, id).first;
          auto tx = material_texture.find(name);
          if (tx != material_texture.end()) LoadTexture(path, id, tx->second);
        }
        current = it->second;
      } else if (!std::strncmp(line, "mtllib ", 7)) {
        ParseMtl(JoinDir(path, trim(line + 7)), &material_texture);
      }
    }
    std::fclose(f);
    LOG(INFO) << "model \"" << path << "\": " << pos.size() << " vertices, " << meshes.size() << " mesh(es), "
              << meshes[0].faces.size() << " faces in mesh 0";
  }

 private:
  static std::string trim(const char *s) {
    std::string r(s);
    while (!r.empty() && (r.back() == '\n' || r.back() == '\r' || r.back() == ' ')) r.pop_back();
    size_t b = 0;
    while (b < r.size() && r[b] == ' ') b++;
    return r.substr(b);
  }
  static std::string JoinDir(const std::string &model_path, const std::string &name) {
    size_t s = model_path.find_last_of("/\\");
    return s == std::string::npos ? name : model_path.substr(0, s + 1) + name;
  }
  static void ParseMtl(const std::string &path, std::map<std::string, std::string> *out) {
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return;
    char line[4096];
    std::string cur;
    while (std::fgets(line, sizeof(line), f)) {
      if (!std::strncmp(line, "newmtl ", 7)) cur = trim(line + 7);
      if (!std::strncmp(line, "map_Kd ", 7) && !cur.empty()) (*out)[cur] = trim(line + 7);
    }
    std::fclose(f);
  }
  void LoadTexture(const std::string &root_path, int id, const std::string &name) {
    std::string image_path = ParentPath(ParentPath(root_path)) + "/textures/" + BaseName(name);
    LOG(INFO) << "loading texture at: \"" << image_path << "\"";
    int channels = 0;
    unsigned char *data = stbi_load(image_path.c_str(), &textures[id].width, &textures[id].height, &channels, 4);
    if (!data) {
      textures[id].width = textures[id].height = 0;
      return;
    }
    textures[id].data = std::string(data, data + (size_t)textures[id].width * textures[id].height * 4);
    stbi_image_free(data);
    meshes[id].texture_id = id;
  }
};
