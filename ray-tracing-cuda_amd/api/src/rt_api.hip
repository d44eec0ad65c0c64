// Host driver of the scene-facing API: Main / DistributedMain (utils.cu:132-242 of the
// reference) over the C ABI of librtmi.so, plus what they need around it — the flatten
// kernel that reads the device-built scene graph, the RNG-state layout conversion, the
// JPEG writer behind WriteImage, and the RCCL exchange of DistributedMain.
//
// Flow of Main (same order as the reference): allocate the four caller-owned device
// buffers -> seed one XORWOW state per pixel (rtmi_rng_init) -> run the user's init_world
// (device-side `new` of the recorder classes) -> synchronise -> flatten the world into C-ABI
// calls -> rtmi_scene_commit -> timed rtmi_render -> un-tile into d_image -> copy to the
// host -> WriteImage("image.jpeg").
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <thread>

#include "../../../include/rtmi.h"
#include "../bvh.cuh"
#include "../camera.cuh"
#include "../dielectric.cuh"
#include "../diffuse_light.cuh"
#include "../hitable_list.cuh"
#include "../lambertian.cuh"
#include "../metal.cuh"
#include "../parallelepiped.cuh"
#include "../parallelogram.cuh"
#include "../sky.cuh"
#include "../sphere.cuh"
#include "../textures/constant_texture.cuh"
#include "../textures/image_texture.cuh"
#include "../triangle.cuh"
#include "../utils.cuh"
#include "rt_jpeg.h"

using glm::vec3;

#define RT_HIP(expr)                                                    \
  do {                                                                  \
    hipError_t e__ = (expr);                                            \
    CHECK(e__ == hipSuccess) << #expr << ": " << hipGetErrorString(e__); \
  } while (0)
#define RT_ABI(expr)                                                \
  do {                                                              \
    int rc__ = (expr);                                              \
    CHECK(rc__ >= 0) << #expr << " failed (" << rc__ << "): " << rtmi_last_error(); \
  } while (0)

// ---------------------------------------------------------------- small host helpers
std::string BaseName(const std::string &path) {
  size_t s = path.find_last_of("/\\");
  return s == std::string::npos ? path : path.substr(s + 1);
}

// Parent directory; a trailing separator is ignored ("a/b/" -> "a").
std::string ParentPath(const std::string &path) {
  if (path.size() < 2) return "";
  size_t s = path.find_last_of("/\\", path.size() - 2);
  return s == std::string::npos ? "" : path.substr(0, s);
}

static int env_int(const char *name, int dflt) {
  const char *v = std::getenv(name);
  return (v && *v) ? std::atoi(v) : dflt;
}

// WriteImage (utils.cu:87-96): float -> uint8 by `* 255` truncation, JPEG quality 100.
void WriteImage(const std::vector<glm::vec3> &pixels, int height, int width, const std::string &path) {
  std::vector<uint8_t> data(pixels.size() * 3);
  for (size_t i = 0; i < pixels.size(); i++)
    for (int j = 0; j < 3; j++) data[i * 3 + j] = (uint8_t)(pixels[i][j] * 255);
  LOG(INFO) << "Writing image to " << path << "...";
  CHECK(rt_write_jpeg(path.c_str(), width, height, data.data())) << "cannot write " << path;
}

// ---------------------------------------------------------------- RNG state layout
// The scene sees curandState as an array of structures (it may draw from &d_states[i] inside
// init_world, scenes/spheres.cu:105); the trace kernel keeps six struct-of-arrays planes
// indexed by work item.  These two kernels move states between the layouts.
__global__ void rt_states_to_aos(rtmi_frame f, const int64_t *pixel_of, int64_t items, const uint32_t *soa,
                                 curandState *aos) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= items) return;
  int64_t px = pixel_of[q];
  if (px < 0) return;
  curandState s;
  s.d = soa[q];
  for (int k = 0; k < 5; k++) s.v[k] = soa[(k + 1) * items + q];
  aos[px] = s;
}
__global__ void rt_states_to_soa(rtmi_frame f, const int64_t *pixel_of, int64_t items, const curandState *aos,
                                 uint32_t *soa) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= items) return;
  int64_t px = pixel_of[q];
  if (px < 0) return;
  curandState s = aos[px];
  soa[q] = s.d;
  for (int k = 0; k < 5; k++) soa[(k + 1) * items + q] = s.v[k];
}

// ---------------------------------------------------------------- flatten
// One record per world-list entry, materials and textures de-duplicated by address.
struct RtObjRec {
  int kind, material, n, has_uv;
  float f[54];
  double radius;
  const void *ptr;
};
struct RtMatRec {
  int kind, texture;
  float rgb[3], param;
  double index;
};
struct RtTexRec {
  int kind;
  float rgb[3];
  RtImageDesc image;
};
struct RtFlat {
  int n_obj, n_mat, n_tex, error;
};
constexpr int kMaxObj = 16384, kMaxMat = 4096, kMaxTex = 4096;  // records, incl. the entries and brackets of nested lists

__device__ int rt_tex_id(const Texture *t, const Texture **seen, RtTexRec *out, RtFlat *fl) {
  for (int i = 0; i < fl->n_tex; i++)
    if (seen[i] == t) return i;
  if (fl->n_tex >= kMaxTex) {
    fl->error = 2;
    return 0;
  }
  int id = fl->n_tex++;
  seen[id] = t;
  RtTexRec r{};
  r.kind = t->rt_kind_;
  if (t->rt_kind_ == rtapi::T_CONSTANT) {
    const ConstantTexture *c = static_cast<const ConstantTexture *>(t);
    r.rgb[0] = c->color_.x, r.rgb[1] = c->color_.y, r.rgb[2] = c->color_.z;
  } else {
    const ImageTexture *im = static_cast<const ImageTexture *>(t);
    r.image = *reinterpret_cast<const RtImageDesc *>((uintptr_t)im->image_texture_);
  }
  out[id] = r;
  return id;
}

__device__ int rt_mat_id(const Material *m, const Material **seen, RtMatRec *out, const Texture **tseen,
                         RtTexRec *tout, RtFlat *fl) {
  if (!m) return -1;
  for (int i = 0; i < fl->n_mat; i++)
    if (seen[i] == m) return i;
  if (fl->n_mat >= kMaxMat) {
    fl->error = 3;
    return 0;
  }
  int id = fl->n_mat++;
  seen[id] = m;
  RtMatRec r{};
  r.kind = m->rt_kind_;
  r.texture = -1;
  switch (m->rt_kind_) {
    case rtapi::M_LAMBERTIAN: {
      const Lambertian *l = static_cast<const Lambertian *>(m);
      if (l->use_constant_tex_) {
        r.rgb[0] = l->color_.color_.x, r.rgb[1] = l->color_.color_.y, r.rgb[2] = l->color_.color_.z;
      } else {
        r.texture = rt_tex_id(l->texture_ptr_, tseen, tout, fl);
      }
      break;
    }
    case rtapi::M_METAL: {
      const Metal *mt = static_cast<const Metal *>(m);
      r.rgb[0] = mt->albedo_.x, r.rgb[1] = mt->albedo_.y, r.rgb[2] = mt->albedo_.z;
      r.param = mt->fuzz_;
      break;
    }
    case rtapi::M_DIELECTRIC: {
      const Dielectric *d = static_cast<const Dielectric *>(m);
      r.rgb[0] = d->attenuation_.x, r.rgb[1] = d->attenuation_.y, r.rgb[2] = d->attenuation_.z;
      r.index = d->refractive_index_;
      break;
    }
    case rtapi::M_DIFFUSE_LIGHT:
      r.texture = rt_tex_id(static_cast<const DiffuseLight *>(m)->texture_ptr_, tseen, tout, fl);
      break;
    default:
      break;
  }
  out[id] = r;
  return id;
}

// Record kinds that bracket the entries of a nested HitableList (hitable_list.cuh:8: a list is a
// Hitable and can be appended to a list); the host replays them as rtmi_list_begin / rtmi_list_end.
enum : int { RT_LIST_BEGIN = 1001, RT_LIST_END = 1002 };
constexpr int kMaxListDepth = 16;

__global__ void rt_flatten(const HitableList *world, RtObjRec *objs, RtMatRec *mats, RtTexRec *texs,
                           const Material **mseen, const Texture **tseen, RtFlat *fl) {
  fl->n_obj = fl->n_mat = fl->n_tex = fl->error = 0;
  // depth-first over the lists with an explicit stack: (list, next entry)
  const HitableList *lists[kMaxListDepth];
  int next[kMaxListDepth];
  int depth = 0;
  lists[0] = world, next[0] = 0;
  while (depth >= 0) {
    const HitableList *cur = lists[depth];
    if (next[depth] >= cur->list_len()) {
      if (depth > 0) {
        if (fl->n_obj >= kMaxObj) {
          fl->error = 1;
          return;
        }
        RtObjRec e{};
        e.kind = RT_LIST_END;
        objs[fl->n_obj++] = e;
      }
      depth--;
      continue;
    }
    const int i = next[depth]++;
    if (fl->n_obj >= kMaxObj) {
      fl->error = 1;
      return;
    }
    if (cur->at(i)->rt_kind_ == rtapi::H_LIST) {
      if (depth + 1 >= kMaxListDepth) {
        fl->error = 5;  // lists nested deeper than this walk's stack
        return;
      }
      RtObjRec b{};
      b.kind = RT_LIST_BEGIN;
      objs[fl->n_obj++] = b;
      depth++;
      lists[depth] = static_cast<const HitableList *>(cur->at(i));
      next[depth] = 0;
      continue;
    }
    const Hitable *h = cur->at(i);
    RtObjRec r{};
    r.kind = h->rt_kind_;
    r.material = -1;
    switch (h->rt_kind_) {
      case rtapi::H_SPHERE: {
        const Sphere *s = static_cast<const Sphere *>(h);
        vec3 c = s->position();
        r.f[0] = c.x, r.f[1] = c.y, r.f[2] = c.z;
        r.radius = s->radius();
        r.material = rt_mat_id(s->material_ptr(), mseen, mats, tseen, texs, fl);
        break;
      }
      case rtapi::H_TRIANGLE: {
        const Triangle *t = static_cast<const Triangle *>(h);
        for (int k = 0; k < 3; k++) r.f[k * 3] = t->p_[k].x, r.f[k * 3 + 1] = t->p_[k].y, r.f[k * 3 + 2] = t->p_[k].z;
        r.material = rt_mat_id(t->material_ptr_, mseen, mats, tseen, texs, fl);
        break;
      }
      case rtapi::H_PARALLELOGRAM: {
        const Parallelogram *t = static_cast<const Parallelogram *>(h);
        for (int k = 0; k < 3; k++) r.f[k * 3] = t->p_[k].x, r.f[k * 3 + 1] = t->p_[k].y, r.f[k * 3 + 2] = t->p_[k].z;
        r.material = rt_mat_id(t->material_ptr_, mseen, mats, tseen, texs, fl);
        break;
      }
      case rtapi::H_PARALLELEPIPED: {
        const Parallelepiped *b = static_cast<const Parallelepiped *>(h);
        for (int fc = 0; fc < 6; fc++)
          for (int k = 0; k < 3; k++) {
            const vec3 &p = b->faces_[fc][k];
            r.f[(fc * 3 + k) * 3] = p.x, r.f[(fc * 3 + k) * 3 + 1] = p.y, r.f[(fc * 3 + k) * 3 + 2] = p.z;
          }
        r.material = rt_mat_id(b->material_ptr_, mseen, mats, tseen, texs, fl);
        break;
      }
      case rtapi::H_SKY:
        break;
      case rtapi::H_BVH: {
        const RtBvhBase *b = static_cast<const RtBvhBase *>(h);
        r.ptr = b->objs_;
        r.n = b->n_;
        r.has_uv = b->has_tex_coord_;
        r.material = rt_mat_id(b->material_ptr_, mseen, mats, tseen, texs, fl);
        break;
      }
      default:
        fl->error = 4;  // an unknown hitable
        return;
    }
    objs[fl->n_obj++] = r;
  }
}

template <class T>
static T *dev_alloc(size_t n) {
  T *p = nullptr;
  RT_HIP(hipMalloc((void **)&p, n * sizeof(T)));
  return p;
}

// Replays the device-built world through the C ABI's scene recorder.
static rtmi_scene *rt_build_scene(const HitableList *d_world, const Camera *d_camera, int height, int width,
                                  bool resolution_overridden) {
  RtObjRec *d_objs = dev_alloc<RtObjRec>(kMaxObj);
  RtMatRec *d_mats = dev_alloc<RtMatRec>(kMaxMat);
  RtTexRec *d_texs = dev_alloc<RtTexRec>(kMaxTex);
  const Material **d_mseen = dev_alloc<const Material *>(kMaxMat);
  const Texture **d_tseen = dev_alloc<const Texture *>(kMaxTex);
  RtFlat *d_fl = dev_alloc<RtFlat>(1);
  hipLaunchKernelGGL(rt_flatten, dim3(1), dim3(1), 0, 0, d_world, d_objs, d_mats, d_texs, d_mseen, d_tseen, d_fl);
  RT_HIP(hipDeviceSynchronize());
  RtFlat fl;
  RT_HIP(hipMemcpy(&fl, d_fl, sizeof(fl), hipMemcpyDeviceToHost));
  CHECK(fl.error == 0) << "cannot flatten the world (code " << fl.error << ")";
  std::vector<RtObjRec> objs(fl.n_obj);
  std::vector<RtMatRec> mats(fl.n_mat);
  std::vector<RtTexRec> texs(fl.n_tex);
  if (fl.n_obj) RT_HIP(hipMemcpy(objs.data(), d_objs, objs.size() * sizeof(RtObjRec), hipMemcpyDeviceToHost));
  if (fl.n_mat) RT_HIP(hipMemcpy(mats.data(), d_mats, mats.size() * sizeof(RtMatRec), hipMemcpyDeviceToHost));
  if (fl.n_tex) RT_HIP(hipMemcpy(texs.data(), d_texs, texs.size() * sizeof(RtTexRec), hipMemcpyDeviceToHost));
  for (void *p : {(void *)d_objs, (void *)d_mats, (void *)d_texs, (void *)d_mseen, (void *)d_tseen, (void *)d_fl})
    (void)hipFree(p);

  rtmi_scene *s = rtmi_scene_create();
  std::vector<int> tex_handle(texs.size(), -1), mat_handle(mats.size(), -1);
  for (size_t i = 0; i < texs.size(); i++) {
    if (texs[i].kind == rtapi::T_CONSTANT) {
      tex_handle[i] = rtmi_constant_texture(s, texs[i].rgb);
    } else {
      const RtImageDesc &im = texs[i].image;
      std::vector<uint8_t> px((size_t)im.height * im.width * 4);
      RT_HIP(hipMemcpy2D(px.data(), (size_t)im.width * 4, im.pixels, im.pitch, (size_t)im.width * 4, im.height,
                         hipMemcpyDeviceToHost));
      tex_handle[i] = rtmi_image_texture(s, px.data(), im.height, im.width, (size_t)im.width * 4);
    }
    RT_ABI(tex_handle[i]);
  }
  for (size_t i = 0; i < mats.size(); i++) {
    const RtMatRec &m = mats[i];
    switch (m.kind) {
      case rtapi::M_LAMBERTIAN:
        mat_handle[i] = m.texture >= 0 ? rtmi_lambertian_tex(s, tex_handle[m.texture]) : rtmi_lambertian(s, m.rgb);
        break;
      case rtapi::M_METAL:
        mat_handle[i] = rtmi_metal(s, m.rgb, m.param);
        break;
      case rtapi::M_DIELECTRIC:
        mat_handle[i] = rtmi_dielectric(s, m.rgb, m.index);
        break;
      case rtapi::M_DIFFUSE_LIGHT:
        mat_handle[i] = rtmi_diffuse_light(s, tex_handle[m.texture]);
        break;
      default:
        CHECK(false) << "unknown material kind " << m.kind;
    }
    RT_ABI(mat_handle[i]);
  }
  for (const RtObjRec &o : objs) {
    const int mat = o.material >= 0 ? mat_handle[o.material] : -1;
    switch (o.kind) {
      case rtapi::H_SPHERE:
        RT_ABI(rtmi_add_sphere(s, o.f, o.radius, mat));
        break;
      case rtapi::H_TRIANGLE:
        RT_ABI(rtmi_add_triangle(s, o.f, mat));
        break;
      case rtapi::H_PARALLELOGRAM:
        RT_ABI(rtmi_add_parallelogram(s, o.f, mat));
        break;
      case rtapi::H_PARALLELEPIPED:
        RT_ABI(rtmi_add_parallelepiped_faces(s, o.f, mat));
        break;
      case rtapi::H_SKY:
        RT_ABI(rtmi_add_sky(s));
        break;
      case RT_LIST_BEGIN:
        RT_ABI(rtmi_list_begin(s));
        break;
      case RT_LIST_END:
        RT_ABI(rtmi_list_end(s));
        break;
      case rtapi::H_BVH: {
        const size_t face_bytes = o.has_uv ? sizeof(Face<true>) : sizeof(Face<false>);
        std::vector<uint8_t> raw((size_t)o.n * face_bytes);
        if (o.n) RT_HIP(hipMemcpy(raw.data(), o.ptr, raw.size(), hipMemcpyDeviceToHost));
        std::vector<float> pos((size_t)o.n * 9), uv;
        if (o.has_uv) uv.resize((size_t)o.n * 6);
        for (int i = 0; i < o.n; i++) {
          const float *f = reinterpret_cast<const float *>(raw.data() + (size_t)i * face_bytes);
          std::memcpy(&pos[(size_t)i * 9], f, 36);
          if (o.has_uv) std::memcpy(&uv[(size_t)i * 6], f + 9, 24);
        }
        RT_ABI(rtmi_add_bvh(s, pos.data(), o.has_uv ? uv.data() : nullptr, o.n, mat, 0));
        break;
      }
    }
  }
  // camera: the object the scene placement-constructed in d_camera
  alignas(Camera) unsigned char cam_raw[sizeof(Camera)];
  RT_HIP(hipMemcpy(cam_raw, d_camera, sizeof(Camera), hipMemcpyDeviceToHost));
  Camera &cam = *reinterpret_cast<Camera *>(cam_raw);
  if (resolution_overridden && cam.aspect_ > 0) {  // re-derive the image plane for the new aspect
    cam.aspect_ = double(width) / height;
    cam.plane();
  }
  const vec3 vs[7] = {cam.position_, cam.lower_left_corner_, cam.horizontal_, cam.vertical_, cam.u_, cam.v_, cam.w_};
  float frame[21];
  for (int i = 0; i < 7; i++) frame[i * 3] = vs[i].x, frame[i * 3 + 1] = vs[i].y, frame[i * 3 + 2] = vs[i].z;
  RT_ABI(rtmi_camera_set(s, frame, cam.is_defocus_camera_ ? 1 : 0, cam.lens_radius_));
  RT_ABI(rtmi_scene_commit(s));
  return s;
}

// ---------------------------------------------------------------- RCCL exchange
// One process per GPU on ONE node (the reference's mpirun ranks, scenes/spheres.cu:83-98).  rank/world
// come from the launcher environment (compat/mpi.h).  Rank 0 hands the ncclUniqueId to the others
// through a small file in RT_RENDEZVOUS_DIR (default /tmp): the file is named after, and carries, a
// token that is the same for the ranks of one launch and different for any other launch -- RT_RUN_ID
// or TORCHELASTIC_RUN_ID if set, else the launcher's pid (the ranks' common parent) -- plus
// MASTER_ADDR:MASTER_PORT.  Rank 0 removes a leftover of that name, creates the file exclusively
// under a temporary name, writes token + id, and renames it into place; the others accept a file only
// if its token matches theirs and it is not older than their own start (minus a minute).  A file left behind by a crashed run, or one of a concurrent job,
// therefore never reaches ncclCommInitRank.
struct RtComm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

struct RtIdFile {
  char magic[8];
  char token[120];
  ncclUniqueId id;
};

static std::string rt_launch_token() {
  const char *run = std::getenv("RT_RUN_ID");
  if (!run) run = std::getenv("TORCHELASTIC_RUN_ID");
  const char *addr = std::getenv("MASTER_ADDR"), *port = std::getenv("MASTER_PORT");
  std::string t = run ? std::string("run-") + run : std::string("ppid-") + std::to_string((long)getppid());
  t += std::string("-") + (addr ? addr : "127.0.0.1") + "-" + (port ? port : "0");
  for (char &c : t)
    if (!(std::isalnum((unsigned char)c) || c == '-' || c == '.')) c = '_';
  if (t.size() > 100) t.resize(100);
  return t;
}

static void rt_comm_init(RtComm *c) {
  c->rank = rt_mpi::rank();
  c->world = rt_mpi::size();
  if (c->world <= 1) return;
  const char *dir = std::getenv("RT_RENDEZVOUS_DIR");
  const std::string token = rt_launch_token();
  const std::string path = std::string(dir ? dir : "/tmp") + "/rtmi_rccl_id_" + token;
  RtIdFile rec;
  std::memset(&rec, 0, sizeof(rec));
  if (c->rank == 0) {
    CHECK(ncclGetUniqueId(&rec.id) == ncclSuccess) << "ncclGetUniqueId";
    std::memcpy(rec.magic, "RTMIID1", 8);
    std::snprintf(rec.token, sizeof(rec.token), "%s", token.c_str());
    (void)::unlink(path.c_str());  // a leftover of a crashed launch with the same token (pid reuse)
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    (void)::unlink(tmp.c_str());
    const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0600);
    CHECK(fd >= 0) << "cannot create " << tmp;
    CHECK(::write(fd, &rec, sizeof(rec)) == (ssize_t)sizeof(rec)) << "short write to " << tmp;
    (void)::fsync(fd);
    ::close(fd);
    CHECK(std::rename(tmp.c_str(), path.c_str()) == 0) << "cannot publish " << path;
  } else {
    bool got = false;
    const time_t not_before = ::time(nullptr) - 60;  // ranks of one launch start within seconds of each other
    for (int tries = 0; tries < 1200 && !got; tries++) {
      RtIdFile in;
      struct stat sb;
      std::ifstream f(path, std::ios::binary);
      if (f && ::stat(path.c_str(), &sb) == 0 && sb.st_mtime >= not_before &&
          f.read(reinterpret_cast<char *>(&in), sizeof(in)) && std::memcmp(in.magic, "RTMIID1", 8) == 0 &&
          std::strncmp(in.token, token.c_str(), sizeof(in.token)) == 0) {
        rec = in;
        got = true;
      }
      if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    CHECK(got) << "timed out waiting for rank 0's " << path;
  }
  CHECK(ncclCommInitRank(&c->comm, c->world, rec.id, c->rank) == ncclSuccess) << "ncclCommInitRank";
  if (c->rank == 0) std::remove(path.c_str());  // every rank holds the id once the communicator exists
}

// ---------------------------------------------------------------- the two entry points
// spp_split: the reference's decomposition (utils.cu:189,220,238) — every rank renders the WHOLE
// frame with GetWorkload(rank, world, spp) samples and no post-process, frames are summed on rank 0
// and the root applies sqrt(clamp(sum / spp)).  All ranks seed identically, as the reference does
// (quirk g11), so this mode reproduces its multi-rank image rather than a better one.
static void rt_run(curandState **d_states, Camera **d_camera, HitableList **d_world, glm::vec3 **d_image,
                   nvstd::function<void(HitableList *, Camera *)> &init_world, int height, int width, int spp,
                   uint64_t seed, RtComm *comm, bool spp_split = false) {
  const bool overridden = std::getenv("RT_WIDTH") || std::getenv("RT_HEIGHT");
  height = env_int("RT_HEIGHT", height);
  width = env_int("RT_WIDTH", width);
  spp = env_int("RT_SPP", spp);
  const int max_depth = env_int("RT_MAX_DEPTH", 10);  // TRACE_DEPTH_LIMIT, ray_tracing.cu:10
  if (const char *sv = std::getenv("RT_SEED")) seed = std::strtoull(sv, nullptr, 10);
  const size_t n_pixels = (size_t)height * width;

  CHECK(rtmi_device_count() > 0) << "no GPU: this renderer has no CPU path";
  RT_HIP(hipMalloc((void **)d_states, sizeof(curandState) * n_pixels));
  RT_HIP(hipMalloc((void **)d_image, sizeof(glm::vec3) * n_pixels));
  RT_HIP(hipMalloc((void **)d_world, sizeof(HitableList)));
  RT_HIP(hipMalloc((void **)d_camera, sizeof(Camera)));

  const int total_spp = spp;
  rtmi_frame frame{height, width, spp, max_depth, 1, comm->rank, comm->world};
  if (spp_split) {
    frame.spp = GetWorkload(comm->rank, comm->world, total_spp);
    frame.post_process = 0;
    frame.rank = 0, frame.world_size = 1;  // every rank owns the whole frame
    LOG(INFO) << "[" << comm->rank << " / " << comm->world << "] workload: " << frame.spp;
  }
  // the scene may draw from any d_states[i] during init_world, so every pixel's state is made
  // available in the scene's layout (rank-independent: a single-shard frame covers all pixels)
  rtmi_frame whole = frame;
  whole.rank = 0, whole.world_size = 1;
  const int64_t whole_items = rtmi_frame_work_items(&whole);
  uint32_t *d_soa_whole = dev_alloc<uint32_t>((size_t)whole_items * RTMI_STATE_WORDS);
  std::vector<int64_t> pixel_of(whole_items);
  RT_ABI(rtmi_frame_pixel_map(&whole, pixel_of.data()));
  int64_t *d_pixel_of = dev_alloc<int64_t>(whole_items);
  RT_HIP(hipMemcpy(d_pixel_of, pixel_of.data(), whole_items * sizeof(int64_t), hipMemcpyHostToDevice));
  RT_ABI(rtmi_rng_init(seed, &whole, d_soa_whole, nullptr));
  const unsigned blocks = (unsigned)((whole_items + 255) / 256);
  hipLaunchKernelGGL(rt_states_to_aos, dim3(blocks), dim3(256), 0, 0, whole, d_pixel_of, whole_items, d_soa_whole,
                     *d_states);
  LOG(INFO) << "random states initialised (seed " << seed << ")";

  init_world(*d_world, *d_camera);
  RT_HIP(hipDeviceSynchronize());
  RT_HIP(hipGetLastError());
  LOG(INFO) << "world initialised";

  rtmi_scene *scene = rt_build_scene(*d_world, *d_camera, height, width, overridden);

  // this rank's shard: states back into planes (they carry whatever init_world consumed)
  const int64_t items = rtmi_frame_work_items(&frame);
  std::vector<int64_t> my_pixel_of(items);
  RT_ABI(rtmi_frame_pixel_map(&frame, my_pixel_of.data()));
  int64_t *d_my_pixel_of = dev_alloc<int64_t>(items);
  RT_HIP(hipMemcpy(d_my_pixel_of, my_pixel_of.data(), items * sizeof(int64_t), hipMemcpyHostToDevice));
  uint32_t *d_soa = dev_alloc<uint32_t>((size_t)items * RTMI_STATE_WORDS);
  RT_HIP(hipMemset(d_soa, 0, (size_t)items * RTMI_STATE_WORDS * sizeof(uint32_t)));
  hipLaunchKernelGGL(rt_states_to_soa, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, 0, frame, d_my_pixel_of,
                     items, *d_states, d_soa);
  float *d_tiles = dev_alloc<float>((size_t)items * 3);
  float *d_all =
      !spp_split && comm->rank == 0 && comm->world > 1 ? dev_alloc<float>((size_t)items * 3 * comm->world) : d_tiles;
  RT_HIP(hipDeviceSynchronize());

  {
    hipEvent_t start, stop;
    RT_HIP(hipEventCreate(&start));
    RT_HIP(hipEventCreate(&stop));
    RT_HIP(hipEventRecord(start, nullptr));
    RT_ABI(rtmi_render(scene, &frame, d_soa, d_tiles, nullptr, nullptr));
    RT_HIP(hipEventRecord(stop, nullptr));
    RT_HIP(hipEventSynchronize(stop));
    float ms = 0;
    RT_HIP(hipEventElapsedTime(&ms, start, stop));
    uint64_t rays = 0;
    RT_ABI(rtmi_last_ray_total(scene, &rays, nullptr));
    LOG(INFO) << "[" << comm->rank << " / " << comm->world << "] Ray tracing finished in " << ms << "ms. (" << rays
              << " rays, " << (rays / (ms * 1e3)) << " Mrays/s)";
  }

  // the exchange goes through the C ABI as well (rtmi_gather / rtmi_reduce_sum on this program's communicator)
  if (spp_split) {
    // MPI_Reduce(SUM) of utils.cu:124-125 on device buffers; `frame` is the whole frame here (every rank owns it)
    if (comm->world > 1) RT_ABI(rtmi_reduce_sum(comm->comm, &frame, d_tiles, 0, nullptr));
    RT_HIP(hipDeviceSynchronize());
  } else {
    RT_ABI(rtmi_gather(comm->comm, &frame, d_tiles, d_all, 0, nullptr));
    RT_HIP(hipDeviceSynchronize());
  }
  if (comm->rank == 0) {
    RT_ABI(rtmi_untile(&frame, spp_split ? d_tiles : d_all, reinterpret_cast<float *>(*d_image), nullptr));
    if (spp_split)  // GatherImageData's root-side step (utils.cu:126-129)
      RT_ABI(rtmi_post_process(reinterpret_cast<float *>(*d_image), (int64_t)n_pixels, total_spp, nullptr));
    std::vector<glm::vec3> image(n_pixels);
    RT_HIP(hipMemcpy(image.data(), *d_image, sizeof(glm::vec3) * n_pixels, hipMemcpyDeviceToHost));
    if (const char *dump = std::getenv("RT_DUMP")) {
      std::ofstream(dump, std::ios::binary).write(reinterpret_cast<const char *>(image.data()),
                                                  (std::streamsize)(n_pixels * sizeof(glm::vec3)));
      LOG(INFO) << "raw float32 frame written to " << dump;
    }
    const char *outp = std::getenv("RT_OUTPUT");
    WriteImage(image, height, width, outp ? outp : "image.jpeg");
  }
  rtmi_scene_destroy(scene);
  for (void *p : {(void *)d_soa_whole, (void *)d_pixel_of, (void *)d_my_pixel_of, (void *)d_soa}) (void)hipFree(p);
  if (d_all != d_tiles) (void)hipFree(d_all);
  (void)hipFree(d_tiles);
}

// Main: single GPU, seed 1024 (utils.cu:146).
__host__ void Main(curandState **d_states, Camera **d_camera, HitableList **d_world, glm::vec3 **d_image,
                   nvstd::function<void(HitableList *world, Camera *camera)> init_world, int height, int width,
                   int spp) {
  RtComm solo;
  rt_run(d_states, d_camera, d_world, d_image, init_world, height, width, spp, 1024, &solo);
}

// DistributedMain: one process per GPU, seed 10086 (utils.cu:202).  The reference splits the
// samples of every pixel over the ranks and sum-reduces full frames on the host; here the ranks
// split the PIXELS (interleaved 8x8 tiles), each pixel gets all its samples on one GPU, and the
// only exchange is an RCCL gather of the tile buffers to rank 0 — the frame equals the
// single-process frame for any number of ranks.  RT_DIST_MODE=spp selects the reference's own
// decomposition instead (sample split + sum-reduce + root post-process).
__host__ void DistributedMain(curandState **d_states, Camera **d_camera, HitableList **d_world, glm::vec3 **d_image,
                              nvstd::function<void(HitableList *world, Camera *camera)> init_world, int height,
                              int width, int spp) {
  RtComm comm;
  rt_comm_init(&comm);
  const char *mode = std::getenv("RT_DIST_MODE");
  const bool spp_split = mode && std::string(mode) == "spp";
  LOG(INFO) << "[" << comm.rank << " / " << comm.world << "] " << (spp_split ? "sample split (reference mode)" : "pixel-tile shard")
            << ", spp " << spp;
  rt_run(d_states, d_camera, d_world, d_image, init_world, height, width, spp, 10086, &comm, spp_split);
  if (comm.comm) ncclCommDestroy(comm.comm);
}
