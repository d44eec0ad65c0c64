// stbi_load / stbi_image_free for the scene sources (scenes/birthday.cu:79-80, model.h).
// stb is an un-fetched submodule of the reference.  Supported inputs: binary PPM (P6) and
// baseline (non-progressive, 8-bit, Huffman) JPEG; always expanded to `req_comp` = 4 (RGBA)
// or 3 (RGB) channels.  Define STB_IMAGE_IMPLEMENTATION in one translation unit, like stb.
#pragma once
#include <stdint.h>

unsigned char *stbi_load(const char *filename, int *x, int *y, int *channels_in_file, int desired_channels);
void stbi_image_free(void *p);

#ifdef STB_IMAGE_IMPLEMENTATION
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace rt_stbi {

struct Huff {
  uint8_t bits[17];
  uint8_t vals[256];
  int mincode[17], maxcode[18], valptr[17];
  void build() {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
      valptr[l] = k;
      mincode[l] = code;
      code += bits[l];
      k += bits[l];
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
  }
};

struct Jpeg {
  const uint8_t *d;
  size_t n, pos = 0;
  int w = 0, h = 0, ncomp = 0;
  int hs[4], vs[4], tq[4], td[4], ta[4], cid[4];
  uint16_t qt[4][64];
  Huff dc[4], ac[4];
  int restart = 0;
  uint32_t bitbuf = 0;
  int bitcnt = 0;
  bool ok = true;
  int u8() { return pos < n ? d[pos++] : (ok = false, 0); }
  int u16() { int a = u8(); return (a << 8) | u8(); }
  int bit() {
    if (!bitcnt) {
      int b = u8();
      if (b == 0xff) {
        int m = u8();
        if (m != 0) { pos -= 2; b = 0; }  // marker: feed zeros
      }
      bitbuf = b;
      bitcnt = 8;
    }
    bitcnt--;
    return (bitbuf >> bitcnt) & 1;
  }
  int bits(int k) { int v = 0; while (k--) v = (v << 1) | bit(); return v; }
  int decode(const Huff &t) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
      code = (code << 1) | bit();
      if (t.maxcode[l] >= 0 && code <= t.maxcode[l] && code >= t.mincode[l]) return t.vals[t.valptr[l] + code - t.mincode[l]];
    }
    ok = false;
    return 0;
  }
  static int extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
};

static const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                    15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
                                    62, 63};

inline void idct8x8(const int *in, uint8_t *out, int stride) {
  static float c[8][8];
  static bool init = false;
  if (!init) {
    for (int x = 0; x < 8; x++)
      for (int u = 0; u < 8; u++) c[x][u] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16);
    init = true;
  }
  float tmp[64];
  for (int y = 0; y < 8; y++)
    for (int x = 0; x < 8; x++) {
      float s = 0;
      for (int u = 0; u < 8; u++) s += c[x][u] * in[y * 8 + u];
      tmp[y * 8 + x] = s;
    }
  for (int x = 0; x < 8; x++)
    for (int y = 0; y < 8; y++) {
      float s = 0;
      for (int v = 0; v < 8; v++) s += c[y][v] * tmp[v * 8 + x];
      int p = (int)std::lround(s + 128.f);
      out[y * stride + x] = (uint8_t)(p < 0 ? 0 : (p > 255 ? 255 : p));
    }
}

inline unsigned char *load_jpeg(const uint8_t *data, size_t n, int *x, int *y, int *comp, int req) {
  Jpeg j;
  j.d = data, j.n = n;
  if (j.u16() != 0xffd8) return nullptr;
  std::vector<uint8_t> planes[4];
  int pw[4], ph[4], hmax = 1, vmax = 1;
  for (;;) {
    int m = j.u16();
    if (!j.ok) return nullptr;
    if (m == 0xffd9) break;
    if ((m & 0xff00) != 0xff00) return nullptr;
    int len = j.u16();
    size_t end = j.pos + len - 2;
    if (m == 0xffdb) {
      while (j.pos < end) {
        int pq = j.u8();
        int id = pq & 15;
        for (int i = 0; i < 64; i++) j.qt[id & 3][kZigzag[i]] = (pq >> 4) ? j.u16() : j.u8();
      }
    } else if (m == 0xffc4) {
      while (j.pos < end) {
        int tc = j.u8();
        Huff &t = (tc >> 4) ? j.ac[tc & 3] : j.dc[tc & 3];
        int total = 0;
        t.bits[0] = 0;
        for (int i = 1; i <= 16; i++) total += (t.bits[i] = j.u8());
        if (total > 256) return nullptr;
        for (int i = 0; i < total; i++) t.vals[i] = j.u8();
        t.build();
      }
    } else if (m == 0xffc0 || m == 0xffc1) {
      if (j.u8() != 8) return nullptr;
      j.h = j.u16(), j.w = j.u16(), j.ncomp = j.u8();
      if (j.ncomp != 1 && j.ncomp != 3) return nullptr;
      for (int i = 0; i < j.ncomp; i++) {
        j.cid[i] = j.u8();
        int s = j.u8();
        j.hs[i] = s >> 4, j.vs[i] = s & 15, j.tq[i] = j.u8() & 3;
        if (j.hs[i] > hmax) hmax = j.hs[i];
        if (j.vs[i] > vmax) vmax = j.vs[i];
      }
    } else if (m == 0xffc2) {
      return nullptr;  // progressive JPEG is not supported
    } else if (m == 0xffdd) {
      j.restart = j.u16();
    } else if (m == 0xffda) {
      int ns = j.u8();
      for (int i = 0; i < ns; i++) {
        int id = j.u8(), t = j.u8();
        for (int c = 0; c < j.ncomp; c++)
          if (j.cid[c] == id) j.td[c] = t >> 4, j.ta[c] = t & 15;
      }
      j.pos += 3;
      int mcuw = 8 * hmax, mcuh = 8 * vmax, mx = (j.w + mcuw - 1) / mcuw, my = (j.h + mcuh - 1) / mcuh;
      for (int c = 0; c < j.ncomp; c++) {
        pw[c] = mx * 8 * j.hs[c], ph[c] = my * 8 * j.vs[c];
        planes[c].assign((size_t)pw[c] * ph[c], 0);
      }
      int pred[4] = {0, 0, 0, 0}, count = 0;
      j.bitcnt = 0;
      for (int my_ = 0; my_ < my && j.ok; my_++)
        for (int mx_ = 0; mx_ < mx && j.ok; mx_++) {
          if (j.restart && count && count % j.restart == 0) {
            j.bitcnt = 0;
            if (j.pos + 1 < j.n && j.d[j.pos] == 0xff && (j.d[j.pos + 1] & 0xf8) == 0xd0) j.pos += 2;
            pred[0] = pred[1] = pred[2] = pred[3] = 0;
          }
          count++;
          for (int c = 0; c < j.ncomp; c++)
            for (int by = 0; by < j.vs[c]; by++)
              for (int bx = 0; bx < j.hs[c]; bx++) {
                int blk[64] = {0};
                int s = j.decode(j.dc[j.td[c] & 3]);
                pred[c] += Jpeg::extend(j.bits(s), s);
                blk[0] = pred[c] * j.qt[j.tq[c]][0];
                for (int k = 1; k < 64;) {
                  int rs = j.decode(j.ac[j.ta[c] & 3]);
                  int r = rs >> 4, sz = rs & 15;
                  if (!sz) {
                    if (r == 15) { k += 16; continue; }
                    break;
                  }
                  k += r;
                  if (k > 63) break;
                  blk[kZigzag[k]] = Jpeg::extend(j.bits(sz), sz) * j.qt[j.tq[c]][kZigzag[k]];
                  k++;
                }
                int ox = (mx_ * j.hs[c] + bx) * 8, oy = (my_ * j.vs[c] + by) * 8;
                idct8x8(blk, &planes[c][(size_t)oy * pw[c] + ox], pw[c]);
              }
        }
      // the scan is followed by EOI (possibly after padding)
      while (j.pos + 1 < j.n && !(j.d[j.pos] == 0xff && j.d[j.pos + 1] == 0xd9)) j.pos++;
      continue;
    }
    j.pos = end;
  }
  if (!j.w || !j.h || planes[0].empty()) return nullptr;
  int oc = req ? req : (j.ncomp == 1 ? 1 : 3);
  unsigned char *out = (unsigned char *)std::malloc((size_t)j.w * j.h * oc);
  for (int yy = 0; yy < j.h; yy++)
    for (int xx = 0; xx < j.w; xx++) {
      float Y = planes[0][(size_t)(yy * j.vs[0] / vmax) * pw[0] + (xx * j.hs[0] / hmax)];
      float r = Y, g = Y, b = Y;
      if (j.ncomp == 3) {
        float cb = planes[1][(size_t)(yy * j.vs[1] / vmax) * pw[1] + (xx * j.hs[1] / hmax)] - 128.f;
        float cr = planes[2][(size_t)(yy * j.vs[2] / vmax) * pw[2] + (xx * j.hs[2] / hmax)] - 128.f;
        r = Y + 1.402f * cr, g = Y - 0.344136f * cb - 0.714136f * cr, b = Y + 1.772f * cb;
      }
      auto cl = [](float v) { int i = (int)std::lround(v); return (unsigned char)(i < 0 ? 0 : (i > 255 ? 255 : i)); };
      unsigned char px[4] = {cl(r), cl(g), cl(b), 255};
      for (int c = 0; c < oc; c++) out[((size_t)yy * j.w + xx) * oc + c] = oc == 1 ? px[0] : px[c];
    }
  *x = j.w, *y = j.h;
  if (comp) *comp = j.ncomp;
  return out;
}

inline unsigned char *load_ppm(const uint8_t *d, size_t n, int *x, int *y, int *comp, int req) {
  size_t p = 2;
  int vals[3], got = 0;
  while (got < 3 && p < n) {
    while (p < n && (d[p] == ' ' || d[p] == '\n' || d[p] == '\r' || d[p] == '\t')) p++;
    if (p < n && d[p] == '#') { while (p < n && d[p] != '\n') p++; continue; }
    int v = 0;
    while (p < n && d[p] >= '0' && d[p] <= '9') v = v * 10 + (d[p++] - '0');
    vals[got++] = v;
  }
  p++;
  int w = vals[0], h = vals[1];
  if (got < 3 || vals[2] != 255 || p + (size_t)w * h * 3 > n) return nullptr;
  int oc = req ? req : 3;
  unsigned char *out = (unsigned char *)std::malloc((size_t)w * h * oc);
  for (size_t i = 0; i < (size_t)w * h; i++)
    for (int c = 0; c < oc; c++) out[i * oc + c] = c < 3 ? d[p + i * 3 + c] : 255;
  *x = w, *y = h;
  if (comp) *comp = 3;
  return out;
}
}  // namespace rt_stbi

unsigned char *stbi_load(const char *filename, int *x, int *y, int *comp, int req) {
  FILE *f = std::fopen(filename, "rb");
  if (!f) return nullptr;
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  size_t k;
  while ((k = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + k);
  std::fclose(f);
  if (buf.size() > 2 && buf[0] == 'P' && buf[1] == '6') return rt_stbi::load_ppm(buf.data(), buf.size(), x, y, comp, req);
  return rt_stbi::load_jpeg(buf.data(), buf.size(), x, y, comp, req);
}
void stbi_image_free(void *p) { std::free(p); }
#endif  // STB_IMAGE_IMPLEMENTATION
