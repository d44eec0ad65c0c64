// Baseline JPEG writer behind WriteImage (the reference calls stbi_write_jpg(..., quality 100),
// utils.cu:95; stb is an un-fetched submodule).  Matches what stb does at quality 100 in the
// ways that matter for the picture: JFIF, 8-bit, YCbCr 4:4:4, both quantisation tables all
// ones.  The entropy coder uses its own (valid, fixed-length) Huffman tables, declared in the
// file's DHT segments, so any baseline decoder reads the result.
#pragma once
#include <stdint.h>

#include <cmath>
#include <cstdio>
#include <vector>

namespace rt_jpeg {

static const uint8_t kZig[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
                                 62, 63};

struct Writer {
  std::vector<uint8_t> out;
  uint32_t acc = 0;
  int nbits = 0;
  void byte(int b) { out.push_back((uint8_t)b); }
  void word(int w) { byte(w >> 8), byte(w & 255); }
  void bits(uint32_t code, int len) {
    for (int i = len - 1; i >= 0; i--) {
      acc = (acc << 1) | ((code >> i) & 1u);
      if (++nbits == 8) {
        byte(acc & 255);
        if ((acc & 255) == 0xff) byte(0);  // byte stuffing
        acc = 0, nbits = 0;
      }
    }
  }
  void flush() {
    while (nbits) bits(1, 1);
  }
};

// AC symbol list: EOB, ZRL, then (run << 4 | size) for run 0..15, size 1..10 — 162 symbols,
// symbol k gets the 8-bit code k.  DC: categories 0..11, category k gets the 4-bit code k.
inline int ac_code(int sym) {
  if (sym == 0x00) return 0;
  if (sym == 0xf0) return 1;
  return 2 + (sym >> 4) * 10 + ((sym & 15) - 1);
}

inline int category(int v) {
  int a = v < 0 ? -v : v, s = 0;
  while (a) s++, a >>= 1;
  return s;
}

inline void fdct(const float *in, int *out) {
  static float c[8][8];
  static bool init = false;
  if (!init) {
    for (int u = 0; u < 8; u++)
      for (int x = 0; x < 8; x++)
        c[u][x] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16);
    init = true;
  }
  float tmp[64];
  for (int y = 0; y < 8; y++)
    for (int u = 0; u < 8; u++) {
      float s = 0;
      for (int x = 0; x < 8; x++) s += c[u][x] * in[y * 8 + x];
      tmp[y * 8 + u] = s;
    }
  for (int u = 0; u < 8; u++)
    for (int v = 0; v < 8; v++) {
      float s = 0;
      for (int y = 0; y < 8; y++) s += c[v][y] * tmp[y * 8 + u];
      int q = (int)std::lround(s);  // quantiser step 1
      if (u | v) q = q > 1023 ? 1023 : (q < -1023 ? -1023 : q);  // baseline AC range
      out[v * 8 + u] = q;
    }
}

inline void encode_block(Writer &w, const int *coef, int *pred) {
  int diff = coef[0] - *pred;
  *pred = coef[0];
  int s = category(diff);
  w.bits((uint32_t)s, 4);
  if (s) w.bits((uint32_t)(diff >= 0 ? diff : diff - 1) & ((1u << s) - 1), s);
  int run = 0;
  for (int k = 1; k < 64; k++) {
    int v = coef[kZig[k]];
    if (v == 0) {
      run++;
      continue;
    }
    while (run > 15) {
      w.bits((uint32_t)ac_code(0xf0), 8);
      run -= 16;
    }
    int sz = category(v);
    w.bits((uint32_t)ac_code((run << 4) | sz), 8);
    w.bits((uint32_t)(v >= 0 ? v : v - 1) & ((1u << sz) - 1), sz);
    run = 0;
  }
  if (run) w.bits((uint32_t)ac_code(0x00), 8);
}

}  // namespace rt_jpeg

// rgb: width*height*3 bytes, row-major, top row first.  Returns false if the file cannot be written.
inline bool rt_write_jpeg(const char *path, int width, int height, const uint8_t *rgb) {
  using namespace rt_jpeg;
  Writer w;
  w.word(0xffd8);
  w.word(0xffe0), w.word(16);
  for (char ch : {'J', 'F', 'I', 'F', '\0'}) w.byte(ch);
  w.byte(1), w.byte(1), w.byte(0), w.word(1), w.word(1), w.byte(0), w.byte(0);
  for (int t = 0; t < 2; t++) {  // DQT: all ones (quality 100)
    w.word(0xffdb), w.word(67), w.byte(t);
    for (int i = 0; i < 64; i++) w.byte(1);
  }
  w.word(0xffc0), w.word(17), w.byte(8), w.word(height), w.word(width), w.byte(3);
  w.byte(1), w.byte(0x11), w.byte(0);
  w.byte(2), w.byte(0x11), w.byte(1);
  w.byte(3), w.byte(0x11), w.byte(1);
  for (int t = 0; t < 2; t++) {
    // DC table t: twelve 4-bit codes
    w.word(0xffc4), w.word(2 + 1 + 16 + 12), w.byte(0x00 | t);
    for (int l = 1; l <= 16; l++) w.byte(l == 4 ? 12 : 0);
    for (int k = 0; k < 12; k++) w.byte(k);
    // AC table t: 162 8-bit codes
    w.word(0xffc4), w.word(2 + 1 + 16 + 162), w.byte(0x10 | t);
    for (int l = 1; l <= 16; l++) w.byte(l == 8 ? 162 : 0);
    w.byte(0x00), w.byte(0xf0);
    for (int run = 0; run < 16; run++)
      for (int sz = 1; sz <= 10; sz++) w.byte((run << 4) | sz);
  }
  w.word(0xffda), w.word(12), w.byte(3);
  w.byte(1), w.byte(0x00), w.byte(2), w.byte(0x11), w.byte(3), w.byte(0x11);
  w.byte(0), w.byte(63), w.byte(0);

  int pred[3] = {0, 0, 0};
  for (int by = 0; by < height; by += 8)
    for (int bx = 0; bx < width; bx += 8) {
      float comp[3][64];
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
          int yy = by + y < height ? by + y : height - 1, xx = bx + x < width ? bx + x : width - 1;
          const uint8_t *p = rgb + ((size_t)yy * width + xx) * 3;
          float r = p[0], g = p[1], b = p[2];
          comp[0][y * 8 + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.f;
          comp[1][y * 8 + x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
          comp[2][y * 8 + x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
        }
      for (int c = 0; c < 3; c++) {
        int coef[64];
        fdct(comp[c], coef);
        encode_block(w, coef, &pred[c]);
      }
    }
  w.flush();
  w.word(0xffd9);
  FILE *f = std::fopen(path, "wb");
  if (!f) return false;
  bool ok = std::fwrite(w.out.data(), 1, w.out.size(), f) == w.out.size();
  std::fclose(f);
  return ok;
}
