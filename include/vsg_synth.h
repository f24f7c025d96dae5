/* Seeded, integer-only synthetic gray frames -- the C mirror of
 * visual_sgraphs_amd/synth.py, byte for byte (tests/test_hostcore.py::test_synth_header_matches_python).
 *
 * A C/C++ host that wants to reproduce the bench/parity inputs without Python includes this header; it has no
 * dependencies beyond <stdint.h>/<stdlib.h>/<string.h> and is NOT part of libvsg_orb.so.
 *
 *   scene    : mid-gray canvas (w+128) x (h+128); K = 400 * canvas_area / 307200 axis-aligned rectangles
 *              (position uniform, sides 4..60, gray 0..255, later ones overwrite earlier) from SplitMix64(seed)
 *   frame t  : the window at offset (64 + 3t mod 64, 64 + 2t mod 64), plus uniform noise +-`noise` from a second
 *              SplitMix64 stream, optional amplitude division around 128 (floor division), clamp to [0,255]
 *   seed     : 0x5EED0000 + sequence number
 */
#ifndef VSG_SYNTH_H
#define VSG_SYNTH_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSG_SYNTH_SEED_BASE 0x5EED0000ull
#define VSG_SYNTH_MARGIN 64

/* output number `index` (1-based, as in the reference SplitMix64 stream) for `seed` */
static inline uint64_t vsg_synth_splitmix64(uint64_t seed, uint64_t index) {
  uint64_t z = seed + index * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static inline int vsg_synth_floordiv(int a, int b) {
  int q = a / b;
  return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

/* Frame `t` of sequence `seq` into out[h][stride].  Returns 0, or -1 on a bad argument / allocation failure. */
static inline int vsg_synth_sequence_frame(int w, int h, uint32_t seq, int t, int amplitude_div, int noise,
                                           uint8_t *out, size_t stride) {
  if (w <= 0 || h <= 0 || !out || stride < (size_t)w || amplitude_div == 0 || noise < 0 || t < 0) return -1;
  const uint64_t seed = VSG_SYNTH_SEED_BASE + seq;
  const int cw = w + 2 * VSG_SYNTH_MARGIN, ch = h + 2 * VSG_SYNTH_MARGIN;
  long long k = (400ll * cw * ch) / 307200;
  if (k < 1) k = 1;
  uint8_t *canvas = (uint8_t *)malloc((size_t)cw * ch);
  if (!canvas) return -1;
  memset(canvas, 128, (size_t)cw * ch);
  for (long long i = 0; i < k; ++i) {
    const int x = (int)(vsg_synth_splitmix64(seed, 5 * i + 1) % (uint64_t)cw);
    const int y = (int)(vsg_synth_splitmix64(seed, 5 * i + 2) % (uint64_t)ch);
    const int rw = (int)(vsg_synth_splitmix64(seed, 5 * i + 3) % 57) + 4;
    const int rh = (int)(vsg_synth_splitmix64(seed, 5 * i + 4) % 57) + 4;
    const uint8_t g = (uint8_t)(vsg_synth_splitmix64(seed, 5 * i + 5) % 256);
    const int x1 = x + rw < cw ? x + rw : cw, y1 = y + rh < ch ? y + rh : ch;
    for (int r = y; r < y1; ++r) memset(canvas + (size_t)r * cw + x, g, (size_t)(x1 - x));
  }
  const int ox = VSG_SYNTH_MARGIN + (3 * t) % VSG_SYNTH_MARGIN, oy = VSG_SYNTH_MARGIN + (2 * t) % VSG_SYNTH_MARGIN;
  const uint64_t nseed = seed ^ (0xA5A5ull << 32) ^ (uint64_t)(t + 1);
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      int v = canvas[(size_t)(oy + r) * cw + ox + c];
      if (noise) v += (int)(vsg_synth_splitmix64(nseed, (uint64_t)r * w + c + 1) % (uint64_t)(2 * noise + 1)) - noise;
      if (amplitude_div != 1) v = 128 + vsg_synth_floordiv(v - 128, amplitude_div);
      out[(size_t)r * stride + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  free(canvas);
  return 0;
}

/* Independent frame number `index` (t = 0 of sequence `index`). */
static inline int vsg_synth_frame(int w, int h, uint32_t index, int amplitude_div, int noise, uint8_t *out,
                                  size_t stride) {
  return vsg_synth_sequence_frame(w, h, index, 0, amplitude_div, noise, out, stride);
}

/* ---- Content classes (the C mirror of synth.content_frame / synth.CONTENT_CLASSES): what real sequences add to the
 * rectangles + noise frames above -- gradients at every scale, defocus, 1-2 px texture, saturation, impulse noise -- and
 * what reaches the paths of the FAST kernel the default class leaves cold (queue overflow, the whole-level minThFAST
 * pass).  Same seeds, same integer arithmetic as the Python side, byte for byte (tests/test_hostcore.py). */
enum vsg_synth_content {
  VSG_CONTENT_RECTANGLES = 0, VSG_CONTENT_VALUE_NOISE, VSG_CONTENT_CHECKER1, VSG_CONTENT_CHECKER2, VSG_CONTENT_GRATING,
  VSG_CONTENT_DEFOCUS, VSG_CONTENT_SATURATED, VSG_CONTENT_RAMP, VSG_CONTENT_SAWTOOTH, VSG_CONTENT_SALT_PEPPER,
  VSG_CONTENT_COUNT
};

/* the rectangle scene of vsg_synth_sequence_frame on its (w + 128) x (h + 128) canvas */
static inline void vsg_synth_scene(int cw, int ch, uint64_t seed, uint8_t *canvas) {
  long long k = (400ll * cw * ch) / 307200;
  if (k < 1) k = 1;
  memset(canvas, 128, (size_t)cw * ch);
  for (long long i = 0; i < k; ++i) {
    const int x = (int)(vsg_synth_splitmix64(seed, 5 * i + 1) % (uint64_t)cw);
    const int y = (int)(vsg_synth_splitmix64(seed, 5 * i + 2) % (uint64_t)ch);
    const int rw = (int)(vsg_synth_splitmix64(seed, 5 * i + 3) % 57) + 4;
    const int rh = (int)(vsg_synth_splitmix64(seed, 5 * i + 4) % 57) + 4;
    const uint8_t g = (uint8_t)(vsg_synth_splitmix64(seed, 5 * i + 5) % 256);
    const int x1 = x + rw < cw ? x + rw : cw, y1 = y + rh < ch ? y + rh : ch;
    for (int r = y; r < y1; ++r) memset(canvas + (size_t)r * cw + x, g, (size_t)(x1 - x));
  }
}

/* Frame `t` of sequence `seq` of content class `kind` into out[h][stride]: the class's canvas translated by (3, 2) px
 * per step, sensor noise +-2 / +-2 / +-1 on the smooth classes (value noise, defocus, ramp).  0, or -1 on a bad argument. */
static inline int vsg_synth_content_frame(int kind, int w, int h, uint32_t seq, int t, uint8_t *out, size_t stride) {
  if (kind == VSG_CONTENT_RECTANGLES) return vsg_synth_sequence_frame(w, h, seq, t, 1, 6, out, stride);
  if (w <= 0 || h <= 0 || !out || stride < (size_t)w || t < 0 || kind < 0 || kind >= VSG_CONTENT_COUNT) return -1;
  const uint64_t seed = VSG_SYNTH_SEED_BASE + seq;
  const int cw = w + 2 * VSG_SYNTH_MARGIN, ch = h + 2 * VSG_SYNTH_MARGIN;
  uint8_t *canvas = (uint8_t *)malloc((size_t)cw * ch);
  if (!canvas) return -1;
  uint64_t r[8];
  for (int i = 0; i < 8; ++i) r[i] = vsg_synth_splitmix64(seed ^ (0xC0DEull << 32), (uint64_t)i + 1);
  int noise = 0;
  if (kind == VSG_CONTENT_VALUE_NOISE) {
    noise = 2;
    long long *acc = (long long *)calloc((size_t)cw * ch, sizeof(long long));
    if (!acc) return free(canvas), -1;
    long long wsum = 0;
    for (int o = 0; o < 6; ++o) {
      const int sp = 64 >> o, gw = cw / sp + 2;
      const uint64_t ls = seed ^ (0x0C7Aull << 32) ^ (uint64_t)(o + 1);
      const long long wgt = 1ll << (5 - o);
      for (int y = 0; y < ch; ++y) {
        const int iy = y / sp, fy = y % sp;
        for (int x = 0; x < cw; ++x) {
          const int ix = x / sp, fx = x % sp;
          const long long a = (long long)(vsg_synth_splitmix64(ls, (uint64_t)iy * gw + ix + 1) % 256);
          const long long b = (long long)(vsg_synth_splitmix64(ls, (uint64_t)iy * gw + ix + 2) % 256);
          const long long c = (long long)(vsg_synth_splitmix64(ls, (uint64_t)(iy + 1) * gw + ix + 1) % 256);
          const long long d = (long long)(vsg_synth_splitmix64(ls, (uint64_t)(iy + 1) * gw + ix + 2) % 256);
          const long long v = ((sp - fy) * ((sp - fx) * a + fx * b) + fy * ((sp - fx) * c + fx * d)) / ((long long)sp * sp);
          acc[(size_t)y * cw + x] += wgt * v;
        }
      }
      wsum += wgt;
    }
    for (size_t i = 0; i < (size_t)cw * ch; ++i) {
      const long long v = acc[i] / wsum;
      canvas[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
    free(acc);
  } else if (kind == VSG_CONTENT_CHECKER1 || kind == VSG_CONTENT_CHECKER2) {
    const int sp = kind == VSG_CONTENT_CHECKER1 ? 1 : 2;
    const int lo = (int)(r[0] % 100), hi = lo + 60 + (int)(r[1] % 96);
    for (int y = 0; y < ch; ++y)
      for (int x = 0; x < cw; ++x) canvas[(size_t)y * cw + x] = (uint8_t)((((x / sp) + (y / sp)) & 1) ? hi : lo);
  } else if (kind == VSG_CONTENT_GRATING) {
    const int period = 2 + (int)(r[0] % 3), lo = (int)(r[1] % 90), hi = lo + 70 + (int)(r[2] % 90);
    for (int y = 0; y < ch; ++y)
      for (int x = 0; x < cw; ++x) canvas[(size_t)y * cw + x] = (uint8_t)((((x + y / 8) % period) * 2 < period) ? hi : lo);
  } else if (kind == VSG_CONTENT_DEFOCUS || kind == VSG_CONTENT_SATURATED) {
    vsg_synth_scene(cw, ch, seed, canvas);
    if (kind == VSG_CONTENT_SATURATED) {
      for (size_t i = 0; i < (size_t)cw * ch; ++i) canvas[i] = canvas[i] >= 128 ? 255 : 0;
    } else {
      noise = 2;  /* 13 x 13 box filter with edge replication, floor division */
      uint8_t *src = (uint8_t *)malloc((size_t)cw * ch);
      if (!src) return free(canvas), -1;
      memcpy(src, canvas, (size_t)cw * ch);
      for (int y = 0; y < ch; ++y)
        for (int x = 0; x < cw; ++x) {
          long long sum = 0;
          for (int dy = -6; dy <= 6; ++dy) {
            const int yy = y + dy < 0 ? 0 : y + dy >= ch ? ch - 1 : y + dy;
            for (int dx = -6; dx <= 6; ++dx) {
              const int xx = x + dx < 0 ? 0 : x + dx >= cw ? cw - 1 : x + dx;
              sum += src[(size_t)yy * cw + xx];
            }
          }
          canvas[(size_t)y * cw + x] = (uint8_t)(sum / 169);
        }
      free(src);
    }
  } else if (kind == VSG_CONTENT_RAMP) {
    noise = 1;
    const long long gx = 1 + (long long)(r[0] % 3), gy = 1 + (long long)(r[1] % 3);
    for (int y = 0; y < ch; ++y)
      for (int x = 0; x < cw; ++x)
        canvas[(size_t)y * cw + x] = (uint8_t)((x * gx * 255 / (3 * cw) + y * gy * 255 / (3 * ch)) % 256);
  } else if (kind == VSG_CONTENT_SAWTOOTH) {
    const int g = 7 + (int)(r[0] % 4);
    for (int y = 0; y < ch; ++y)
      for (int x = 0; x < cw; ++x) canvas[(size_t)y * cw + x] = (uint8_t)((x * g + y * g) % 256);
  } else {  /* salt and pepper */
    const uint64_t us = seed ^ (0x5A17ull << 32);
    for (size_t i = 0; i < (size_t)cw * ch; ++i) {
      const uint64_t u = vsg_synth_splitmix64(us, (uint64_t)i + 1) % 100;
      canvas[i] = u == 0 ? 0 : u == 1 ? 255 : 128;
    }
  }
  const int ox = VSG_SYNTH_MARGIN + (3 * t) % VSG_SYNTH_MARGIN, oy = VSG_SYNTH_MARGIN + (2 * t) % VSG_SYNTH_MARGIN;
  const uint64_t nseed = seed ^ (0xA5A5ull << 32) ^ (uint64_t)(t + 1);
  for (int rr = 0; rr < h; ++rr)
    for (int c = 0; c < w; ++c) {
      int v = canvas[(size_t)(oy + rr) * cw + ox + c];
      if (noise) v += (int)(vsg_synth_splitmix64(nseed, (uint64_t)rr * w + c + 1) % (uint64_t)(2 * noise + 1)) - noise;
      out[(size_t)rr * stride + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  free(canvas);
  return 0;
}

/* A DBoW2 ORB vocabulary image in the reference's binary layout (TemplatedVocabulary.h:1495-1547) -- the C mirror of
 * synth.synthetic_vocabulary: int k, L, scoring, weighting; then per node (BFS order, node i has parent (i - 1) / k)
 * int parent, uint8 isLeaf, uint8[32] descriptor, double weight = 45 bytes.  Full k-ary tree; node descriptors and
 * leaf weights from SplitMix64(0x5EED0000 ^ 0xB0C ^ seed << 12), five outputs per node; leaves whose draw falls below
 * stop_fraction get weight 0 ("stopped" words).  k = 10, L = 6 (the shape of ORBvoc.txt.bin) is 49 999 966 bytes.
 * Returns the byte count; with out == NULL (or cap too small) nothing is written. */
static inline size_t vsg_synth_vocabulary(int k, int L, uint32_t seed, int scoring, int weighting,
                                          double stop_fraction, uint8_t *out, size_t cap) {
  if (k < 2 || L < 1) return 0;
  uint64_t n_nodes = 0, p = 1, first_leaf = 0;
  for (int l = 0; l <= L; ++l) {
    if (l == L) first_leaf = n_nodes;
    n_nodes += p;
    p *= (uint64_t)k;
  }
  const size_t bytes = 16 + (size_t)(n_nodes - 1) * 45;
  if (!out || cap < bytes) return bytes;
  const int32_t hdr[4] = {k, L, scoring, weighting};
  memcpy(out, hdr, 16);
  const uint64_t s = (VSG_SYNTH_SEED_BASE ^ 0xB0Cull) ^ ((uint64_t)seed << 12);
  uint8_t *q = out + 16;
  for (uint64_t i = 1; i < n_nodes; ++i, q += 45) {
    const int32_t parent = (int32_t)((i - 1) / (uint64_t)k);
    const int leaf = i >= first_leaf;
    memcpy(q, &parent, 4);
    q[4] = (uint8_t)leaf;
    for (int j = 0; j < 4; ++j) {
      const uint64_t r = vsg_synth_splitmix64(s, 5 * (i - 1) + 1 + (uint64_t)j);
      memcpy(q + 5 + 8 * j, &r, 8);
    }
    double w = 0.0;
    if (leaf) {
      const uint64_t u = vsg_synth_splitmix64(s, 5 * (i - 1) + 5) % 10000;
      w = (double)u < stop_fraction * 10000 ? 0.0 : 0.5 + (double)u / 1000.0;
    }
    memcpy(q + 37, &w, 8);
  }
  return bytes;
}

#ifdef __cplusplus
}
#endif
#endif /* VSG_SYNTH_H */
