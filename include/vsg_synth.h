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
