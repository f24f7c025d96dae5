// Host build of the PRODUCT's device-agnostic core (visual_sgraphs_amd/csrc/*.h) for CPU unit tests:
// the octree algorithm runs here as a 1-thread group; geometry/tables are the same code the runtime
// uses.  This library is test-only -- the shipped path is the HIP library and has no CPU fallback.
#include <algorithm>
#include <cstring>
#include <vector>

#include "vsg_geometry.h"
#include "vsg_introsort.h"
#include "vsg_math.h"
#include "vsg_octree_core.h"
#include "../../include/vsg_synth.h"

using namespace vsg;

static Geometry g_geom;
static ExtractorTables g_tab;

extern "C" {

int hc_build(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh, int rows, int cols) {
  if (!build_tables(g_tab, nfeatures, scaleFactor, nlevels, iniTh, minTh)) return -1;
  const uint16_t taps[7] = {18, 34, 49, 55, 49, 34, 18};
  return build_geometry(g_geom, g_tab, rows, cols, 0, 0, taps);
}

// out: w,h,pitch,nCols,nRows,wCell,hCell,quota,cand_cap,sel_cap,nIni,oct_width,oct_height,cell_base,kp_size(int)
int hc_level(int l, int *out) {
  const LevelGeom &L = g_geom.fg.lv[l];
  int v[] = {L.w, L.h, L.pitch, L.nCols, L.nRows, L.wCell, L.hCell, L.quota, L.cand_cap, L.sel_cap,
             L.nIni, L.oct_width, L.oct_height, L.cell_base, (int)L.kp_size};
  memcpy(out, v, sizeof(v));
  return (int)(sizeof(v) / sizeof(int));
}
void hc_tables(float *scale, float *inv, int *quota, int *umax) {
  for (int i = 0; i < g_tab.nlevels; i++) {
    scale[i] = g_tab.scale[i];
    inv[i] = g_tab.invScale[i];
    quota[i] = g_tab.quota[i];
  }
  for (int i = 0; i < 16; i++) umax[i] = g_tab.umax[i];
}
// fused-pyramid tiling i of the geometry built by hc_build: {ntiles, lds bytes, ok, tabMax, ldsA, ldsB}
void hc_pyr_tiling(int i, int *out) {
  const PyrTiling &P = g_geom.pyr[i];
  out[0] = (int)P.tiles.size(), out[1] = P.lds_bytes(), out[2] = P.ok, out[3] = P.tabMax, out[4] = P.ldsA, out[5] = P.ldsB;
}
int hc_total_cells() { return g_geom.fg.total_cells; }
int hc_out_cap() { return g_geom.fg.out_cap; }
void hc_cell(int i, int *out) {
  const CellDesc &c = g_geom.cells[i];
  out[0] = c.level, out[1] = c.x0, out[2] = c.y0, out[3] = c.x1, out[4] = c.y1;
}
// the FAST record of cell i (FastCellRec, vsg_common.h: 8 words) plus {img_off, cand_off of its level}
void hc_fast_rec(int i, unsigned *out) {
  memcpy(out, g_geom.fastRecs[i].w, 32);
  const LevelGeom &L = g_geom.fg.lv[g_geom.cells[i].level];
  out[8] = (unsigned)L.img_off, out[9] = (unsigned)L.cand_off, out[10] = (unsigned)g_geom.cells[i].cand_off;
}
// resize tables of level l (from l-1): xs[4*w], ys[4*h]
void hc_resize_tables(int l, short *xs, short *ys) {
  const LevelGeom &L = g_geom.fg.lv[l];
  memcpy(xs, &g_geom.resizeTab[L.tab_x_off], sizeof(Short4) * L.w);
  memcpy(ys, &g_geom.resizeTab[L.tab_y_off], sizeof(Short4) * L.h);
}

// octree of level l of the geometry built by hc_build; candidates given in ANY order
int hc_octree(int l, const int *x, const int *y, const int *resp, int n, int N_override, int *outPacked) {
  const LevelGeom &L = g_geom.fg.lv[l];
  octree::Params P;
  P.N = N_override >= 0 ? N_override : L.quota;
  P.height = L.oct_height;
  P.nIni = L.nIni;
  P.iniUL = L.iniUL;
  P.iniThresh = L.iniThresh;
  P.nCols = L.nCols;
  P.wCell = L.wCell;
  P.hCell = L.hCell;
  std::vector<uint32_t> cand(n > 0 ? n : 1);
  for (int i = 0; i < n; i++) cand[i] = pack_cand(x[i], y[i], resp[i]);
  std::vector<uint16_t> node_of(n > 0 ? n : 1);
  const int cap = octree::node_capacity(P.N);
  std::vector<uint64_t> buf(octree::work_bytes(cap) / 8 + 2);
  octree::Work W;
  octree::carve(W, buf.data(), cap);
  std::vector<uint32_t> sel(cap);
  octree::SerialGroup g;
  int m = octree::distribute(g, P, cand.data(), n, node_of.data(), W, sel.data());
  for (int i = 0; i < m; i++) outPacked[i] = (int)sel[i];
  // the register-resident points policy (the device's common path) must give the same list
  if (n <= 8192) {
    std::vector<uint64_t> buf2(buf.size());
    octree::Work W2;
    octree::carve(W2, buf2.data(), cap);
    std::vector<uint32_t> sel2(cap);
    const int m2 = octree::distribute_reg<8192>(g, P, cand.data(), n, W2, sel2.data());
    if (m2 != m) return -1000;
    for (int i = 0; i < m; i++)
      if (sel2[i] != sel[i]) return -1001;
  }
  {  // the memory form with its sweeps taken 8 points at a time (what the stand-alone k_octree launches)
    std::vector<uint64_t> buf3(buf.size());
    octree::Work W3;
    octree::carve(W3, buf3.data(), cap);
    std::vector<uint32_t> sel3(cap);
    std::vector<uint16_t> node3(n > 0 ? n : 1);
    const int m3 = octree::distribute<8>(g, P, cand.data(), n, node3.data(), W3, sel3.data());
    if (m3 != m) return -1002;
    for (int i = 0; i < m; i++)
      if (sel3[i] != sel[i]) return -1003;
  }
  return m;
}

void hc_sort(uint64_t *items, int n) { introsort::sort(items, n); }
// the real thing, for comparison (same libstdc++ the oracle is built with)
void hc_std_sort(uint64_t *items, int n) {
  std::sort(items, items + n, [](uint64_t a, uint64_t b) { return (uint32_t)(a >> 32) < (uint32_t)(b >> 32); });
}
// the split the device uses: serial partition phase, then a stable rank of what it left
void hc_sort_depth(uint64_t *items, int n, int depth) { introsort::sort(items, n, depth); }
void hc_sort_split_depth(uint64_t *items, int n, int depth);
void hc_sort_split(uint64_t *items, int n) { hc_sort_split_depth(items, n, -1); }
void hc_sort_split_depth(uint64_t *items, int n, int depth) {
  if (n <= 0) return;
  introsort::partition_phase(items, n, depth);
  std::vector<uint64_t> tmp(items, items + n);
  for (int t = 0; t < n; t++) {
    // windowed stable rank, exactly as vsg_octree_core.h computes it
    const int lo = t > 15 ? t - 15 : 0, hi = t + 15 < n - 1 ? t + 15 : n - 1;
    int rank = lo;
    for (int j = lo; j <= hi; j++) {
      const uint32_t kj = (uint32_t)(tmp[j] >> 32), kt = (uint32_t)(tmp[t] >> 32);
      rank += (kj < kt) | ((kj == kt) & (j < t));
    }
    items[rank] = tmp[t];
  }
}
float hc_fast_atan2(float y, float x) { return fast_atan2_deg(y, x); }
int hc_synth_frame(int w, int h, unsigned seq, int t, int div, int noise, uint8_t *out, size_t stride) {
  return vsg_synth_sequence_frame(w, h, seq, t, div, noise, out, stride);
}
int hc_synth_content_frame(int kind, int w, int h, unsigned seq, int t, uint8_t *out, size_t stride) {
  return vsg_synth_content_frame(kind, w, h, seq, t, out, stride);
}
size_t hc_synth_vocabulary(int k, int L, unsigned seed, int scoring, int weighting, double stop_fraction, uint8_t *out,
                           size_t cap) {
  return vsg_synth_vocabulary(k, L, seed, scoring, weighting, stop_fraction, out, cap);
}
void hc_brief_rotation(float angle, float *a, float *b) { brief_rotation(angle, a, b); }
void hc_brief_offset(int px, int py, float a, float b, int *dx, int *dy) { brief_offset(px, py, a, b, dx, dy); }
float hc_sinf(float x, int fma) { return fma ? SinCosF<true>::eval(x, false) : SinCosF<false>::eval(x, false); }
void hc_sincos_pair(float x, int fma, float *c, float *s) {
  if (fma) sincos_pair<true>(x, c, s); else sincos_pair<false>(x, c, s);
}
// every float in [lo, hi] (bit patterns lo_bits .. hi_bits): the pair form against the two eval() calls; returns mismatches
long hc_sincos_pair_sweep(uint32_t lo_bits, uint32_t hi_bits, int fma) {
  long bad = 0;
  for (uint32_t u = lo_bits; u <= hi_bits; u++) {
    union { uint32_t u; float f; } v;
    v.u = u;
    float c, s, c0, s0;
    if (fma) {
      sincos_pair<true>(v.f, &c, &s);
      c0 = SinCosF<true>::eval(v.f, true), s0 = SinCosF<true>::eval(v.f, false);
    } else {
      sincos_pair<false>(v.f, &c, &s);
      c0 = SinCosF<false>::eval(v.f, true), s0 = SinCosF<false>::eval(v.f, false);
    }
    bad += (f2u(c) != f2u(c0)) + (f2u(s) != f2u(s0));
    if (u == 0xFFFFFFFFu) break;
  }
  return bad;
}
float hc_fast_atan2_sel(float y, float x) { return fast_atan2_deg_sel(y, x); }
void hc_brief_rotation_of_moments(float m01, float m10, float *angle, float *a, float *b) {
  brief_rotation_of_moments(m01, m10, angle, a, b);
}
float hc_cosf(float x, int fma) { return fma ? SinCosF<true>::eval(x, true) : SinCosF<false>::eval(x, true); }
}
