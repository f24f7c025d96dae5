// vsg_geometry.h -- host-side tables of the extractor (no GPU code): constructor tables of
// ORBextractor (ORBextractor.cc:411-470) and the per-image-size geometry the kernels consume.
// Float arithmetic follows the reference's types step by step (float vs double), because level
// sizes, quotas and cell grids are all derived through float rounding.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "vsg_common.h"

namespace vsg {

struct ExtractorTables {
  int nfeatures = 0, nlevels = 0, iniTh = 0, minTh = 0;
  double scaleFactor = 0;  // ORBextractor.h:105: a double member holding the float argument
  std::vector<float> scale, invScale, sigma2, invSigma2;
  std::vector<int> quota;
  int umax[16];
};

inline int cv_round_f(float v) { return (int)lrintf(v); }  // SSE cvtss2si, round-half-even
inline int cv_round_d(double v) { return (int)lrint(v); }

inline bool build_tables(ExtractorTables &T, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
  if (nlevels < 1 || nlevels > kMaxLevels || nfeatures < 0 || !(scaleFactor > 1.0f)) return false;
  T.nfeatures = nfeatures;
  T.nlevels = nlevels;
  T.iniTh = iniTh;
  T.minTh = minTh;
  T.scaleFactor = scaleFactor;
  T.scale.assign(nlevels, 1.0f);
  T.sigma2.assign(nlevels, 1.0f);
  for (int i = 1; i < nlevels; i++) {  // :419-423  float = float * double
    T.scale[i] = (float)((double)T.scale[i - 1] * T.scaleFactor);
    T.sigma2[i] = T.scale[i] * T.scale[i];
  }
  T.invScale.resize(nlevels);
  T.invSigma2.resize(nlevels);
  for (int i = 0; i < nlevels; i++) {  // :427-431
    T.invScale[i] = 1.0f / T.scale[i];
    T.invSigma2[i] = 1.0f / T.sigma2[i];
  }
  // :436-446 quotas
  const float factor = (float)(1.0 / T.scaleFactor);
  const float denom = 1.0f - (float)std::pow((double)factor, (double)nlevels);
  float desired = (float)nfeatures * (1.0f - factor) / denom;
  T.quota.assign(nlevels, 0);
  int sum = 0;
  for (int l = 0; l + 1 < nlevels; l++) {
    T.quota[l] = cv_round_f(desired);
    sum += T.quota[l];
    desired *= factor;
  }
  T.quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
  // :454-469 umax
  const int hp = kHalfPatch;
  const float half_diag = (float)hp * std::sqrt(2.f) / 2;
  const int vmax = (int)std::floor(half_diag + 1), vmin = (int)std::ceil(half_diag);
  for (int v = 0; v <= vmax; ++v) T.umax[v] = cv_round_d(std::sqrt((double)hp * hp - (double)v * v));
  for (int v = hp, v0 = 0; v >= vmin; --v) {
    while (T.umax[v0] == T.umax[v0 + 1]) ++v0;
    T.umax[v] = v0;
    ++v0;
  }
  return true;
}

// resize table entries: dst column -> {sx, a0, a1, 0}; dst row -> {sy0, sy1, b0, b1}  ([OCV] resize.cpp)
struct Short4 {
  int16_t a, b, c, d;
};

struct CellDesc {  // one FAST cell: valid region [x0,x1) x [y0,y1) in level coordinates
  int16_t level, x0, y0, x1, y1, pad;
  int32_t cand_off;  // the cell's own segment of its level's candidate slice (its worst-case survivor count long)
};

// Fused pyramid: one workgroup computes, for one tile of the TOP level, the regions of every lower level that
// feed it ("need"), and writes the part of each level it owns ("own").  Rects are {x0, y0, x1, y1}, x1/y1 exclusive.
struct PyrTile {
  int16_t need[kMaxLevels][4];
  int16_t own[kMaxLevels][4];
  int32_t tab_off, tab_n;  // this tile's slice of Geometry::pyrTileTab
};
// Candidate top-level tile edges of the fused pyramid.  A larger tile recomputes less halo but needs more LDS and
// yields fewer workgroups: 36 is ahead for >= ~2000 workgroups per launch (0.214 -> 0.194 ms per 256 C2 frames),
// 32 for small batches and for geometries whose 36-tiling would not leave three workgroups per CU.
// Tiling 2 (16-pixel top tiles) is for the LATENCY path: one frame is 30 workgroups at edge 32 on a 256-CU part, each
// walking a ~130 x 130-pixel level-0 cone alone; at edge 16 it is 108 workgroups with a third of the work each (the
// recomputed halo grows, which a launch that does not fill the chip does not pay for).
enum { kPyrTilings = 3, kPyrTilingSmall = 2 };
constexpr int kPyrTileEdge[kPyrTilings] = {32, 36, 16};

struct PyrTiling {
  std::vector<PyrTile> tiles;
  // Per tile, per level 1..top: the tile's slice of the resize tables, already rebased to the tile's LDS images:
  //   x entries {sx - sx0a, a0, a1, sx1 - sx0a}   then   y entries {(sy0 - y0) * spitch, (sy1 - y0) * spitch, b0, b1}
  std::vector<Short4> tab;
  int ldsA = 0, ldsB = 0;  // LDS bytes for even / odd levels of the ping-pong
  int tabMax = 0;          // max over tiles of staged table entries
  bool ok = false;         // usable by k_pyramid (16-bit LDS row offsets, column span)
  int lds_bytes() const { return ((ldsA + 15) & ~15) + ((ldsB + 15) & ~15) + 8 * (tabMax + 1); }
};

struct Geometry {
  FrameGeom fg;
  PyrTiling pyr[kPyrTilings];
  int maxCellsPerLevel = 0;
  int fastMaxVh = 0, fastMaxVw = 0, fastMaxArea = 0;  // over all FAST cells: rows, columns and pixels of the valid region (LDS sizing)
  std::vector<Short4> resizeTab;  // all levels, x tables then y tables (offsets in LevelGeom)
  std::vector<CellDesc> cells;
  std::vector<FastCellRec> fastRecs;  // one per cell, same order
  int maxQuota = 0;
};


inline int16_t sat_short(float v) {
  int iv = cv_round_f(v);
  return (int16_t)(iv < -32768 ? -32768 : iv > 32767 ? 32767 : iv);
}

inline void resize_axis_table(int dn, int sn, bool is_x, std::vector<Short4> &out) {
  const double scale = 1.0 / ((double)dn / sn);
  for (int d = 0; d < dn; d++) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)std::floor(f);
    f -= s;
    Short4 e;
    if (is_x) {
      if (s < 0) {
        f = 0;
        s = 0;
      }
      if (s >= sn - 1) {
        f = 0;
        s = sn - 1;
      }
      e.a = (int16_t)s;
      e.b = sat_short((1.f - f) * 2048);
      e.c = sat_short(f * 2048);
      e.d = (int16_t)(s + 1 < sn ? s + 1 : sn - 1);
    } else {
      int s0 = s < 0 ? 0 : s >= sn ? sn - 1 : s;
      int s1 = s + 1 < 0 ? 0 : s + 1 >= sn ? sn - 1 : s + 1;
      e.a = (int16_t)s0;
      e.b = (int16_t)s1;
      e.c = sat_short((1.f - f) * 2048);
      e.d = sat_short(f * 2048);
    }
    out.push_back(e);
  }
}

// Returns 0 on success, <0 if the image size / parameters cannot be processed the way the reference
// would (the reference itself divides by zero on such inputs).
inline int build_geometry(Geometry &G, const ExtractorTables &T, int rows, int cols, int lap0, int lap1,
                          const uint16_t taps[7]) {
  FrameGeom &fg = G.fg;
  memset(&fg, 0, sizeof(fg));
  G.resizeTab.clear();
  G.cells.clear();
  G.fastMaxVh = G.fastMaxVw = G.fastMaxArea = 0;
  G.maxCellsPerLevel = 0;
  G.maxQuota = 0;
  fg.nlevels = T.nlevels;
  fg.rows = rows;
  fg.cols = cols;
  fg.iniTh = T.iniTh;
  fg.minTh = T.minTh;
  fg.lap0 = lap0;
  fg.lap1 = lap1;
  for (int k = 0; k < 7; k++) fg.taps[k] = taps[k];
  if (T.iniTh < 1 || T.minTh < 1 || T.iniTh > 255 || T.minTh > 255) return -3;
  int img_off = 0, cand_off = 0, sel_off = 0, out_cap = 0, blur_blocks = 0, blur_off = 0;
  int prev_w = 0, prev_h = 0;
  for (int l = 0; l < T.nlevels; l++) {
    LevelGeom &L = fg.lv[l];
    L.w = cv_round_f((float)cols * T.invScale[l]);  // :1175-1176
    L.h = cv_round_f((float)rows * T.invScale[l]);
    if (L.w > 4095 || L.h > 4095) return -3;
    L.pitch = (L.w + 63) & ~63;
    L.img_off = img_off;
    img_off += L.pitch * L.h;
    // FAST cell grid :795-809
    const int minB = kFastBorder, maxBX = L.w - kFastBorder, maxBY = L.h - kFastBorder;
    const float width = (float)(maxBX - minB), height = (float)(maxBY - minB);
    const float W = 35;
    L.nCols = (int)(width / W);
    L.nRows = (int)(height / W);
    if (L.nCols < 1 || L.nRows < 1) return -3;  // reference: division by zero
    L.wCell = (int)std::ceil(width / L.nCols);
    L.hCell = (int)std::ceil(height / L.nRows);
    if ((long long)L.nCols * L.nRows * L.wCell * L.hCell > 0xFFFFFF) return -3;
    if (L.nCols * L.nRows > G.maxCellsPerLevel) G.maxCellsPerLevel = L.nCols * L.nRows;
    L.cell_base = (int)G.cells.size();
    int cand_cap = 0;
    for (int i = 0; i < L.nRows; i++) {
      for (int j = 0; j < L.nCols; j++) {
        const int iniY = minB + i * L.hCell, iniX = minB + j * L.wCell;
        int maxY = iniY + L.hCell + 6, maxX = iniX + L.wCell + 6;
        if (maxY > maxBY) maxY = maxBY;
        if (maxX > maxBX) maxX = maxBX;
        CellDesc c;
        c.level = (int16_t)l;
        c.x0 = (int16_t)(iniX + 3);
        c.y0 = (int16_t)(iniY + 3);
        c.x1 = (int16_t)(maxX - 3);
        c.y1 = (int16_t)(maxY - 3);
        c.pad = 0;
        c.cand_off = 0;
        if (c.x1 < c.x0) c.x1 = c.x0;  // empty (the reference `continue`s or FAST finds nothing)
        if (c.y1 < c.y0) c.y1 = c.y0;
        G.cells.push_back(c);
        if (c.y1 - c.y0 > G.fastMaxVh) G.fastMaxVh = c.y1 - c.y0;
        if (c.x1 - c.x0 > G.fastMaxVw) G.fastMaxVw = c.x1 - c.x0;
        if ((c.x1 - c.x0) * (c.y1 - c.y0) > G.fastMaxArea) G.fastMaxArea = (c.x1 - c.x0) * (c.y1 - c.y0);
        // a strict 3x3 local maximum occupies a 2x2 block: worst-case survivors per cell
        G.cells.back().cand_off = cand_cap;
        cand_cap += ((c.x1 - c.x0 + 1) / 2) * ((c.y1 - c.y0 + 1) / 2);
      }
    }
    L.quota = T.quota[l];
    if (L.quota > kMaxQuota) return -3;
    if (L.quota > G.maxQuota) G.maxQuota = L.quota;
    L.cand_off = cand_off;
    L.cand_cap = (cand_cap + 63) & ~63;
    cand_off += L.cand_cap;
    L.sel_off = sel_off;
    // final list size: <= quota+3 from the careful phase (:692,:753-754), or <= 4*nIni when the very
    // first pass already reaches the quota; filled in below once nIni is known
    // resize tables
    L.tab_x_off = L.tab_y_off = 0;
    if (l > 0) {
      L.tab_x_off = (int)G.resizeTab.size();
      resize_axis_table(L.w, prev_w, true, G.resizeTab);
      L.tab_y_off = (int)G.resizeTab.size();
      resize_axis_table(L.h, prev_h, false, G.resizeTab);
    }
    prev_w = L.w;
    prev_h = L.h;
    // blur launch: thread = 4 px x kBlurStrip rows, 256 threads per workgroup, workgroups never straddle levels
    // (strips are 9 tile rows; the four column groups of a tile column are four adjacent lanes)
    static_assert(kBlurStrip % kBlurTileH == 0, "a blur strip is a whole number of tile rows");
    L.btx = (L.w + kBlurTileW - 1) / kBlurTileW;
    L.bty = (L.h + kBlurTileH - 1) / kBlurTileH;
    L.boff = blur_off;
    blur_off += L.btx * L.bty * kBlurTileBytes;
    L.blur_int_tc = L.w >= 36 ? (L.w - 20) / 16 : 0;  // tile columns tc >= 1 with 16 tc + 12 + 8 <= w
    L.blur_block_base = blur_blocks;
    L.blur_nxg = 4 * L.btx;
    L.blur_nys = (L.h + kBlurStrip - 1) / kBlurStrip;
    blur_blocks += (L.blur_nxg * L.blur_nys + 255) / 256;
    L.scale = T.scale[l];
    L.kp_size = (float)(int)((float)(2 * kHalfPatch + 1) * T.scale[l]);  // :884
    // octree initial nodes :566-593
    L.oct_width = maxBX - minB;
    L.oct_height = maxBY - minB;
    L.nIni = (int)std::round((float)L.oct_width / (float)L.oct_height);
    if (L.nIni < 1 || L.nIni > kMaxIniNodes) return -3;  // reference: hX = inf / out-of-range index
    const float hX = (float)L.oct_width / (float)L.nIni;
    for (int i = 0; i <= L.nIni; i++) L.iniUL[i] = (int)(hX * (float)i);
    for (int i = 0; i < L.nIni; i++) L.iniThresh[i] = 0x7FFFFFFF;
    for (int x = L.oct_width - 1; x >= 0; x--) {
      int idx = (int)((float)x / hX);
      if (idx >= L.nIni) return -3;  // reference: out-of-range vpIniNodes index
      L.iniThresh[idx] = x;          // ends at the smallest x of every index
    }
    for (int i = 1; i < L.nIni; i++) {  // monotone fill for indices no x maps to
      if (L.iniThresh[i] == 0x7FFFFFFF) L.iniThresh[i] = 0x7FFFFFFE;
    }
    L.iniThresh[0] = 0;
    {
      int worst = L.quota + 3 > 4 * L.nIni ? L.quota + 3 : 4 * L.nIni;
      L.sel_cap = (worst + 3) & ~3;
      sel_off += L.sel_cap;
      out_cap += L.sel_cap;
    }
  }
  // ---- fused-pyramid tiles
  bool spanOk = true;
  for (int l = 1; l < T.nlevels; l++) {
    // k_pyramid picks the (sx, sx + 1) pairs of 4 adjacent columns out of 8 source bytes: sx may advance by at
    // most 6 over 3 columns (any scale factor up to 2); larger ratios take the per-level kernel.
    const LevelGeom &D = fg.lv[l];
    const Short4 *tabx = &G.resizeTab[D.tab_x_off];
    for (int x = 0; x + 1 < D.w; x++) {
      const int x3 = x + 3 < D.w ? x + 3 : D.w - 1;
      if (tabx[x3].a - tabx[x].a > 6) spanOk = false;
    }
  }
  for (int ti = 0; ti < kPyrTilings; ti++) {
    PyrTiling &PT = G.pyr[ti];
    PT = PyrTiling();
    if (T.nlevels <= 1) continue;
    PT.ok = spanOk;
    const int kPyrTile = kPyrTileEdge[ti];
    const int top = T.nlevels - 1;
    // balanced split of the top level into tiles of at most kPyrTile x kPyrTile
    const int tw = fg.lv[top].w, thh = fg.lv[top].h;
    const int ntx = (tw + kPyrTile - 1) / kPyrTile, nty = (thh + kPyrTile - 1) / kPyrTile;
    for (int ty = 0; ty < nty; ty++)
      for (int tx = 0; tx < ntx; tx++) {
        PyrTile t;
        memset(&t, 0, sizeof(t));
        int16_t *o = t.own[top];
        o[0] = (int16_t)((long long)tx * tw / ntx);
        o[1] = (int16_t)((long long)ty * thh / nty);
        o[2] = (int16_t)((long long)(tx + 1) * tw / ntx);
        o[3] = (int16_t)((long long)(ty + 1) * thh / nty);
        memcpy(t.need[top], o, sizeof(t.need[top]));
        for (int l = top; l >= 1; l--) {
          const LevelGeom &D = fg.lv[l], &S = fg.lv[l - 1];
          const Short4 *tabx = &G.resizeTab[D.tab_x_off], *taby = &G.resizeTab[D.tab_y_off];
          const int16_t *od = t.own[l], *nd = t.need[l];
          int16_t *os = t.own[l - 1], *ns = t.need[l - 1];
          // ownership boundaries follow the (monotone) source index of the destination boundary
          os[0] = (int16_t)(tx == 0 ? 0 : tabx[od[0]].a);
          os[2] = (int16_t)(tx + 1 == ntx ? S.w : tabx[od[2]].a);
          os[1] = (int16_t)(ty == 0 ? 0 : taby[od[1]].a);
          os[3] = (int16_t)(ty + 1 == nty ? S.h : taby[od[3]].a);
          // needed source region = sources of the needed destination region, plus what this tile owns
          int nx0 = tabx[nd[0]].a, nx1 = tabx[nd[2] - 1].d + 1, ny0 = taby[nd[1]].a, ny1 = taby[nd[3] - 1].b + 1;
          ns[0] = (int16_t)(nx0 < os[0] ? nx0 : os[0]);
          ns[2] = (int16_t)(nx1 > os[2] ? nx1 : os[2]);
          ns[1] = (int16_t)(ny0 < os[1] ? ny0 : os[1]);
          ns[3] = (int16_t)(ny1 > os[3] ? ny1 : os[3]);
        }
        int tabn = 0;
        for (int l = 1; l <= top; l++) tabn += (t.need[l][2] - t.need[l][0]) + (t.need[l][3] - t.need[l][1]);
        if (tabn > PT.tabMax) PT.tabMax = tabn;
        t.tab_off = (int32_t)PT.tab.size();
        t.tab_n = tabn;
        for (int l = 1; l <= top; l++) {
          const LevelGeom &D = fg.lv[l];
          const int sx0a = t.need[l - 1][0] & ~3, sy0 = t.need[l - 1][1];
          const int spitch = ((t.need[l - 1][2] - sx0a) + 3) & ~3;
          for (int x = t.need[l][0]; x < t.need[l][2]; x++) {
            Short4 e = G.resizeTab[D.tab_x_off + x];
            e.a = (int16_t)(e.a - sx0a);
            e.d = (int16_t)(e.d - sx0a);
            PT.tab.push_back(e);
          }
          for (int y = t.need[l][1]; y < t.need[l][3]; y++) {
            Short4 e = G.resizeTab[D.tab_y_off + y];
            if ((e.b - sy0) * spitch > 32767) PT.ok = false;  // LDS row offsets are kept in 16 bits
            e.a = (int16_t)((e.a - sy0) * spitch);
            e.b = (int16_t)((e.b - sy0) * spitch);
            PT.tab.push_back(e);
          }
        }
        for (int l = 0; l <= top; l++) {
          // LDS image of a level: columns start at the 4-byte aligned column below need.x0, pitch multiple of 4
          const int x0a = t.need[l][0] & ~3;
          const int pitch = ((t.need[l][2] - x0a) + 3) & ~3;
          const int bytes = pitch * (t.need[l][3] - t.need[l][1]);
          if (l & 1) {
            if (bytes > PT.ldsB) PT.ldsB = bytes;
          } else {
            if (bytes > PT.ldsA) PT.ldsA = bytes;
          }
        }
        PT.tiles.push_back(t);
      }
  }
  fg.pyr_frame_bytes = (img_off + 255) & ~255;
  // FastCellRec per cell (layout: vsg_common.h); the run / mask arithmetic is the kernel's, done here once
  // + one empty record past the end: k_fast_cells loads the record of cell ci + 1 unconditionally
  G.fastRecs.assign(G.cells.size() + 1, FastCellRec());
  memset(&G.fastRecs.back(), 0, sizeof(FastCellRec));
  for (size_t i = 0; i < G.cells.size(); i++) {
    const CellDesc &c = G.cells[i];
    const LevelGeom &L = fg.lv[c.level];
    FastCellRec &R = G.fastRecs[i];
    memset(&R, 0, sizeof(R));
    const int vw = c.x1 - c.x0, vh = c.y1 - c.y0;
    R.w[1] = (uint32_t)L.img_off;
    R.w[2] = (uint32_t)(c.level == 0 ? 0 : L.pitch) | ((uint32_t)c.level << 20);
    R.w[6] = (uint32_t)(L.cand_off + c.cand_off);
    R.w[7] = (uint32_t)(uint16_t)c.x0 | ((uint32_t)(uint16_t)c.y0 << 16);
    if (vw <= 0 || vh <= 0) continue;
    if (L.pitch >= (1 << 20) || vw > 127 || vh > 127 || c.x0 < 3 || c.y0 < 3) return -3;
    const int ax = (c.x0 - 3) & ~3, ox = (c.x0 - 3) - ax;
    const int tdw = (ox + vw + 6 + 3) >> 2, nq4 = (tdw + 3) >> 2;
    const int g0 = (3 + ox) >> 2, g1 = (3 + ox + vw + 3) >> 2, ng = g1 - g0, nrun = (ng + 1) >> 1;
    if (tdw > 31 || nq4 > 7 || nrun > 15 || g0 > 1) return -3;
    // first run of a row: its pixel 0 is valid-region column cb0 >= -3; last run: `over` of its 8 columns lie beyond vw
    const int cb0 = 4 * g0 - 3 - ox, over = 4 * (g0 + (nrun - 1) * 2) - 3 - ox + 8 - vw;
    auto mask8 = [](uint32_t m8) {  // pixel p < 4 -> byte p, bits 5 (dark) / 4 (bright); p >= 4 -> byte p - 4, bits 7 / 6
      uint32_t f = 0;
      for (int p = 0; p < 8; p++)
        if (m8 & (1u << p)) f |= (p < 4 ? 0x30u : 0xC0u) << (8 * (p & 3));
      return f;
    };
    R.w[0] = (uint32_t)ax | ((uint32_t)(c.y0 - 3) << 16);
    R.w[3] = (uint32_t)vw | ((uint32_t)vh << 7) | ((uint32_t)ox << 14) | ((uint32_t)tdw << 16) | ((uint32_t)nq4 << 21) |
             ((uint32_t)g0 << 24) | ((uint32_t)nrun << 25);
    R.w[4] = mask8(cb0 < 0 ? (0xFFu << (-cb0)) & 0xFFu : 0xFFu);
    R.w[5] = mask8(over <= 0 ? 0xFFu : over >= 8 ? 0u : (1u << (8 - over)) - 1u);
  }
  {
    // k_fast_cells: a queue entry holds a run's dword index in the tile in 11 bits, a surviving pixel's byte offset in
    // 13 (vsg_kernels.hip); the tile pitch classes are launch_fast's.  The reference's 35 px cell grid gives <= 69 x 69.
    const int tp = G.fastMaxVw <= 40 ? 52 : G.fastMaxVw <= 56 ? 68 : 84;
    if (G.fastMaxVw + 9 > tp || (G.fastMaxVh + 7) * tp > 8192) return -3;
  }
  fg.cand_frame = cand_off;
  fg.sel_frame = sel_off;
  fg.total_cells = (int)G.cells.size();
  // candidates per cell in fixed segments (no returning atomic in k_fast_cells; the octree gathers them through a
  // prefix of the cell counts in LDS) as long as a level's cells fit that prefix; beyond: one list per level
  fg.cand_segmented = G.maxCellsPerLevel <= kOctMaxCells ? 1 : 0;
  fg.total_blur_blocks = blur_blocks;
  fg.blur_frame_bytes = blur_off;
  // output capacity: what operator() can produce = sum over levels of final list sizes
  fg.out_cap = out_cap;
  return 0;
}

}  // namespace vsg
