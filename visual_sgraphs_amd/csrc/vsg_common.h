// vsg_common.h -- PODs shared by the host runtime, the HIP kernels and the host-side unit tests.
#pragma once
#include <stdint.h>

namespace vsg {

enum {
  kMaxLevels = 16,
  kOctMaxCells = 4096,   // FAST cells per level whose counts k_octree can prefix in LDS (segmented candidate lists)
  kMaxIniNodes = 16,     // octree: round(width/height) initial nodes (ORBextractor.cc:566)
  kEdgeThreshold = 19,   // ORBextractor.cc:71
  kHalfPatch = 15,       // ORBextractor.cc:70
  kFastBorder = 16,      // EDGE_THRESHOLD - 3 (ORBextractor.cc:795)
  kCellMax = 70,         // valid FAST cell extent is < 70 px: ceil(d / floor(d/35)) for d >= 35
  kMaxQuota = 2500,      // per-level feature quota: the octree workspace (63 B per node) must fit the 160 KB LDS
};

// FAST candidate / selected keypoint packed in 32 bits: x | y<<12 | response<<24.
// x,y are relative to (16,16) of the level, as in vToDistributeKeys (ORBextractor.cc:868-873).
__attribute__((always_inline)) inline
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t
    pack_cand(int x, int y, int resp) {
  return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)resp << 24);
}
#define VSG_CAND_X(c) ((int)((c)&0xFFFu))
#define VSG_CAND_Y(c) ((int)(((c) >> 12) & 0xFFFu))
#define VSG_CAND_R(c) ((int)((c) >> 24))

// Per-level geometry, computed once per (image size, extractor parameters) on the host.
struct LevelGeom {
  int w, h, pitch;          // level image, row pitch in bytes (multiple of 64)
  int img_off;              // byte offset of the level inside one frame's pyramid block
  int nCols, nRows, wCell, hCell;  // FAST cell grid (ORBextractor.cc:803-809)
  int cell_base;            // index of this level's first cell in the global cell list
  int quota;                // mnFeaturesPerLevel[level]
  int cand_off, cand_cap;   // slice of the per-frame candidate array (uint32 units)
  int sel_off, sel_cap;     // slice of the per-frame selected-keypoint array
  int tab_x_off, tab_y_off; // resize tables (short4 units) for producing THIS level from level-1
  int blur_block_base;      // first workgroup of this level in the blur launch
  int blur_nxg, blur_nys;   // 4-pixel column groups per row (4 per tile column, padding included) / row strips of kBlurStrip rows
  // The BLURRED level is stored as 16 x 4-pixel tiles, one 64-byte line each (x fastest inside a tile, tiles row-major):
  // k_orient_desc's 37 x 37 patch then touches ~33 lines instead of ~58 row segments (tools/ubench_tile.hip).
  int btx, bty;             // tile columns = ceil(w / 16), tile rows = ceil(h / 4)
  int boff;                 // byte offset of the level inside one frame's blurred block (blur_frame_bytes)
  int blur_int_tc;          // tile columns 1 .. blur_int_tc are interior for the blur (all 12 source bytes of their four
                            // column groups exist): their threads come first, the edge tile columns' threads last
  float scale;              // mvScaleFactor[level]
  float kp_size;            // (float)(int)(31 * scale)   (ORBextractor.cc:884,893)
  // octree (ORBextractor.cc:562-593)
  int oct_width, oct_height;         // maxX-minX, maxY-minY = w-32, h-32
  int nIni;
  int iniUL[kMaxIniNodes + 1];       // UL.x of initial node i; [nIni] = UR.x of the last one
  int iniThresh[kMaxIniNodes];       // smallest x with (int)((float)x / hX) >= i
};

struct FrameGeom {
  int nlevels;
  int rows, cols;
  int pyr_frame_bytes;   // bytes of one frame's pyramid block (all levels)
  int blur_frame_bytes;  // bytes of one frame's blurred block (all levels, tiled: LevelGeom::boff)
  int cand_frame;        // uint32 per frame in the candidate array
  int sel_frame;         // uint32 per frame in the selected array
  int total_cells;       // FAST cells per frame (all levels)
  int cand_segmented;    // 1: k_fast_cells writes per-cell segments + counts, 0: one atomically appended list per level
  int total_blur_blocks;
  int out_cap;           // keypoint capacity per frame in the output arrays
  int iniTh, minTh;
  int lap0, lap1;
  uint16_t taps[8];      // 7 taps of the 8.8 fixed-point Gaussian (+1 pad)
  LevelGeom lv[kMaxLevels];
};

// Everything k_fast_cells needs to know about one cell, derived on the host once per geometry and read with ONE 32-byte
// scalar load (cell descriptor -> level geometry -> derived constants was a chain of dependent scalar loads and ~40
// scalar instructions in front of every cell).
//   w0 = ax | ty << 16        tile origin in the level image: ax = (x0 - 3) & ~3, ty = y0 - 3
//   w1 = byte offset of the level inside a frame's pyramid block (level 0: unused, the frame is read in place)
//   w2 = pitch | level << 20  row pitch of the level in bytes (level 0: 0 = take the caller's pitch)
//   w3 = vw | vh << 7 | ox << 14 | tdw << 16 | nq4 << 21 | g0 << 24 | nrun << 25     (all 0: empty cell)
//   w4, w5 = flag masks of the first / last run of a row (k_fast_cells: fast_flag_mask8)
//   w6 = the cell's segment in the frame's candidate array (level slice + cell offset, uint32 units)
//   w7 = x0 | y0 << 16        valid region origin in level coordinates
struct FastCellRec {
  uint32_t w[8];
};

// Level 0 is read in place from the caller's (or the staging) buffer: no ingest copy.
struct Src0 {
  const uint8_t *base;   // frame 0, 4-byte aligned
  size_t frame_stride;   // bytes between frames, multiple of 4
  int pitch;             // bytes per row, multiple of 4
};

// pyramid of one frame as raw level pointers (level 0 may live in the caller's buffer)
struct PyrView {
  const uint8_t *lvl[kMaxLevels];
  int pitch[kMaxLevels];
  int w[kMaxLevels], h[kMaxLevels];
};

enum { kBlurStrip = 36 };  // output rows per thread in k_blur (36 + 6 halo rows = 6 x 7-row window turns; 9 tile rows)
enum { kBlurTileW = 16, kBlurTileH = 4, kBlurTileBytes = 64 };
// byte offset of pixel (x, y) inside a tiled blurred level with `btx` tile columns
__attribute__((always_inline)) inline
#if defined(__HIPCC__)
    __host__ __device__
#endif
    int
    blur_tiled_offset(int x, int y, int btx) {
  return ((y >> 2) * btx + (x >> 4)) * kBlurTileBytes + (y & 3) * kBlurTileW + (x & 15);
}

// cv::KeyPoint-compatible record (28 bytes): pt.x pt.y size angle response octave class_id
struct KeyPointPOD {
  float x, y, size, angle, response;
  int32_t octave, class_id;
};

// per-frame header written by the slot kernel
struct FrameHeader {
  int n;                        // total keypoints
  int mono;                     // monoIndex (return value of operator())
  int level_start[kMaxLevels + 1];  // prefix of per-level counts
};

}  // namespace vsg
