// vsg_kernels.hip -- gfx950 kernels of the ORB extractor.  One launch per stage covers every pyramid
// level and every frame of the batch (grid.y / grid.z = frame), so a 64-frame batch fills 256 CUs.
//
//   k_pyramid       ORBextractor::ComputePyramid            ORBextractor.cc:1171-1195  ([OCV] resize INTER_LINEAR 8U)
//                   (all levels in one launch; k_resize = one level per launch, for scale factors > 2)
//   k_fast_cells    per-cell cv::FAST(20) else cv::FAST(7)  ORBextractor.cc:787-876    ([OCV] fast.cpp / fast_score.cpp)
//   k_octree        DistributeOctTree                       ORBextractor.cc:562-785    (vsg_octree_core.h)
//   k_blur          GaussianBlur 7x7 sigma 2                ORBextractor.cc:1129-1130  ([OCV] 8.8 fixed point)
//   k_slots         output ordering / lapping area          ORBextractor.cc:1117-1168
//   k_orient_desc   IC_Angle + computeOrbDescriptor         ORBextractor.cc:73-149, 472-480, 1074-1081
//
// All image arithmetic is integer; the float pieces live in vsg_math.h with explicit non-contracting ops.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "vsg_common.h"
#include "vsg_ctx.h"
#include "vsg_geometry.h"
#include "vsg_math.h"
#include "vsg_octree_core.h"
#include "vsg_kernels.h"

namespace vsg {

// ------------------------------------------------------------------------------------------------
// Level l of frame f: level 0 is read in place from the caller's buffer (Src0), levels >= 1 from the pyramid.
__device__ __forceinline__ const uint8_t *level_ptr(const FrameGeom *fg, const Src0 &s0, const uint8_t *pyr, int frame,
                                                    int level, int &pitch) {
  if (level == 0) {
    pitch = s0.pitch;
    return s0.base + (size_t)frame * s0.frame_stride;
  }
  pitch = fg->lv[level].pitch;
  return pyr + (size_t)frame * fg->pyr_frame_bytes + fg->lv[level].img_off;
}

// ------------------------------------------------------------------------------------------------
// (work item, frame) grids: workgroups are dealt round-robin over the 8 XCDs by linear block id (observed, not
// contractual: MI355X_MICROARCH.md "Workgroup dispatch"), so with the plain mapping the cells / tiles / keypoints of
// ONE frame are spread over all eight private L2s and every L2 fetches the frame (and its halos) again.  The remap
// gives the blocks that share an XCD whole frames: block L of the launch works on frame 8 * (slot / nx) + L % 8, item
// slot % nx (slot = L / 8).  Bijective over the first ny & ~7 frames; the tail frames keep the plain mapping.  A
// pure speed choice: results do not depend on it.  VSG_NO_XCD_REMAP builds without it (A/B).
struct BlockXY {
  int x, y;
};
__device__ __forceinline__ BlockXY frame_major_block() {
  BlockXY b = {(int)blockIdx.x, (int)blockIdx.y};
#ifndef VSG_NO_XCD_REMAP
  const unsigned nx = gridDim.x, ny8 = gridDim.y & ~7u;
  if (blockIdx.y < ny8) {
    const unsigned L = blockIdx.y * nx + blockIdx.x, slot = L >> 3, q = slot / nx;
    b.y = (int)(q * 8 + (L & 7));
    b.x = (int)(slot - q * nx);
  }
#endif
  return b;
}

// Pyramid: level `level` from level-1, chained like the reference (resize of the previous LEVEL).
// Thread = 4 horizontally adjacent destination pixels -> one aligned 32-bit store.
__global__ __launch_bounds__(256) void k_resize(uint8_t *__restrict__ pyr, const FrameGeom *__restrict__ fg,
                                                const Short4 *__restrict__ tab, Src0 s0, int level) {
  const LevelGeom &D = fg->lv[level];
  const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int y = blockIdx.y * 4 + threadIdx.y;
  if (x4 >= D.w || y >= D.h) return;
  uint8_t *frame = pyr + (size_t)blockIdx.z * fg->pyr_frame_bytes;
  int spitch;
  const uint8_t *S = level_ptr(fg, s0, pyr, blockIdx.z, level - 1, spitch);
  const Short4 ty = tab[D.tab_y_off + y];
  const uint8_t *S0 = S + (size_t)ty.a * spitch;
  const uint8_t *S1 = S + (size_t)ty.b * spitch;
  const int b0 = ty.c, b1 = ty.d;
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int x = x4 + k;
    if (x < D.w) {
      const Short4 tx = tab[D.tab_x_off + x];
      const int h0 = (int)S0[tx.a] * tx.b + (int)S0[tx.d] * tx.c;
      const int h1 = (int)S1[tx.a] * tx.b + (int)S1[tx.d] * tx.c;
      const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
      out |= (uint32_t)(v & 0xFF) << (8 * k);
    }
  }
  *(uint32_t *)(frame + D.img_off + (size_t)y * D.pitch + x4) = out;
}

// ------------------------------------------------------------------------------------------------
// floor(i / d) for 0 <= i < 2^16 and 1 <= d <= 128 with inv = 1.0f / d: (i + 0.5) / d stays >= 0.5/d away from
// every integer, far more than the float rounding error, so the truncation is exact (3 VALU ops, no v_rcp chain).
__device__ __forceinline__ int div_small(int i, float inv) { return (int)(((float)i + 0.5f) * inv); }

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Fused pyramid: ONE launch builds levels 1..L-1 of every frame.  A workgroup owns one tile of the top level and
// walks DOWN the dependency cone: it stages the level-0 region that feeds the tile in LDS, then produces level 1,
// 2, ... in LDS ping-pong buffers (each level resized from the previous LEVEL, exactly like the reference's chain,
// ORBextractor.cc:1184), writing to HBM only the part of each level it owns.  Ownership boundaries are the source
// indices of the destination boundaries, so tiles partition every level without gaps or overlaps; the ~25 % halo
// is recomputed instead of communicated.  HBM traffic: level 0 read once (+halo), levels 1.. written once.
#ifndef VSG_PYR_NT
#define VSG_PYR_NT 256
#endif
constexpr int kPyrThreads = VSG_PYR_NT;  // threads per pyramid tile (512: 0.196 -> 0.203 ms, 1024: 0.31 ms per 256 C2 frames)

__global__ __launch_bounds__(kPyrThreads) void k_pyramid(uint8_t *__restrict__ pyr, const FrameGeom *__restrict__ fg,
                                                 const Short4 *__restrict__ tile_tab, Src0 s0,
                                                 const PyrTile *__restrict__ tiles, int ldsA, int ldsAB,
                                                 int *__restrict__ cand_count) {
  extern __shared__ __attribute__((aligned(16))) uint8_t pyr_lds[];
  const BlockXY blk = frame_major_block();
  const PyrTile &T = tiles[blk.x];
  const int frame = blk.y, tid = threadIdx.x;
  // the per-level FAST candidate counters of this frame start the call at zero (k_fast_cells adds to them after this
  // kernel): one workgroup per frame clears them here instead of a launch of its own
  if (cand_count && blk.x == 0 && tid < kMaxLevels) cand_count[frame * kMaxLevels + tid] = 0;
  uint8_t *buf0 = pyr_lds, *buf1 = pyr_lds + ldsA;
  Short4 *s_tab = (Short4 *)(pyr_lds + ldsAB);  // the tile's slices of the resize tables, all levels
  // One round of independent global loads: the tile's (pre-rebased, contiguous) table slice and the level-0 region.
  // Written as "request everything, then store": as a plain loop the compiler waited for every load before the next one
  // was issued (load, vmcnt(0), ds_write, branch) -- 4-5 dependent trips to L2 / HBM for the region and 3 more for the
  // table at the head of every workgroup.  Every lane loads (clamped index) so that the requests stay one block.
  {
    constexpr int kTabAhead = 4, kRegAhead = 6;  // per thread: table entries (8 B) / region chunks (16 B) requested at once
    const Short4 *tt = tile_tab + T.tab_off;
    const int tab_n = T.tab_n;
    Short4 tv[kTabAhead];
#pragma unroll
    for (int k = 0; k < kTabAhead; k++) tv[k] = tt[min(tid + k * kPyrThreads, tab_n - 1)];
    const int x0a = T.need[0][0] & ~3, y0 = T.need[0][1];
    const int w4 = ((T.need[0][2] - x0a) + 3) >> 2, hh = T.need[0][3] - y0, pitch = 4 * w4;
    const uint8_t *src = s0.base + (size_t)frame * s0.frame_stride + (size_t)y0 * s0.pitch + x0a;
#ifdef VSG_PYR_STAGE1
    const bool narrow = true;
#else
    const bool narrow = w4 < 4;
#endif
    if (narrow) {
      const float inv = __builtin_amdgcn_rcpf((float)w4);
      for (int i = tid; i < w4 * hh; i += kPyrThreads) {
        const int r = div_small(i, inv), c = i - r * w4;
        // wave-uniform base + 32-bit lane offset: no 64-bit multiply per element
        *(uint32_t *)(buf0 + r * pitch + 4 * c) = *(const uint32_t *)(src + (uint32_t)(r * s0.pitch + 4 * c));
      }
    } else {
      // 16 bytes per lane; the last load of a row is pulled back so that it ends with the row (see k_fast_cells)
      typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
      const int nq4 = (w4 + 3) >> 2, n = nq4 * hh;
      const float inv = __builtin_amdgcn_rcpf((float)nq4);
      u32x4u rv[kRegAhead];
      int ro[kRegAhead];
#pragma unroll
      for (int k = 0; k < kRegAhead; k++) {
        const int i = min(tid + k * kPyrThreads, n - 1);
        const int r = div_small(i, inv), c = min(4 * (i - r * nq4), w4 - 4);
        rv[k] = *(const u32x4u *)(src + (uint32_t)(r * s0.pitch + 4 * c));
        ro[k] = r * pitch + 4 * c;
      }
#pragma unroll
      for (int k = 0; k < kRegAhead; k++)
        if (tid + k * kPyrThreads < n) {
          uint32_t *d = (uint32_t *)(buf0 + ro[k]);
          d[0] = rv[k].x, d[1] = rv[k].y, d[2] = rv[k].z, d[3] = rv[k].w;
        }
      for (int i = tid + kRegAhead * kPyrThreads; i < n; i += kPyrThreads) {  // regions beyond 24 KB
        const int r = div_small(i, inv), c = min(4 * (i - r * nq4), w4 - 4);
        const u32x4u v = *(const u32x4u *)(src + (uint32_t)(r * s0.pitch + 4 * c));
        uint32_t *d = (uint32_t *)(buf0 + r * pitch + 4 * c);
        d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
      }
    }
#pragma unroll
    for (int k = 0; k < kTabAhead; k++)
      if (tid + k * kPyrThreads < tab_n) s_tab[tid + k * kPyrThreads] = tv[k];
    for (int i = tid + kTabAhead * kPyrThreads; i < tab_n; i += kPyrThreads) s_tab[i] = tt[i];
  }
  __syncthreads();
  uint8_t *frame_base = pyr + (size_t)frame * fg->pyr_frame_bytes;
  int toff = 0;
  for (int l = 1; l < fg->nlevels; l++) {
    const LevelGeom &D = fg->lv[l];
    const uint8_t *src = (l & 1) ? buf0 : buf1;
    uint8_t *dst = (l & 1) ? buf1 : buf0;
    const int dx0 = T.need[l][0], dy0 = T.need[l][1], dw = T.need[l][2] - dx0, dh = T.need[l][3] - dy0;
    const int dx0a = dx0 & ~3, dpitch = ((T.need[l][2] - dx0a) + 3) & ~3;
    const Short4 *txv = &s_tab[toff], *tyv = &s_tab[toff + dw];
    toff += dw + dh;
    // Thread = 4 adjacent destination columns (one aligned dword of the LDS image and of the level in HBM)
    // walking DOWN a chunk of rows.
    //  * a source row is fetched as 12 aligned bytes, shifted to start at the first source column, and the
    //    (sx, sx+1) byte pair of each column is picked with a per-thread v_perm selector; the horizontal 11-bit
    //    interpolation of a pair is then ONE v_dot2_u32_u16 (weights a0 | a1 << 16).  Where the reference clamps
    //    sx1 to sx (image edge) its weight a1 is 0, so reading sx + 1 instead changes nothing.
    //  * consecutive destination rows share a source row 5 times out of 6 (scale 1.2): the horizontal sums of the
    //    lower row are kept in registers and reused as the upper row of the next destination row.
    //  * the dword goes to the LDS image (source of the next level) and, where this tile OWNS it, straight to HBM.
    // threads = row groups x column groups with the EXACT column-group count (a power-of-two split left 13-44 % of
    // the lanes idle on most levels: 78 % -> 97 % busy lanes over the chain)
    const int ncg = dpitch >> 2;  // column groups (<= 64)
    const int nrg = kPyrThreads / ncg;
    const int rg = div_small(tid, __builtin_amdgcn_rcpf((float)ncg)), cg = tid - rg * ncg;
    const int chunk = (dh + nrg - 1) / nrg, ya = rg * chunk, yb = min(dh, ya + chunk);
    if (rg < nrg && ya < yb) {
      uint32_t sel[4];
      u16x2 wgt[4];
      int o = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int ci = min(max(4 * cg + k - (dx0 - dx0a), 0), dw - 1);  // columns outside the needed range: any value
        // the whole 8-byte entry in ONE aligned LDS read: left alone the compiler fetches the three fields it needs as
        // ds_read_i16 + ds_read_b32 at offset 2 -- a mis-aligned dword read, which the LDS serves one lane at a time
        // (64 cycles, profiles/r03_g_ubench_lds_rates.txt); the empty asm makes it need all 64 bits
        uint2 tw = *(const uint2 *)&txv[ci];
        asm volatile("" : "+v"(tw.x), "+v"(tw.y));
        const Short4 tx = {(int16_t)(tw.x & 0xFFFFu), (int16_t)(tw.x >> 16), (int16_t)(tw.y & 0xFFFFu), 0};
        if (k == 0) o = tx.a;
        const uint32_t f = (uint32_t)(tx.a - o);  // 0..6 (vsg_geometry.h checks the span)
        sel[k] = f | (0x0Cu << 8) | ((f + 1) << 16) | (0x0Cu << 24);
        wgt[k] = (u16x2){(unsigned short)tx.b, (unsigned short)tx.c};
      }
      const uint8_t *swin = src + (o & ~3);
      const uint32_t sh = (uint32_t)(o & 3);
      auto hrow = [&](int off, uint32_t (&H)[4]) {
        const uint32_t *q = (const uint32_t *)(swin + off);
        const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
        const uint32_t A = __builtin_amdgcn_alignbyte(w1, w0, sh), B = __builtin_amdgcn_alignbyte(w2, w1, sh);
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
          H[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(B, A, sel[k])), wgt[k], 0u,
                                        false) >> 4;
          asm volatile("" : "+v"(H[k]));  // the shifted sum is THE value: otherwise the compiler carries the un-shifted one
                                          // across rows and shifts it again where it is reused
        }
      };
      // ownership of this column group: all 4 bytes, some (a seam between tiles), or none
      const int gx = dx0a + 4 * cg;  // level column of byte 0
      const int ox0 = T.own[l][0], oy0 = T.own[l][1], ox1 = T.own[l][2], oy1 = T.own[l][3];
      const bool full = gx >= ox0 && gx + 4 <= ox1, part = !full && gx < ox1 && gx + 4 > ox0;
      // Two destination rows per trip with the roles of the two horizontal-sum sets swapped (the lower source row of one
      // destination row is the upper one of the next 5 times out of 6: no register copies), the next row's table entry
      // requested before the current row is worked on, and the lower row's sums formed unconditionally (where the
      // reference clamps sy + 1 to sy the same row is summed twice: same value).  Ownership as one unsigned range test
      // per row, the level's address as a uniform base + a 32-bit lane offset.
      uint32_t HA[4] = {0, 0, 0, 0}, HB[4];
      int offc = -1;  // LDS row offset whose horizontal sums are cached (in the set that is "upper" next)
      uint8_t *drow = dst + ya * dpitch + 4 * cg;
      uint8_t *gbase = frame_base + D.img_off;
      uint32_t goff = (uint32_t)((dy0 + ya) * D.pitch + gx);
      const uint32_t gpitch = (uint32_t)D.pitch;
      const int own_lo = oy0 - dy0;
      const uint32_t own_n = (uint32_t)max(oy1 - oy0, 0);
      auto emit = [&](int y, const Short4 &ty, const uint32_t (&H0)[4], const uint32_t (&H1)[4]) {
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t v = ((__umul24((uint32_t)ty.c, H0[k]) >> 16) + (__umul24((uint32_t)ty.d, H1[k]) >> 16) + 2u) >> 2;
          out |= v << (8 * k);
        }
        *(uint32_t *)drow = out;
        if ((uint32_t)(y - own_lo) < own_n) {
          if (full) {
            *(uint32_t *)(gbase + goff) = out;
          } else if (part) {
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (gx + j >= ox0 && gx + j < ox1) gbase[goff + j] = (uint8_t)(out >> (8 * j));
          }
        }
        drow += dpitch;
        goff += gpitch;
      };
      // (the entry after a chunk's last row is read and not used: within the table, or the one entry of padding the
      // launch allocates after it)
      const Short4 *typ = tyv + ya;
      Short4 ty = typ[0];
      for (int y = ya; y < yb; y += 2, typ += 2) {
        const Short4 ty1 = typ[1];
        if (ty.a != offc) hrow(ty.a, HA);
        hrow(ty.b, HB);
        emit(y, ty, HA, HB);
        offc = ty.b;
        if (y + 1 >= yb) break;
        ty = typ[2];
        if (ty1.a != offc) hrow(ty1.a, HB);
        hrow(ty1.b, HA);
        emit(y + 1, ty1, HB, HA);
        offc = ty1.b;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// FAST-9-16 ([OCV] fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>) on an LDS tile of row pitch kTileP.
//   score = max over the 16 contiguous 9-arcs of min(+-(v - ring)) - 1;  corner-at-t <=> score >= t, and the
//   score does not depend on t.
// The LDS row pitches are template parameters (52, 68, 84 bytes for cells up to 40, 56, 70 px wide) so that
// the 16 ring offsets stay instruction immediates while the footprint follows the geometry.

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 pk_min(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s16x2 pk_swap(s16x2 a) { return a.yx; }

// inclusive wave scan on the DPP path (no LDS round trips): Hillis-Steele inside each 16-lane row with zero-filled
// row shifts, then lane 15 / lane 31 broadcasts carry the row totals across rows
__device__ __forceinline__ int wave_inclusive_scan_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

// phase 1b's trigger (see there): NUM x (runs with a passer) > DEN x (runs of the cell), per pass: more than half of the
// runs at iniThFAST, more than a quarter in the minThFAST pass -- there the two-pair test passes 17-50 % of a natural
// cell's pixels and the four-pair test removes 60 % of them for the price of 4 % of the pass (in-run A/Bs of eight
// settings: profiles/r05_a_fast_pass_ablation.txt, r05_d_fast_stage_b_trigger_ab.txt)
#ifndef VSG_FAST_STAGE_B_NUM
#define VSG_FAST_STAGE_B_NUM(pass) ((pass) ? 4 : 2)
#define VSG_FAST_STAGE_B_DEN(pass) 1
#endif
#ifndef VSG_OD_SPATIAL
#define VSG_OD_SPATIAL 0  // k_orient_desc processes a level's keypoints in Morton order of their cells (experiment)
#endif
#ifndef VSG_FAST_STAGE_C
#define VSG_FAST_STAGE_C 0  // phase 1c: 0 = off (default), 1 = with phase 1b in both passes, 2 = minThFAST pass only, 3 = iniThFAST pass only
#endif
#ifndef VSG_FAST_NT
#define VSG_FAST_NT 128  // threads per FAST cell: 2 waves keep more cells resident per CU than 4 (measured: 0.47 ->
                         // 0.38 ms per 256 frames); 1 wave runs out of LDS before it runs out of wave slots
#endif

// ------------------------------------------------------------------------------------------------
// The cell kernel.  Per cell: the valid region (3 px inside the reference's sub-image) staged in LDS, a cheap necessary
// test whose passers are COMPACTED into an LDS queue, the exact score and the non-max suppression on dense wavefronts,
// survivors into the cell's own segment of the level's candidate slice.  Like the reference the cell is first searched at
// iniThFAST and only if that yields nothing at minThFAST (ORBextractor.cc:832-851); NMS only looks at neighbours INSIDE
// the valid region (outside counts as 0, exactly like the zeroed row buffers of FAST_t).  Emission order is irrelevant
// (the octree ranks candidates).
//
//  * The necessary test runs on 6-BIT pixels, four per 32-bit operation.  With q(x) = x >> 2 per byte,
//        n < v - t   =>   q(v) - q(n) >= T6 := ceil((t - 2) / 4)        (4 (qv - qn) >= v - n - 3 >= t - 2)
//    and byte-wise  (q(v) + 128 - T6) - q(n)  stays inside [1, 191] -- no borrow between bytes -- with bit 7 set exactly
//    when q(v) - q(n) >= T6.  So the dark flags of four pixels against one ring pixel are ONE 32-bit subtract, the
//    bright flags one add (against 128 - T6 - q(v)), and the pairing (N or S) and (W or E) is two v_bitop3 per side.
//    The 6-bit pixels are a second copy of the tile, written by the staging pass (every tile dword is read ~3 times by
//    the test, once by the staging).  The 6-bit test is slightly weaker than the exact one (it lets differences of
//    t - 3 .. t through); every pixel it passes still gets the exact score, so the result is unchanged.
//  * The test keeps the POLARITY of every passer (bit 7 / bit 6 of its byte = dark / bright side possible).  A 9-arc
//    of darker and a 9-arc of brighter ring pixels cannot coexist (9 + 9 > 16), so the exact score of a passer is
//    the score of its possible side: max over arcs of the min of +-(v - ring), 48 packed min/max instead of 96 (the
//    sign rides on the v_pk_mad_i16 that forms the differences).  A pixel that passes on both sides (noise at low
//    thresholds) takes one queue entry per side; when that would overflow the queue the cell is unpacked again with one
//    entry per pixel and a bright-side retry pass.
//  * Runs of 8 pixels with a passer are appended (flag word + run index) by one ballot + one LDS add per wave; a dense
//    pass unpacks them into the pixel queue through a wave scan of their popcounts; queue entries name the run and the
//    flag's bit, and the lane that scores an entry turns it into (row, column).
//  * Workgroups of several consecutive cells with the next cell's tile in flight, one 32-byte record per cell
//    (k_fast_cells below), and the register caps that keep 8 waves per SIMD resident.
// c0 = the centre pixel's address - (3 P + 1): every ring byte then sits at a non-negative offset from ONE register
// (LDS instructions take unsigned immediates; from the centre the compiler rebased and added per negative offset)
template <int kTileP>
__device__ __forceinline__ int fast_score_side_raw(const uint8_t *c0, uint32_t dark) {
  const int P = kTileP;
  const uint8_t *c = c0 + (3 * P + 1);
  // (ds_read_u8_d16 / _d16_hi into the two halves of one register would save the packing, but with SRAM ECC on -- as on
  // this part -- a D16 load clears the other half instead of keeping it)
  // the centre and the 16 ring bytes are requested back to back and waited for once (the empty asm needs all of them):
  // left alone the scheduler interleaves read / wait / use pairs, one LDS round trip after the other
  const int olo[8] = {3 * P, 3 * P + 1, 2 * P + 2, P + 3, 3, -P + 3, -2 * P + 2, -3 * P + 1};
  uint32_t lo[8], hi[8], ctr = c[0];
#pragma unroll
  for (int k = 0; k < 8; k++) lo[k] = c[olo[k]], hi[k] = c[-olo[k]];
  asm volatile("" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]),
                    "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]), "+v"(hi[6]), "+v"(hi[7]),
                    "+v"(ctr));
  // lane pair (d[k], d[k+8]) of side-signed differences, one v_pk_mad_i16 each: ring * (-1) + v (dark), ring - v (bright)
  const short sv = dark ? (short)ctr : (short)-(int)ctr, sn = dark ? (short)-1 : (short)1;
  const s16x2 vs = {sv, sv}, ns = {sn, sn};
  s16x2 D[8];
#pragma unroll
  for (int k = 0; k < 8; k++) D[k] = __builtin_bit_cast(s16x2, lo[k] | (hi[k] << 16)) * ns + vs;
  // Minimum over every 9-arc with prefix / suffix minima of the two half rings (the packed lanes): with G[i] = min of
  // D[0..i] and H[i] = min of D[i..7] (low half: ring 0..7, high half: ring 8..15), the arc that starts at ring position
  // i is H_lo[i] with G_hi[i] and the one that starts at i + 8 is H_hi[i] with G_lo[i] -- pk_min(H[i], swap(G[i])) holds
  // both.  7 + 6 + 8 packed minima and 8 maxima instead of the 8 + 8 + 8 + 8 + 8 of the doubling scheme (FAST 0.418 -> 0.403 ms).
  s16x2 G[8], H[8];
  G[0] = D[0];
#pragma unroll
  for (int k = 1; k < 8; k++) G[k] = pk_min(G[k - 1], D[k]);
  H[7] = D[7];
#pragma unroll
  for (int k = 6; k >= 1; k--) H[k] = pk_min(H[k + 1], D[k]);
  H[0] = G[7];  // both are the minimum of a whole half ring
  s16x2 A = (s16x2){-256, -256};
#pragma unroll
  for (int k = 0; k < 8; k++) A = pk_max(A, pk_min(H[k], pk_swap(G[k])));
  return max((int)A.x, (int)A.y) - 1;
}
template <int kTileP>
__device__ __forceinline__ int fast_score_side(const uint8_t *c0, int floor_t, uint32_t dark) {
  const int s = fast_score_side_raw<kTileP>(c0, dark);
  return s >= floor_t ? s : 0;
}

// any 3-input boolean function in one v_bitop3_b32: the truth table is the function applied to 0xF0, 0xCC, 0xAA
#define VSG_BITOP3(a, b, c, expr) \
  __builtin_amdgcn_bitop3_b32((a), (b), (c), (uint32_t)([](uint32_t A, uint32_t B, uint32_t C) constexpr { return (expr); }(0xF0u, 0xCCu, 0xAAu)) & 0xFFu)
// One ds_add_rtn_u32 by the calling lane.  atomicAdd() on LDS goes through the compiler's atomic optimizer, which
// wraps the single-lane add in 8 more vector instructions (mbcnt over exec, a multiply, a second readfirstlane).
__device__ __forceinline__ int lds_add_rtn(int *p, int v) {
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int *)p;
  int old;
  asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(a), "v"(v) : "memory");
  return old;
}
// number of set bits of a wave mask below the calling lane
__device__ __forceinline__ int mbcnt64(uint64_t m) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// flag word of a run of 8 pixels: pixel p < 4 -> byte p, bits 5 (dark) / 4 (bright); pixel p >= 4 -> byte p - 4, bits 7 / 6
__device__ __forceinline__ uint32_t fast_flag_mask8(uint32_t m8) {
  uint32_t f = 0;
#pragma unroll
  for (int p = 0; p < 8; p++)
    if (m8 & (1u << p)) f |= (p < 4 ? 0x30u : 0xC0u) << (8 * (p & 3));
  return f;
}

// More than 80 SGPRs (VCC and the reserved ones included) cost a wave slot per SIMD on gfx950: with 75-77 numbered SGPRs
// the one-cell form of this kernel ran at 0.470 ms per 512 C2 frames, capped (a few scalars live in VGPR lanes) at 0.440.
#ifndef VSG_FAST_SGPRS
#define VSG_FAST_SGPRS 80
#endif
// One cell's record (FastCellRec, vsg_common.h) as ONE 32-byte scalar load, and what the staging code derives from it.
struct FastCell {
  uint32_t w3, first_mask, last_mask, cand_off, xy;
  const uint8_t *tsrc;  // image address of tile row 0, column 0
  int pitch, level;
  __device__ __forceinline__ int vw() const { return (int)(w3 & 127u); }
  __device__ __forceinline__ int vh() const { return (int)((w3 >> 7) & 127u); }
  __device__ __forceinline__ int ox() const { return (int)((w3 >> 14) & 3u); }
  __device__ __forceinline__ int tdw() const { return (int)((w3 >> 16) & 31u); }
  __device__ __forceinline__ int nq4() const { return (int)((w3 >> 21) & 7u); }
  __device__ __forceinline__ int g0() const { return (int)((w3 >> 24) & 1u); }
  __device__ __forceinline__ int nrun() const { return (int)(w3 >> 25); }
  __device__ __forceinline__ int th() const { return vh() + 6; }
  __device__ __forceinline__ bool live() const { return w3 != 0; }
};
__device__ __forceinline__ FastCell load_fast_cell(const FastCellRec *__restrict__ rec, const Src0 &s0, const uint8_t *pyr,
                                                   int pyr_frame_bytes, int frame) {
  typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
  const u32x8 r = *reinterpret_cast<const u32x8 *>(rec);
  FastCell c;
  c.w3 = r[3], c.first_mask = r[4], c.last_mask = r[5], c.cand_off = r[6], c.xy = r[7];
  c.level = (int)(r[2] >> 20);
  const uint8_t *img;
  if (c.level == 0) {
    c.pitch = s0.pitch;
    img = s0.base + (size_t)frame * s0.frame_stride;
  } else {
    c.pitch = (int)(r[2] & 0xFFFFFu);
    img = pyr + (size_t)frame * pyr_frame_bytes + r[1];
  }
  c.tsrc = img + (size_t)((r[0] >> 16) * (uint32_t)c.pitch + (r[0] & 0xFFFFu));
  return c;
}

// The kernel's parameter list as it lies in the kernarg segment.  Launch-invariant scalars that the cell loop needs once per
// cell or per pass (pointers, per-frame strides, thresholds) are RE-LOADED from there at their point of use -- scalar loads
// out of the constant cache on the scalar unit's own issue port -- instead of being kept alive across the loop: under the
// 80-SGPR cap that keeps 8 waves per SIMD they did not fit, the compiler parked 31 of them in the lanes of a VGPR and the
// loop read them back with ~45 v_readlane_b32 per cell, each a slot of the VALU issue port this kernel is bound by
// (VERDICT r5 #2a).  The pointer passes through an empty asm so that every use site loads afresh (nothing to hoist or merge).
// A comparison on a value that passed through an empty asm is not loop-invariant to the compiler: it is made where it is
// used (one v_cmp) instead of once in front of the cell loop, parked as a 64-bit lane mask in two VGPR lanes and read back
// with two v_readlane_b32 at every use (the SGPR cap again)
#ifndef VSG_FAST_OPQ
#define VSG_FAST_OPQ 1
#endif
#ifndef VSG_FAST_HINTS
#define VSG_FAST_HINTS 1
#endif
#if VSG_FAST_OPQ
#define VSG_OPQ(x) ([&] { int t_ = (x); asm volatile("" : "+v"(t_)); return t_; }())
#else
#define VSG_OPQ(x) (x)
#endif
// block-frequency hints for the paths that almost never run (slivers, tiles that were not prefetched, the one-entry-per-pixel
// re-unpack, levels of more than 4096 cells): the register allocator weighs spills by block frequency
#if VSG_FAST_HINTS
#define VSG_COLD(x) __builtin_expect(!!(x), 0)
#define VSG_HOT(x) __builtin_expect(!!(x), 1)
#else
#define VSG_COLD(x) (x)
#define VSG_HOT(x) (x)
#endif
#ifndef VSG_FAST_KA
#define VSG_FAST_KA 3   // bit 0: the next cell's image bases, 1: the cell counter, 2: the candidate segment, 3: the record array
#endif
struct FastKArgs {
  const uint8_t *pyr;
  const FrameGeom *fg;
  const FastCellRec *recs;
  Src0 s0;
  uint32_t *cand;
  int *cand_count, *cell_count;
  int tile_bytes, score_bytes, queue_cap, cells_per_wg, pyr_frame_bytes, total_cells, cand_frame, th_pack;
};
typedef const FastKArgs __attribute__((address_space(4))) *FastKArgsPtr;
__device__ __forceinline__ FastKArgsPtr fast_kargs() {
  FastKArgsPtr p = (FastKArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

// A workgroup works through `cells_per_wg` consecutive cells of one frame.  The tile of the NEXT cell is fetched into
// registers (kPre 16-byte chunks per thread) while the current cell is tested, scored and suppressed, and written to LDS
// when the current cell is done: the kernel is bound by the workgroup slots of a CU (16: LDS and wave slots) times the
// life of a workgroup, and ~3 us of a one-cell workgroup's 6 us were launch + descriptor chain + the tile's trip from
// L2 (profiles/r03_c_fast_ablation.txt: 0.065 / 0.074 / 0.108 / 0.221 ms for workgroups that end at once / after the
// descriptors / after a staging pass without loads / after the real staging pass, of 0.441 ms).
template <int NT, int kTileP, int kScoreP, int kPre>
#ifndef VSG_FAST_WAVES_MIN
#define VSG_FAST_WAVES_MIN 8
#define VSG_FAST_WAVES_MAX 8
#endif
__global__ __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(VSG_FAST_SGPRS), amdgpu_waves_per_eu(VSG_FAST_WAVES_MIN, VSG_FAST_WAVES_MAX))) void k_fast_cells(
    const uint8_t *__restrict__ pyr, const FrameGeom *__restrict__ fg, const FastCellRec *__restrict__ recs, Src0 s0,
    uint32_t *__restrict__ cand, int *__restrict__ cand_count, int *__restrict__ cell_count, int tile_bytes,
    int score_bytes, int queue_cap, int cells_per_wg, int pyr_frame_bytes, int total_cells, int cand_frame, int th_pack) {
  static_assert(kScoreP == kTileP, "a pixel's score offset is its tile offset minus a constant");
  static_assert(offsetof(FastKArgs, s0) == 24 && offsetof(FastKArgs, cand) == 48 && offsetof(FastKArgs, cell_count) == 64 &&
                    offsetof(FastKArgs, tile_bytes) == 72 && offsetof(FastKArgs, th_pack) == 100,
                "FastKArgs mirrors this parameter list");
  extern __shared__ __attribute__((aligned(16))) uint8_t fast_lds[];
  uint8_t *tile = fast_lds, *score = fast_lds + tile_bytes;
  uint16_t *queue = (uint16_t *)(fast_lds + tile_bytes + score_bytes);  // queue_cap entries = the largest cell's pixels
  // the tile once more as 6-bit pixels (x >> 2 per byte), read by the necessary test only: it lives where the pixel
  // queue is built afterwards (the launcher sizes that region for both)
  uint8_t *qtile = (uint8_t *)queue;
  // flag word + index of every run with a passer; both lists are consumed before the score rows they alias are cleared
  uint32_t *runF = (uint32_t *)score;
  __shared__ int s_cnt[8];  // [0]=NMS survivors [1]=queue length [2]=emit cursor [3]=global base [4]=run entries
  const BlockXY blk = frame_major_block();
  const int frame = blk.y;
  const int tid = threadIdx.x, lane = tid & 63;
  // th_pack = iniThFAST | minThFAST << 8 | cand_segmented << 16 (launch-invariant, ONE register for the whole loop)
  const int c_begin = blk.x * cells_per_wg, c_end = min(c_begin + cells_per_wg, total_cells);
  typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
  u32x4u pre[kPre];
  int pre_off[kPre];  // where the chunk goes in the tile
  auto put = [&](int off, const u32x4u &v) {
    uint32_t *d = (uint32_t *)&tile[off];
    d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    uint32_t *dq = (uint32_t *)&qtile[off];
    dq[0] = (v.x >> 2) & 0x3F3F3F3Fu, dq[1] = (v.y >> 2) & 0x3F3F3F3Fu;
    dq[2] = (v.z >> 2) & 0x3F3F3F3Fu, dq[3] = (v.w >> 2) & 0x3F3F3F3Fu;
  };
  // 16 bytes per lane and chunk, the last chunk of a row pulled back so that it ENDS with the row (it re-writes a few dwords
  // of its neighbour with the same values instead of reading past the row -- level 0 is the caller's buffer).
  // Every lane loads (the ones past the last chunk load it again): the kPre loads of a thread stay one block of
  // back-to-back instructions -- behind per-load branches the compiler put a vmcnt(0) in front of each.
  // Not for slivers (rows shorter than a chunk) or tiles beyond kPre chunks per thread: staged when their turn comes.
  // (row, first dword) of a thread's chunks only depend on the chunks per row, which rarely change from cell to cell: kept
  // across cells in the narrow tile class (the wider ones have no registers to spare) and clamped per cell instead of
  // divided out again
  constexpr bool kKeepRC = kPre * NT == 256;
  int rc_nq4 = 0, rc_r[kKeepRC ? kPre : 1], rc_c[kKeepRC ? kPre : 1];
  auto fetch = [&](const FastCell &N) -> bool {
    const int n = N.nq4() * N.th();
    if (!N.live() || N.tdw() < 4 || n > kPre * NT) return false;
    if constexpr (kKeepRC) {
      if (N.nq4() != rc_nq4) {  // workgroup-uniform
        rc_nq4 = N.nq4();
        const float inv_nq4 = __builtin_amdgcn_rcpf((float)rc_nq4);
#pragma unroll
        for (int k = 0; k < kPre; k++) {
          rc_r[k] = div_small(tid + k * NT, inv_nq4);
          rc_c[k] = 4 * (tid + k * NT - rc_r[k] * rc_nq4);
        }
      }
#pragma unroll
      for (int k = 0; k < kPre; k++) {
        // chunks past the tile's last (tid + k NT >= n: never committed) re-read a chunk of its last row
        const int r = min(rc_r[k], N.th() - 1), c = min(rc_c[k], N.tdw() - 4);
        pre[k] = *(const u32x4u *)(N.tsrc + (uint32_t)(r * N.pitch + 4 * c));
        pre_off[k] = r * kTileP + 4 * c;
      }
    } else {
      const float inv_nq4 = __builtin_amdgcn_rcpf((float)N.nq4());
#pragma unroll
      for (int k = 0; k < kPre; k++) {
        const int i = min(tid + k * NT, n - 1);
        const int r = div_small(i, inv_nq4), c = min(4 * (i - r * N.nq4()), N.tdw() - 4);
        pre[k] = *(const u32x4u *)(N.tsrc + (uint32_t)(r * N.pitch + 4 * c));
        pre_off[k] = r * kTileP + 4 * c;
      }
    }
    return true;
  };
  FastCell C = load_fast_cell(recs + c_begin, s0, pyr, pyr_frame_bytes, frame);
  bool have_pre = fetch(C);  // the first cell of the workgroup waits for its tile
  for (int ci = c_begin; ci < c_end; ci++) {
    // the record of the next cell: its scalar load runs beside the LDS writes below.  Its inputs come from the kernarg
    // segment again (see FastKArgs); the record array carries one empty record past its end, so no clamp of the index
    FastCell N;
#if VSG_FAST_KA & 1
    {
      const FastKArgsPtr ka = fast_kargs();
      const Src0 ks0 = {ka->s0.base, ka->s0.frame_stride, ka->s0.pitch};
#if VSG_FAST_KA & 8
      N = load_fast_cell(ka->recs + (ci + 1), ks0, ka->pyr, ka->pyr_frame_bytes, frame);
#else
      N = load_fast_cell(recs + (ci + 1), ks0, ka->pyr, ka->pyr_frame_bytes, frame);
#endif
    }
#else
    N = load_fast_cell(recs + (ci + 1), s0, pyr, pyr_frame_bytes, frame);
#endif
    const bool live = C.live();
    if (live) {
      if (VSG_COLD(C.tdw() < 4)) {  // a sliver of a cell at the right edge of a level (cell-uniform)
        const float inv_tdw = __builtin_amdgcn_rcpf((float)C.tdw());
        for (int i = tid; i < C.tdw() * C.th(); i += NT) {
          const int r = div_small(i, inv_tdw), c = i - r * C.tdw();
          const uint32_t v = *(const uint32_t *)(C.tsrc + (uint32_t)(r * C.pitch + 4 * c));
          *(uint32_t *)&tile[r * kTileP + 4 * c] = v;
          *(uint32_t *)&qtile[r * kTileP + 4 * c] = (v >> 2) & 0x3F3F3F3Fu;
        }
      } else {
        const int n = C.nq4() * C.th();
        const float inv_nq4 = __builtin_amdgcn_rcpf((float)C.nq4());
        if (VSG_HOT(have_pre)) {
#pragma unroll
          for (int k = 0; k < kPre; k++)
            if (tid + k * NT < n) put(pre_off[k], pre[k]);
        } else {
          for (int i = tid; i < n; i += NT) {
            const int r = div_small(i, inv_nq4), c = min(4 * (i - r * C.nq4()), C.tdw() - 4);
            put(r * kTileP + 4 * c, *(const u32x4u *)(C.tsrc + (uint32_t)(r * C.pitch + 4 * c)));
          }
        }
      }
    }
    // the next cell's tile is on its way while this one is worked on
    have_pre = ci + 1 < c_end && fetch(N);
    const bool seg = (th_pack >> 16) != 0;
    auto my_count = [&]() -> int * {  // the cell's counter, from the kernarg segment at the moment it is written
#if VSG_FAST_KA & 2
      const FastKArgsPtr ka = fast_kargs();
      return ka->cell_count + ((size_t)frame * ka->total_cells + ci);
#else
      return cell_count + ((size_t)frame * total_cells + ci);
#endif
    };
    if (!live) {  // empty cell
      if (seg && VSG_OPQ(tid) == 0) *my_count() = 0;
      C = N;
      continue;
    }
    {
      const int iniTh = th_pack & 0xFF, minTh = (th_pack >> 8) & 0xFF;
      const int vw = C.vw(), vh = C.vh(), ox = C.ox(), th = C.th();
      const int cell_x0 = (int)(C.xy & 0xFFFFu), cell_y0 = (int)(C.xy >> 16), cell_level = C.level;
        // centre pixels live in tile columns [3 + ox, 3 + ox + vw): dword groups g0 .. g1-1, runs of 2 dwords per row
        const int g0 = C.g0(), nrun = C.nrun(), nruns = nrun * vh;
        uint16_t *runI = (uint16_t *)(runF + nruns);  // 6 bytes per run <= the score row pitch (static_assert at the launcher)
        const float inv_nrun = __builtin_amdgcn_rcpf((float)nrun);
        // Phase 1 lane layout, fixed for the cell: lane -> (row r0 of a band of rows_it rows, run rr of the row); a lane's
        // run masks, tile offset and the "left of column 0" fix are then per-cell values and the loop over bands adds a
        // uniform stride -- per iteration that was a division, three multiplies and six mask instructions per lane.
        const int rows_it = __builtin_amdgcn_readfirstlane(div_small(NT, inv_nrun));  // whole rows per iteration (nrun <= 15: >= 8)
        const int p1_row = div_small(tid, inv_nrun), p1_rr = tid - p1_row * nrun;
        const int p1_r0 = p1_row < rows_it ? p1_row : 1 << 20;  // lanes past the last whole row idle
        const uint32_t p1_mask = (p1_rr == 0 ? C.first_mask : 0xF0F0F0F0u) & (p1_rr == nrun - 1 ? C.last_mask : 0xF0F0F0F0u);
        const uint32_t p1_left = g0 == 0 && p1_rr == 0 ? 0u : ~0u;  // and-mask of the dword left of the run (see below)
        const int p1_off = (p1_row + 3) * kTileP + 4 * (g0 + 2 * p1_rr);
        uint32_t keep = 0;
        int nq = 0, thr = iniTh;
#ifdef VSG_FAST_PASS_UNROLL
#pragma unroll VSG_FAST_PASS_UNROLL
#endif
        for (int pass = 0; pass < 2; pass++) {
          if (VSG_OPQ(tid) < 5) s_cnt[tid] = 0;
          __syncthreads();  // the tile is staged / the previous pass is done with the score rows
          // ---- phase 1: 6-bit necessary test, 8 pixels (2 dwords) per thread; runs with a passer are appended (flag word
          // + run index) to the run list: one ballot and one LDS atomic per wave and iteration
          {
            const int t6 = thr <= 2 ? 0 : min(64, (thr + 1) >> 2);  // ceil((t - 2) / 4)
            const uint32_t K = (uint32_t)(128 - t6) * 0x01010101u, H = 0x80808080u;
            for (int rb = 0; rb < vh; rb += rows_it) {
              uint32_t F = 0;
              const int runB = p1_off + rb * kTileP;  // the run's first byte in the tile
              if (rb + p1_r0 < vh) {
                const uint32_t *pc = (const uint32_t *)&qtile[runB];
                const uint32_t *pu = pc - 3 * (kTileP / 4), *pd = pc + 3 * (kTileP / 4);
                uint32_t qc[4] = {pc[-1], pc[0], pc[1], pc[2]};
                // Bytes of the 6-bit tile that were never staged (beyond a row's last staged dword) may hold anything, and
                // a byte above 63 carries into the bytes ABOVE it.  Those are pixels further right, invalid like the
                // byte itself -- except in the dword left of tile column 0, whose bytes sit BELOW a valid pixel's neighbour
                qc[0] &= p1_left;
                uint32_t f[2];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                  const uint32_t qu = pu[k], qd = pd[k];
                  const uint32_t qw = __builtin_amdgcn_alignbyte(qc[k + 1], qc[k], 1);      // columns -3
                  const uint32_t qe = __builtin_amdgcn_alignbyte(qc[k + 2], qc[k + 1], 3);  // columns +3
                  const uint32_t A = qc[k + 1] + K, B = K - qc[k + 1];
                  const uint32_t dark = VSG_BITOP3(A - qu, A - qd, (A - qw) | (A - qe), (A | B) & C);
                  const uint32_t bright = VSG_BITOP3(B + qu, B + qd, (B + qw) | (B + qe), (A | B) & C);
                  f[k] = VSG_BITOP3(dark, bright >> 1, H, (A & C) | (B & ~C));  // bit 7 dark, bit 6 bright, the rest: anything
                }
                F = VSG_BITOP3(f[1], f[0] >> 2, 0xC0C0C0C0u, (A & C) | (B & ~C));
                // the run masks hold flag bits only (0xF0 per byte), so they also clear what the merges above left below them
                F &= p1_mask;
              }
              const uint64_t hit = __ballot(F != 0);
              if (hit) {
                int base = 0;
                if (lane == 0) base = lds_add_rtn(&s_cnt[4], __popcll(hit));
                const int slot = __builtin_amdgcn_readfirstlane(base) + mbcnt64(hit);
                if (F) runF[slot] = F, runI[slot] = (uint16_t)runB;
              }
            }
          }
          __syncthreads();
          // No pixel of the cell passes the necessary test at iniThFAST (every cell of smooth natural content: value noise,
          // defocus -- the tracking thresholds of ORBextractor.cc:832-851 were chosen so that such cells take the second
          // threshold): nothing to unpack, score or suppress, and the 6-bit tile is still whole (the queue that aliases it
          // was never written) -- straight to the minThFAST sweep.  Cell-uniform.
          if (pass == 0 && s_cnt[4] == 0 && minTh < thr) {
            thr = minTh;
            continue;
          }
#ifndef VSG_FAST_NO_STAGE_B
          // ---- phase 1b, dense cells only: the same necessary test on the two DIAGONAL ring pairs (ring positions 2 / 10
          // and 6 / 14: offsets (+-2, +-2)), run entry by run entry.  A 9-arc holds one pixel of every opposite pair, so a
          // corner on a side needs that side on all four tested pairs.  On the frames the pipeline is tuned on a cell
          // queues 60-190 pixels and this pass would cost more than the scores it saves (the reason the test above stops at
          // two pairs); on fine texture -- 1-2 px checkerboards, gratings, steep ramps -- EVERY pixel passes the N / S / W / E
          // test, on ramps on both sides, and the cell scored its whole area two or three times per threshold (FAST launch
          // 1.05-2.47 ms instead of 0.39 per 512 frames: profiles/r04_b_bench_content_sweep_before_stage_b.json).  The
          // diagonals see the other phase of a checkerboard and the iso-line of a ramp.  Cell-uniform trigger: the share of
          // the cell's runs that hold a passer (VSG_FAST_STAGE_B_NUM / _DEN above).
          if (VSG_FAST_STAGE_B_NUM(pass) * s_cnt[4] > VSG_FAST_STAGE_B_DEN(pass) * nruns) {
            const int nr1 = s_cnt[4];
            const int t6 = thr <= 2 ? 0 : min(64, (thr + 1) >> 2);
            const uint32_t K = (uint32_t)(128 - t6) * 0x01010101u, H = 0x80808080u;
            for (int e = tid; e < nr1; e += NT) {
              const uint32_t F0 = runF[e];
              const int runB = runI[e];
              const uint32_t *pc = (const uint32_t *)&qtile[runB];
              const uint32_t *pu = pc - 2 * (kTileP / 4), *pd = pc + 2 * (kTileP / 4);
              uint32_t qu[4] = {pu[-1], pu[0], pu[1], pu[2]}, qd[4] = {pd[-1], pd[0], pd[1], pd[2]};
              const uint32_t qc[2] = {pc[0], pc[1]};
              // the dword left of tile column 0 (see phase 1): only the first run of a row reads it, and only when g0 == 0
              const int col = runB - kTileP * div_small(runB, 1.0f / kTileP);
              const uint32_t left = (g0 == 0 && col == 0) ? 0u : ~0u;
              qu[0] &= left, qd[0] &= left;
              uint32_t g[2];
#pragma unroll
              for (int k = 0; k < 2; k++) {
                const uint32_t nw = __builtin_amdgcn_alignbyte(qu[k + 1], qu[k], 2);      // row -2, columns -2
                const uint32_t ne = __builtin_amdgcn_alignbyte(qu[k + 2], qu[k + 1], 2);  // row -2, columns +2
                const uint32_t sw = __builtin_amdgcn_alignbyte(qd[k + 1], qd[k], 2);      // row +2, columns -2
                const uint32_t se = __builtin_amdgcn_alignbyte(qd[k + 2], qd[k + 1], 2);  // row +2, columns +2
                const uint32_t A = qc[k] + K, B = K - qc[k];
                const uint32_t dark = VSG_BITOP3(A - nw, A - se, (A - ne) | (A - sw), (A | B) & C);
                const uint32_t bright = VSG_BITOP3(B + nw, B + se, (B + ne) | (B + sw), (A | B) & C);
                g[k] = VSG_BITOP3(dark, bright >> 1, H, (A & C) | (B & ~C));
              }
              // same flag layout as F; F0 already carries the run masks, so the bits the merge leaves below the flags die
              runF[e] = F0 & VSG_BITOP3(g[1], g[0] >> 2, 0xC0C0C0C0u, (A & C) | (B & ~C));
            }
            __syncthreads();
#if VSG_FAST_STAGE_C
            // ---- phase 1c (round 6 experiment, VERDICT r5 #2b; OFF by default -- measured: profiles/r06_d_fast_stage_c_ab.txt):
            // the same necessary test on the remaining FOUR ring pairs (1 / 9, 3 / 11, 5 / 13, 7 / 15: rows +-3 at columns -+1,
            // rows +-1 at columns -+3), run entry by run entry, for cells that took phase 1b.  ~95 vector instructions per run
            // entry against 62 per queued pixel it removes.
            if (VSG_FAST_STAGE_C == 1 || (VSG_FAST_STAGE_C == 2 && pass == 1) || (VSG_FAST_STAGE_C == 3 && pass == 0)) {
              for (int e = tid; e < nr1; e += NT) {
                const uint32_t F0 = runF[e];
                if (!F0) continue;
                const int runB = runI[e];
                const uint32_t *pc = (const uint32_t *)&qtile[runB];
                const uint32_t *p3u = pc - 3 * (kTileP / 4), *p3d = pc + 3 * (kTileP / 4), *p1u = pc - (kTileP / 4),
                               *p1d = pc + (kTileP / 4);
                uint32_t u3[4] = {p3u[-1], p3u[0], p3u[1], p3u[2]}, d3[4] = {p3d[-1], p3d[0], p3d[1], p3d[2]};
                uint32_t u1[4] = {p1u[-1], p1u[0], p1u[1], p1u[2]}, d1[4] = {p1d[-1], p1d[0], p1d[1], p1d[2]};
                const uint32_t qc[2] = {pc[0], pc[1]};
                const int col = runB - kTileP * div_small(runB, 1.0f / kTileP);
                const uint32_t left = (g0 == 0 && col == 0) ? 0u : ~0u;  // the dword left of tile column 0 (see phase 1)
                u3[0] &= left, d3[0] &= left, u1[0] &= left, d1[0] &= left;
                uint32_t g[2];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                  const uint32_t n9 = __builtin_amdgcn_alignbyte(u3[k + 1], u3[k], 3);       // row -3, columns -1
                  const uint32_t n7 = __builtin_amdgcn_alignbyte(u3[k + 2], u3[k + 1], 1);   // row -3, columns +1
                  const uint32_t n15 = __builtin_amdgcn_alignbyte(d3[k + 1], d3[k], 3);      // row +3, columns -1
                  const uint32_t n1 = __builtin_amdgcn_alignbyte(d3[k + 2], d3[k + 1], 1);   // row +3, columns +1
                  const uint32_t n11 = __builtin_amdgcn_alignbyte(u1[k + 1], u1[k], 1);      // row -1, columns -3
                  const uint32_t n5 = __builtin_amdgcn_alignbyte(u1[k + 2], u1[k + 1], 3);   // row -1, columns +3
                  const uint32_t n13 = __builtin_amdgcn_alignbyte(d1[k + 1], d1[k], 1);      // row +1, columns -3
                  const uint32_t n3 = __builtin_amdgcn_alignbyte(d1[k + 2], d1[k + 1], 3);   // row +1, columns +3
                  const uint32_t A = qc[k] + K, B = K - qc[k];
                  const uint32_t dA = VSG_BITOP3((A - n1) | (A - n9), (A - n7) | (A - n15), (A - n3) | (A - n11), A & B & C);
                  const uint32_t dark = dA & ((A - n5) | (A - n13));
                  const uint32_t bA = VSG_BITOP3((B + n1) | (B + n9), (B + n7) | (B + n15), (B + n3) | (B + n11), A & B & C);
                  const uint32_t bright = bA & ((B + n5) | (B + n13));
                  g[k] = VSG_BITOP3(dark, bright >> 1, H, (A & C) | (B & ~C));
                }
                runF[e] = F0 & VSG_BITOP3(g[1], g[0] >> 2, 0xC0C0C0C0u, (A & C) | (B & ~C));
              }
              __syncthreads();
            }
#endif
          }
#endif
          // ---- run list -> pixel queue.  Entry = the run's dword index in the tile << 5 | bit position of the flag in the
          // run's word (bit 0 of the position: dark side; bit 2, set in every flag position, cleared = retry); phase 2 turns
          // an entry into the pixel's byte offset in the tile -- two shifts and two masks, no division by the run count, no
          // multiply by the row pitch -- once per lane instead of once per entry of the unpack loop.  A pixel that passed the necessary test on both sides (noise
          // at low thresholds) normally gets one entry per side -- a spare lane in phase 2, at most one of the two can score.
          // That can exceed the queue (one slot per pixel of the largest cell) when most of a cell passes on both sides;
          // such a cell is unpacked again with ONE entry per pixel: both-sided pixels flagged dark + retry, scored dark
          // first and, in a second pass of their own, bright.
          const int nr = s_cnt[4];
          auto unpack = [&](const bool single) {
            for (int e0 = 0; e0 < nr; e0 += NT) {
              const int e0t = e0 + tid;
              const uint32_t F = e0t < nr ? runF[e0t] : 0u;
              const uint32_t e = e0t < nr ? runI[e0t] : 0u;
              uint32_t P = single ? (F | (F >> 1)) & 0x50505050u : F;  // one bit per pixel (at its bright flag) / per flag
              const int cnt = __popc(P);
              const int incl = wave_inclusive_scan_i32(cnt);
              const int wtotal = __builtin_amdgcn_readlane(incl, 63);
              if (wtotal) {
                int base = 0;
                if (lane == 0) base = lds_add_rtn(&s_cnt[1], wtotal);
                const int wbase = __builtin_amdgcn_readfirstlane(base);
                // a wave whose entries would run past the queue writes none of them (wave-uniform test, nothing per entry):
                // the total then exceeds the capacity and the cell is unpacked again below
                if (wbase + wtotal <= queue_cap) {
                  uint16_t *qp = queue + (wbase + incl - cnt);
                  const uint32_t ent0 = e << 3;  // e = the run's byte offset in the tile, a multiple of 4: dword index << 5
                  while (P) {
                    const uint32_t b = (uint32_t)__builtin_ctz(P);
                    P &= P - 1;
                    uint32_t ent = ent0 + b;  // bit 0 of b is the dark flag's bit of the pair
                    if (single) {
                      const uint32_t fb = (F >> b) & 3u;  // b = the bright flag's (even) position: bit 1 dark, bit 0 bright
                      ent += (fb >> 1) - ((fb + 1u) & 4u);  // both sides: dark first, bit 2 cleared = retry bright
                    }
                    *qp++ = (uint16_t)ent;
                  }
                }
              }
            }
          };
          unpack(false);
          __syncthreads();
          const bool single = s_cnt[1] > queue_cap;  // cell-uniform; only cells where most pixels pass on both sides
          if (VSG_COLD(single)) {
            __syncthreads();
            if (VSG_OPQ(tid) == 0) s_cnt[1] = 0;
            __syncthreads();
            unpack(true);
            __syncthreads();
          }
          nq = s_cnt[1];
          for (int i = tid; i < ((vh + 2) * kScoreP + 15) / 16; i += NT) ((uint4 *)score)[i] = make_uint4(0, 0, 0, 0);
          __syncthreads();
          // ---- phase 2: exact score of the queued pixels on their flagged side; entries that do not score are dropped
          // (retry-flagged ones -- single-entry cells only -- become bright-side entries for the second pass below)
          const int thr1 = max(thr, 1);
          // score rows have the tile's pitch, so pixel (r, c) of the cell -- tile byte (r + 3) * P + c + 3 + ox -- has its
          // score at (r + 1) * P + c + 1 = tile offset - soff0: one subtraction apart
          // fast_score_side_raw's bias, hidden from the compiler: it would fold "- bias + (bias + ring offset)" back into
          // signed offsets from the centre and spend an add on every negative one
          int kBias = 3 * kTileP + 1;
          asm volatile("" : "+s"(kBias));
          const uint8_t *tile_b = tile - kBias;
          const int soff0 = 2 * kTileP + 2 + ox;
          for (int q = tid; q < nq; q += NT) {
            const uint32_t ent = queue[q];
            // ent >> 3 = dword index * 4 + byte (b >> 3) inside its dword; + 4 for the second dword of the run (bit 1 of b)
            const uint32_t t0 = (ent >> 3) + ((ent << 1) & 4u);
            const int s = fast_score_side_raw<kTileP>(&tile_b[t0], ent & 1u);
            const uint32_t so = t0 - (uint32_t)soff0;
            // phase 3 reads the score offset; 0xFFFF = dropped; bit 14 = bright-side retry (single-entry cells only)
            if (s >= thr1) {  // thr1 = max(thr, 1): a corner at the threshold, and never a zero score
              score[so] = (uint8_t)s;
              queue[q] = (uint16_t)so;
            } else {
              queue[q] = (uint16_t)((ent & 4u) ? 0xFFFFu : (so | 0x4000u));
            }
          }
          __syncthreads();
          if (VSG_COLD(single)) {
            for (int q = tid; q < nq; q += NT) {
              const uint32_t ent = queue[q];
              if ((ent >> 14) != 1u) continue;  // bright side, retry flag: the dark side did not score
              const uint32_t so = ent & 0x3FFFu;
              const int s = fast_score_side<kTileP>(&tile_b[so + (uint32_t)soff0], thr, 0u);
              if (s)
                score[so] = (uint8_t)s, queue[q] = (uint16_t)so;
              else
                queue[q] = 0xFFFFu;
            }
            __syncthreads();
          }
          // ---- phase 3: non-max suppression inside the cell
          uint32_t *seg_out;
#if VSG_FAST_KA & 4
          {
            const FastKArgsPtr ka = fast_kargs();
            seg_out = ka->cand + ((size_t)frame * ka->cand_frame + C.cand_off);
          }
#else
          seg_out = cand + ((size_t)frame * cand_frame + C.cand_off);
#endif
          keep = 0;  // bit per loop iteration: queued pixel survives NMS
          int it = 0;
          int nbias = kScoreP + 1;  // the 3 x 3 neighbourhood at non-negative offsets from one register (see kBias)
          asm volatile("" : "+s"(nbias));
          const uint8_t *score_b = score - nbias;
          for (int q = tid; q < nq; q += NT, it++) {
            const uint32_t ent = queue[q];
            if (ent == 0xFFFFu) continue;
            const uint8_t *sp = &score_b[ent] + (kScoreP + 1);
            // nine reads back to back, one wait (see fast_score_side_raw)
            int s = sp[0], n0 = sp[-kScoreP - 1], n1 = sp[-kScoreP], n2 = sp[-kScoreP + 1], n3 = sp[-1], n4 = sp[1],
                n5 = sp[kScoreP - 1], n6 = sp[kScoreP], n7 = sp[kScoreP + 1];
            asm volatile("" : "+v"(s), "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6), "+v"(n7));
            const int mx = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
            if (s > mx) {
              if (VSG_HOT(seg)) {
                const int r1 = div_small((int)ent, 1.0f / kScoreP), c1 = (int)ent - r1 * kScoreP;  // row + 1, column + 1
                seg_out[atomicAdd(&s_cnt[0], 1)] = pack_cand(cell_x0 + c1 - 1 - kFastBorder, cell_y0 + r1 - 1 - kFastBorder, s);
              } else {
                keep |= 1u << it;
                atomicAdd(&s_cnt[0], 1);
              }
            }
          }
          __syncthreads();
          if (s_cnt[0] > 0 || pass == 1 || minTh >= thr) break;
          thr = minTh;  // vKeysCell.empty() -> retry with minThFAST (:848-851)
          __syncthreads();
          // the queue was built over the 6-bit tile: derive it again (whole rows: unstaged bytes come out as 6-bit values too)
          for (int i = tid; i < ((th + 1) * kTileP) / 4; i += NT)
            ((uint32_t *)qtile)[i] = (((const uint32_t *)tile)[i] >> 2) & 0x3F3F3F3Fu;
        }


      const int nEmit = s_cnt[0];
      if (VSG_HOT(seg)) {
        if (VSG_OPQ(tid) == 0) *my_count() = nEmit;
      } else if (nEmit) {  // levels of more than 4096 cells: one unordered list per level (rare geometries)
        const FastKArgsPtr ka = fast_kargs();
        const LevelGeom &L = ka->fg->lv[cell_level];
        if (tid == 0) s_cnt[3] = atomicAdd(&ka->cand_count[frame * kMaxLevels + cell_level], nEmit);
        __syncthreads();
        const int base = s_cnt[3];
        uint32_t *out = ka->cand + (size_t)frame * ka->cand_frame + L.cand_off;
        int it = 0;
        for (int q = tid; q < nq; q += NT, it++) {
          if (!(keep & (1u << it))) continue;
          const uint32_t ent = queue[q];
          const int r1 = div_small((int)ent, 1.0f / kScoreP), c1 = (int)ent - r1 * kScoreP;
          const int s = score[ent];
          const int slot = base + atomicAdd(&s_cnt[2], 1);
          if (slot < L.cand_cap) out[slot] = pack_cand(cell_x0 + c1 - 1 - kFastBorder, cell_y0 + r1 - 1 - kFastBorder, s);
        }
      }
    }
    __syncthreads();  // the cell is done with the tile, the score rows and the queue: the next one may be written
    C = N;
  }
}

// ------------------------------------------------------------------------------------------------
// Workgroup implementation of the octree Group concept.
enum { kSortStack = 24, kMaxWaves = 16, kSortFrontier = 128, kSortStackInts = 4 + 2 * kSortFrontier };  // kMaxWaves: wavefronts per workgroup the block scans support  // >= 2 * lg(n) + 1 pending ranges for n < 2048

struct BlockGroup {
  int tid, nthreads;
  int *wtot;  // LDS, one int per wave
  int *stk;   // LDS, kSortStackInts ints: the quicksort range stack (3 * kSortStack) / the range frontiers of sort_partition_phase
  __device__ void sync() { __syncthreads(); }
  __device__ int atomic_add(int *p, int v) { return atomicAdd(p, v); }
  __device__ void atomic_max(uint32_t *p, uint32_t v) { atomicMax(p, v); }
  __device__ void atomic_max64(uint64_t *p, uint64_t v) { atomicMax((unsigned long long *)p, (unsigned long long)v); }
  __device__ void atomic_min(int *p, int v) { atomicMin(p, v); }
  static __device__ __forceinline__ int wave_inclusive_scan(int v) { return wave_inclusive_scan_i32(v); }
  // p[0..4) += the workgroup's sums of four per-thread counts: wave sums, then at most four atomics per wave
  __device__ void add4(int *p, int c0, int c1, int c2, int c3) {
    const int s0 = __builtin_amdgcn_readlane(wave_inclusive_scan_i32(c0), 63);
    const int s1 = __builtin_amdgcn_readlane(wave_inclusive_scan_i32(c1), 63);
    const int s2 = __builtin_amdgcn_readlane(wave_inclusive_scan_i32(c2), 63);
    const int s3 = __builtin_amdgcn_readlane(wave_inclusive_scan_i32(c3), 63);
    if ((tid & 63) == 0) {
      if (s0) atomicAdd(&p[0], s0);
      if (s1) atomicAdd(&p[1], s1);
      if (s2) atomicAdd(&p[2], s2);
      if (s3) atomicAdd(&p[3], s3);
    }
  }
  __device__ int exclusive_scan(int *a, int n) {
    const int per = (n + nthreads - 1) / nthreads;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    int s = 0;
    for (int i = lo; i < hi; i++) s += a[i];
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    const int incl = wave_inclusive_scan(s);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < nwaves; w++) {
      const int t = wtot[w];
      if (w < wave) base += t;
      total += t;
    }
    int run = base + incl - s;
    for (int i = lo; i < hi; i++) {
      const int v = a[i];
      a[i] = run;
      run += v;
    }
    __syncthreads();
    return total;
  }
  // the same over two arrays at once (one pair of barriers); returns total of a, *total_b = total of b
  __device__ int exclusive_scan2(int *a, int *b, int n, int *total_b) {
    const int per = (n + nthreads - 1) / nthreads;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    int sa = 0, sb = 0;
    for (int i = lo; i < hi; i++) sa += a[i], sb += b[i];
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    const int ia = wave_inclusive_scan(sa), ib = wave_inclusive_scan(sb);
    if (lane == 63) wtot[wave] = ia, wtot[kMaxWaves + wave] = ib;
    __syncthreads();
    int basea = 0, ta = 0, baseb = 0, tb = 0;
    for (int w = 0; w < nwaves; w++) {
      const int x = wtot[w], y = wtot[kMaxWaves + w];
      if (w < wave) basea += x, baseb += y;
      ta += x, tb += y;
    }
    int ra = basea + ia - sa, rb = baseb + ib - sb;
    for (int i = lo; i < hi; i++) {
      const int va = a[i], vb = b[i];
      a[i] = ra, b[i] = rb;
      ra += va, rb += vb;
    }
    __syncthreads();
    *total_b = tb;
    return ta;
  }

  // one value pair per thread: exclusive prefixes over the workgroup, ONE barrier (the caller's next barrier frees wtot)
  __device__ int exclusive_scan2_one(int a, int b, int *ex_a, int *ex_b, int *total_b) {
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    const int ia = wave_inclusive_scan(a), ib = wave_inclusive_scan(b);
    if (lane == 63) wtot[wave] = ia, wtot[kMaxWaves + wave] = ib;
    __syncthreads();
    int basea = 0, ta = 0, baseb = 0, tb = 0;
    for (int w = 0; w < nwaves; w++) {
      const int x = wtot[w], y = wtot[kMaxWaves + w];
      if (w < wave) basea += x, baseb += y;
      ta += x, tb += y;
    }
    *ex_a = basea + ia - a, *ex_b = baseb + ib - b;
    *total_b = tb;
    return ta;
  }

  // introsort::partition_phase replayed by the 64 lanes of wave 0 with the SAME resulting array (bit for bit).
  // The serial loop `while (*first < pivot) ++first; --last; while (pivot < *last) --last; swap` visits, in the
  // ORIGINAL data of the range, the ascending positions A_0 < A_1 < ... of the elements >= pivot and the descending
  // positions B_0 > B_1 > ... of the elements <= pivot (then the pivot slot itself), swapping (A_k, B_k) while
  // A_k < B_k; it returns min(A_K, B_{K-1}) at the first K with A_K >= B_K.  The pairs are disjoint, so ranks from
  // wave ballots, one parallel round of swaps and a popcount reproduce a whole partition step.  Median selection
  // and the (rare) heapsort fallback stay on lane 0.  Other waves fall through; the caller's barrier orders them.
#ifndef VSG_OCT_RANK_UNROLL
#define VSG_OCT_RANK_UNROLL 6  // neighbour keys in flight at a time in the register form's stable ranks
#endif
  // n <= 64 (the careful phase of every level whose list holds at most 64 splittable nodes: all of C2's): the array lives
  // in the 64 lanes' REGISTERS.  Median selection reads three lanes (v_readlane, scalar compares), the A / B position
  // lists are ranks of two ballots, lane k learns (A_k, B_k) through two ds_permute, the partners through two more and
  // the items cross through two ds_bpermute: no LDS array, no fence, no dependent LDS round trips per step (the LDS form
  // below took 11.2 k cycles for 64 items on the one-frame path, profiles/r05_c_octree_stamps_one_frame.txt).
  // proc != nullptr: std::sort's final insertion pass as well -- a stable sort of what the partition phase leaves, i.e.
  // a rank among the 15 neighbours on either side (vsg_octree_core.h) -- on the lanes' registers (30 ds_bpermute of the
  // keys, all in flight together), written straight into the back-to-front processing order: the items never return to LDS
  __device__ void sort_partition_phase_regs(introsort::item_t *a, int n, uint16_t *proc = nullptr) {
    if (tid >= 64) return;
    const int lane = tid;
    introsort::item_t item = lane < n ? a[lane] : ~(introsort::item_t)0;
    const uint64_t below = (1ull << lane) - 1, above = lane == 63 ? 0ull : ~((2ull << lane) - 1);
    auto lane_item = [&](int p) {
      const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)item, p);
      const uint32_t hi32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(item >> 32), p);
      return ((introsort::item_t)hi32 << 32) | lo32;
    };
    int sp = 0, lo = 0, hi = n, depth = 2 * introsort::lg_(n > 0 ? n : 1);
    while (true) {
      while (hi - lo > 16) {
        if (depth == 0) {  // heapsort fallback (never seen on real lists): through LDS, on lane 0
          if (lane < n) a[lane] = item;
          __threadfence_block();
          if (lane == 0) introsort::heap_sort_(a + lo, hi - lo);
          __threadfence_block();
          item = lane < n ? a[lane] : ~(introsort::item_t)0;
          break;
        }
        --depth;
        // __move_median_to_first(first, first + 1, mid, last - 1): the median of three goes to position lo
        const int pa = lo + 1, pb = lo + (hi - lo) / 2, pc = hi - 1;
        const introsort::item_t ia = lane_item(pa), ib = lane_item(pb), ic = lane_item(pc), i0 = lane_item(lo);
        const uint32_t ka = (uint32_t)(ia >> 32), kb = (uint32_t)(ib >> 32), kc = (uint32_t)(ic >> 32);
        int pm;
        if (ka < kb)
          pm = kb < kc ? pb : ka < kc ? pc : pa;
        else
          pm = ka < kc ? pa : kb < kc ? pc : pb;
        const introsort::item_t im = pm == pa ? ia : pm == pb ? ib : ic;
        if (lane == lo) item = im;
        if (lane == pm) item = i0;
        const uint32_t P = (uint32_t)(im >> 32);
        const uint32_t key = (uint32_t)(item >> 32);
        // __unguarded_partition(first + 1, last, first): A = ascending positions of the elements >= pivot, B = descending
        // positions of the elements <= pivot, then the pivot slot itself (see the LDS form below)
        const bool in = lane > lo && lane < hi;
        const bool fa = in && key >= P, fb = (in && key <= P) || lane == lo;
        const uint64_t ba = __ballot(fa), bb = __ballot(fb);
        const int nA = __popcll(ba), nB = __popcll(bb);
        const int rA = __popcll(ba & below), rB = __popcll(bb & above);
        // lane k receives A_k and B_k (exactly one sender per k < nA / k < nB; other lanes' values are not used)
        // (ds_permute is a push and every lane pushes: the lanes outside the list take the slots behind it, in lane order,
        // so that the destinations are a permutation of the 64 lanes)
        const int Ak = __builtin_amdgcn_ds_permute((fa ? rA : nA + __popcll(~ba & below)) << 2, lane);
        const int Bk = __builtin_amdgcn_ds_permute((fb ? rB : nB + __popcll(~bb & below)) << 2, lane);
        const bool sw = lane < nA && lane < nB && Ak < Bk;
        const int K = __popcll(__ballot(sw));  // the pairs are monotone: valid exactly for k < K
        // the swapping lanes PULL their partners from the lane of their own rank, then the partner's item
        const int partA = __builtin_amdgcn_ds_bpermute(rA << 2, Bk);
        const int partB = __builtin_amdgcn_ds_bpermute(rB << 2, Ak);
        const bool isA = fa && rA < K, isB = fb && rB < K;
        const int partner = isA ? partA : isB ? partB : lane;
        const uint32_t xlo = (uint32_t)__builtin_amdgcn_ds_bpermute(partner << 2, (int)(uint32_t)item);
        const uint32_t xhi = (uint32_t)__builtin_amdgcn_ds_bpermute(partner << 2, (int)(uint32_t)(item >> 32));
        if (isA || isB) item = ((introsort::item_t)xhi << 32) | xlo;
        const int cutA = K < nA ? __builtin_amdgcn_readlane(Ak, K) : 0x7FFFFFFF;
        const int cutB = K >= 1 ? __builtin_amdgcn_readlane(Bk, K - 1) : hi;
        const int cut = cutA < cutB ? cutA : cutB;
        if (lane == 0) stk[3 * sp] = lo, stk[3 * sp + 1] = cut, stk[3 * sp + 2] = depth;  // uniform values
        sp++;
        lo = cut;
      }
      if (sp == 0) break;
      --sp;
      __threadfence_block();
      lo = stk[3 * sp], hi = stk[3 * sp + 1], depth = stk[3 * sp + 2];
    }
    if (proc) {
      const uint32_t key = (uint32_t)(item >> 32);
      const int lo_t = lane > 15 ? lane - 15 : 0, hi_t = lane + 15 < n - 1 ? lane + 15 : n - 1;
      int rank = lo_t;
      // (six neighbours in flight at a time: all thirty at once cost the whole kernel ~ 150 spilled registers)
#pragma unroll VSG_OCT_RANK_UNROLL
      for (int i = 0; i < 30; i++) {
        const int d = i < 15 ? i - 15 : i - 14;
        const int j = lane + d;
        const uint32_t kj = (uint32_t)__builtin_amdgcn_ds_bpermute((j & 63) << 2, (int)key);
        rank += (j >= lo_t) & (j <= hi_t) & ((kj < key) | ((kj == key) & (d < 0)));
      }
      if (lane < n) proc[n - 1 - rank] = (uint16_t)(uint32_t)item;
      return;
    }
    if (lane < n) a[lane] = item;
    __threadfence_block();
  }
  __device__ bool sort_to_proc(introsort::item_t *a, int n, uint16_t *proc) {
#ifndef VSG_OCT_NO_REGSORT
    if (n <= 64) {
      sort_partition_phase_regs(a, n, proc);
      return true;
    }
#endif
    return false;
  }
  __device__ void sort_partition_phase(introsort::item_t *a, int n, uint16_t *posA, uint16_t *posB) {
#ifndef VSG_OCT_NO_REGSORT  // A/B builds: the LDS form for every n
    if (n <= 64) {
      sort_partition_phase_regs(a, n);
      return;
    }
#endif
#ifndef VSG_OCT_SERIAL_SORT
    // More than 64 items (1200+ features per frame, photographs, 1280x720): the ranges the quicksort loop leaves behind are
    // disjoint and each is partitioned from its own data alone, so the ORDER in which introsort works them off does not
    // change the result -- all ranges of one recursion depth are partitioned at once, one range per wave: as many rounds as
    // the recursion is deep (5-9 on real node lists) instead of one step per range (n / 12: 47 k cycles for the ~150 nodes of a 1250-feature
    // frame's level 0, profiles/r06_i_*).  A round's ranges lie in a frontier list (lo | hi << 12 | depth << 24), the children
    // go to the other list; three counters rotate (read / filled / zeroed) so that a round needs ONE barrier.
    if (n < 4096 && n <= 17 * (kSortFrontier - 1) && nthreads >= 128) {
      const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
      const uint64_t lt = (1ull << lane) - 1;
      uint32_t *F0 = (uint32_t *)(stk + 4), *F1 = F0 + kSortFrontier;
      if (tid == 0) {
        F0[0] = 0u | ((uint32_t)n << 12) | ((uint32_t)(2 * introsort::lg_(n)) << 24);
        stk[0] = 1, stk[1] = 0, stk[2] = 0;
      }
      __syncthreads();
      for (int round = 0;; round++) {
        const int ic = round % 3, in = (round + 1) % 3, iz = (round + 2) % 3;
        const int R = stk[ic];
        if (R == 0) break;  // workgroup-uniform
        const uint32_t *Fc = (round & 1) ? F1 : F0;
        uint32_t *Fn = (round & 1) ? F0 : F1;
        if (tid == 0) stk[iz] = 0;  // the counter the round after this one fills (last read a round ago)
        for (int r = wave; r < R; r += nwaves) {
          const uint32_t w = Fc[r];
          const int lo = (int)(w & 0xFFFu), hi = (int)((w >> 12) & 0xFFFu);
          int depth = (int)(w >> 24);
          if (depth == 0) {
            if (lane == 0) introsort::heap_sort_(a + lo, hi - lo);
            continue;
          }
          --depth;
          // A step is a chain of dependent LDS round trips (~300 cycles each for one wave of a nearly idle CU), so it is written
          // to need few: ONE round of reads brings the four items the median rule looks at and the lane's items of the first
          // chunk from either end; every lane evaluates __move_median_to_first (stl_algo.h) itself and patches what it holds,
          // lane 0 writes the swap; cut and children come out of registers where the lists fit one chunk.
          const int s = lo + 1, e = hi, pmid = lo + (hi - lo) / 2;
          const introsort::item_t i0 = a[lo], iA = a[lo + 1], iB = a[pmid], iC = a[hi - 1];
          const int qa0 = s + lane, qb0 = e - 1 - lane;
          introsort::item_t xa = a[qa0 < e ? qa0 : e - 1], xb = a[qb0 >= s ? qb0 : s];
          const uint32_t ka = (uint32_t)(iA >> 32), kb = (uint32_t)(iB >> 32), kc = (uint32_t)(iC >> 32);
          int pm;
          if (ka < kb)
            pm = kb < kc ? pmid : ka < kc ? hi - 1 : lo + 1;
          else
            pm = ka < kc ? lo + 1 : kb < kc ? hi - 1 : pmid;
          const introsort::item_t im = pm == lo + 1 ? iA : pm == pmid ? iB : iC;
          if (lane == 0) a[lo] = im, a[pm] = i0;
          __threadfence_block();
          const uint32_t P = (uint32_t)(im >> 32);
          uint16_t *pA = posA + lo, *pB = posB + lo;  // <= hi - lo - 1 / hi - lo entries: inside the range's own slots
          int nA = 0, nB = 0;
          for (int c = 0; s + c < e; c += 64) {
            const int ia = s + c + lane, ib = e - 1 - c - lane;
            if (c) xa = a[ia < e ? ia : e - 1], xb = a[ib >= s ? ib : s];  // (behind lane 0's swap: LDS keeps a wave's order)
            if (ia == pm) xa = i0;
            if (ib == pm) xb = i0;
            const bool fa = ia < e && (uint32_t)(xa >> 32) >= P;
            const bool fb = ib >= s && (uint32_t)(xb >> 32) <= P;
            const uint64_t ba = __ballot(fa), bb = __ballot(fb);
            if (fa) pA[nA + __popcll(ba & lt)] = (uint16_t)ia;
            if (fb) pB[nB + __popcll(bb & lt)] = (uint16_t)ib;
            nA += __popcll(ba);
            nB += __popcll(bb);
          }
          if (lane == 0) pB[nB] = (uint16_t)lo;  // the pivot slot stops the downward scan
          nB++;
          __threadfence_block();
          const int np = nA < nB ? nA : nB;
          int K = 0, cut;
          if (nA < 64 && nB <= 64) {  // both lists in one chunk: the cut comes out of the lanes' registers
            const int vA = lane < nA ? (int)pA[lane] : 0x7FFFFFFF, vB = lane < nB ? (int)pB[lane] : -1;
            const bool sw = lane < np && vA < vB;
            K = __popcll(__ballot(sw));
            if (sw) {
              const introsort::item_t t = a[vA];
              a[vA] = a[vB];
              a[vB] = t;
            }
            const int cutA = __builtin_amdgcn_readlane(vA, K);  // K <= np <= nA < 64; lanes >= nA hold "none"
            const int cutB = K >= 1 ? __builtin_amdgcn_readlane(vB, K - 1) : e;
            cut = cutA < cutB ? cutA : cutB;
          } else {
            for (int c = 0; c < np; c += 64) {
              const int j = c + lane;
              const bool sw = j < np && pA[j] < pB[j];
              const uint64_t bs = __ballot(sw);
              if (sw) {
                const int ia = pA[j], ib = pB[j];
                const introsort::item_t t = a[ia];
                a[ia] = a[ib];
                a[ib] = t;
              }
              K += __popcll(bs);
            }
            const int cutA = K < nA ? (int)pA[K] : 0x7FFFFFFF, cutB = K >= 1 ? (int)pB[K - 1] : e;
            cut = cutA < cutB ? cutA : cutB;
          }
          if (lane == 0) {
            const int cl = cut - lo > 16, cr = hi - cut > 16;
            if (cl + cr) {
              const int at = atomicAdd(&stk[in], cl + cr);
              if (cl) Fn[at] = (uint32_t)lo | ((uint32_t)cut << 12) | ((uint32_t)depth << 24);
              if (cr) Fn[at + cl] = (uint32_t)cut | ((uint32_t)hi << 12) | ((uint32_t)depth << 24);
            }
          }
        }
        __syncthreads();
      }
      return;
    }
#endif
    if (tid >= 64) return;
    const int lane = tid;
    const uint64_t lt = (1ull << lane) - 1;
    int sp = 0, lo = 0, hi = n, depth = 2 * introsort::lg_(n > 0 ? n : 1);
    while (true) {
      while (hi - lo > 16) {
        if (depth == 0) {
          if (lane == 0) introsort::heap_sort_(a + lo, hi - lo);
          __threadfence_block();
          break;
        }
        --depth;
        if (lane == 0) introsort::move_median_to_first_(a + lo, a + lo + 1, a + lo + (hi - lo) / 2, a + hi - 1);
        __threadfence_block();
        const uint32_t P = (uint32_t)(a[lo] >> 32);
        const int s = lo + 1, e = hi;
        int nA = 0, nB = 0;
        for (int c = 0; s + c < e; c += 64) {
          const int ia = s + c + lane, ib = e - 1 - c - lane;
          const bool fa = ia < e && (uint32_t)(a[ia] >> 32) >= P;
          const bool fb = ib >= s && (uint32_t)(a[ib] >> 32) <= P;
          const uint64_t ba = __ballot(fa), bb = __ballot(fb);
          if (fa) posA[nA + __popcll(ba & lt)] = (uint16_t)ia;
          if (fb) posB[nB + __popcll(bb & lt)] = (uint16_t)ib;
          nA += __popcll(ba);
          nB += __popcll(bb);
        }
        if (lane == 0) posB[nB] = (uint16_t)lo;  // the pivot slot stops the downward scan
        nB++;
        __threadfence_block();
        const int np = nA < nB ? nA : nB;
        int K = 0;
        for (int c = 0; c < np; c += 64) {
          const int j = c + lane;
          const bool sw = j < np && posA[j] < posB[j];
          const uint64_t bs = __ballot(sw);
          if (sw) {
            const int ia = posA[j], ib = posB[j];
            const introsort::item_t t = a[ia];
            a[ia] = a[ib];
            a[ib] = t;
          }
          K += __popcll(bs);
        }
        const int cutA = K < nA ? (int)posA[K] : 0x7FFFFFFF, cutB = K >= 1 ? (int)posB[K - 1] : e;
        const int cut = cutA < cutB ? cutA : cutB;
        __threadfence_block();
        if (lane == 0) stk[3 * sp] = lo, stk[3 * sp + 1] = cut, stk[3 * sp + 2] = depth;  // uniform values
        sp++;
        lo = cut;
      }
      if (sp == 0) break;
      --sp;
      __threadfence_block();
      lo = stk[3 * sp], hi = stk[3 * sp + 1], depth = stk[3 * sp + 2];
    }
  }
};

#ifndef VSG_OCT_K
#define VSG_OCT_K 8
#endif
constexpr int kOctRegPts = VSG_OCT_K;  // candidates per thread held in registers (x 256 threads = 2048 per level; 12 costs a wave of occupancy and loses)

#ifndef VSG_OCT_NT
#define VSG_OCT_NT 256
#endif
constexpr int kOctThreads = VSG_OCT_NT;  // threads per (frame, level) octree

// 5 waves per SIMD = 5 octrees per CU (96 VGPRs, 10 dwords spilled off the hot paths): the kernel is latency-bound,
// so residency is worth more than registers, and 96-register waves leave room for the blur waves that run beside it
// (4 waves / 127 VGPRs: 0.167 ms per 512 frames and 256 k frames/s; 5: 0.160 ms and 260 k; 6: 255 k; 8: 0.185 ms)
// Candidates of a level whose cells wrote their survivors into fixed segments.  LDS holds the exclusive prefix of the
// cell counts and the segment offsets; candidate p lives in the cell found by bisection of the prefix, at segment
// offset + (p - prefix).
struct SegSrc {
  const uint32_t *level_cand;  // the level's slice of the frame's candidate array
  const int *prefix;           // LDS [ncells + 1]
  const int *segoff;           // LDS [ncells]
  int ncells;
  __device__ int cell_of(int p) const {
    int lo = 0, hi = ncells;  // invariant: prefix[lo] <= p < prefix[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (prefix[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
  }
  __device__ uint32_t operator()(int p) const {
    const int ci = cell_of(p);
    return level_cand[segoff[ci] + (p - prefix[ci])];
  }
};
// The octree's register-resident point set (octree::RegPts) fed from the segments: a thread takes a CONTIGUOUS run of
// candidate indices (which thread holds which point is irrelevant to the algorithm), so it bisects once and then
// walks along the cells; its K loads are independent and in flight together.
template <int K>
struct SegRegPts {
  SegSrc src;
  uint32_t c[K];
  int n[K];
  int cnt;
  template <class G>
  __device__ void load(G &g, int npts) {
    const int per = (npts + g.nthreads - 1) / g.nthreads;  // <= K by the caller's choice of this form
    const int p0 = min(g.tid * per, npts);
    cnt = min(per, npts - p0);
    int ci = cnt > 0 ? src.cell_of(p0) : 0;
    int next = src.prefix[ci + 1];
    int off = src.segoff[ci] - src.prefix[ci];  // candidate p of this cell sits at level_cand[off + p]
#pragma unroll
    for (int k = 0; k < K; k++) {
      n[k] = 0;
      c[k] = 0u;
      if (k < cnt) {
        const int p = p0 + k;
        while (p >= next) {  // next non-empty cell
          ci++;
          next = src.prefix[ci + 1];
          off = src.segoff[ci] - src.prefix[ci];
        }
        c[k] = src.level_cand[off + p];
      }
    }
  }
  template <class G, class F>
  __device__ void for_each(G &, int, F f) {
#pragma unroll
    for (int k = 0; k < K; k++)
      if (k < cnt) f(c[k], n[k]);
  }
  template <class G, class F>
  __device__ void init_each(G &g, int npts, F f) {
    for_each(g, npts, f);
  }
};

#ifndef VSG_OCT_FEW
#define VSG_OCT_FEW 8
#endif
constexpr int kOctFewFrames = VSG_OCT_FEW;  // calls of up to this many frames take the 4-waves-per-SIMD octree launch (k_octree_few; the fused launch is compiled for 4 waves anyway)
struct OctArgs {
  const FrameGeom *fg;
  const uint32_t *cand;
  const int *cand_count;
  const CellDesc *cells;
  const int *cell_count;
  uint32_t *cand2;
  uint16_t *node_of;
  uint32_t *sel;
  int *sel_count;
  int cap, prefix_off;
  int lab_off, lab_cap;  // node labels of levels that overflow the registers: u16[lab_cap] behind the workspace (0: none)
  int hist_big;          // the workspace class (vsg_octree_core.h work_bytes)
};

// kMemU: points per batch of the memory-form sweeps (octree::MemPtsT)
template <int kMemU>
__device__ __forceinline__ void octree_block(const OctArgs &a, int level, int frame, uint8_t *oct_lds, int *wtot,
                                             int *sort_stack) {
  const FrameGeom *__restrict__ fg = a.fg;
  const uint32_t *__restrict__ cand = a.cand;
  const int *__restrict__ cand_count = a.cand_count;
  const CellDesc *__restrict__ cells = a.cells;
  const int *__restrict__ cell_count = a.cell_count;
  uint32_t *__restrict__ cand2 = a.cand2;
  uint16_t *__restrict__ node_of = a.node_of;
  uint32_t *__restrict__ sel = a.sel;
  int *__restrict__ sel_count = a.sel_count;
  const int cap = a.cap, prefix_off = a.prefix_off;
  const LevelGeom &L = fg->lv[level];
  octree::Params P;
  P.N = L.quota;
  P.height = L.oct_height;
  P.nIni = L.nIni;
  P.iniUL = L.iniUL;
  P.iniThresh = L.iniThresh;
  P.nCols = L.nCols;
  P.wCell = L.wCell;
  P.hCell = L.hCell;
  octree::Work W;
  octree::carve(W, oct_lds, cap, a.hist_big != 0);
  BlockGroup g;
  g.tid = threadIdx.x;
  g.nthreads = blockDim.x;
  g.wtot = wtot;
  g.stk = sort_stack;
  const size_t coff = (size_t)frame * fg->cand_frame + L.cand_off;
  uint32_t *out = sel + (size_t)frame * fg->sel_frame + L.sel_off;
  int n;
  if (fg->cand_segmented) {
    // per-cell segments: exclusive prefix of this level's cell counts in LDS (behind the octree workspace)
    const int ncells = L.nCols * L.nRows;
    int *prefix = (int *)(oct_lds + prefix_off), *segoff = prefix + ncells + 1;
    const int *cc = cell_count + (size_t)frame * fg->total_cells + L.cell_base;
    for (int i = g.tid; i < ncells; i += g.nthreads) {
      prefix[i] = cc[i];
      segoff[i] = cells[L.cell_base + i].cand_off;
    }
    __syncthreads();
    const int npts = g.exclusive_scan(prefix, ncells);
    if (g.tid == 0) prefix[ncells] = npts;
    __syncthreads();
    const SegSrc src = {cand + coff, prefix, segoff, ncells};
    if (npts <= kOctRegPts * kOctThreads) {
      SegRegPts<kOctRegPts> pts;
      pts.src = src;
      n = octree::distribute_pts(g, P, pts, npts, W, out);
    } else {
      // too many for the registers (a real photograph's level 0 holds 3-6 k candidates at 640x480): compact once into the
      // scratch list, then the memory-resident form -- with the node labels in LDS when they fit behind the workspace
      // (round 6: the launch pads a workgroup's LDS up to the share four workgroups per CU leave it anyway; the labels are
      // what every pass reads AND rewrites, the candidate words are read-only and stream through L2)
      uint32_t *list = cand2 + coff;
      {
        // a thread copies a CONTIGUOUS run of the concatenated list: one bisection for its first point, then it walks along
        // the cell boundaries (a bisection per point -- eight dependent LDS reads each -- was 36 k of the 165 k cycles a
        // photograph's level 0 took)
        const int chunk = (npts + g.nthreads - 1) / g.nthreads;
        const int p0 = g.tid * chunk, p1 = min(p0 + chunk, npts);
        if (p0 < p1) {
          // U points at a time: their source positions from the LDS walk, then U loads in flight, then U stores (one point per
          // trip waited ~1.5 k cycles for its own load: 116 k cycles for a 1280x720 photograph's level 0, profiles/r06_u_*)
          constexpr int U = 8;
          int ci = src.cell_of(p0);
          int p = p0;
          for (; p + U <= p1; p += U) {
            int so[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
              while (p + u >= src.prefix[ci + 1]) ci++;
              so[u] = src.segoff[ci] + (p + u - src.prefix[ci]);
            }
            uint32_t v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = src.level_cand[so[u]];
#pragma unroll
            for (int u = 0; u < U; u++) list[p + u] = v[u];
          }
          for (; p < p1; p++) {
            while (p >= src.prefix[ci + 1]) ci++;
            list[p] = src.level_cand[src.segoff[ci] + (p - src.prefix[ci])];
          }
        }
      }
      __threadfence_block();
      __syncthreads();
      if (npts <= a.lab_cap)  // (two call sites: the compiler has to SEE that the labels are LDS to emit ds_ instead of flat_ accesses)
        n = octree::distribute<kMemU>(g, P, list, npts, (uint16_t *)(oct_lds + a.lab_off), W, out);
      else
        n = octree::distribute<kMemU>(g, P, list, npts, node_of + coff, W, out);
    }
  } else {
    int npts = cand_count[frame * kMaxLevels + level];
    if (npts > L.cand_cap) npts = L.cand_cap;
    // candidates stay in registers across the passes when they fit (the common case); node_of[] is only touched by
    // the fallback
    n = npts <= kOctRegPts * kOctThreads ? octree::distribute_reg<kOctRegPts>(g, P, cand + coff, npts, W, out)
                                         : octree::distribute<kMemU>(g, P, cand + coff, npts, node_of + coff, W, out);
  }
  if (threadIdx.x == 0) sel_count[frame * kMaxLevels + level] = n;
#if VSG_OD_SPATIAL
  // Experiment (OFF by default: profiles/r06_e_orient_desc_order_ab.txt): the PROCESSING order of the level's
  // keypoints for k_orient_desc -- Morton order of their 32 x 32 px cells, so that the twelve keypoints of one of its workgroups
  // are spatial neighbours -- as a permutation in the level's slice of the (by now idle) node_of scratch; the OUTPUT slot of a
  // keypoint stays its position in the octree's list.  Rank by counting: n <= sel_cap keys, n^2 / 256 comparisons per thread.
  {
    __syncthreads();  // out[0..n) is written (same workgroup: the barrier orders it)
    auto key_of = [&](int i) -> uint32_t {
      const uint32_t c = out[i];
      uint32_t x = VSG_CAND_X(c) >> 5, y = VSG_CAND_Y(c) >> 5, m = 0;
#pragma unroll
      for (int b = 0; b < 6; b++) m |= ((x >> b) & 1u) << (2 * b) | ((y >> b) & 1u) << (2 * b + 1);
      return (m << 16) | (uint32_t)i;  // unique: ties by list position
    };
    uint16_t *ord = node_of + coff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const uint32_t ki = key_of(i);
      int r = 0;
      for (int j = 0; j < n; j++) r += key_of(j) < ki;
      ord[r] = (uint16_t)i;
    }
  }
#endif
}

#ifndef VSG_OCT_WAVES
#define VSG_OCT_WAVES 5
#endif
// Points per batch of the memory-form sweeps (octree::MemPtsT), per kernel: the stand-alone launch (1280x720 and wider: every
// level 0 is memory-resident) and the few-frame launch of geometries too large for the fused one take 8; the launch that also
// carries the blur keeps 1 -- their register
// allocation is what the default content's register-form levels run on (C2 / 1024, rectangles: octree stage 0.2037 -> 0.2105 ms with
// 4, unchanged with 1; C4 / 256: 0.318 -> 0.252 with 1, 0.205 with 8; profiles/r06_u_*)
#ifndef VSG_OCT_MEM_BATCH
#define VSG_OCT_MEM_BATCH 8
#endif
#ifndef VSG_OCT_MEM_BATCH_FUSED
#define VSG_OCT_MEM_BATCH_FUSED 1
#endif

__global__ __launch_bounds__(kOctThreads) __attribute__((amdgpu_waves_per_eu(VSG_OCT_WAVES, VSG_OCT_WAVES))) void k_octree(OctArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t oct_lds[];
  __shared__ int wtot[2 * kMaxWaves];
  __shared__ int sort_stack[kSortStackInts];
  octree_block<VSG_OCT_MEM_BATCH>(a, blockIdx.x, blockIdx.y, oct_lds, wtot, sort_stack);
}
// calls of a few frames: registers instead of residency
template <int kMemU>
__global__ __launch_bounds__(kOctThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_octree_few(OctArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t oct_lds[];
  __shared__ int wtot[2 * kMaxWaves];
  __shared__ int sort_stack[kSortStackInts];
  octree_block<kMemU>(a, blockIdx.x, blockIdx.y, oct_lds, wtot, sort_stack);
}

// Test hook (vsg_debug_device_sort): the octree's std::sort replay -- wave-parallel partition phase + stable rank --
// on arbitrary items, so that tests can compare it with the real std::sort directly.
__global__ __launch_bounds__(256) void k_debug_sort(uint64_t *items, int n) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dbg_lds[];
  __shared__ int wtot[2 * kMaxWaves];
  __shared__ int sort_stack[kSortStackInts];
  introsort::item_t *buf = (introsort::item_t *)dbg_lds;
  uint16_t *posA = (uint16_t *)(buf + n), *posB = posA + n + 2;
  BlockGroup g;
  g.tid = threadIdx.x;
  g.nthreads = blockDim.x;
  g.wtot = wtot;
  g.stk = sort_stack;
  for (int i = g.tid; i < n; i += g.nthreads) buf[i] = items[i];
  __syncthreads();
  {
    // the form the octree takes for <= 64 nodes: partition phase AND stable ranks on wave 0's registers, the processing order
    // (back to front) as the result.  Run on a copy whose payload is the item's index, then gathered back.
    introsort::item_t *tmp = (introsort::item_t *)(((uintptr_t)(posB + n + 2) + 7) & ~(uintptr_t)7);
    for (int i = g.tid; i < n; i += g.nthreads) tmp[i] = (buf[i] & 0xFFFFFFFF00000000ull) | (uint32_t)i;
    __syncthreads();
    if (g.sort_to_proc(tmp, n, posA)) {
      __syncthreads();
      for (int t = g.tid; t < n; t += g.nthreads) items[n - 1 - t] = buf[posA[t]];
      return;
    }
  }
  g.sort_partition_phase(buf, n, posA, posB);
  __syncthreads();
  for (int t = g.tid; t < n; t += g.nthreads) {
    const introsort::item_t it = buf[t];
    const uint32_t key = (uint32_t)(it >> 32);
    const int lo = t > 15 ? t - 15 : 0, hi = t + 15 < n - 1 ? t + 15 : n - 1;
    int rank = lo;
    for (int j = lo; j <= hi; j++) {
      const uint32_t kj = (uint32_t)(buf[j] >> 32);
      rank += (kj < key) | ((kj == key) & (j < t));
    }
    items[rank] = it;
  }
}

// ------------------------------------------------------------------------------------------------
// 7x7 Gaussian, sigma 2, 8.8 fixed point, BORDER_REFLECT_101 at the level's own edges (the reference blurs a
// border-less clone, ORBextractor.cc:1129-1130).  No LDS: a thread owns 4 adjacent columns and walks down
// kBlurStrip + 6 rows with a 7-row register window.  Per row it loads 3 aligned dwords (12 px), forms the 4
// horizontal 8.8 sums with v_alignbyte + v_dot4_u32_u8, and once the window is full emits 4 output bytes
// (one 32-bit store) from 7 v_mad_u32_u24 per pixel.  Adjacent lanes own adjacent column groups, so every
// wave-level load/store is one contiguous 256-byte row segment.
typedef uint32_t u32_unaligned __attribute__((aligned(1)));  // gfx950 global dword loads need no alignment
// The level-0 pointer comes out of a by-value kernel-argument struct (Src0), which the compiler types as a FLAT
// pointer; an align-1 flat load is split into four byte loads before address-space inference turns it global.
// Loads through this type say "global" up front and stay one dword instruction.
typedef __attribute__((address_space(1))) const u32_unaligned u32_global_unaligned;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x2 u32x2_unaligned __attribute__((aligned(1)));
typedef __attribute__((address_space(1))) const u32x2_unaligned u64_global_unaligned;
typedef u32x2 u32x2_4aligned __attribute__((aligned(4)));
typedef __attribute__((address_space(1))) const u32x2_4aligned u64_global_4aligned;
typedef __attribute__((address_space(1))) const uint32_t u32_global;

__device__ __forceinline__ int reflect101(int p, int len) {
  while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

// rows whose loads a blur thread keeps in flight ahead of the row it works on: 12 inside the octree's launch (compiled for 4
// waves per SIMD = 128 VGPRs; rounds 3-4: 7 rows at 5 waves / 96 VGPRs), 10 in the blur's own launch: with the round-5
// arithmetic (shifted tap words, clamp bit, two-offset row addresses) that is 92 VGPRs = 5 waves per SIMD, 0.216 ms per
// 512 C2 frames against 0.220-0.226 with 14 rows ahead (104 VGPRs, 4 waves) and 0.253 for the round-4 arithmetic (in-run
// A/B, profiles/r05_p_*)
#ifndef VSG_BLUR_AHEAD
#define VSG_BLUR_AHEAD 12
#endif
#ifndef VSG_BLUR_AHEAD_ALONE
#define VSG_BLUR_AHEAD_ALONE 10
#endif
template <int kBlurAhead>
__device__ __forceinline__ void blur_block(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur,
                                           const FrameGeom *__restrict__ fg, const Src0 &s0, BlockXY blk) {
  int level = 0;
  while (level + 1 < fg->nlevels && blk.x >= fg->lv[level + 1].blur_block_base) level++;
  const LevelGeom &L = fg->lv[level];
  const int t = (blk.x - L.blur_block_base) * 256 + (int)threadIdx.x;
  if (t >= L.blur_nxg * L.blur_nys) return;
  // Four adjacent lanes own the four column groups of one 16-pixel TILE column (the output is tiled: 16 x 4 pixels per
  // 64-byte line), so a quad's stores of one row are 16 contiguous bytes and four rows fill the line.  The threads of
  // the interior tile columns (all four groups read 12 existing bytes) come first, those of the edge tile columns --
  // column 0, the last one or two, with the REFLECT_101 selectors and, past the image, padding -- last, so that whole
  // wavefronts take either the fast or the slow load path instead of every wave diverging at a row end.
  const int nI = 4 * L.blur_int_tc, nInt = nI * L.blur_nys;
  int gx, sy;
  if (t < nInt) {
    sy = t / nI;
    gx = 4 + (t - sy * nI);
  } else {
    const int nE = L.blur_nxg - nI, e = t - nInt;
    sy = e / nE;
    const int k = e - sy * nE;
    gx = k < 4 ? k : nI + k;
  }
  const int x0 = gx * 4, y0 = sy * kBlurStrip;
  const int w = L.w, h = L.h;
  int spitch;
  const uint8_t *img = level_ptr(fg, s0, pyr, blk.y, level, spitch);
  // Stores: four output rows of a quad are a 4 x 4 matrix of dwords (row, column group) = one tile.  It is transposed
  // inside the quad (two rounds of DPP quad permutes + selects) so that lane q holds ROW q of the tile -- 16 contiguous
  // bytes -- and the quad writes the whole 64-byte line with one dwordx4 store per lane: a wave's store is 16 full lines.
  // (Storing each row's dword where it belongs -- 16-byte pieces of 16 different lines per instruction, 4 times the
  // write requests -- cost the blur 0.267 -> 0.323 ms per 512 C2 frames: profiles/r04_d_*.)
  const int q = gx & 3;
  uint8_t *dst = blur + (size_t)blk.y * fg->blur_frame_bytes + L.boff + (uint32_t)blur_tiled_offset(x0 & ~15, y0, L.btx) +
                 q * kBlurTileW;
  const uint32_t tile_row_bytes = (uint32_t)L.btx * kBlurTileBytes;
  const bool q_odd = q & 1, q_hi = q & 2;
  uint32_t trow[4] = {0, 0, 0, 0};  // the thread's dwords of the four rows of the current tile
  uint32_t k[7];
#pragma unroll
  for (int j = 0; j < 7; j++) k[j] = fg->taps[j];
  // Horizontal pass without byte alignment: the 12 source bytes stay in their three dwords {e0, e1, e2} and the TAPS are
  // shifted instead -- pixel j's window is bytes j+1 .. j+7, so its seven taps sit at byte offsets that differ per pixel:
  // ten v_dot4_u32_u8 against ten scalar tap words (2 + 3 + 3 + 2) instead of eight and six v_alignbyte_b32.
  auto tw = [&](int a, int b, int c, int d) {  // tap word: tap index per byte, -1 = no tap
    return (a < 0 ? 0u : k[a]) | (b < 0 ? 0u : k[b] << 8) | (c < 0 ? 0u : k[c] << 16) | (d < 0 ? 0u : k[d] << 24);
  };
  const uint32_t TA0 = tw(-1, 0, 1, 2), TA1 = tw(3, 4, 5, 6);                          // pixel 0: e0, e1
  const uint32_t TB0 = tw(-1, -1, 0, 1), TB1 = tw(2, 3, 4, 5), TB2 = tw(6, -1, -1, -1);  // pixel 1: e0, e1, e2
  const uint32_t TC0 = tw(-1, -1, -1, 0), TC1 = tw(1, 2, 3, 4), TC2 = tw(5, 6, -1, -1);  // pixel 2
  const uint32_t TD1 = tw(0, 1, 2, 3), TD2 = tw(4, 5, 6, -1);                          // pixel 3: e1, e2
  const bool interior = x0 >= 4 && x0 + 8 <= w;  // all 12 source bytes exist: aligned dword loads
  // column groups past the image (the padding of the last tile column) compute on clamped addresses and store garbage
  // into bytes nothing reads
  // Edge groups (the first and the last two of a row) also fetch 12 contiguous bytes per row -- columns 0..11 on
  // the left, w-12..w-1 on the right (w >= 16, checked with the geometry) -- which hold every REFLECT_101 source
  // column they need; the 12 wanted bytes are then picked with v_perm selectors that are computed once per thread.
  const int base = interior ? x0 - 4 : (x0 == 0 ? 0 : w - 12);
  const int xr = x0 < w ? x0 : w - 4;  // reflected columns of a padding group: anything inside the row
  uint32_t selLo[3] = {0, 0, 0}, selHi[3] = {0, 0, 0};
  if (!interior) {
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const uint32_t sidx = (uint32_t)(reflect101(xr - 4 + 4 * q + b, w) - base);  // 0..11
        selLo[q] |= (sidx < 8 ? sidx : 0x0Cu) << (8 * b);
        selHi[q] |= (sidx >= 8 ? sidx - 8 : 0x0Cu) << (8 * b);
      }
  }
  // Vertical pass on row PAIRS: a row's horizontal sums fit 16 bits (taps sum <= 257), so row r is kept as
  // P[r] = H[r-1] | H[r] << 16 and an output row is three v_dot2_u32_u16 (taps 0-1, 2-3, 4-5 against P[r-5], P[r-3],
  // P[r-1]) plus one multiply-add for tap 6 on the newest row: 4 + 1 (the pack) operations per pixel instead of 7.
  uint32_t pw[7][4], hprev[4] = {0, 0, 0, 0};
  // The vertical sums are formed 256 times too large (taps << 8, rounding constant 32768 << 8): the output byte is byte 3
  // of the sum, and saturate_cast<uchar> -- a sum >= 2^24, which only a tap table that sums to 257 reaches, on 255-valued
  // areas -- is the CLAMP bit of the v_dot2_u32_u16 (an unsigned overflow saturates to 0xFFFFFFFF, byte 3 = 255) instead
  // of a v_min_u32 per pixel.  Taps are <= 255 and sum to <= 257 (vsg_orb_set_blur_taps), so the multiply-add of tap 6
  // cannot overflow by itself: 65 280 x 65 535 + 2^23 < 2^32.
  const u16x2 K01 = {(unsigned short)(k[0] << 8), (unsigned short)(k[1] << 8)},
              K23 = {(unsigned short)(k[2] << 8), (unsigned short)(k[3] << 8)},
              K45 = {(unsigned short)(k[4] << 8), (unsigned short)(k[5] << 8)};
  // tap 6 in a vector register and the rounding constant in a scalar one: v_mad_u32_u24 reads one scalar operand, and
  // left alone the compiler keeps the tap scalar and re-materialises the literal with a v_mov_b32 in every row
  uint32_t k6s, round_s;
  asm("v_mov_b32 %0, %1" : "=v"(k6s) : "s"(k[6] << 8));
  asm("s_mov_b32 %0, 0x800000" : "=s"(round_s));
  // The 36 + 6 source rows of the strip as a software pipeline: the three dwords of row r + kBlurAhead are requested
  // before row r is worked on, so a wave always has kBlurAhead rows' loads in flight under its arithmetic (the first form
  // fetched 7 rows, waited for them, worked through them and only then fetched the next 7: every 7th row paid a full
  // trip to L2 / HBM with nothing of its own to do meanwhile).  Fully unrolled: the arrays are registers, the ring
  // indices constants.
  constexpr int kRowsIn = kBlurStrip + 6;
  uint32_t d0[kRowsIn], d1[kRowsIn], d2[kRowsIn];
  // Row addresses: source row y0 - 3 + rr, REFLECT_101 without a loop (rows start at >= -3 and overshoot the bottom by
  // < h: levels are >= 64 rows, a strip is 36 + 6, so one fold per side is exact; rows past the fold are never stored).
  // Only the first three rows can fold at the top; from the fourth on the byte offset is min(down, up) of two offsets
  // that move by one row pitch per row -- an add, a subtract (of a scalar multiple of the pitch) and a minimum.
  const uint32_t off_dn = (uint32_t)((y0 - 3) * spitch + base), off_up = (uint32_t)((2 * h - 2 - (y0 - 3)) * spitch + base);
  auto fetch_row = [&](int rr) {
    uint32_t off;
    if (rr < 3) {
      const int ysrc = y0 == 0 ? 3 - rr : y0 - 3 + rr;
      off = (uint32_t)(ysrc * spitch + base);
    } else {
      off = min(off_dn + (uint32_t)(rr * spitch), off_up - (uint32_t)(rr * spitch));
    }
    const uint8_t *row = img + off;  // uniform base + 32-bit lane offset
    d0[rr] = *(const u32_unaligned *)(row);
    d1[rr] = *(const u32_unaligned *)(row + 4);
    d2[rr] = *(const u32_unaligned *)(row + 8);
  };
#pragma unroll
  for (int rr = 0; rr < kBlurAhead; rr++) fetch_row(rr);
#pragma unroll
  for (int rr = 0; rr < kRowsIn; rr++) {  // source row y0 - 3 + rr
    if (rr + kBlurAhead < kRowsIn) fetch_row(rr + kBlurAhead);
    const int s = rr % 7;
    uint32_t e0 = d0[rr], e1 = d1[rr], e2 = d2[rr];
    if (!interior) {
      const uint32_t b0 = __builtin_amdgcn_perm(e1, e0, selLo[0]) | __builtin_amdgcn_perm(0u, e2, selHi[0]);
      const uint32_t b1 = __builtin_amdgcn_perm(e1, e0, selLo[1]) | __builtin_amdgcn_perm(0u, e2, selHi[1]);
      const uint32_t b2 = __builtin_amdgcn_perm(e1, e0, selLo[2]) | __builtin_amdgcn_perm(0u, e2, selHi[2]);
      // written back INTO the registers the loads filled, so that the interior waves (which skip this block) need no
      // copies to meet the edge waves' values in other registers
      asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "+v"(e0), "+v"(e1), "+v"(e2) : "v"(b0), "v"(b1), "v"(b2));
    }
    // pixel j: taps over bytes j+1 .. j+7 of {e0,e1,e2}
    uint32_t H[4];
    H[0] = __builtin_amdgcn_udot4(e0, TA0, __builtin_amdgcn_udot4(e1, TA1, 0u, false), false);
    H[1] = __builtin_amdgcn_udot4(e0, TB0, __builtin_amdgcn_udot4(e1, TB1, __builtin_amdgcn_udot4(e2, TB2, 0u, false), false), false);
    H[2] = __builtin_amdgcn_udot4(e0, TC0, __builtin_amdgcn_udot4(e1, TC1, __builtin_amdgcn_udot4(e2, TC2, 0u, false), false), false);
    H[3] = __builtin_amdgcn_udot4(e1, TD1, __builtin_amdgcn_udot4(e2, TD2, 0u, false), false);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      pw[s][j] = hprev[j] | (H[j] << 16);
      hprev[j] = H[j];
    }
    if (rr >= 6) {
      const int yo = y0 + rr - 6;
      if (yo < h) {
        uint32_t cl[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          uint32_t acc = __umul24(k6s, H[j]) + round_s;  // 32768 << 8
          acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pw[(s + 2) % 7][j]), K01, acc, true);  // rows r-6, r-5
          acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pw[(s + 4) % 7][j]), K23, acc, true);  // rows r-4, r-3
          cl[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pw[(s + 6) % 7][j]), K45, acc, true);  // rows r-2, r-1
        }
        const uint32_t out = __builtin_amdgcn_perm(cl[1], cl[0], 0x0C0C0703u) |
                             (__builtin_amdgcn_perm(cl[3], cl[2], 0x0C0C0703u) << 16);
        trow[(rr - 6) & 3] = out;
      }
      const int o = rr - 6;  // output row of the strip: tile row o / 4, line row o % 4
      if ((o & 3) == 3 && y0 + o - 3 < h) {  // the tile row is complete (rows past the level's last one: padding)
        auto x1 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); };  // quad_perm [1,0,3,2]: every lane has a source
        auto x2 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); };  // quad_perm [2,3,0,1]
        const uint32_t p01 = x1(trow[1]), p10 = x1(trow[0]), p23 = x1(trow[3]), p32 = x1(trow[2]);
        const uint32_t c0 = q_odd ? p01 : trow[0], c1 = q_odd ? trow[1] : p10;
        const uint32_t c2 = q_odd ? p23 : trow[2], c3 = q_odd ? trow[3] : p32;
        const uint32_t r02 = x2(c2), r13 = x2(c3), r20 = x2(c0), r31 = x2(c1);
        u32x4 line;
        line.x = q_hi ? r02 : c0, line.y = q_hi ? r13 : c1, line.z = q_hi ? c2 : r20, line.w = q_hi ? c3 : r31;
        *(u32x4 *)(dst + (uint32_t)(o >> 2) * tile_row_bytes) = line;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur,
                                              const FrameGeom *__restrict__ fg, Src0 s0) {
  blur_block<VSG_BLUR_AHEAD_ALONE>(pyr, blur, fg, s0, frame_major_block());
}

// Latency path of the blocking single-frame calls: the blur workgroups ride in the octree's launch (blocks
// [nlevels, nlevels + blur blocks) of a frame), hidden under the octree's 30-odd microseconds, instead of on a second
// stream whose fork / join events cost the chain as much GPU idle time as the blur itself takes (measured: 5-7 us at
// each of the two cross-stream waits against 12 us of blur).  Both need only the pyramid.  Throughput batches keep the
// two kernels apart: there the blur is issue-bound work that wants its own occupancy.
// (VSG_OB_WAVES, vsg_kernels.h: 4 waves per SIMD = 128 registers.  With the round-5 blur arithmetic the blur's waves are
// bound by memory latency more than by issue slots, and 12 rows of loads in flight at 4 waves beat 7 rows at 5 waves and the
// octree's 59 spilled registers: + 1.0-1.2 % on the 512-frame step, 3 waves - 2.5 %: profiles/r05_p_*)
template <int kMemU>
__device__ __forceinline__ void octree_blur_body(const OctArgs &a, const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur,
                                                 const Src0 &s0, int nlevels, int nframes, int lead, uint8_t *oct_lds, int *wtot,
                                                 int *sort_stack) {
  static_assert(kOctThreads == 256, "the blur body is written for 256-thread workgroups");
  // frame-major over the combined grid: the workgroups that share an XCD (and its L2) work on whole frames
  BlockXY blk = frame_major_block();
  if (lead > 0) {
    // The octree workgroups run `lead` rows of the grid AHEAD of the blur workgroups (grid rows = frames + lead): an
    // octree workgroup lives several times as long as a blur workgroup, and dealt out frame by frame the last frames'
    // octrees were a tail of the launch with most workgroup slots empty (3.5 of 5 waves per SIMD resident on average).
    // Row r: octree of frame r (if there is one), blur of frame r - lead (if there is one).
    if (blk.x >= nlevels) blk.y -= lead;
    if (blk.y < 0 || blk.y >= nframes) return;
  }
  if (blk.x < nlevels)
    octree_block<kMemU>(a, blk.x, blk.y, oct_lds, wtot, sort_stack);
  else
    blur_block<VSG_BLUR_AHEAD>(pyr, blur, a.fg, s0, BlockXY{blk.x - nlevels, blk.y});
}
template <int kMemU>
__global__ __launch_bounds__(kOctThreads) __attribute__((amdgpu_waves_per_eu(VSG_OB_WAVES, VSG_OB_WAVES))) void k_octree_blur(
    OctArgs a, const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur, Src0 s0, int nlevels, int nframes, int lead) {
  extern __shared__ __attribute__((aligned(16))) uint8_t oct_lds[];
  __shared__ int wtot[2 * kMaxWaves];
  __shared__ int sort_stack[kSortStackInts];
  octree_blur_body<kMemU>(a, pyr, blur, s0, nlevels, nframes, lead, oct_lds, wtot, sort_stack);
}
// (Rounds 3-5 had a second instantiation of this launch for calls of a few frames, compiled for 4 waves per SIMD; since round 5
// k_octree_blur itself is compiled for VSG_OB_WAVES = 4 waves -- 125 registers, nothing in scratch -- the two were the same code
// object twice; one-frame calls take k_octree_blur.)

// ------------------------------------------------------------------------------------------------
// Output slots: keypoints are visited level by level in octree order; those inside the lapping area
// fill the arrays from the back, the others from the front (ORBextractor.cc:1119,1152-1163).
__device__ __forceinline__ void slots_of_frame(const FrameGeom *__restrict__ fg, const uint32_t *__restrict__ sel,
                                               const int *__restrict__ sel_count, int *__restrict__ flags,
                                               int4 *__restrict__ slots, FrameHeader *__restrict__ hdr, int lap0, int lap1,
                                               int frame, int *wtot, int *lstart) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    int s = 0;
    for (int l = 0; l < fg->nlevels; l++) {
      lstart[l] = s;
      s += min(sel_count[frame * kMaxLevels + l], fg->lv[l].sel_cap);
    }
    for (int l = fg->nlevels; l <= kMaxLevels; l++) lstart[l] = s;
  }
  __syncthreads();
  const int n = lstart[fg->nlevels];
  int *fl = flags + (size_t)frame * fg->out_cap;
  int4 *sl = slots + (size_t)frame * fg->out_cap;
  const float flap0 = (float)lap0, flap1 = (float)lap1;
  for (int i = tid; i < n; i += 256) {
    int l = 0;
    while (i >= lstart[l + 1]) l++;
    const uint32_t c = sel[(size_t)frame * fg->sel_frame + fg->lv[l].sel_off + (i - lstart[l])];
    float x = (float)(VSG_CAND_X(c) + kFastBorder);
    if (l != 0) x = fmul(x, fg->lv[l].scale);  // keypoint->pt *= scale (:1147-1150)
    fl[i] = (x >= flap0 && x <= flap1) ? 1 : 0;
  }
  __syncthreads();
  BlockGroup g;
  g.tid = tid;
  g.nthreads = 256;
  g.wtot = wtot;
  g.stk = nullptr;
  const int T = g.exclusive_scan(fl, n);  // fl[i] = lapping keypoints before i
  for (int i = tid; i < n; i += 256) {
    const int before = fl[i];
    const int after = (i + 1 < n) ? fl[i + 1] : T;
    // everything k_orient_desc needs to start on keypoint i in ONE load whose address it knows at launch: the selected
    // candidate, its output slot (stereoIndex-- / monoIndex++) and its level
    int l = 0;
    while (i >= lstart[l + 1]) l++;
    const uint32_t c = sel[(size_t)frame * fg->sel_frame + fg->lv[l].sel_off + (i - lstart[l])];
    sl[i] = make_int4((int)c, (after - before) ? (n - 1 - before) : (i - before), l, 0);
  }
  if (tid == 0) {
    hdr[frame].n = n;
    hdr[frame].mono = n - T;
    for (int l = 0; l <= kMaxLevels; l++) hdr[frame].level_start[l] = lstart[l];
  }
}

__global__ __launch_bounds__(256) void k_slots(const FrameGeom *__restrict__ fg, const uint32_t *__restrict__ sel,
                                               const int *__restrict__ sel_count, int *__restrict__ flags,
                                               int4 *__restrict__ slots, FrameHeader *__restrict__ hdr, int lap0,
                                               int lap1) {
  __shared__ int wtot[2 * kMaxWaves];
  __shared__ int lstart[kMaxLevels + 1];
  slots_of_frame(fg, sel, sel_count, flags, slots, hdr, lap0, lap1, blockIdx.x, wtot, lstart);
}

// ------------------------------------------------------------------------------------------------
// One wavefront per keypoint: intensity-centroid angle on the un-blurred level, steered rBRIEF-256 on
// the blurred level, keypoint record + 32 descriptor bytes written to the keypoint's output slot.
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
// rBRIEF patch radius / width; the LDS copy holds the 10 x (3 or 4) blurred TILES the patch touches: 40 rows of 64 bytes at
// a row pitch of 80.  The 512 test points of a keypoint are byte reads at rotated pattern offsets clustered around the
// centre: at a pitch of 64 bytes (16 dwords) every second row lies on the same 16 banks and a 32-lane group of one
// ds_read_u8 meets 4.6 distinct dwords on its busiest bank on average (the real pattern under random rotations, simulated);
// at 80 bytes (20 dwords: eight row classes) 3.1 -- what 32 random addresses give.  The test reads are two thirds of the
// LDS pipe's busy time in this kernel.  80 keeps the tile rows 16-byte aligned for the b128 commits.
#ifndef VSG_OD_PITCH
#define VSG_OD_PITCH 80
#endif
enum { kPatchR = 18, kPatchW = 2 * kPatchR + 1, kPatchP = VSG_OD_PITCH, kPatchRows = 40 };
static_assert(kPatchP % 16 == 0 && kPatchP >= 64 && kPatchP < 256, "tile rows are committed as aligned 16-byte stores");

// IC_Angle lane layout: 16 rows x 4 (unaligned) qwords per load instruction (lane = 4 * row + column), 2 instructions
// cover the 31 x 31 patch (rows 16 * it + row - 15, pixels u = 8 * column + 4 * half + b - 15).  Per (it, lane, half)
// the table holds byte weights for the 4 pixels of that dword: wp = 1 inside the 749-px disc (|u| <= umax[|v|],
// ORBextractor.cc:454-469), wu = (u + 15) inside, else 0 -- so  sum p = udot4(px, wp)  and
// sum (u + 15) p = udot4(px, wu).  (tools/ubench_tcp.hip: the L1 moves this patch fastest as qwords, 4 lanes per
// row -- 82 cycles per CU against 106 for dwords, 8 lanes per row -- and that layout does not care about the byte
// alignment of the patch origin.)
struct alignas(16) DiscTable {
  uint32_t w[2][64][4];  // [it][lane] = {wp(half 0), wu(half 0), wp(half 1), wu(half 1)}
};
constexpr DiscTable make_disc_table() {
  DiscTable t{};
  for (int it = 0; it < 2; it++)
    for (int lane = 0; lane < 64; lane++) {
      const int v = 16 * it + (lane >> 2) - kHalfPatch, av = v < 0 ? -v : v;
      for (int h = 0; h < 2; h++)
        for (int b = 0; b < 4; b++) {
          const int u = 8 * (lane & 3) + 4 * h + b - kHalfPatch, au = u < 0 ? -u : u;
          if (v > kHalfPatch || u > kHalfPatch) continue;
          const int um = (int)((0x3689ABCDDEEEFFFFull >> (4 * (av & 15))) & 15);  // umax[|v|]
          if (au <= um) {
            t.w[it][lane][2 * h + 0] |= 1u << (8 * b);
            t.w[it][lane][2 * h + 1] |= (uint32_t)(u + kHalfPatch) << (8 * b);
          }
        }
    }
  return t;
}
__constant__ DiscTable c_disc = make_disc_table();

// Sum over the 64 lanes of a wavefront, returned wave-uniform (SGPR): two quad permutes and two row mirrors on
// the DPP path give every lane its 16-lane row sum, four v_readlane + scalar adds finish.  Integer, exact.
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);  // row_mirror
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
         __builtin_amdgcn_readlane(v, 48);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef VSG_OD_G
#define VSG_OD_G 3
#endif
// keypoints a wavefront works through one after the other: amortises the workgroup prologue (pattern + disc tables
// into LDS, one barrier).  1: 0.343, 2: 0.296, 3 and 4: 0.291 ms per 512 C2 frames; workgroups of hundreds of keypoints (a
// launch that fills the chip once) lose badly: 0.34-0.44 ms, the frames of the resident workgroups thrash the XCD's L2;
// requesting the first keypoint's patches before the prologue's barrier: 0.293 -> 0.300 (profiles/r04_d_*, steps 4-5)
constexpr int kOdKpPerWave = VSG_OD_G;

#ifndef VSG_OD_SGPRS
#define VSG_OD_SGPRS 72
#endif
// kSelf (latency path, no lapping area: RGB-D / stereo, vLappingArea = {0, 0}): the kernel derives what k_slots would have
// handed it -- the level starts from the octree's per-level counts, slot = keypoint index -- in its own prologue, and the
// 8 us k_slots launch leaves the one-frame chain (two dependent loads in front of a keypoint instead of a kernel).
template <bool kMirror, bool kSelf>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(VSG_OD_SGPRS))) void k_orient_desc(const uint8_t *__restrict__ pyr, const uint8_t *__restrict__ blur,
                                                     const FrameGeom *__restrict__ fg, Src0 s0,
                                                     const uint32_t *__restrict__ sel, const int *__restrict__ sel_count,
                                                     const int4 *__restrict__ slots, const FrameHeader *__restrict__ hdr,
                                                     const int8_t *__restrict__ pattern, KeyPointPOD *__restrict__ kps,
                                                     uint8_t *__restrict__ desc, int *__restrict__ counts, int capacity,
                                                     OutMirror mir) {
  __shared__ f32x4 patf[256];            // test k: (x0, x1, y0, y1) of bit_pattern_31_ as floats: the packed operands as they are used
  __shared__ int s_hdr[kMaxLevels + 3];  // n, mono, level_start[0..kMaxLevels]
  __shared__ __attribute__((aligned(16))) uint8_t s_patch[4 * kPatchRows * kPatchP];
  __shared__ u32x4 s_ic[2 * 64];  // c_disc, one 16-byte entry per (it, lane)
  const BlockXY blk = frame_major_block();
  const int frame = blk.y, tid = threadIdx.x;
  // The wave's keypoints (k_slots' records: candidate, slot, level), fetched at an address known at launch, beside the
  // table loads below: header -> level search -> selected list -> keypoint used to be a chain of dependent round trips
  // in front of every keypoint's patch loads.  Records at or past the frame's count are stale but inside the array.
  int4 rec[kOdKpPerWave];
  if (!kSelf) {
    const int cap1 = fg->out_cap - 1;
#pragma unroll
    for (int j = 0; j < kOdKpPerWave; j++) {
      const int g = __builtin_amdgcn_readfirstlane((blk.x * kOdKpPerWave + j) * 4 + (tid >> 6));
      rec[j] = slots[(size_t)frame * fg->out_cap + min(g, cap1)];
    }
  }
  if (tid < 128) s_ic[tid] = ((const u32x4 *)c_disc.w)[tid];
  {
    const uint32_t pw = ((const uint32_t *)pattern)[tid];
    patf[tid] = (f32x4){(float)(int8_t)pw, (float)(int8_t)(pw >> 16), (float)(int8_t)(pw >> 8), (float)(int8_t)(pw >> 24)};
  }
  if (kSelf) {
    if (tid == 0) {  // the slot kernel's scan (slots_of_frame) with an empty lapping set: n = monoIndex = the total
      int cnt[kMaxLevels];
#pragma unroll
      for (int l = 0; l < kMaxLevels; l++) cnt[l] = l < fg->nlevels ? min(sel_count[frame * kMaxLevels + l], fg->lv[l].sel_cap) : 0;
      int run = 0;
#pragma unroll
      for (int l = 0; l < kMaxLevels; l++) {
        s_hdr[2 + l] = run;
        run += cnt[l];
      }
      s_hdr[2 + kMaxLevels] = run;
      s_hdr[0] = run, s_hdr[1] = run;
    }
  } else if (tid < kMaxLevels + 3) {
    s_hdr[tid] = ((const int *)&hdr[frame])[tid];
  }
  __syncthreads();
  const int n = s_hdr[0];
  if (blk.x == 0 && tid == 0) {
    counts[frame * 2 + 0] = n;
    counts[frame * 2 + 1] = s_hdr[1];
    if (kMirror) mir.counts[frame * 2 + 0] = n, mir.counts[frame * 2 + 1] = s_hdr[1];
  }
  const int lane = tid & 63;
  // everything derived from the keypoint index is wave-uniform: keep it in SGPRs so the per-level geometry comes
  // through the scalar cache in one round trip instead of a chain of dependent vector loads
  uint8_t *patch = &s_patch[(tid >> 6) * (kPatchRows * kPatchP)];
  // Three passes over the wave's keypoints: (A) the moments of every keypoint, (B) fastAtan2 -> cosf / sinf of ALL of them in
  // one evaluation -- keypoint j in lane j, the branch-free forms of vsg_math.h -- and (C) the descriptors.  The rotation
  // is ~40 dependent FP32 / FP64 instructions (an IEEE division, the glibc kernels in double) that were issued once per
  // keypoint with every lane computing the same value; FP64 issues at a fraction of the FP32 rate on this part.
  int k_l[kOdKpPerWave], k_slot[kOdKpPerWave], k_m01[kOdKpPerWave], k_m10[kOdKpPerWave], nk = 0;
  uint32_t k_c[kOdKpPerWave];
  // the workgroup's keypoints: wave w takes blk.x * 4 * G + 4 * j + w  (adjacent keypoints run side by side)
  const int g0 = __builtin_amdgcn_readfirstlane(blk.x * kOdKpPerWave * 4 + (tid >> 6));
  if (g0 >= n) return;  // nothing for this wave (no barrier follows)
  nk = min(kOdKpPerWave, (n - g0 + 3) >> 2);  // wave-uniform
  // (level, candidate, slot) of every keypoint of the wave first -- past the wave's last keypoint: the last one again, so
  // that every address below is valid and the loads need no branch -- then ALL their patch loads in one block of
  // back-to-back instructions: one trip to L2 / HBM for the wave's three IC_Angle patches instead of three dependent ones
  // (load, reduce, next keypoint's load; 59 % of this kernel's wave-cycles were parked at waits, profiles/r04_r_pmc_*)
#pragma unroll
  for (int j = 0; j < kOdKpPerWave; j++) {
    const int jj = min(j, nk - 1);
    const int g = g0 + 4 * jj;
    int l, slot;
    uint32_t c;
    if (kSelf) {
      l = 0;
      while (g >= s_hdr[3 + l]) l++;  // wave-uniform (g < n = s_hdr[2 + kMaxLevels])
      l = __builtin_amdgcn_readfirstlane(l);
#if VSG_OD_SPATIAL
      // processing position g - level start -> the keypoint's position in the octree's list (the octree's epilogue wrote
      // the permutation into node_of, which travels in the unused `slots` argument of this launch form)
      const int i = ((const uint16_t *)slots)[(size_t)frame * fg->cand_frame + fg->lv[l].cand_off + (g - s_hdr[2 + l])];
      c = sel[(size_t)frame * fg->sel_frame + fg->lv[l].sel_off + i];
      slot = s_hdr[2 + l] + i;
#else
      c = sel[(size_t)frame * fg->sel_frame + fg->lv[l].sel_off + (g - s_hdr[2 + l])];
      slot = g;
#endif
    } else {
      int4 r = rec[0];
#pragma unroll
      for (int q = 1; q < kOdKpPerWave; q++)
        if (q == jj) r = rec[q];
      l = r.z, c = (uint32_t)r.x, slot = r.y;
    }
    k_l[j] = l, k_slot[j] = slot, k_c[j] = c;
  }
#pragma unroll
  for (int j = 0; j < kOdKpPerWave; j++) {
    k_l[j] = __builtin_amdgcn_readfirstlane(k_l[j]);
    k_c[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)k_c[j]);
    k_slot[j] = __builtin_amdgcn_readfirstlane(k_slot[j]);
  }
  // ---- IC_Angle: 16 rows x 4 (unaligned) qwords per load instruction, 2 instructions per patch; the disc mask and
  // the column weights are v_dot4_u32_u8 operands.  The row past v = 15 (it = 1, row = 15) is masked by the weights;
  // those lanes re-read row 30 so the address stays inside the image.
  {
    const int row = lane >> 2, col = lane & 3;
    u32x2 p0[kOdKpPerWave], p1[kOdKpPerWave];
#pragma unroll
    for (int j = 0; j < kOdKpPerWave; j++) {
      const int cx = VSG_CAND_X(k_c[j]) + kFastBorder, cy = VSG_CAND_Y(k_c[j]) + kFastBorder;
      int upitch;
      const uint8_t *unblurred = level_ptr(fg, s0, pyr, frame, k_l[j], upitch);
      const uint8_t *ptr = unblurred + (ptrdiff_t)(cy - kHalfPatch) * upitch + (cx - kHalfPatch) + 8 * col;
      p0[j] = *(const u64_global_unaligned *)(ptr + (ptrdiff_t)row * upitch);
      p1[j] = *(const u64_global_unaligned *)(ptr + (ptrdiff_t)min(row + 16, 2 * kHalfPatch) * upitch);
    }
    const u32x4 w0 = s_ic[lane], w1 = s_ic[64 + lane];
#pragma unroll
    for (int j = 0; j < kOdKpPerWave; j++) {
      // S = sum p, U = sum (u + 15) p over the lane's 2 x 8 pixels;  v = 16 it + row - 15
      const uint32_t s0p = __builtin_amdgcn_udot4(p0[j].y, w0.z, __builtin_amdgcn_udot4(p0[j].x, w0.x, 0u, false), false);
      const uint32_t s1p = __builtin_amdgcn_udot4(p1[j].y, w1.z, __builtin_amdgcn_udot4(p1[j].x, w1.x, 0u, false), false);
      uint32_t su = __builtin_amdgcn_udot4(p0[j].x, w0.y, 0u, false);
      su = __builtin_amdgcn_udot4(p0[j].y, w0.w, su, false);
      su = __builtin_amdgcn_udot4(p1[j].x, w1.y, su, false);
      su = __builtin_amdgcn_udot4(p1[j].y, w1.w, su, false);
      const int S = (int)(s0p + s1p);
      k_m10[j] = wave_sum_i32((int)su - kHalfPatch * S);
      k_m01[j] = wave_sum_i32((row - kHalfPatch) * S + 16 * (int)s1p);
    }
  }
  // ---- descriptor patch.  The rotated pattern stays within +-18 px of the centre (|p| <= 18.38).  The blurred level is TILED
  // (16 x 4 pixels per 64-byte line, LevelGeom::btx): the 37 x 37 patch touches 10 tile rows x 3 or 4 tile columns, and
  // the load's lanes are mapped to whole tiles -- lane = 4 * tile + row of the tile, 16 bytes per lane -- so every group of
  // four lanes reads ONE full line and a keypoint costs ~33 line requests in 2 or 3 load instructions instead of ~58
  // row segments in 4 (tools/ubench_tile.hip: 175 against 358 CU-cycles per patch with the LDS side included).  The
  // tiles land in LDS where they lie (aligned b128 rows of a 40 x 64-byte image), so no byte shifting is left.
  struct TileLoad {
    u32x4 pv[3];
    int lo[3], nt;
  };
  const int sub = lane & 3, tg = lane >> 2;
  auto issue_tiles = [&](int l, uint32_t c, TileLoad &T) {
    const LevelGeom &L = fg->lv[l];
    const int px0 = VSG_CAND_X(c) + kFastBorder - kPatchR, py0 = VSG_CAND_Y(c) + kFastBorder - kPatchR;
    const int tx0 = px0 >> 4, ty0 = py0 >> 2;
    const int ntx = ((px0 + 2 * kPatchR) >> 4) - tx0 + 1;  // 3 (three quarters of the origins) or 4; always 10 tile rows
    const int nt = 10 * ntx;
    const float inv_ntx = ntx == 3 ? 1.0f / 3.0f : 0.25f;
    const uint8_t *gp = blur + (size_t)frame * fg->blur_frame_bytes + L.boff + (uint32_t)((ty0 * L.btx + tx0) * kBlurTileBytes) +
                        sub * kBlurTileW;
    const uint32_t trow = (uint32_t)L.btx * kBlurTileBytes;
    T.nt = nt;
#pragma unroll
    for (int it = 0; it < 3; it++) {
      if (it == 2 && ntx == 3) break;  // wave-uniform: 30 tiles fit two instructions
      const int t = min(it * 16 + tg, nt - 1);  // lanes past the last tile re-read it (never stored)
      const int tyi = div_small(t, inv_ntx), txi = t - tyi * ntx;
      T.pv[it] = *(const __attribute__((address_space(1))) u32x4 *)(gp + (uint32_t)tyi * trow + (uint32_t)(txi * kBlurTileBytes));
      T.lo[it] = (4 * tyi + sub) * kPatchP + kBlurTileW * txi;
    }
  };
  auto commit_tiles = [&](const TileLoad &T) {
#pragma unroll
    for (int it = 0; it < 3; it++) {
      if (it == 2 && T.nt == 30) break;
      if (it * 16 + tg < T.nt) *(u32x4 *)(patch + T.lo[it]) = T.pv[it];
    }
  };
  // the first keypoint's tiles are requested before the rotations are evaluated (~120 instructions with nothing in flight);
  // keypoint j + 1's are requested right after keypoint j's have been written to LDS, so they travel while the 256 tests
  // of keypoint j run (two register sets, alternating: the kernel holds 8 waves per SIMD on 36 registers, 16 more are free)
  TileLoad TA, TB;
  issue_tiles(k_l[0], k_c[0], TA);
  float ang_v, a_v, b_v;
  {
    float fm01 = 0.0f, fm10 = 0.0f;
#pragma unroll
    for (int j = 0; j < kOdKpPerWave; j++)
      if (lane == j) fm01 = (float)k_m01[j], fm10 = (float)k_m10[j];
    brief_rotation_of_moments(fm01, fm10, &ang_v, &a_v, &b_v);
  }
#pragma unroll
  for (int j = 0; j < kOdKpPerWave; j++) {
  if (j >= nk) break;
  const int l = k_l[j], slot = k_slot[j];
  const uint32_t c = k_c[j];
  const LevelGeom &L = fg->lv[l];
  const int cx = VSG_CAND_X(c) + kFastBorder, cy = VSG_CAND_Y(c) + kFastBorder;
  const float angle = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ang_v), j));
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a_v), j));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b_v), j));
  const int tx0 = (cx - kPatchR) >> 4, ty0 = (cy - kPatchR) >> 2;
  if ((j & 1) == 0) {
    commit_tiles(TA);
    if (j + 1 < kOdKpPerWave && j + 1 < nk) issue_tiles(k_l[j + 1], k_c[j + 1], TB);
  } else {
    commit_tiles(TB);
    if (j + 1 < kOdKpPerWave && j + 1 < nk) issue_tiles(k_l[j + 1], k_c[j + 1], TA);
  }
  // the patch is private to this wavefront: LDS writes complete in order before the reads below
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint8_t *center = patch + (cy - 4 * ty0) * kPatchP + (cx - 16 * tx0);
  // both points of a test at once on the packed-fp32 path; every product and sum is rounded separately (no contraction:
  // the file is built with -ffp-contract=off), exactly like the scalar code
  //   dy = cvRound(x*b + y*a), dx = cvRound(x*a - y*b)     (ORBextractor.cc:113-115)
  // lane handles tests lane, lane + 64, lane + 128, lane + 192
  uint64_t word = 0;
  // cvRound (round half to even) of |v| <= 26 as ONE float addition: v + 1.5 * 2^23 has an ulp of 1, so the sum's rounding
  // IS the rounding to the nearest integer, ties to even, and its bit pattern is 0x4B400000 + round(v) =: M + round(v).
  // Both points at once (v_pk_add_f32), and the constant leaves through the base address: with By = M + dy, Bx = M + dx the
  // byte of (dy, dx) sits at centre + P dy + dx = (centre - P (M & 0xFFFFFF) - M) + P By[23:0] + Bx modulo 2^32 -- the low
  // 24 bits of By are 0x400000 + dy, a positive 24-bit number, so the row term is ONE v_mad_u32_u24 for any pitch P
  // (v_rndne + v_cvt_i32 per coordinate before: 8 instructions per pair of points, now 2).
  {
    const uint32_t cbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)center -
                           (uint32_t)kPatchP * 0x400000u - 0x4B400000u;
    const f32x2 kMagic = {12582912.0f, 12582912.0f};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const f32x4 pt = patf[r * 64 + lane];
      const f32x2 X = pt.xy, Y = pt.zw;
      const f32x2 ry = (X * b + Y * a) + kMagic, rx = (X * a - Y * b) + kMagic;
      // (the components through named floats: __builtin_bit_cast of `ry.y` itself read component x)
      const float ry0 = ry.x, ry1 = ry.y, rx0 = rx.x, rx1 = rx.y;
      const uint32_t o0 = cbase + (__umul24(__builtin_bit_cast(uint32_t, ry0), (uint32_t)kPatchP) + __builtin_bit_cast(uint32_t, rx0));
      const uint32_t o1 = cbase + (__umul24(__builtin_bit_cast(uint32_t, ry1), (uint32_t)kPatchP) + __builtin_bit_cast(uint32_t, rx1));
      const uint32_t t0 = *(const __attribute__((address_space(3))) uint8_t *)(uintptr_t)o0;
      const uint32_t t1 = *(const __attribute__((address_space(3))) uint8_t *)(uintptr_t)o1;
      const uint64_t m = __ballot(t0 < t1);  // 64 descriptor bits per ballot
      if (lane == r) word = m;
    }
  }
  if (slot < capacity) {
    if (lane < 4) *(uint64_t *)(desc + ((size_t)frame * capacity + slot) * 32 + lane * 8) = word;
    if (lane == 0) {
      KeyPointPOD kp;
      kp.x = (float)cx;
      kp.y = (float)cy;
      if (l != 0) {
        kp.x = fmul(kp.x, L.scale);
        kp.y = fmul(kp.y, L.scale);
      }
      kp.size = L.kp_size;
      kp.angle = angle;
      kp.response = (float)VSG_CAND_R(c);
      kp.octave = l;
      kp.class_id = -1;
      kps[(size_t)frame * capacity + slot] = kp;
      if (kMirror && slot < mir.capacity) mir.kps[(size_t)frame * mir.capacity + slot] = kp;
    }
    if (kMirror && slot < mir.capacity && lane < 4)
      *(uint64_t *)(mir.desc + ((size_t)frame * mir.capacity + slot) * 32 + lane * 8) = word;
  }
  // the next keypoint overwrites the patch: this wave's LDS reads above are issued (and, in order, completed) first
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// mvImagePyramid[level] with its 19 px BORDER_REFLECT_101 frame (copyMakeBorder, :1186-1192), on demand.
__global__ void k_border_copy(const uint8_t *__restrict__ img, int w, int h, int pitch, uint8_t *__restrict__ dst,
                              int dpitch, int b) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w + 2 * b) return;
  dst[(size_t)y * dpitch + x] = img[(size_t)reflect101(y - b, h) * pitch + reflect101(x - b, w)];
}

// ------------------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches (Frame.cc:957-1127), one wavefront per LEFT keypoint (no greedy state: left
// keypoints are independent).  Row-band candidates are found by scanning the right keypoints in index order
// (= the push order of vRowIndices), packed keys keep "first minimum wins"; the 11x11 L1 SAD over 11 shifts runs
// on the un-blurred pyramid level of the left keypoint's octave; the sub-pixel parabola and the disparity/depth
// arithmetic are plain non-contracted float ops.  The median-based outlier cut (:1113-1126) is done by the host.
struct StereoParams {
  float mb, mbf;
  float scale[kMaxLevels], invScale[kMaxLevels];
  int nRows;
};

__global__ __launch_bounds__(256) void k_stereo(PyrView pl, PyrView pr, StereoParams sp, const KeyPointPOD *kpsL,
                                                const uint8_t *descL, int nL, const KeyPointPOD *kpsR,
                                                const uint8_t *descR, int nR, float *uRight, float *depth,
                                                int *sadBest) {
  const int lane = threadIdx.x & 63;
  const int iL = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (iL >= nL) return;
  const KeyPointPOD kpL = kpsL[iL];
  const int levelL = kpL.octave;
  const float vL = kpL.y, uL = kpL.x;
  float outU = -1.0f, outD = -1.0f;
  int outS = -1;
  const float minZ = sp.mb, minD = 0.f;
  const float maxD = fdiv(sp.mbf, minZ);
  const float minU = fsub(uL, maxD), maxU = fsub(uL, minD);
  const int rowL = (int)vL;
  uint32_t k = 0xFFFFFFFFu;
  if (rowL >= 0 && rowL < sp.nRows && !(maxU < 0)) {
    const uint4 *dl = (const uint4 *)(descL + (size_t)iL * 32);
    const uint4 a0 = dl[0], a1 = dl[1];
    for (int iR = lane; iR < nR; iR += 64) {
      const KeyPointPOD kpR = kpsR[iR];
      const float r = fmul(2.0f, sp.scale[kpR.octave]);
      const int maxr = (int)ceilf(fadd(kpR.y, r)), minr = (int)floorf(fsub(kpR.y, r));
      if (rowL < minr || rowL > maxr) continue;
      if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
      if (!(kpR.x >= minU && kpR.x <= maxU)) continue;
      const uint4 *dr = (const uint4 *)(descR + (size_t)iR * 32);
      const uint4 b0 = dr[0], b1 = dr[1];
      const int dist = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                       __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
      if (dist < 100) k = min(k, ((uint32_t)dist << 20) | (uint32_t)iR);  // bestDist starts at TH_HIGH (:1013)
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) k = min(k, __shfl_xor(k, d));
  const int thOrbDist = (100 + 50) / 2;
  if (k != 0xFFFFFFFFu && (int)(k >> 20) < thOrbDist) {
    const int bestIdxR = (int)(k & 0xFFFFF);
    const float uR0 = kpsR[bestIdxR].x;
    const float sf = sp.invScale[levelL];
    const float scaleduL = roundf(fmul(kpL.x, sf)), scaledvL = roundf(fmul(kpL.y, sf)), scaleduR0 = roundf(fmul(uR0, sf));
    const int w = 5, L = 5;
    const int ily = (int)fsub(scaledvL, (float)w), ilx = (int)fsub(scaleduL, (float)w);
    const float iniu = fsub(fadd(scaleduR0, (float)L), (float)w);
    const float endu = fadd(fadd(fadd(scaleduR0, (float)L), (float)w), 1.0f);
    if (!(iniu < 0 || endu >= (float)pr.w[levelL])) {
      const uint8_t *PL = pl.lvl[levelL], *PR = pr.lvl[levelL];
      const int pL = pl.pitch[levelL], pR = pr.pitch[levelL];
      // this lane's pixels of the 11x11 window: p = lane and lane + 64
      int py[2], px[2], lv[2];
      bool ok[2];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int p = lane + 64 * t;
        ok[t] = p < 121;
        py[t] = ok[t] ? p / 11 : 0;
        px[t] = ok[t] ? p - py[t] * 11 : 0;
        lv[t] = PL[(size_t)(ily + py[t]) * pL + ilx + px[t]];
      }
      int bestDistS = 0x7FFFFFFF, bestincR = 0;
      float vDists[11];
#pragma unroll
      for (int incR = -5; incR <= 5; incR++) {
        const int irx = (int)fsub(fadd(scaleduR0, (float)incR), (float)w);
        int s = 0;
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const int rv = PR[(size_t)(ily + py[t]) * pR + irx + px[t]];
          const int dd = lv[t] - rv;
          s += ok[t] ? (dd < 0 ? -dd : dd) : 0;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
        const float dist = (float)s;  // cv::norm(IL, IR, NORM_L1)
        if (dist < (float)bestDistS) {
          bestDistS = (int)dist;
          bestincR = incR;
        }
        vDists[incR + 5] = dist;
      }
      if (!(bestincR == -L || bestincR == L)) {
        float dist1 = 0, dist2 = 0, dist3 = 0;
#pragma unroll
        for (int j = 1; j < 10; j++)
          if (j == bestincR + 5) {
            dist1 = vDists[j - 1];
            dist2 = vDists[j];
            dist3 = vDists[j + 1];
          }
        const float deltaR = fdiv(fsub(dist1, dist3), fmul(2.0f, fsub(fadd(dist1, dist3), fmul(2.0f, dist2))));
        if (!(deltaR < -1 || deltaR > 1)) {
          float bestuR = fmul(sp.scale[levelL], fadd(fadd(scaleduR0, (float)bestincR), deltaR));
          float disparity = fsub(uL, bestuR);
          if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) {
              disparity = 0.01f;                                   // disparity = 0.01 (double literal -> float)
              bestuR = (float)((double)uL - 0.01);                 // bestuR = uL - 0.01
            }
            outD = fdiv(sp.mbf, disparity);
            outU = bestuR;
            outS = bestDistS;
          }
        }
      }
    }
  }
  if (lane == 0) {
    uRight[iL] = outU;
    depth[iL] = outD;
    sadBest[iL] = outS;
  }
}
void launch_stereo(hipStream_t s, const PyrView &pl, const PyrView &pr, float mb, float mbf, const float *scale,
                   const float *invScale, int nlevels, const KeyPointPOD *kpsL, const uint8_t *descL, int nL,
                   const KeyPointPOD *kpsR, const uint8_t *descR, int nR, float *uRight, float *depth, int *sadBest) {
  StereoParams sp;
  sp.mb = mb;
  sp.mbf = mbf;
  for (int l = 0; l < kMaxLevels; l++) {
    sp.scale[l] = l < nlevels ? scale[l] : 1.f;
    sp.invScale[l] = l < nlevels ? invScale[l] : 1.f;
  }
  sp.nRows = pl.h[0];
  hipLaunchKernelGGL(k_stereo, dim3((nL + 3) / 4), dim3(256), 0, s, pl, pr, sp, kpsL, descL, nL, kpsR, descR, nR, uRight,
                     depth, sadBest);
}

// ------------------------------------------------------------------------------------------------
// cv::cvtColor(..., COLOR_{RGB,BGR,RGBA,BGRA}2GRAY) of Tracking::GrabImage* (Tracking.cc:1595-1608) fused into the
// level-0 staging: thread = 4 output pixels -> one 32-bit store into the gray staging buffer.
__global__ __launch_bounds__(256) void k_cvt_gray(const uint8_t *__restrict__ src, size_t sframe, int spitch,
                                                  int channels, int ri, int bi, int rows, int cols,
                                                  uint8_t *__restrict__ dst, size_t dframe, int dpitch, int cr, int cg,
                                                  int cb, int shift) {
  const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
  if (x4 >= cols) return;
  const uint8_t *s = src + (size_t)blockIdx.z * sframe + (size_t)y * spitch + (size_t)x4 * channels;
  uint32_t out = 0;
  const int half = 1 << (shift - 1);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (x4 + k < cols) {
      const uint8_t *p = s + k * channels;
      const int v = ((int)p[ri] * cr + (int)p[1] * cg + (int)p[bi] * cb + half) >> shift;
      out |= (uint32_t)(v & 0xFF) << (8 * k);
    }
  }
  *(uint32_t *)(dst + (size_t)blockIdx.z * dframe + (size_t)y * dpitch + x4) = out;
}
void launch_cvt_gray(hipStream_t s, const uint8_t *src, size_t sframe, int spitch, int channels, int rgb_order, int rows,
                     int cols, uint8_t *dst, size_t dframe, int dpitch, const int coeffs[3], int shift, int nframes) {
  dim3 grid((cols + 1023) / 1024, rows, nframes), block(256);
  hipLaunchKernelGGL(k_cvt_gray, grid, block, 0, s, src, sframe, spitch, channels, rgb_order ? 0 : 2, rgb_order ? 2 : 0,
                     rows, cols, dst, dframe, dpitch, coeffs[0], coeffs[1], coeffs[2], shift);
}

// ------------------------------------------------------------------------------------------------
// Output export of the host API (vsg_orb_submit_batch / vsg_orb_wait): frame f's n keypoint records and descriptors
// go from the device outputs ([f][src_cap]) to PINNED HOST memory ([f][dst_cap]; the slot's staging or the caller's
// own registered arrays) as coalesced dword stores over PCIe -- n rows per frame, not the capacity, and no DMA
// descriptor per frame.  counts ({n, monoIndex} per frame) go along.
__global__ __launch_bounds__(256) void k_export_outputs(const KeyPointPOD *__restrict__ kps, const uint8_t *__restrict__ desc,
                                                        const int *__restrict__ counts, int src_cap,
                                                        uint32_t *__restrict__ h_kps, uint32_t *__restrict__ h_desc,
                                                        int *__restrict__ h_counts, int dst_cap) {
  const int f = blockIdx.y;
  int n = counts[2 * f];
  if (blockIdx.x == 0 && threadIdx.x < 2) h_counts[2 * f + threadIdx.x] = counts[2 * f + threadIdx.x];
  if (n > dst_cap) n = dst_cap;  // the host reports VSG_ERR_CAPACITY from the count
  const uint32_t *sk = (const uint32_t *)(kps + (size_t)f * src_cap);
  const uint32_t *sd = (const uint32_t *)(desc + (size_t)f * src_cap * 32);
  uint32_t *dk = h_kps + (size_t)f * dst_cap * 7, *dd = h_desc + (size_t)f * dst_cap * 8;
  const int nk = n * 7, nd = n * 8, stride = gridDim.x * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nd; i += stride) {
    dd[i] = sd[i];
    if (i < nk) dk[i] = sk[i];
  }
}
void launch_export(hipStream_t s, const KeyPointPOD *kps, const uint8_t *desc, const int *counts, int src_cap,
                   void *h_kps, void *h_desc, int *h_counts, int dst_cap, int nframes) {
  hipLaunchKernelGGL(k_export_outputs, dim3(4, nframes), dim3(256), 0, s, kps, desc, counts, src_cap, (uint32_t *)h_kps,
                     (uint32_t *)h_desc, h_counts, dst_cap);
}

// Level-0 ingest of the blocking entry points: the device itself reads the caller's (or the slot's) PINNED host image
// over PCIe and writes the level-0 staging -- no copy-engine start-up on the latency path of a single-frame operator()
// (a 307 KB hipMemcpyAsync costs ~15 us end to end, this kernel ~6).  Thread = 16 bytes of a row; rows that are not a
// multiple of 16 bytes end with narrower moves.
__global__ __launch_bounds__(256) void k_ingest(const uint8_t *__restrict__ src, size_t sframe, int spitch,
                                                uint8_t *__restrict__ dst, size_t dframe, int dpitch, int rows, int cols) {
  const int chunks = (cols + 15) >> 4;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= chunks * rows) return;
  const int y = i / chunks, c = i - y * chunks, x = c << 4;
  const uint8_t *s = src + (size_t)blockIdx.y * sframe + (size_t)y * spitch + x;
  uint8_t *d = dst + (size_t)blockIdx.y * dframe + (size_t)y * dpitch + x;
  if (x + 16 <= cols) {
    typedef uint32_t u32x4a1 __attribute__((ext_vector_type(4), aligned(1)));
    typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
    *(u32x4a4 *)d = *(const u32x4a1 *)s;
  } else {
    for (int k = 0; x + k < cols; k++) d[k] = s[k];
  }
}
void launch_ingest(hipStream_t s, const uint8_t *src, size_t sframe, int spitch, uint8_t *dst, size_t dframe, int dpitch,
                   int rows, int cols, int nframes) {
  const int n = ((cols + 15) >> 4) * rows;
  hipLaunchKernelGGL(k_ingest, dim3((n + 255) / 256, nframes), dim3(256), 0, s, src, sframe, spitch, dst, dframe, dpitch,
                     rows, cols);
}

// per-call reset of the candidate / selection counters (a kernel rather than hipMemsetAsync so that it is
// ordered like every other stage on the stream and the stage-timing events bracket real work)
__global__ void k_zero(int *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}
void launch_zero(hipStream_t s, int *p, int n) {
  hipLaunchKernelGGL(k_zero, dim3((n + 255) / 256), dim3(256), 0, s, p, n);
}

// ------------------------------------------------------------------------------------------------
// launchers (host)
void launch_resize(hipStream_t s, uint8_t *pyr, const FrameGeom *d_fg, const Short4 *d_tab, const Src0 &s0,
                   const FrameGeom &fg, int level, int nframes) {
  const LevelGeom &D = fg.lv[level];
  dim3 grid((D.w + 255) / 256, (D.h + 3) / 4, nframes), block(64, 4);
  hipLaunchKernelGGL(k_resize, grid, block, 0, s, pyr, d_fg, d_tab, s0, level);
}
void launch_pyramid(hipStream_t s, uint8_t *pyr, const FrameGeom *d_fg, const Short4 *d_tile_tab, const Src0 &s0,
                    const PyrTile *d_tiles, int ntiles, int ldsA, int ldsB, int tabMax, int nframes, int *cand_count) {
  const int a16 = (ldsA + 15) & ~15, ab16 = a16 + ((ldsB + 15) & ~15);
  const size_t lds = ab16 + (tabMax + 1) * sizeof(Short4);  // + 1: the row loop reads one entry ahead
  // the raised limit is a per-DEVICE attribute of the function (vsg_ctx.h lds_limit_ensure)
  int dev = 0;
  hipGetDevice(&dev);
  lds_limit_ensure(0, dev, (const void *)k_pyramid, lds);
  hipLaunchKernelGGL(k_pyramid, dim3(ntiles, nframes), dim3(kPyrThreads), lds, s, pyr, d_fg,
                     d_tile_tab, s0, d_tiles, a16, ab16, cand_count);
}
template <int NT, int TP, int SP, int PRE>
static void launch_fast_t(hipStream_t s, const uint8_t *pyr, const FrameGeom *d_fg, const FastCellRec *d_recs,
                          const Src0 &s0, uint32_t *cand, int *cand_count, int *cell_count, const FrameGeom &fg, int maxVh,
                          int maxArea, int nframes, int cus) {
  // Cells per workgroup: the first cell of a workgroup waits for its tile, the others find theirs fetched -- 2, 3 and 4
  // measure the same (0.425 ms per 512 C2 frames against 0.446 with 1; 6: 0.432, 8: 0.440, the tail of a launch grows).
  // Launches that would not fill the workgroup slots a few times over (16 per CU) keep one cell per workgroup: a
  // single frame is 577 workgroups on 256 CUs.  `cus` = the CU count of the handle's device (queried once per handle).
  // VSG_FAST_K overrides (tests/test_gpu_fast_multicell.py runs every count against the oracle).
  static const int kenv = getenv("VSG_FAST_K") ? atoi(getenv("VSG_FAST_K")) : 0;
  const long long cells_total = (long long)fg.total_cells * nframes;
  const int kauto = cells_total >= 3LL * 4 * 16 * (cus > 0 ? cus : 256) ? 3 : 1;
  const int cells_per_wg = std::max(1, std::min(kenv > 0 ? kenv : kauto, fg.total_cells));
  dim3 grid((fg.total_cells + cells_per_wg - 1) / cells_per_wg, nframes), block(NT);
  // + one spare row: the necessary test reads (masked) dwords just past the last tile row
  const int tile_bytes = ((maxVh + 6 + 1) * TP + 15) & ~15, score_bytes = ((maxVh + 2) * SP + 15) & ~15;
  // the pixel queue's region also holds the 6-bit copy of the tile during the necessary test
  const size_t queue_bytes = std::max<size_t>(((size_t)maxArea * 2 + 15) & ~(size_t)15, (size_t)tile_bytes);
  const size_t lds = (size_t)tile_bytes + score_bytes + queue_bytes;
  const int th_pack = (fg.iniTh & 0xFF) | ((fg.minTh & 0xFF) << 8) | ((fg.cand_segmented ? 1 : 0) << 16);
  hipLaunchKernelGGL((k_fast_cells<NT, TP, SP, PRE>), grid, block, lds, s, pyr, d_fg, d_recs, s0, cand, cand_count,
                     cell_count, tile_bytes, score_bytes, maxArea, cells_per_wg, fg.pyr_frame_bytes, fg.total_cells,
                     fg.cand_frame, th_pack);
}
void launch_fast(hipStream_t s, const uint8_t *pyr, const FrameGeom *d_fg, const FastCellRec *d_recs, const Src0 &s0,
                 uint32_t *cand, int *cand_count, int *cell_count, const FrameGeom &fg, int maxVh, int maxVw, int maxArea,
                 int nframes, int cus) {
  // tile row = up to 3 alignment bytes + vw + 6 ring bytes, rounded up to dwords; score rows (vw + 2 used) have the
  // tile's pitch: a pixel's score sits a constant away from its tile byte (k_fast_cells phase 2)
  if (maxVw <= 40)
    launch_fast_t<VSG_FAST_NT, 52, 52, 2 * (128 / VSG_FAST_NT)>(s, pyr, d_fg, d_recs, s0, cand, cand_count, cell_count, fg, maxVh, maxArea, nframes, cus);
  else if (maxVw <= 56)
    launch_fast_t<VSG_FAST_NT, 68, 68, 3 * (128 / VSG_FAST_NT)>(s, pyr, d_fg, d_recs, s0, cand, cand_count, cell_count, fg, maxVh, maxArea, nframes, cus);
  else
    launch_fast_t<VSG_FAST_NT, 84, 84, 4 * (128 / VSG_FAST_NT)>(s, pyr, d_fg, d_recs, s0, cand, cand_count, cell_count, fg, maxVh, maxArea, nframes, cus);
}
// dynamic LDS of one octree workgroup (also what every blur workgroup of the fused launch is charged)
size_t octree_lds_bytes(const FrameGeom &fg, int maxQuota, int maxCellsPerLevel) {
  const int cap = octree::node_capacity(maxQuota);
  // the workspace with the SMALL histogram: what the fused-launch gate of vsg_orb.hip was measured on (the launch itself
  // takes the big one where a workgroup's share of the CU has the room: launch_octree)
  const size_t prefix_off = (octree::work_bytes(cap, false) + 15) & ~(size_t)15;
  return prefix_off + (fg.cand_segmented ? (2 * (size_t)maxCellsPerLevel + 1) * 4 : 0);
}

void launch_octree(hipStream_t s, const FrameGeom *d_fg, const uint32_t *cand, const int *cand_count,
                   const CellDesc *d_cells, const int *cell_count, uint32_t *cand2, uint16_t *node_of, uint32_t *sel,
                   int *sel_count, const FrameGeom &fg, int maxQuota, int maxCellsPerLevel, int nframes,
                   const uint8_t *blur_pyr, uint8_t *blur_out, const Src0 *blur_s0) {
  const int cap = octree::node_capacity(maxQuota);
  const size_t prefix_bytes = fg.cand_segmented ? (2 * (size_t)maxCellsPerLevel + 1) * 4 : 0;
  // Node labels of levels whose candidates overflow the registers (octree_block), and the workspace class: every octree
  // launch form runs four workgroups per CU at best (4 waves per SIMD, or the workspace's own size), so a workgroup may as
  // well own a quarter of the CU's 160 KB (less the static part and a margin): the big histogram where it fits that share
  // (640x480 / 1000: 29 KB), and ~3 k labels behind it; geometries whose node arrays fill the share (1280x720 / 2000) take the
  // small histogram and keep overflowing levels' labels in global memory.
  // (measured, photo_china / rectangles at C2 / 1024: no labels 336.6 / 387.5 k frames/s, a 28 KB share 340.6 / 385.6, 32 KB
  // 346.6 / 386.9, 36 KB 346.4 / 386.4, 39 KB 346.3 / 387.2 -- profiles/r06_h_octree_large_levels.txt)
  const size_t share = 36 * 1024;
  const bool hist_big = ((octree::work_bytes(cap, true) + 15) & ~(size_t)15) + prefix_bytes <= share;
  const int prefix_off = (int)((octree::work_bytes(cap, hist_big) + 15) & ~(size_t)15);
  const size_t work_lds = (size_t)prefix_off + prefix_bytes;
  const size_t lab_off = (work_lds + 15) & ~(size_t)15;
  const size_t lab_bytes = work_lds <= 64 * 1024 && lab_off + 4096 <= share ? share - lab_off : 0;
  const size_t lds = lab_bytes ? lab_off + lab_bytes : work_lds;
  // beyond 64 KB of dynamic LDS the launch needs the limit raised (quotas above ~1000 per level, e.g. a single
  // level holding every feature); gfx950 has 160 KB per workgroup
  int dev = 0;
  hipGetDevice(&dev);
  const OctArgs a = {d_fg, cand, cand_count, d_cells, cell_count, cand2, node_of, sel, sel_count, cap, prefix_off,
                     (int)lab_off, (int)(lab_bytes / 2), hist_big ? 1 : 0};
  const bool few = nframes <= kOctFewFrames;  // a call waits for ONE workgroup per level
  if (blur_out) {  // the blur of the same frames as extra workgroups of this launch (latency path)
    // octree workgroups `lead` frames ahead of the blur's (see the kernel); only for launches long enough to have a tail
    // (64 rows, a multiple of 8 = whole XCD rounds: 317.5 -> 320.5 k frames/s at 512 C2 frames; 32 and 128 measured the same +-0.3 %)
    constexpr int kLead = 64;
    const int lead = nframes >= 4 * kLead ? kLead : 0;
    // Few-frame calls of the geometries whose levels are memory-resident as a rule (1280x720 / 2000: the workspace size that also
    // keeps their BATCHES out of this launch) take the instantiation with batched memory-form sweeps: one 1280x720 frame per
    // call 0.180 -> 0.173 ms, the building photograph 0.362 -> 0.302.  Everything else keeps the other one: with batched sweeps
    // the one-frame call of a 640x480 photograph gets 2 % shorter and the register-form calls -- every 640x480 / 752x480
    // frame of the default content -- 1-2 % LONGER, the same register-allocation effect as in the batch launch (profiles/r06_u_*)
    const dim3 grid(fg.nlevels + fg.total_blur_blocks, nframes + lead);
    if (few && 5 * work_lds > 160 * 1024) {
      lds_limit_ensure(5, dev, (const void *)k_octree_blur<VSG_OCT_MEM_BATCH>, lds);
      hipLaunchKernelGGL(k_octree_blur<VSG_OCT_MEM_BATCH>, grid, dim3(kOctThreads), lds, s, a, blur_pyr, blur_out, *blur_s0,
                         fg.nlevels, nframes, lead);
    } else {
      lds_limit_ensure(2, dev, (const void *)k_octree_blur<VSG_OCT_MEM_BATCH_FUSED>, lds);
      hipLaunchKernelGGL(k_octree_blur<VSG_OCT_MEM_BATCH_FUSED>, grid, dim3(kOctThreads), lds, s, a, blur_pyr, blur_out, *blur_s0,
                         fg.nlevels, nframes, lead);
    }
    return;
  }
  dim3 grid(fg.nlevels, nframes), block(kOctThreads);
  // The 4-waves-per-SIMD instantiation (126 VGPRs, no spills, no scratch) for calls of a few frames AND for every geometry
  // whose workspace lets five workgroups share a CU's LDS (640x480 / 1000: 0.110 ms per 512 frames against 0.137-0.161 for the
  // 5-waves one with its 59 spilled registers -- its launch time was bimodal from run to run, the only scratch user of the
  // chain); where the workspace caps a CU at four workgroups anyway (1280x720 / 2000: 33 KB) the 5-waves form measures 0.8 %
  // more frames/s (115.3-116.8 against 115.1-115.6 k, profiles/r05_p_*) and stays.
  if (few) {
    lds_limit_ensure(6, dev, (const void *)k_octree_few<VSG_OCT_MEM_BATCH>, lds);
    hipLaunchKernelGGL(k_octree_few<VSG_OCT_MEM_BATCH>, grid, block, lds, s, a);
    return;
  }
  if (5 * work_lds <= 160 * 1024) {
    lds_limit_ensure(4, dev, (const void *)k_octree_few<VSG_OCT_MEM_BATCH_FUSED>, lds);
    hipLaunchKernelGGL(k_octree_few<VSG_OCT_MEM_BATCH_FUSED>, grid, block, lds, s, a);
    return;
  }
  lds_limit_ensure(1, dev, (const void *)k_octree, lds);
  hipLaunchKernelGGL(k_octree, grid, block, lds, s, a);
}
void launch_debug_sort(hipStream_t s, uint64_t *d_items, int n) {
  hipLaunchKernelGGL(k_debug_sort, dim3(1), dim3(256), (size_t)n * 16 + 2 * (n + 2) * 2 + 32, s, d_items, n);
}
void launch_blur(hipStream_t s, const uint8_t *pyr, uint8_t *blur, const FrameGeom *d_fg, const Src0 &s0,
                 const FrameGeom &fg, int nframes) {
  dim3 grid(fg.total_blur_blocks, nframes), block(256);
  hipLaunchKernelGGL(k_blur, grid, block, 0, s, pyr, blur, d_fg, s0);
}
void launch_slots(hipStream_t s, const FrameGeom *d_fg, const uint32_t *sel, const int *sel_count, int *flags,
                  int4 *slots, FrameHeader *hdr, int lap0, int lap1, int nframes) {
  hipLaunchKernelGGL(k_slots, dim3(nframes), dim3(256), 0, s, d_fg, sel, sel_count, flags, slots, hdr, lap0, lap1);
}
void launch_orient_desc(hipStream_t s, const uint8_t *pyr, const uint8_t *blur, const FrameGeom *d_fg, const Src0 &s0,
                        const uint32_t *sel, const int *sel_count, const int4 *slots, const FrameHeader *hdr,
                        const int8_t *pattern, KeyPointPOD *kps, uint8_t *desc, int *counts, int capacity,
                        const FrameGeom &fg, int nframes, const OutMirror &mir, bool self_slots) {
  dim3 grid((fg.out_cap + 4 * kOdKpPerWave - 1) / (4 * kOdKpPerWave), nframes), block(256);
  if (self_slots && mir.kps)
    hipLaunchKernelGGL((k_orient_desc<true, true>), grid, block, 0, s, pyr, blur, d_fg, s0, sel, sel_count, slots, hdr, pattern,
                       kps, desc, counts, capacity, mir);
  else if (self_slots)
    hipLaunchKernelGGL((k_orient_desc<false, true>), grid, block, 0, s, pyr, blur, d_fg, s0, sel, sel_count, slots, hdr, pattern,
                       kps, desc, counts, capacity, mir);
  else if (mir.kps)
    hipLaunchKernelGGL((k_orient_desc<true, false>), grid, block, 0, s, pyr, blur, d_fg, s0, sel, sel_count, slots, hdr, pattern,
                       kps, desc, counts, capacity, mir);
  else
    hipLaunchKernelGGL((k_orient_desc<false, false>), grid, block, 0, s, pyr, blur, d_fg, s0, sel, sel_count, slots, hdr, pattern,
                       kps, desc, counts, capacity, mir);
}
void launch_border_copy(hipStream_t s, const uint8_t *img, int w, int h, int pitch, uint8_t *dst, int dpitch, int b) {
  dim3 grid((w + 2 * b + 255) / 256, h + 2 * b), block(256);
  hipLaunchKernelGGL(k_border_copy, grid, block, 0, s, img, w, h, pitch, dst, dpitch, b);
}

}  // namespace vsg
