// vsg_kernels.h -- host-callable launchers of the extractor kernels (vsg_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "vsg_common.h"
#include "vsg_geometry.h"

namespace vsg {
// second destination of k_orient_desc's records: pinned host memory the device writes directly on the latency path of
// the blocking entry points (only the n records of a frame cross PCIe, and no export launch follows); all null: none
struct OutMirror {
  KeyPointPOD *kps = nullptr;
  uint8_t *desc = nullptr;
  int *counts = nullptr;
  int capacity = 0;
};
void launch_stereo(hipStream_t s, const PyrView &pl, const PyrView &pr, float mb, float mbf, const float *scale,
                   const float *invScale, int nlevels, const KeyPointPOD *kpsL, const uint8_t *descL, int nL,
                   const KeyPointPOD *kpsR, const uint8_t *descR, int nR, float *uRight, float *depth, int *sadBest);
void launch_cvt_gray(hipStream_t s, const uint8_t *src, size_t sframe, int spitch, int channels, int rgb_order, int rows,
                     int cols, uint8_t *dst, size_t dframe, int dpitch, const int coeffs[3], int shift, int nframes);
void launch_zero(hipStream_t s, int *p, int n);
void launch_ingest(hipStream_t s, const uint8_t *src, size_t sframe, int spitch, uint8_t *dst, size_t dframe, int dpitch,
                   int rows, int cols, int nframes);
void launch_export(hipStream_t s, const KeyPointPOD *kps, const uint8_t *desc, const int *counts, int src_cap,
                   void *h_kps, void *h_desc, int *h_counts, int dst_cap, int nframes);
void launch_resize(hipStream_t s, uint8_t *pyr, const FrameGeom *d_fg, const Short4 *d_tab, const Src0 &s0,
                   const FrameGeom &fg, int level, int nframes);
void launch_pyramid(hipStream_t s, uint8_t *pyr, const FrameGeom *d_fg, const Short4 *d_tile_tab, const Src0 &s0,
                    const PyrTile *d_tiles, int ntiles, int ldsA, int ldsB, int tabMax, int nframes, int *cand_count);
void launch_fast(hipStream_t s, const uint8_t *pyr, const FrameGeom *d_fg, const FastCellRec *d_recs, const Src0 &s0,
                 uint32_t *cand, int *cand_count, int *cell_count, const FrameGeom &fg, int maxVh, int maxVw, int maxArea,
                 int nframes, int cus);
void launch_octree(hipStream_t s, const FrameGeom *d_fg, const uint32_t *cand, const int *cand_count,
                   const CellDesc *d_cells, const int *cell_count, uint32_t *cand2, uint16_t *node_of, uint32_t *sel,
                   int *sel_count, const FrameGeom &fg, int maxQuota, int maxCellsPerLevel, int nframes,
                   const uint8_t *blur_pyr = nullptr, uint8_t *blur_out = nullptr, const Src0 *blur_s0 = nullptr);
size_t octree_lds_bytes(const FrameGeom &fg, int maxQuota, int maxCellsPerLevel);
// waves per SIMD the fused octree + blur launch is compiled for = its 256-thread workgroups per CU (the LDS they need decides
// whether a geometry takes that launch: vsg_orb.hip)
#ifndef VSG_OB_WAVES
#define VSG_OB_WAVES 4
#endif
constexpr int kOctBlurWaves = VSG_OB_WAVES;
void launch_debug_sort(hipStream_t s, uint64_t *d_items, int n);
void launch_blur(hipStream_t s, const uint8_t *pyr, uint8_t *blur, const FrameGeom *d_fg, const Src0 &s0,
                 const FrameGeom &fg, int nframes);
void launch_slots(hipStream_t s, const FrameGeom *d_fg, const uint32_t *sel, const int *sel_count, int *flags,
                  int4 *slots, FrameHeader *hdr, int lap0, int lap1, int nframes);
void launch_orient_desc(hipStream_t s, const uint8_t *pyr, const uint8_t *blur, const FrameGeom *d_fg, const Src0 &s0,
                        const uint32_t *sel, const int *sel_count, const int4 *slots, const FrameHeader *hdr,
                        const int8_t *pattern, KeyPointPOD *kps, uint8_t *desc, int *counts, int capacity,
                        const FrameGeom &fg, int nframes, const OutMirror &mir, bool self_slots);
void launch_border_copy(hipStream_t s, const uint8_t *img, int w, int h, int pitch, uint8_t *dst, int dpitch, int b);
}  // namespace vsg
