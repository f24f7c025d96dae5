// vsg_frame_int.h -- internal layout of a device-resident Frame / KeyFrame feature set (include/vsg_orb.h: vsg_frame)
// and the window-search launcher shared by vsg_frame.hip, vsg_match.hip, vsg_bow.hip and vsg_orb.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/vsg_orb.h"
#include "vsg_common.h"
#include "vsg_ctx.h"
#include "vsg_walks.h"

enum { kGridCols = 64, kGridRows = 48, kGridCells = kGridCols * kGridRows };  // FRAME_GRID_COLS / ROWS (Frame.h:49-50)
enum { kGridLdsMax = 4096 };  // keypoints per frame up to which k_frame_grid_build orders the cells in LDS

namespace vsg {
// One entry of a grid cell's vector (mGrid[ix][iy][j], Frame.h:290) with the keypoint fields GetFeaturesInArea tests
// stored inline: a window walk reads cell_start -> entries and never chases the keypoint array.
struct GridEnt {
  float x, y;   // kpUn.pt
  uint32_t io;  // feature index (grid-local) | octave << 16
};
}  // namespace vsg

// What the searches read of one Frame (Frame.h:280-290) or KeyFrame, resident on the device.
struct vsg_frame {
  int device = 0, capacity = 0;
  int n = 0, nleft = -1;  // N, Nleft (right-camera features are [nleft, n))
  bool has_uright = false;
  float minX = 0, minY = 0, maxX = 0, maxY = 0, invW = 0, invH = 0;  // mnMinX.., mfGridElementWidthInv / HeightInv
  uint8_t *d_block = nullptr;  // one allocation behind all of the pointers below
  vsg::KeyPointPOD *d_kps = nullptr;
  uint8_t *d_desc = nullptr;
  float *d_uright = nullptr;
  int *d_cell_start[2] = {nullptr, nullptr};  // [0] mGrid, [1] mGridRight: CSR over cells ix * 48 + iy
  vsg::GridEnt *d_ent[2] = {nullptr, nullptr};
  std::vector<vsg_keypoint> h_kps;  // host mirror (angle / octave for the ordered host passes)
  // Frame::mFeatVec (Frame.h:196), resident since round 6: written by the assembly kernel of ComputeBoW (vsg_bow.hip), joined
  // with another frame's by the SearchByBoW kernels without a host round trip.  hdr = {nodes, features listed}
  int *d_fv_hdr = nullptr, *d_fv_node = nullptr, *d_fv_off = nullptr, *d_fv_idx = nullptr;
  bool fv_valid = false;  // a ComputeBoW of the CURRENT features has been enqueued on the owning thread's stream
  int fv_bound = 0;       // upper bound of the FeatureVector's node count (what the join kernels launch for)
};

namespace vsg {

// device view of a frame, passed to kernels by value
struct FrameDev {
  const KeyPointPOD *kps;
  const uint8_t *desc;
  const float *uright;  // nullptr: every mvuRight is -1
  const int *cell_start[2];
  const GridEnt *ent[2];
  int n, nleft;
  float minX, minY, invW, invH;
};
FrameDev frame_dev(const vsg_frame *f);

// One GetFeaturesInArea window + the static candidate filters of a search routine.
struct WinQuery {
  float x, y, r;    // Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel, bRight)   (Frame.cc:802-868)
  int minL, maxL;   //   (-1, -1 = KeyFrame::GetFeaturesInArea, KeyFrame.cc:834-874)
  int lo, hi;       // kpLevel < lo || kpLevel > hi -> skip (hi < 0: no such filter)       e.g. ORBmatcher.cc:506-509
  float ur, gate;   // projected right coordinate + threshold of the stereo gates           e.g. ORBmatcher.cc:97-102
  int flags;        // bit 0: bRight (mGridRight, indices + Nleft); bit 1: inactive query (empty list)
  int pad0, pad1;
};
static_assert(sizeof(WinQuery) == 48, "WinQuery layout");

enum { kWinList = 0, kWinBest = 1 };
enum { kGateNone = 0, kGateUr = 1, kGateChi2 = 2 };

// One window-search call on the calling thread's stream: begin() lays the pinned arena out and returns host pointers
// for the caller to fill (queries, descriptors); launch() enqueues the kernel; finish() syncs and, for lists that
// overflowed their stride, re-runs with a larger one.
struct WindowCall {
  ThreadCtx *c = nullptr;
  int nq = 0, mode = kWinList, cap = 0;  // cap: entries the compact candidate array holds (list mode)
  size_t oQ = 0, oD = 0, oOff = 0, oCnt = 0, oOut = 0;
  bool with_desc = true;
  // launch parameters remembered for the overflow re-run
  const vsg_frame *frame = nullptr;
  int gate_mode = kGateNone, best_init = 256;
  float inv_sigma2[16] = {};
  size_t base = 0;  // offset of this call's blocks inside the arena (several calls can share one arena)
  bool range_open = false;  // a roctx range pushed by begin() and not yet popped (every exit path pops exactly once)
  ~WindowCall();

  int begin(int device, int nq, int mode, bool with_desc, size_t arena_base = 0, size_t arena_extra = 0);
  WinQuery *queries() const { return (WinQuery *)(c->h_pin + base + oQ); }
  uint8_t *desc() const { return c->h_pin + base + oD; }
  int launch(const vsg_frame *f, int gate_mode, int best_init, const float *inv_sigma2, int nlevels,
             const uint8_t *qdesc_dev = nullptr);  // qdesc_dev: query descriptors already on the device
  int finish();  // completion (pinned flag written by the kernel's last wave, or the stream) + overflow handling
  size_t bytes() const;
  walk::CandView lists() const;
  const int32_t *best() const { return (const int32_t *)(c->h_pin + base + oOut); }  // pairs {idx, dist}
};

// One ComputeBoW in flight on a thread's arena (vsg_bow.hip): enqueue -> [other work of the same Frame] -> ONE wait -> finish
struct BowCall {
  vsg_vocab *voc = nullptr;
  ThreadCtx *c = nullptr;
  int n = 0;
  bool active = false, device_assembly = false;
  size_t pin_base = 0, oW = 0, oWord = 0, oNode = 0, oHdr = 0, oBowId = 0, oBowVal = 0, oFvNode = 0, oFvOff = 0, oFvIdx = 0;
};
void bow_sizes(int n, bool host_desc, size_t *pin_bytes, size_t *dev_bytes);
int bow_enqueue(BowCall *b, vsg_vocab *voc, const uint8_t *desc, const uint8_t *d_desc, int n, int levelsup,
                vsg_frame *resident, ThreadCtx *c, size_t pin_base, size_t dev_base);
int bow_finish(BowCall *b, int32_t *bow_ids, double *bow_vals, int bow_cap, int *n_bow, int32_t *fv_node, int32_t *fv_off,
               int32_t *fv_idx, int fv_cap, int *n_fv, int32_t *word_of, int32_t *node_of, double *weight_of);
int vocab_device(const vsg_vocab *v);

// One SearchByBoW on two frames with resident FeatureVectors (vsg_match.hip), in the same two halves
struct BowSearchCall {
  ThreadCtx *c = nullptr;
  vsg_frame *A = nullptr, *B = nullptr;
  int mode = 0, nOut = 0;
  bool active = false;
  size_t pin_base = 0, oOut = 0;
};
void bow_search_sizes(int nA, int nB, int mode, size_t *pin_bytes, size_t *dev_bytes);
int bow_search_enqueue(BowSearchCall *s, int mode, vsg_frame *A, const uint8_t *validA, vsg_frame *B, const uint8_t *validB,
                       float nnratio, ThreadCtx *c, size_t pin_base, size_t dev_base);
int bow_search_finish(BowSearchCall *s, int check_orientation, int32_t *out);

// view of the handle's outputs of the last extract call (vsg_orb.hip)
struct OrbOutputView {
  const KeyPointPOD *d_kps = nullptr;  // frame `index`
  const uint8_t *d_desc = nullptr;
  const int *d_counts = nullptr;       // {n, monoIndex}
  hipEvent_t done = nullptr;           // recorded after the last kernel of that call
  int device = 0;
};

}  // namespace vsg

int vsg_orb_output_view(vsg_orb *h, int index, vsg::OrbOutputView *v);
int vsg_orb_device_of(const vsg_orb *h);  // the device the handle lives on
// Work a blocking extract call enqueues on the handle's stream BEHIND its stage chain and in front of the completion event
// the call waits for (vsg_orb_extract_to_frame: the resident frame's grid launch rides in operator()'s one wait).  The
// hook is consumed by the next submit; `v` = frame 0 of that call's outputs.
typedef int (*vsg_post_chain_fn)(void *ctx, hipStream_t stream, const vsg::OrbOutputView &v);
void vsg_orb_set_post_chain(vsg_orb *h, vsg_post_chain_fn fn, void *ctx);
