/*
 * vsg_orb.h -- C ABI of the MI355X-native ORB front-end (libvsg_orb.so).
 *
 * Drop-in boundary for the per-frame hot path of snt-arg/visual_sgraphs (vS-Graphs on
 * ORB-SLAM3).  Each entry point names the reference interface it replaces (paths relative to
 * the reference tree).  Plain pointers and sizes only; nothing throws across this boundary;
 * every function returns a negative VSG_ERR_* code on failure.  There is NO CPU fallback:
 * without a HIP device every compute entry point fails with VSG_ERR_NO_DEVICE.
 *
 * Threading: a handle is not re-entrant (like one ORBextractor instance, which mutates
 * mvImagePyramid); distinct handles may be used concurrently from different host threads,
 * on the same or different GPUs (the stereo Frame ctor does exactly that, Frame.cc:129-132).
 * Matcher entry points are stateless and thread-safe.
 */
#ifndef VSG_ORB_H
#define VSG_ORB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSG_OK 0
#define VSG_ERR_EMPTY_IMAGE (-1)   /* operator() returns -1 on an empty image (ORBextractor.cc:1087-1088) */
#define VSG_ERR_CAPACITY (-2)      /* caller's keypoint/descriptor capacity is too small */
#define VSG_ERR_UNSUPPORTED (-3)   /* parameters/image size the reference itself cannot process, or over a limit */
#define VSG_ERR_NO_DEVICE (-4)     /* no usable HIP device / HIP runtime error at start-up */
#define VSG_ERR_HIP (-5)           /* a HIP call failed; see vsg_last_error() */
#define VSG_ERR_INVALID (-6)       /* bad argument */
#define VSG_ERR_BUSY (-7)          /* submitted batches are still un-waited: vsg_orb_wait them, then retry (nothing to grow) */

/* cv::KeyPoint-compatible record: pt.x pt.y size angle response octave class_id (28 bytes),
 * so a C++ adaptor can memcpy into std::vector<cv::KeyPoint>. */
typedef struct vsg_keypoint {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} vsg_keypoint;

typedef struct vsg_orb vsg_orb; /* one ORBextractor instance + its device buffers and streams */

const char *vsg_last_error(void);   /* thread-local description of the last failure */
int vsg_device_count(void);         /* number of HIP devices, or VSG_ERR_NO_DEVICE */

/* ---- ORBextractor ------------------------------------------------------------------------- */

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * (ORBextractor.h:51-52, ORBextractor.cc:411-470).  `device` = HIP device ordinal;
 * `max_batch` = frames processed per launch by the batch entry points (>= 1). */
int vsg_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast, int device,
                   int max_batch, vsg_orb **out);
void vsg_orb_destroy(vsg_orb *h);

/* GetLevels/GetScaleFactors/GetInverseScaleFactors/GetScaleSigmaSquares/GetInverseScaleSigmaSquares
 * (ORBextractor.h:63-91) plus mnFeaturesPerLevel and umax[16]; any pointer may be NULL.
 * Returns nlevels.  No device needed. */
int vsg_orb_get_tables(const vsg_orb *h, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                       int *features_per_level, int *umax16);

/* The 7 taps of cv::GaussianBlur(7x7, sigma 2)'s 8.8 fixed-point kernel (ORBextractor.cc:1130).
 * Default {18,34,49,55,49,34,18} (OpenCV 4.2.0 independent rounding); OpenCV >= 4.3 normalises to
 * {18,34,48,56,48,34,18}.  Version-sensitive, hence data. */
int vsg_orb_set_blur_taps(vsg_orb *h, const uint16_t taps[7]);

/* Launch form of ComputePyramid (ORBextractor.cc:1171-1195): -1 (default) = the fused chain kernel with the tiling the
 * handle picks (the one it timed faster for its image size; 16-pixel top tiles for calls of one or two frames),
 * 0 / 1 / 2 = the fused kernel with the 32- / 36- / 16-pixel top-level tiling, 3 = one launch per level.  Every form
 * produces the same bytes (tests/test_gpu_switches.py runs all five against the oracle); the setter exists so that each
 * form stays a tested path instead of an environment switch. */
int vsg_orb_set_pyramid_tiling(vsg_orb *h, int which);

/* Keypoint capacity a caller must provide per frame for images of this size (>= nfeatures + 3*nlevels,
 * ORBextractor.cc:692,753-754), or a VSG_ERR_* code. */
int vsg_orb_capacity(vsg_orb *h, int rows, int cols);

/* int ORBextractor::operator()(image, mask, keypoints, descriptors, vLappingArea)
 * (ORBextractor.h:59-61, ORBextractor.cc:1083-1169).  gray: CV_8UC1 host image, `stride` bytes per row.
 * lap0/lap1 = vLappingArea.  kps[capacity], desc[capacity*32] are host buffers owned by the caller.
 * Returns monoIndex (>= 0) like the reference, VSG_ERR_EMPTY_IMAGE (-1) for an empty image, or another
 * VSG_ERR_*.  *n = number of keypoints (rows of the descriptor Mat; 0 => the reference release()s it). */
int vsg_orb_extract(vsg_orb *h, const uint8_t *gray, int rows, int cols, int stride, int lap0, int lap1,
                    vsg_keypoint *kps, uint8_t *desc, int capacity, int *n);

/* Batched operator(): nframes (<= max_batch) images of equal size at gray + i*frame_stride.
 * Outputs are [nframes][capacity] records; n[i], mono_index[i] per frame.  Host pointers. */
int vsg_orb_extract_batch(vsg_orb *h, const uint8_t *gray, int nframes, size_t frame_stride, int rows, int cols,
                          int stride, int lap0, int lap1, vsg_keypoint *kps, uint8_t *desc, int capacity, int *n,
                          int *mono_index);

/* Device-resident batched operator() for throughput pipelines and multi-GPU sharding:
 * d_gray, d_kps ([nframes][capacity]), d_desc ([nframes][capacity][32]) and d_counts ([nframes][2] =
 * {n, monoIndex}) are DEVICE pointers on the handle's device.  Work is enqueued asynchronously behind
 * `stream` (a hipStream_t) and completes in stream order.  stream == NULL means the caller's NULL stream: the chain is
 * ordered behind everything the NULL stream holds at entry and the NULL stream waits for it at exit, so
 * producer -> extract -> vsg_hamming_block_best2_device(stream = NULL) sequences are ordered.  Level 0 is read in
 * place: d_gray must stay alive until the work has completed (and for image_pyramid(0) / stereo read-back after it).
 * Descriptors stay resident for the matcher entry points. */
int vsg_orb_extract_batch_device(vsg_orb *h, const uint8_t *d_gray, int nframes, size_t frame_stride, int rows,
                                 int cols, int stride, int lap0, int lap1, vsg_keypoint *d_kps, uint8_t *d_desc,
                                 int *d_counts, int capacity, void *stream);

/* Colour input: Tracking::GrabImage{Monocular,Stereo,RGBD} converts with cv::cvtColor(...,
 * COLOR_{RGB,BGR,RGBA,BGRA}2GRAY) before building the Frame (Tracking.cc:1526-1551, 1595-1608, 1643-1656).  This
 * entry takes the interleaved 8-bit colour frames (channels = 3 or 4; rgb_order != 0 for RGB(A), 0 for BGR(A) =
 * Tracking::mbRGB) on the device and fuses the conversion into the level-0 staging.  Coefficients are data
 * (default = OpenCV 4.2: R2Y 4899, G2Y 9617, B2Y 1868, shift 14); they must sum to 1 << shift. */
int vsg_orb_set_gray_coeffs(vsg_orb *h, const int coeffs[3], int shift);
int vsg_orb_extract_batch_device_color(vsg_orb *h, const uint8_t *d_img, int channels, int rgb_order, int nframes,
                                       size_t frame_stride, int rows, int cols, int stride, int lap0, int lap1,
                                       vsg_keypoint *d_kps, uint8_t *d_desc, int *d_counts, int capacity,
                                       void *stream);
/* same with host pointers (colour cv::Mat frames in, keypoints/descriptors out) */
int vsg_orb_extract_batch_color(vsg_orb *h, const uint8_t *img, int channels, int rgb_order, int nframes,
                                size_t frame_stride, int rows, int cols, int stride, int lap0, int lap1,
                                vsg_keypoint *kps, uint8_t *desc, int capacity, int *n, int *mono_index);

/* mvImagePyramid[level] (public member, ORBextractor.h:93; read by Frame::ComputeStereoMatches,
 * Frame.cc:964,1054-1069) of frame `frame` of the last call.  with_border != 0 copies the
 * (w+38)x(h+38) buffer including the 19 px BORDER_REFLECT_101 frame (ORBextractor.cc:1186-1192). */
int vsg_orb_level_size(vsg_orb *h, int level, int *w, int *ht);
int vsg_orb_copy_pyramid_level(vsg_orb *h, int frame, int level, int with_border, uint8_t *dst, int dst_stride);

/* All levels of mvImagePyramid of frame `frame` of the last call in ONE device-to-host copy, each WITH its 19 px
 * border, packed back to back: level l starts at offsets[l] (nlevels entries) with a row stride of width + 38.  With
 * dst == NULL (or dst_bytes too small) nothing is written and the byte count needed is returned; otherwise the bytes
 * written.  This is what the adaptor's DownloadPyramid() uses. */
int vsg_orb_copy_pyramid(vsg_orb *h, int frame, uint8_t *dst, size_t dst_bytes, size_t *offsets);

/* Stage read-back of the last call, for stage-by-stage parity tests: the blurred level
 * (ORBextractor.cc:1129-1130), the FAST candidates (vToDistributeKeys, :868-873; unordered,
 * packed x | y<<12 | response<<24 relative to (16,16)) and the octree selection in list order. */
int vsg_orb_copy_blurred_level(vsg_orb *h, int frame, int level, uint8_t *dst, int dst_stride);
int vsg_orb_copy_candidates(vsg_orb *h, int frame, int level, uint32_t *dst, int cap);
int vsg_orb_copy_selected(vsg_orb *h, int frame, int level, uint32_t *dst, int cap);

/* Average device time per stage of the last N timed calls (HIP events on the handle's streams).
 * names: "pyramid","fast","octree","blur","slots","orient_desc","total".  Returns number of stages.
 * vsg_orb_enable_timing: 0 = off; 1 = events around every stage (the blur then runs as its own launch on its own stream
 * instead of riding in the octree's launch, so the chain is a few % slower than an untimed call); 2 = events around the
 * FAST launch only, everything else exactly as in an untimed call (only "fast" is filled in). */
/* Tracing: VSG_ROCTX=1 in the environment wraps the stage chain and the entry points in roctx ranges for
 * rocprofv3 --marker-trace.  vsg_orb_time_stats is the REGISTER_TIMES analogue (Settings.h:23): mean / std in ms of
 * the host wall time of the blocking operator() calls so far ("ORB Extraction" in Tracking::PrintTimeStats,
 * Tracking.cc:291-330; Frame::mTimeORB_Ext, Frame.cc:126-137).  Returns the number of calls; reset != 0 clears. */
int vsg_orb_time_stats(vsg_orb *h, double *mean_ms, double *std_ms, int reset);
int vsg_orb_enable_timing(vsg_orb *h, int enable);
/* serialize != 0: enqueue every kernel on one stream (no blur overlap), so that stage timings and rocprof kernel
 * durations are free of interference from concurrent kernels.  Results are identical either way. */
int vsg_orb_set_serialize(vsg_orb *h, int serialize);
int vsg_orb_get_timing(vsg_orb *h, float *ms_out, int cap);

/* void Frame::ComputeStereoMatches() (Frame.cc:957-1127) for a rectified pair.  hl/hr = the
 * extractors that just processed the left/right image (their mvImagePyramid is read on the device; one handle
 * with a 2-frame batch works too: frame_l/frame_r index the batch).  kps/desc = the operator() outputs (host).
 * u_right[n_l], depth[n_l] receive mvuRight / mvDepth (-1 where no match); mb, mbf as in Frame.  Returns the
 * number of stereo matches kept. */
int vsg_stereo_matches(vsg_orb *hl, int frame_l, vsg_orb *hr, int frame_r, const vsg_keypoint *kps_l,
                       const uint8_t *desc_l, int n_l, const vsg_keypoint *kps_r, const uint8_t *desc_r, int n_r,
                       float mb, float mbf, float *u_right, float *depth);

/* ---- ORBmatcher (flattened POD views; pointer-graph walking and geometry stay in the C++ adaptor) ---- */

/* static int ORBmatcher::DescriptorDistance(a, b) (ORBmatcher.h:40, ORBmatcher.cc:2047-2063)
 * for npairs row pairs: dist[i] = hamming(a[ia[i]], b[ib[i]]).  Host pointers. */
int vsg_hamming_pairs(int device, const uint8_t *a, int na, const uint8_t *b, int nb, const int32_t *ia,
                      const int32_t *ib, int npairs, int32_t *dist);

/* Brute-force best / second-best of every row of A against all rows of B: the inner loop of
 * SearchByBoW (ORBmatcher.cc:276-299, 810-841) for one vocabulary node without greedy state.
 * Ties resolve to the lowest B index (strict '<' scan).  Host pointers. */
int vsg_hamming_block_best2(int device, const uint8_t *a, int na, const uint8_t *b, int nb, int32_t *best,
                            int32_t *second, int32_t *argbest);
/* Same on device pointers, batched over `nblocks` independent (A_i, B_i) blocks laid out with fixed
 * strides (rows per block: na[i], nb[i] from d_counts[i*count_stride]); asynchronous on `stream`. */
int vsg_hamming_block_best2_device(int device, const uint8_t *d_a, const uint8_t *d_b, size_t block_stride_bytes,
                                   const int32_t *d_counts_a, const int32_t *d_counts_b, int count_stride,
                                   int nblocks, int max_rows, int32_t *d_best, int32_t *d_second,
                                   int32_t *d_argbest, void *stream);

/* int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vpMapPointMatches) (ORBmatcher.h:64,
 * ORBmatcher.cc:226-428), F.Nleft == -1.  FeatureVectors (DBoW2 std::map<NodeId, vector<unsigned>>)
 * are CSR: node ids ascending, offsets[nodes+1], indices.  kf_valid[i] = (pMP && !pMP->isBad()).
 * match_f[i] = KF feature index whose MapPoint F feature i received, or -1.  Returns nmatches. */
int vsg_search_by_bow_kf_f(int device, const uint8_t *kf_desc, const float *kf_angle, const uint8_t *kf_valid,
                           int n_kf, const int32_t *kf_node_id, const int32_t *kf_off, const int32_t *kf_idx,
                           int kf_nodes, const uint8_t *f_desc, const float *f_angle, int n_f,
                           const int32_t *f_node_id, const int32_t *f_off, const int32_t *f_idx, int f_nodes,
                           float nnratio, int check_orientation, int32_t *match_f);

/* The same for a fisheye-stereo Frame (F.Nleft != -1, ORBmatcher.cc:277-326, 362-389): features [0, f_nleft) come
 * from the left camera, [f_nleft, n_f) from the right (descriptors vconcat'ed, Frame.cc:296).  Left and right
 * candidates keep separate best / second-best pairs; the right best is accepted at dist <= TH_LOW without a ratio
 * test, but only when the left best also is <= TH_LOW (the reference nests the block).  f_nleft = -1 is the call
 * above. */
int vsg_search_by_bow_kf_f_stereo(int device, const uint8_t *kf_desc, const float *kf_angle, const uint8_t *kf_valid,
                                  int n_kf, const int32_t *kf_node_id, const int32_t *kf_off, const int32_t *kf_idx,
                                  int kf_nodes, const uint8_t *f_desc, const float *f_angle, int n_f, int f_nleft,
                                  const int32_t *f_node_id, const int32_t *f_off, const int32_t *f_idx, int f_nodes,
                                  float nnratio, int check_orientation, int32_t *match_f);

/* int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12) (ORBmatcher.h:65,
 * ORBmatcher.cc:758-900), NLeft == -1.  matches12[idx1] = idx2 or -1. */
int vsg_search_by_bow_kf_kf(int device, const uint8_t *desc1, const float *angle1, const uint8_t *valid1, int n1,
                            const int32_t *node_id1, const int32_t *off1, const int32_t *idx1, int nodes1,
                            const uint8_t *desc2, const float *angle2, const uint8_t *valid2, int n2,
                            const int32_t *node_id2, const int32_t *off2, const int32_t *idx2, int nodes2,
                            float nnratio, int check_orientation, int32_t *matches12);

/* int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, vMatchedPairs, bOnlyStereo, bCoarse)
 * (ORBmatcher.h:72, ORBmatcher.cc:902-1146), NLeft == -1.  eligible1[i] = !pKF1->GetMapPoint(i) && (!bOnlyStereo ||
 * stereo(i)) (:969-979), eligible2 likewise (:998-1005; the reference never sets vbMatched2).  The geometric
 * predicate of :1031-1071 (epipole distance gate, then bCoarse || epipolarConstrain) is a pure function of the pair
 * and stays with the adaptor: for the s-th SHARED vocabulary node (ascending node id = merge-join order), bit
 * pair_off[s] + i1 * n2(s) + i2 of pair_ok says whether (node list position i1 of KF1, i2 of KF2) passes;
 * pair_off has (shared nodes + 1) entries; pair_ok == NULL means every pair passes.  Among the passing candidates
 * with dist <= TH_LOW the smallest distance wins, the LATER one on ties (`dist > bestDist` skips, :1015).
 * matches12[idx1] = idx2 or -1 (vMatchedPairs = the non-negative entries in index order).  Returns nmatches after
 * the rotation-histogram filter. */
int vsg_search_for_triangulation(int device, const uint8_t *desc1, const float *angle1, const uint8_t *eligible1,
                                 int n1, const int32_t *node_id1, const int32_t *off1, const int32_t *idx1, int nodes1,
                                 const uint8_t *desc2, const float *angle2, const uint8_t *eligible2, int n2,
                                 const int32_t *node_id2, const int32_t *off2, const int32_t *idx2, int nodes2,
                                 const uint32_t *pair_ok, const int32_t *pair_off, int check_orientation,
                                 int32_t *matches12);

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)
 * (ORBmatcher.h:50, ORBmatcher.cc:1667-1878), Nleft == -1.  The adaptor projects the map points and
 * runs Frame::GetFeaturesInArea; query q brings its descriptor, keypoint angle and candidate list
 * cand_idx[cand_off[q] .. cand_off[q+1]) in GetFeaturesInArea order.  train_blocked[i] =
 * (CurrentFrame.mvpMapPoints[i] && Observations() > 0) on entry, updated in place; query_blocks[q] =
 * (pMP_q->Observations() > 0).  train_match[i] = q or left untouched (caller initialises to -1).
 * th_high = TH_HIGH (100).  Returns nmatches. */
int vsg_search_by_projection_last(int device, const uint8_t *q_desc, const float *q_angle,
                                  const uint8_t *query_blocks, int n_q, const int32_t *cand_off,
                                  const int32_t *cand_idx, const uint8_t *t_desc, const float *t_angle,
                                  uint8_t *train_blocked, int n_t, int th_high, int check_orientation,
                                  int32_t *train_match);

/* int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &, th, ...) (ORBmatcher.h:46,
 * ORBmatcher.cc:42-216), left block, Nleft == -1: best + second best with octave-aware ratio test. */
int vsg_search_by_projection_local(int device, const uint8_t *q_desc, const uint8_t *query_blocks, int n_q,
                                   const int32_t *cand_off, const int32_t *cand_idx, const uint8_t *t_desc,
                                   const int32_t *t_octave, uint8_t *train_blocked, int n_t, float nnratio,
                                   int32_t *train_match);

/* Common core of the searches that keep only the best candidate per projected MapPoint:
 *   SearchByProjection(KeyFrame*, Sim3, vpPoints, vpMatched, th, ratioHamming)   (ORBmatcher.h:58, .cc:430-528)
 *   SearchByProjection(KeyFrame*, Sim3, vpPoints, vpPointsKFs, ...)             (ORBmatcher.h:62, .cc:530-641)
 *   SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)           (ORBmatcher.h:54, .cc:1880-2000; the
 *                                                  rotation check is vsg_search_by_projection_last's check_orientation)
 *   SearchBySim3's two directional passes (.cc:1500-1640) and Fuse's candidate loops (.cc:1148-1446), whose
 *   geometric predicates (level window, chi2 of the reprojection) filter the candidate lists in the adaptor.
 * For each query in order: best (strict '<', earliest wins) over the non-blocked candidates; q_best_idx/q_best_dist per
 * query (-1 / 256 if none); if dist <= th_high the match is accepted: train_match[best] = q, train_blocked[best] =
 * query_blocks[q] (NULL: never blocks).  train_blocked / train_match may be NULL.  Returns the number accepted. */
int vsg_search_window(int device, const uint8_t *q_desc, const uint8_t *query_blocks, int n_q,
                      const int32_t *cand_off, const int32_t *cand_idx, const uint8_t *t_desc, uint8_t *train_blocked,
                      int n_t, int th_high, int32_t *q_best_idx, int32_t *q_best_dist, int32_t *train_match);

/* void MapPoint::ComputeDistinctiveDescriptors() (MapPoint.cc:340-415) for ngroups map points at once:
 * group g = descriptor rows off[g] .. off[g+1]-1 (the observations' descriptors in the order the reference pushes
 * them); best[g] = index inside the group of the descriptor with the least median distance to the rest (first one
 * on ties), -1 for an empty group.  Groups of more than 128 descriptors return VSG_ERR_UNSUPPORTED. */
int vsg_distinctive_descriptors(int device, const uint8_t *desc, const int32_t *off, int ngroups, int32_t *best);

/* ---- Frame grid: Frame::AssignFeaturesToGrid / PosInGrid (Frame.cc:521-553, 870-880) and
 * Frame::GetFeaturesInArea / KeyFrame::GetFeaturesInArea (Frame.cc:802-868, KeyFrame.cc:834-874) on the device.
 * 64 x 48 cells (Frame.h:49-50).  kps = the (undistorted) keypoints the reference indexes (host pointer). */
typedef struct vsg_grid vsg_grid;
int vsg_grid_build(int device, const vsg_keypoint *kps, int n, float min_x, float min_y, float max_x, float max_y,
                   vsg_grid **out);
void vsg_grid_destroy(vsg_grid *g);
/* nq queries (x, y, r, minLevel, maxLevel) -> CSR candidate lists in the reference's order (cells ix outer, iy
 * inner, insertion order inside a cell): cand_off[nq+1], cand_idx[cap].  min_level/max_level may be NULL (= -1,
 * KeyFrame::GetFeaturesInArea has no level filter).  Returns the total number of candidates (which may exceed
 * cap: nothing beyond cap is written) or a VSG_ERR_* code. */
int vsg_grid_query(vsg_grid *g, const float *x, const float *y, const float *r, const int32_t *min_level,
                   const int32_t *max_level, int nq, int32_t *cand_off, int32_t *cand_idx, int cap);

/* ---- DBoW2 vocabulary: ORBVocabulary::loadFromBinFile (TemplatedVocabulary.h:1478-1552, called at
 * System.cc:105-112) from an in-memory image of the .bin file, and transform(features, BowVector, FeatureVector,
 * levelsup) (TemplatedVocabulary.h:1139-1212) as called by Frame::ComputeBoW / KeyFrame::ComputeBoW with
 * levelsup = 4 (Frame.cc:882-889).  BowVector: ascending (word id, value) pairs; FeatureVector: CSR with ascending
 * node ids (the form vsg_search_by_bow_* takes); fv_idx must hold n entries, fv_off fv_cap+1.  word_of / node_of /
 * weight_of (optional) are the per-feature leaf word, node at level L - levelsup and word weight. */
typedef struct vsg_vocab vsg_vocab;
int vsg_vocab_load(int device, const uint8_t *blob, size_t size, vsg_vocab **out);
void vsg_vocab_destroy(vsg_vocab *v);
int vsg_vocab_info(const vsg_vocab *v, int *k, int *L, int *scoring, int *weighting, int *nnodes, int *nwords);
int vsg_bow_transform(vsg_vocab *voc, const uint8_t *desc, int n, int levelsup, int32_t *bow_ids, double *bow_vals,
                      int bow_cap, int *n_bow, int32_t *fv_node, int32_t *fv_off, int32_t *fv_idx, int fv_cap,
                      int *n_fv, int32_t *word_of, int32_t *node_of, double *weight_of);

/* int ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)
 * (ORBmatcher.h:68, ORBmatcher.cc:643-756); candidate lists from F2.GetFeaturesInArea per F1 keypoint. */
int vsg_search_for_initialization(int device, const uint8_t *desc1, const float *angle1, const int32_t *octave1,
                                  int n1, const int32_t *cand_off, const int32_t *cand_idx, const uint8_t *desc2,
                                  const float *angle2, int n2, float nnratio, int check_orientation,
                                  int32_t *matches12);


/* ---- Threads, streams, staging ("re-entrant, thread-safe, stream per calling thread") ----------------
 * Every matcher / grid / BoW / stereo / frame entry point runs on the CALLING THREAD's own non-blocking HIP stream and
 * stages through that thread's pinned + device arenas, which only grow: no hipMalloc / hipFree / NULL-stream launch
 * in steady state (a hipFree in LoopClosing's thread would otherwise stall Tracking's extractor streams).
 * vsg_thread_release() frees the calling thread's streams and arenas (optional; call before the thread exits).
 * vsg_thread_arena_growths(device) = number of arena (re)allocations so far on this thread (constant in steady state). */
int vsg_thread_release(void);
int vsg_thread_arena_growths(int device);
/* hipMemcpyAsync(dst, src, bytes, DeviceToDevice, stream) on `device`, for hosts that hold raw device pointers (the
 * records of vsg_shard_record) and should not bind a second copy of the HIP runtime for one copy. */
int vsg_copy_d2d_async(int device, void *dst, const void *src, size_t bytes, void *stream);

/* ---- Pinned caller memory (the buffers behind cv::Mat::data of a reused frame, Frame::mvKeys / mDescriptors rings;
 * the reference hands operator() plain heap memory: ORBextractor.h:59-61, Frame.cc:555-563) ---------------------------
 * vsg_host_alloc / vsg_host_free: pinned host memory from the runtime's own allocator (hipHostMalloc), what the handle's
 * staging slots use themselves.  Memory from vsg_host_alloc is the memory the device reads and writes IN PLACE
 * (vsg_orb_submit_batch / vsg_orb_wait, the blocking entry points): no staging copy on either side.  So is hipHostMalloc
 * memory the caller obtained elsewhere.  vsg_host_free releases ONLY pointers vsg_host_alloc returned (VSG_ERR_INVALID for
 * anything else, also for hipHostMalloc memory of another owner: since round 5 the library keeps a registry of its own blocks and
 * does not forward foreign pointers to hipHostFree).
 * vsg_host_register / vsg_host_unregister: pin ordinary heap memory after the fact (hipHostRegister).  Such memory is a
 * user-pointer mapping of heap pages; on the round-4 test pool a page of a live, registered, in-use range lost its
 * device mapping under heap churn (a fatal `Memory access fault by GPU`; profiles/r04_q_open_issue_gpu_fault.txt,
 * tools/repro_hostregister.cpp).  By DEFAULT the library therefore treats registered memory like pageable memory: it is
 * staged through the slot's own pinned buffers by the calling thread and the device never touches it.
 * vsg_orb_set_direct_registered(h, 1) opts a handle in to in-place access of registered memory as well (hosts that
 * register page-aligned, long-lived, mlock-ed arenas and have qualified their kernel / driver).
 * vsg_host_kind(ptr, bytes): how the library classifies the WHOLE range [ptr, ptr + bytes) -- both ends must be pinned
 * the same way under one contiguous device alias, otherwise the range counts as pageable. */
#define VSG_HOST_PAGEABLE 0      /* staged through the slot's pinned buffers */
#define VSG_HOST_LIB_ALLOC 1     /* inside one vsg_host_alloc range: in place */
#define VSG_HOST_HIPHOSTMALLOC 2 /* somebody else's hipHostMalloc memory: in place */
#define VSG_HOST_REGISTERED 3    /* hipHostRegister-ed heap memory: staged unless vsg_orb_set_direct_registered(h, 1) */
int vsg_host_register(void *ptr, size_t bytes);
int vsg_host_unregister(void *ptr);
int vsg_host_alloc(size_t bytes, void **out);
int vsg_host_free(void *ptr);
int vsg_host_kind(const void *ptr, size_t bytes);
int vsg_orb_set_direct_registered(vsg_orb *h, int enable);

/* ---- Asynchronous operator() batches (caller: Frame::ExtractORB, Frame.cc:555-563, in a throughput pipeline) -----
 * vsg_orb_submit_batch enqueues H2D + the stage chain + the output export of one batch into one of the handle's
 * vsg_orb_slots() (= 3) pipeline slots and returns at once with a ticket (>= 0); VSG_ERR_BUSY when every slot
 * holds a batch that has not been waited for, and when the image size differs from the handle's current one while any
 * ticket is still un-waited (a new size rebuilds the slots those tickets live in).  H2D of batch k+1 runs beside the kernels of batch k and the export of
 * batch k-1 (three streams).  kps / desc ([nframes][capacity] records, host memory) must stay valid until the wait
 * returns; if they are memory the device may touch in place (vsg_host_alloc / hipHostMalloc memory, see "Pinned caller
 * memory" above) the device writes the n[i] records of every frame straight into them, otherwise they are filled from
 * the slot's pinned staging inside vsg_orb_wait -- either way only n[i] records per frame cross PCIe, not `capacity`.
 * `gray` must stay valid until the wait returns if it is such memory; any other
 * input is copied before submit returns (batches of 16 frames and more: by the calling thread and three helper
 * threads the handle keeps asleep between batches).  vsg_orb_wait(ticket) blocks until that batch is complete and delivers n[]
 * and mono_index[] (nframes entries each).  Tickets are waited for in submission order. */
int vsg_orb_slots(const vsg_orb *h);
/* Latency path of the BLOCKING entry points (vsg_orb_extract, vsg_orb_extract_batch with <= 8 frames; the reference
 * calls operator() once per frame: System::TrackRGBD -> Frame::ExtractORB, System.cc:359, Frame.cc:555-563): the pinned
 * image is read by a kernel instead of a copy-engine transfer and k_orient_desc writes the records into the pinned
 * destination itself (no export launch, no DMA).  With VSG_GRAPH=1 in the environment the stream work of a call is additionally recorded as
 * a hipGraph on the second call with the same source / destination / lapping area and replayed with one
 * hipGraphLaunch from the third on (measured: no faster than the eager launches on ROCm 7.0, hence opt-in).
 * Results are identical in every mode.  Returns the number of graph launches so far. */
long vsg_orb_chain_graph_launches(const vsg_orb *h);
int vsg_orb_submit_batch(vsg_orb *h, const uint8_t *gray, int nframes, size_t frame_stride, int rows, int cols,
                         int stride, int lap0, int lap1, vsg_keypoint *kps, uint8_t *desc, int capacity);
int vsg_orb_wait(vsg_orb *h, int ticket, int *n, int *mono_index);

/* ---- Device-resident Frame / KeyFrame features (SURVEY 8b last row; Frame.h:280,290) ---------------------------
 * A vsg_frame keeps what the searches read of one Frame or KeyFrame on the device between calls: the keypoints the
 * grid indexes (mvKeysUn; mvKeys || mvKeysRight when Nleft != -1), mDescriptors, mvuRight and the 64 x 48 mGrid (+
 * mGridRight) as CSR -- so that a search uploads only the projected positions and the map points' descriptors.
 * A small host mirror of the keypoints serves the ordered host-side passes (rotation histogram, level ratio test).
 * Frames are immutable after upload and may be searched concurrently from several threads. */
typedef struct vsg_frame vsg_frame;
/* Limits (the packed candidate entries are index : 15 | distance : 9 | octave : 4 bits): capacity <= 32767 features per
 * frame (vsg_frame_create returns VSG_ERR_INVALID beyond, the host-candidate forms vsg_search_* return
 * VSG_ERR_UNSUPPORTED for n_t > 32767) and keypoint octaves 0..15 (vsg_frame_upload: VSG_ERR_UNSUPPORTED; the
 * extractor itself is limited to 16 levels).  The reference's settings use 1000-2000 features and 8 levels.  A frame
 * that was created but never uploaded is empty: every search on it finds nothing. */
int vsg_frame_create(int device, int capacity, vsg_frame **out);
void vsg_frame_destroy(vsg_frame *f);
/* Frame::Frame(...) after ExtractORB + UndistortKeyPoints: keys = mvKeysUn (Nleft == -1) or mvKeys followed by
 * mvKeysRight (nleft = Nleft, Frame.cc:296); u_right = mvuRight (NULL: all -1); grid bounds mnMinX.. (Frame.cc:378).
 * Builds mGrid / mGridRight exactly like AssignFeaturesToGrid (Frame.cc:521-553). */
int vsg_frame_upload(vsg_frame *f, const vsg_keypoint *keys, const uint8_t *desc, const float *u_right, int n,
                     int nleft, float min_x, float min_y, float max_x, float max_y);
/* The same straight out of the extractor, device to device, for frame `index` of the handle's last
 * vsg_orb_extract / _batch / _submit_batch call (no distortion: mvKeysUn = mvKeys, Frame.cc UndistortKeyPoints with
 * mDistCoef == 0); the grid is built by a kernel.  The host mirror is filled from kps_host (the records the extract
 * call delivered) -- nothing is downloaded. */
int vsg_frame_from_extractor(vsg_frame *f, vsg_orb *h, int index, const vsg_keypoint *kps_host, int n, float min_x,
                             float min_y, float max_x, float max_y);
/* Frame::ComputeImageBounds (Frame.cc:924-955) for a pinhole camera: K4 = {fx, fy, cx, cy} (the floats of mK), dist = the
 * 4 or 5 floats of mDistCoef (k1, k2, p1, p2[, k3]; Tracking.cc:742-787); out = {mnMinX, mnMinY, mnMaxX, mnMaxY} -- the
 * image rectangle when mDistCoef(0) == 0, otherwise the four corners through cv::undistortPoints(.., K, D, Mat(), K)
 * ([OCV 4.2]: five fixed iterations in double).  Four points, once per camera (mbInitialComputations, Frame.cc:373-392):
 * host arithmetic, no device needed. */
int vsg_camera_image_bounds(int cols, int rows, const float K4[4], const float *dist, int ndist, float out[4]);
/* vsg_frame_from_extractor for a DISTORTED camera (mDistCoef(0) != 0: config/RGB-D/TUM1.yaml, config/RGB-D-Inertial/
 * RealSense_D435i.yaml -- BASELINE configs C1 and C5): Frame::UndistortKeyPoints (Frame.cc:891-921) runs on the device,
 * in double, inside the launch that builds the grid -- the frame's keypoints are mvKeysUn, the grid indexes them with
 * the bounds of vsg_camera_image_bounds, and keys_un_out (n records, may be NULL) receives mvKeysUn for the host's own
 * use (Tracking, the optimizer).  The frame stays resident: no D2H -> cv::undistortPoints -> H2D hop.  With
 * mDistCoef(0) == 0 this is vsg_frame_from_extractor (keys_un_out = kps_host). */
int vsg_frame_from_extractor_undistort(vsg_frame *f, vsg_orb *h, int index, const vsg_keypoint *kps_host, int n,
                                       const float K4[4], const float *dist, int ndist, float min_x, float min_y,
                                       float max_x, float max_y, vsg_keypoint *keys_un_out);
/* The Frame constructor's front end in ONE call and ONE wait (Frame.cc:344-358: ExtractORB -> UndistortKeyPoints ->
 * AssignFeaturesToGrid): vsg_orb_extract + vsg_frame_from_extractor_undistort, with the grid launch enqueued on the
 * extractor's stream right behind its stage chain (it reads the keypoint count on the device), so the blocking call's
 * single wait covers both.  Arguments as the two calls'; K4 == NULL: no distortion.  Returns monoIndex like
 * vsg_orb_extract, *n = the keypoint count; VSG_ERR_CAPACITY when the frame's capacity is below *n. */
int vsg_orb_extract_to_frame(vsg_orb *h, const uint8_t *gray, int rows, int cols, int stride, int lap0, int lap1,
                             vsg_keypoint *kps, uint8_t *desc, int capacity, int *n, vsg_frame *f, const float K4[4],
                             const float *dist, int ndist, float min_x, float min_y, float max_x, float max_y,
                             vsg_keypoint *keys_un_out);
int vsg_frame_size(const vsg_frame *f);
/* test / debug read-back of the device copy: grid CSR (cell_start[64*48+1], entries[n]) of the left (0) or right (1) grid */
int vsg_frame_copy_grid(vsg_frame *f, int right, int32_t *cell_start, int32_t *entries);

/* Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel, bRight) (Frame.cc:802-868) / KeyFrame::GetFeaturesInArea
 * (KeyFrame.cc:834-874; pass min_level = max_level = NULL) for nq windows on the resident grid; CSR out like
 * vsg_grid_query.  Indices are grid-local (right grid: i - Nleft), exactly what the reference's vectors hold. */
int vsg_frame_features_in_area(vsg_frame *f, const float *x, const float *y, const float *r, const int32_t *min_level,
                               const int32_t *max_level, int right, int nq, int32_t *cand_off, int32_t *cand_idx,
                               int cap);

/* int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th, bFarPoints, thFarPoints)
 * (ORBmatcher.h:46, ORBmatcher.cc:42-216), BOTH blocks (left :59-143, right camera :146-214 when F.Nleft != -1).
 * One entry per map point that survives the caller-side tests (:50-57: in view, far points, isBad):
 *   in_view / proj_x / proj_y / proj_xr / scale_level / view_cos   = mbTrackInView, mTrackProjX/Y, mTrackProjXR,
 *   mnTrackScaleLevel, mTrackViewCos;  the *_r arrays (NULL when Nleft == -1) = the ...R members of the right camera;
 *   mp_desc = pMP->GetDescriptor(); mp_observed = pMP->Observations() > 0.
 * scale_factors = F.mvScaleFactors.  left_to_right / right_to_left = F.mvLeftToRightMatch / mvRightToLeftMatch
 * (Nleft != -1).  train_blocked[i] = (F.mvpMapPoints[i] && Observations() > 0) on entry, updated in place;
 * train_match[i] = index of the map point assigned to feature i (caller initialises to -1).  Returns nmatches. */
int vsg_frame_search_by_projection(vsg_frame *F, int n_mp, const uint8_t *mp_desc, const uint8_t *mp_observed,
                                   const uint8_t *in_view, const float *proj_x, const float *proj_y,
                                   const float *proj_xr, const int32_t *scale_level, const float *view_cos,
                                   const uint8_t *in_view_r, const float *proj_x_r, const float *proj_y_r,
                                   const int32_t *scale_level_r, const float *view_cos_r, float th, float nnratio,
                                   const float *scale_factors, int nlevels, const int32_t *left_to_right,
                                   const int32_t *right_to_left, uint8_t *train_blocked, int32_t *train_match);

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) (ORBmatcher.h:50,
 * ORBmatcher.cc:1667-1878), both blocks (right camera :1786-1853 when CurrentFrame.Nleft != -1).  One entry per
 * LastFrame map point that projects into the image (:1690-1711):  u, v = uv;  ur = uv(0) - mbf * invzc (stereo gate
 * :1741-1747, used when Nleft == -1);  u_r, v_r = projection through Trl (:1788-1789; NULL when Nleft == -1);
 * last_octave = nLastOctave;  last_angle = kpLF.angle;  mp_desc / mp_observed as above.  direction: 0 = neither,
 * 1 = bForward, 2 = bBackward (:1683-1684) selects the level window of GetFeaturesInArea.  Returns nmatches after
 * the rotation-consistency filter; a match dropped by it also loses its blocked flag (mvpMapPoints[i] = NULL, :1869). */
int vsg_frame_search_by_projection_last(vsg_frame *cur, int n_q, const uint8_t *mp_desc, const uint8_t *mp_observed,
                                        const float *u, const float *v, const float *ur, const float *u_r,
                                        const float *v_r, const int32_t *last_octave, const float *last_angle,
                                        float th, int direction, const float *scale_factors, int nlevels,
                                        int check_orientation, uint8_t *train_blocked, int32_t *train_match);

/* int ORBmatcher::SearchByProjection(KeyFrame *pKF, Sim3f &Scw, vpPoints, vpMatched, th, ratioHamming)
 * (ORBmatcher.h:58, ORBmatcher.cc:430-528) and its twin that also records the source KeyFrame (ORBmatcher.h:62,
 * .cc:530-641: the adaptor sets vpMatchedKF[i] = vpPointsKFs[matched[i]]).  One entry per point that passes the
 * geometry (:446-487): u, v, radius = th * mvScaleFactors[nPredictedLevel], predicted_level.  matched[i] != -1 on
 * entry = vpMatched[i] already set (skipped as a candidate, :501); on return matched[i] = query index for the new
 * assignments (old entries keep their value).  Accept rule: bestDist <= TH_LOW * ratioHamming (:520).  Returns nmatches. */
int vsg_frame_search_by_projection_sim3(vsg_frame *kf, int n_q, const uint8_t *mp_desc, const float *u, const float *v,
                                        const float *radius, const int32_t *predicted_level, float ratio_hamming,
                                        int32_t *matched);

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th,
 * ORBdist) (ORBmatcher.h:54, ORBmatcher.cc:1880-2000).  One entry per KeyFrame map point that is not bad, not in
 * sAlreadyFound and projects into the frame (:1905-1932): u, v, radius, predicted_level (window levels
 * [level-1, level+1], :1934), kf_angle = pKF->mvKeysUn[i].angle.  occupied[i] = (CurrentFrame.mvpMapPoints[i] != NULL)
 * on entry (:1946), updated in place incl. the rotation filter's reset (:1990); train_match as above. */
int vsg_frame_search_by_projection_kf(vsg_frame *cur, int n_q, const uint8_t *mp_desc, const float *u, const float *v,
                                      const float *radius, const int32_t *predicted_level, const float *kf_angle,
                                      int orb_dist, int check_orientation, uint8_t *occupied, int32_t *train_match);

/* int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12, S12, th) (ORBmatcher.h:76,
 * ORBmatcher.cc:1448-1665).  Direction 1 (:1489-1565): for each KF1 feature i1 with a usable, not yet matched map
 * point that projects into KF2: idx1[k] = i1, its descriptor, (u, v, radius, predicted level) in KF2; direction 2
 * (:1568-1643) likewise from KF2 into KF1.  best-only scans from INT_MAX, accept <= TH_HIGH, then the agreement pass
 * (:1646-1662): matches12[i1] = i2 where both directions agree, else -1.  n1 / n2 = feature counts of the two
 * KeyFrames (matches12 has n1 entries).  Returns nFound. */
int vsg_frame_search_by_sim3(vsg_frame *kf1, vsg_frame *kf2, int nq1, const int32_t *idx1, const uint8_t *desc1,
                             const float *u1, const float *v1, const float *radius1, const int32_t *level1, int nq2,
                             const int32_t *idx2, const uint8_t *desc2, const float *u2, const float *v2,
                             const float *radius2, const int32_t *level2, int32_t *matches12);

/* int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th, bRight) (ORBmatcher.h:84,
 * ORBmatcher.cc:1148-1329): the search.  One entry per map point that passes :1176-1236: u, v, ur = u - bf * invz,
 * radius, predicted_level.  Candidates: GetFeaturesInArea(u, v, radius, bRight), level window, then the chi-square
 * gate on the reprojection error (:1267-1292: 7.8 with mvuRight >= 0, 5.99 without) with inv_level_sigma2 =
 * pKF->mvInvLevelSigma2.  best_idx[k] (index into mDescriptors, i.e. + NLeft for bRight) / best_dist[k]; -1 / 256 when
 * nothing qualifies.  A match is bestDist <= TH_LOW (:1308); the replace-vs-add decision on the MapPoint graph
 * (:1310-1324) is vsg_fuse_decide below. Returns the number of entries with bestDist <= TH_LOW. */
int vsg_frame_fuse(vsg_frame *kf, int n_q, const uint8_t *mp_desc, const float *u, const float *v, const float *ur,
                   const float *radius, const int32_t *predicted_level, int right, const float *inv_level_sigma2,
                   int nlevels, int32_t *best_idx, int32_t *best_dist);
/* int ORBmatcher::Fuse(KeyFrame *pKF, Sim3f &Scw, vpPoints, th, vpReplacePoint) (ORBmatcher.h:87,
 * ORBmatcher.cc:1331-1446): the search (scan from INT_MAX, level window only, no chi-square gate). */
int vsg_frame_fuse_sim3(vsg_frame *kf, int n_q, const uint8_t *mp_desc, const float *u, const float *v,
                        const float *radius, const int32_t *predicted_level, int32_t *best_idx, int32_t *best_dist);
/* The ordered decision pass of both Fuse overloads on flattened state, for callers that keep map points as ids:
 * slot_mp[i] = id of the map point in pKF->GetMapPoint(i) or -1 (updated in place as AddMapPoint /
 * ReplaceMapPointMatch would); mp_obs[id] = Observations(), mp_bad[id] = isBad() (updated by Replace as
 * MapPoint::Replace does for THIS keyframe: the survivor takes over the slot and the loser's observation count, the
 * loser turns bad).  query_mp[k] = id of the k-th query's map point.  action[k]: 0 none (bestDist > TH_LOW),
 * 1 AddObservation + AddMapPoint (:1321-1322), 2 pMP->Replace(pMPinKF) (:1315), 3 pMPinKF->Replace(pMP) (:1317),
 * 4 counted but nothing done (pMPinKF is bad), 5 (sim3 form) vpReplacePoint[k] = pMPinKF (:1436).  sim3_form selects
 * :1429-1444.  Returns nFused. */
int vsg_fuse_decide(int n_q, const int32_t *query_mp, const int32_t *best_idx, const int32_t *best_dist, int sim3_form,
                    int32_t *slot_mp, int n_slots, int32_t *mp_obs, uint8_t *mp_bad, int n_mp, int32_t *action,
                    int32_t *other_mp);

/* int ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.h:68,
 * ORBmatcher.cc:643-756) with both frames resident: only vbPrevMatched (x, y per F1 keypoint) goes up;
 * F2.GetFeaturesInArea(x, y, windowSize, 0, 0) (:663) runs on F2's resident grid for the level-0 keypoints of F1. */
int vsg_frame_search_for_initialization(vsg_frame *f1, vsg_frame *f2, const float *prev_x, const float *prev_y,
                                        int window_size, float nnratio, int check_orientation, int32_t *matches12);

/* SearchByBoW(KeyFrame*, Frame&) / (KeyFrame*, KeyFrame*) (ORBmatcher.cc:226-428, 758-900) on resident descriptors:
 * only the FeatureVectors and the validity flags go up.  Same outputs as vsg_search_by_bow_kf_f_stereo / _kf_kf.
 * With ALL FeatureVector arrays NULL the FeatureVectors both frames keep resident since their vsg_frame_bow_transform
 * (Frame::mFeatVec; frames of at most 2048 features) are joined on the device and only the flags go up;
 * VSG_ERR_INVALID when either frame never had its ComputeBoW since its features were last written. */
int vsg_frame_search_by_bow_kf_f(vsg_frame *kf, const uint8_t *kf_valid, const int32_t *kf_node_id,
                                 const int32_t *kf_off, const int32_t *kf_idx, int kf_nodes, vsg_frame *f,
                                 const int32_t *f_node_id, const int32_t *f_off, const int32_t *f_idx, int f_nodes,
                                 float nnratio, int check_orientation, int32_t *match_f);
int vsg_frame_search_by_bow_kf_kf(vsg_frame *kf1, const uint8_t *valid1, const int32_t *node_id1, const int32_t *off1,
                                  const int32_t *idx1, int nodes1, vsg_frame *kf2, const uint8_t *valid2,
                                  const int32_t *node_id2, const int32_t *off2, const int32_t *idx2, int nodes2,
                                  float nnratio, int check_orientation, int32_t *matches12);
/* SearchForTriangulation(KeyFrame*, KeyFrame*, ...) (ORBmatcher.cc:902-1146) with both KeyFrames resident (LocalMapping::
 * CreateNewMapPoints calls it for the current keyframe against each of its 10-20 best covisible neighbours,
 * LocalMapping.cc:389: the current keyframe's descriptors go up once, not once per neighbour).  Arguments and result as
 * vsg_search_for_triangulation; angles come from the frames' host mirrors. */
int vsg_frame_search_for_triangulation(vsg_frame *kf1, const uint8_t *eligible1, const int32_t *node_id1,
                                       const int32_t *off1, const int32_t *idx1, int nodes1, vsg_frame *kf2,
                                       const uint8_t *eligible2, const int32_t *node_id2, const int32_t *off2,
                                       const int32_t *idx2, int nodes2, const uint32_t *pair_ok, const int32_t *pair_off,
                                       int check_orientation, int32_t *matches12);
/* Frame::ComputeBoW (Frame.cc:882-889) on the resident descriptors; outputs as vsg_bow_transform. */
int vsg_frame_bow_transform(vsg_vocab *voc, vsg_frame *f, int levelsup, int32_t *bow_ids, double *bow_vals,
                            int bow_cap, int *n_bow, int32_t *fv_node, int32_t *fv_off, int32_t *fv_idx, int fv_cap,
                            int *n_fv, int32_t *word_of, int32_t *node_of, double *weight_of);
/* Frame::ComputeStereoMatches (Frame.cc:957-1127) on two resident feature sets (left / right eye of one stereo
 * Frame, straight out of the two extractors): nothing but mvuRight / mvDepth crosses PCIe. */
int vsg_frame_stereo_matches(vsg_orb *hl, int frame_l, vsg_orb *hr, int frame_r, vsg_frame *fl, vsg_frame *fr,
                             float mb, float mbf, float *u_right, float *depth);

/* Frame::ComputeStereoMatches (Frame.cc:957-1127) + Frame::ComputeBoW (Frame.cc:882-889) + ORBmatcher::SearchByBoW(KeyFrame*,
 * Frame&, ...) (ORBmatcher.cc:226-428) of one stereo Frame in ONE enqueue and ONE wait: what depends only on two resident
 * frames shares a stream round trip instead of paying three.  Arguments as vsg_frame_stereo_matches (fl / fr = the resident
 * left / right eye; *n_stereo = matches kept by the median cut), vsg_frame_bow_transform (of fl; its FeatureVector also
 * stays resident in fl) and vsg_frame_search_by_bow_kf_f with NULL FeatureVector arrays (kf = a frame whose own ComputeBoW
 * ran earlier; kf == NULL: no search, match_f / n_match are not touched).  Results are those of the three blocking calls. */
int vsg_frame_stereo_bow_search(vsg_orb *hl, int frame_l, vsg_orb *hr, int frame_r, vsg_frame *fl, vsg_frame *fr, float mb,
                                float mbf, float *u_right, float *depth, int *n_stereo, vsg_vocab *voc, int levelsup,
                                int32_t *bow_ids, double *bow_vals, int bow_cap, int *n_bow, int32_t *fv_node,
                                int32_t *fv_off, int32_t *fv_idx, int fv_cap, int *n_fv, vsg_frame *kf,
                                const uint8_t *kf_valid, float nnratio, int check_orientation, int32_t *match_f,
                                int *n_match);

/* ---- Frame sharding over the GPUs of one node (one process per GPU) -------------------------------
 * The reference has no distributed layer.  Extraction shards by frame (or camera stream) with no collective; the
 * neighbour's features that matching frame t against t-1 needs travel as fixed-capacity per-frame records
 *   { int32 n, int32 monoIndex, uint32 flags, pad to 16 | KeyPoint[cap] (padded to 16) | uint8 desc[cap][32] | pad to 64 }
 * (flags bit 0 = VSG_SHARD_FLAG_TRUNCATED: the frame held more keypoints than the record's capacity; n and monoIndex
 * are clamped to it.  Slots of a partial batch -- nframes < frames_per_rank -- are sent with n = 0.)
 * in ONE ncclAllGather (RCCL over xGMI) per batch, enqueued on the caller's stream.  RCCL is loaded on first use
 * (dlopen); VSG_ERR_UNSUPPORTED when it is not available.
 *   vsg_shard_unique_id   rank 0 creates the 128-byte ncclUniqueId; the caller hands it to every rank (MPI, a file,
 *                         torch.distributed.broadcast, ...)
 *   vsg_shard_create      ncclCommInitRank + the send / receive record buffers for `frames_per_rank` frames per batch
 *   vsg_shard_all_gather  packs this rank's batch (the device outputs of vsg_orb_extract_batch_device: d_counts
 *                         [nframes][2], d_kps / d_desc [nframes][src_capacity]) and all-gathers; asynchronous on `stream`
 *   vsg_shard_record      device pointers of frame `frame` of rank `rank` in the gathered block: they can be passed
 *                         straight to vsg_hamming_block_best2_device / vsg_frame_* (valid until the next all-gather)
 *   vsg_shard_frame_owner / vsg_shard_stream_owner   the partition: frame f -> rank f mod world; camera stream s of
 *                         n_streams -> rank s (world <= n_streams: s mod world), a stream's frames round-robin over the
 *                         ranks {s, s + n_streams, ...} when there are more ranks than streams (4 cameras on 8 GPUs) */
typedef struct vsg_shard vsg_shard;
#define VSG_SHARD_FLAG_TRUNCATED 1u
const char *vsg_shard_last_error(void);
size_t vsg_shard_record_bytes(int capacity);
size_t vsg_shard_record_desc_offset(int capacity);
int vsg_shard_frame_owner(int frame, int world);
int vsg_shard_stream_owner(int stream, int n_streams, int world, int frame);
int vsg_shard_unique_id(uint8_t id[128]);
int vsg_shard_create(int device, int rank, int world, const uint8_t id[128], int capacity, int frames_per_rank,
                     vsg_shard **out);
void vsg_shard_destroy(vsg_shard *s);
int vsg_shard_all_gather(vsg_shard *s, const int *d_counts, const vsg_keypoint *d_kps, const uint8_t *d_desc,
                         int src_capacity, int nframes, void *stream);
int vsg_shard_record(vsg_shard *s, int rank, int frame, const int **d_counts, const vsg_keypoint **d_kps,
                     const uint8_t **d_desc);
/* Size of the communicator and this process's rank in it AS RCCL REPORTS THEM (ncclCommCount / ncclCommUserRank on the
 * live communicator -- not an echo of what vsg_shard_create was given); VSG_ERR_UNSUPPORTED if the loaded RCCL lacks them. */
int vsg_shard_world(const vsg_shard *s);
int vsg_shard_rank(const vsg_shard *s);
/* Neighbour-only exchange for chunk-partitioned sequences (rank r owns frames [r B, (r + 1) B) of a step): matching
 * needs ONE remote record per rank and step, the predecessor rank's last frame.  Packs frame `frame` of this rank's
 * batch and sends it to rank + 1 while receiving the record of rank - 1 (cyclic): one ncclSend / ncclRecv pair on the
 * caller's stream instead of an all-gather of every record.  vsg_shard_boundary_record: device pointers of the
 * received record (valid until the next exchange). */
int vsg_shard_send_recv_boundary(vsg_shard *s, const int *d_counts, const vsg_keypoint *d_kps, const uint8_t *d_desc,
                                 int src_capacity, int frame, void *stream);
int vsg_shard_boundary_record(vsg_shard *s, const int **d_counts, const vsg_keypoint **d_kps, const uint8_t **d_desc);

#ifdef __cplusplus
}
#endif
#endif /* VSG_ORB_H */
