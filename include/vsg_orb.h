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
 * `stream` (a hipStream_t, may be NULL = the handle's own stream) and completes in stream order.
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

/* Stage read-back of the last call, for stage-by-stage parity tests: the blurred level
 * (ORBextractor.cc:1129-1130), the FAST candidates (vToDistributeKeys, :868-873; unordered,
 * packed x | y<<12 | response<<24 relative to (16,16)) and the octree selection in list order. */
int vsg_orb_copy_blurred_level(vsg_orb *h, int frame, int level, uint8_t *dst, int dst_stride);
int vsg_orb_copy_candidates(vsg_orb *h, int frame, int level, uint32_t *dst, int cap);
int vsg_orb_copy_selected(vsg_orb *h, int frame, int level, uint32_t *dst, int cap);

/* Average device time per stage of the last N timed calls (HIP events on the handle's streams).
 * names: "pyramid","fast","octree","blur","slots","orient_desc","total".  Returns number of stages. */
int vsg_orb_enable_timing(vsg_orb *h, int enable);
/* serialize != 0: enqueue every kernel on one stream (no blur overlap), so that stage timings and rocprof kernel
 * durations are free of interference from concurrent kernels.  Results are identical either way. */
int vsg_orb_set_serialize(vsg_orb *h, int serialize);
int vsg_orb_get_timing(vsg_orb *h, float *ms_out, int cap);

/* Test hook: sorts items[0..n) (n <= 2048) by their upper 32 bits with the device code DistributeOctTree uses for
 * `std::sort(vSizeAndPointerToNode...)` (ORBextractor.cc:707): a replay of libstdc++'s introsort whose result --
 * including the order of equal keys -- must equal std::sort's.  Lets tests compare the two directly. */
int vsg_debug_device_sort(int device, uint64_t *items, int n);

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

#ifdef __cplusplus
}
#endif
#endif /* VSG_ORB_H */
