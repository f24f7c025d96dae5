/*
 * orb_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain C++ restatement, without OpenCV, of the per-frame ORB front-end of
 * snt-arg/visual_sgraphs (vS-Graphs on ORB-SLAM3):
 *   orb_slam3/src/ORBextractor.cc  (ORBextractor::operator() and helpers)
 *   orb_slam3/src/ORBmatcher.cc    (DescriptorDistance, SearchBy*)
 *   orb_slam3/src/Frame.cc         (grid: AssignFeaturesToGrid/GetFeaturesInArea)
 * and of the OpenCV 4.2 routines those files call (cv::FAST, cv::resize
 * INTER_LINEAR 8U, cv::GaussianBlur 8U fixed point, cv::copyMakeBorder,
 * cv::fastAtan2, cvRound).  OpenCV is an un-vendored third-party dependency
 * (CMakeLists.txt:35 `find_package(OpenCV 4.2)`), absent from this image, so
 * its published algorithms are restated from knowledge of the upstream source.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
 * this path and neither the reference nor OpenCV can
 * be built here.  Two version-sensitive tables are kept as data: the 8.8
 * fixed-point Gaussian taps (see or_set_blur_taps) and nothing else inside
 * operator().
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use this library, and only as the checker / reported baseline.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint layout: pt.x pt.y size angle response octave class_id = 28 B */
typedef struct OrKeyPoint {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} OrKeyPoint;

typedef struct OrExtractor OrExtractor;

/* ---- ORBextractor (ORBextractor.cc:411-470 ctor) ---- */
OrExtractor *or_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
void or_destroy(OrExtractor *e);
/* 7 taps of the 8.8 fixed-point Gaussian (default {18,34,49,55,49,34,18}). */
void or_set_blur_taps(OrExtractor *e, const uint16_t taps[7]);
/* getters (ORBextractor.h:63-91) + quotas + umax[16] */
int or_get_tables(const OrExtractor *e, float *scale, float *invScale, float *sigma2, float *invSigma2,
                  int *featuresPerLevel, int *umax16);

/* ORBextractor::operator() (ORBextractor.cc:1083-1169).  Returns monoIndex, or
 * -1 for an empty image, -2 if capacity is too small. *n = number of keypoints. */
int or_extract(OrExtractor *e, const uint8_t *gray, int rows, int cols, int stride, int lap0, int lap1,
               OrKeyPoint *kps, uint8_t *desc, int capacity, int *n);

/* ---- stage read-back after or_extract (for stage-by-stage parity) ---- */
int or_level_size(const OrExtractor *e, int level, int *w, int *h);
/* copy pyramid level; with_border!=0 -> (w+38)x(h+38) incl. the 19px REFLECT_101 frame */
int or_get_pyramid_level(const OrExtractor *e, int level, uint8_t *dst, int dst_stride, int with_border);
int or_get_blurred_level(const OrExtractor *e, int level, uint8_t *dst, int dst_stride);
/* FAST candidates of a level in reference order (vToDistributeKeys, coords relative to (16,16)) */
int or_get_candidates(const OrExtractor *e, int level, int *x, int *y, int *response, int cap);
/* keypoints of a level after octree+orientation, level-local coords (incl +16), in octree order */
int or_get_level_keypoints(const OrExtractor *e, int level, OrKeyPoint *kps, int cap);

/* ---- stand-alone restatements of the OpenCV pieces ---- */
int or_cv_round_f(float v);
int or_cv_round_d(double v);
float or_fast_atan2(float y, float x);
/* cv::resize(..., INTER_LINEAR) for CV_8UC1 */
void or_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh, int dstride);
/* cv::copyMakeBorder(..., BORDER_REFLECT_101) equal border b on all sides; dst is (w+2b)x(h+2b) */
void or_copy_make_border101(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride, int b);
/* cv::GaussianBlur(7x7, sigma 2, REFLECT_101) 8U fixed point with the given taps */
void or_gaussian_blur7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride,
                          const uint16_t taps[7]);
/* cv::FAST(img, kps, threshold, nonmax) TYPE_9_16; returns count; outputs in OpenCV order */
int or_fast9_16(const uint8_t *img, int w, int h, int stride, int threshold, int nonmax, int *x, int *y,
                int *score, int cap);
/* ORBextractor::DistributeOctTree (ORBextractor.cc:562-785) on (x,y,response) lists */
int or_distribute_octree(const int *x, const int *y, const int *response, int n, int minX, int maxX, int minY,
                         int maxY, int N, int *outIndex, int cap);
/* IC_Angle (ORBextractor.cc:73-100) at integer centre on an image with given stride; needs 15px margin */
float or_ic_angle(const uint8_t *img, int stride, int cx, int cy);
/* computeOrbDescriptor (ORBextractor.cc:103-149) */
void or_orb_descriptor(const uint8_t *blurred, int stride, int cx, int cy, float angleDeg, uint8_t desc[32]);

/* ---- DBoW2 (vendored): vocabulary in the reference's binary layout and transform() (bow_oracle.cpp) ---- */
typedef struct OrVocab OrVocab;
OrVocab *or_vocab_load(const uint8_t *blob, size_t size);
void or_vocab_destroy(OrVocab *v);
int or_vocab_info(const OrVocab *v, int *k, int *L, int *scoring, int *weighting, int *nnodes, int *nwords);
/* transform(features, BowVector, FeatureVector, levelsup): BowVector as ascending (id, value) pairs, FeatureVector
 * as CSR (ascending node ids); fv_idx must hold n ints; optional per-feature word / node / weight. */
int or_vocab_transform(const OrVocab *voc, const uint8_t *desc, int n, int levelsup, int *bow_ids, double *bow_vals,
                       int bow_cap, int *n_bow, int *fv_node, int *fv_off, int *fv_idx, int fv_cap, int *n_fv,
                       int *word_of, int *node_of, double *weight_of);

/* cv::cvtColor(COLOR_{RGB,BGR,RGBA,BGRA}2GRAY) for 8-bit images as called in Tracking::GrabImage*
 * (Tracking.cc:1526-1551, 1595-1608, 1643-1656): gray = (R*cr + G*cg + B*cb + (1 << (shift-1))) >> shift.
 * [OCV] 4.2 coefficients: cr,cg,cb = 4899, 9617, 1868, shift 14 (version-sensitive => data). */
void or_cvt_gray_u8(const uint8_t *src, int rows, int cols, int sstride, int channels, int rgb_order, uint8_t *dst,
                    int dstride, const int coeffs[3], int shift);

/* Frame::ComputeStereoMatches (Frame.cc:957-1127) for a rectified pair whose two extractors have just run
 * or_extract (their mvImagePyramid is read, Frame.cc:964,1054-1069).  kps/desc = the operator() outputs.
 * uRight[nL], depth[nL] receive mvuRight / mvDepth (-1 where no match). */
void or_stereo_matches(const OrExtractor *left, const OrExtractor *right, const OrKeyPoint *kpsL, const uint8_t *descL,
                       int nL, const OrKeyPoint *kpsR, const uint8_t *descR, int nR, float mb, float mbf,
                       float *uRight, float *depth);

/* ---- CPU baseline (bench.py cpu_baseline leg): `nthreads` independent extractors, each on its own host thread
 * (the reference is single-threaded per ORBextractor, two threads for stereo: Frame.cc:129-132), looping over the
 * given frames: operator() + brute-force best/second-best match against the thread's previous frame when
 * do_match != 0.  Runs for about `seconds`; returns frames/s, *frames_done = frames processed. */
double or_bench_throughput(const uint8_t *frames, int nframes, int rows, int cols, int nfeatures, float scaleFactor,
                           int nlevels, int iniThFAST, int minThFAST, int nthreads, double seconds, int do_match,
                           long *frames_done);

/* ---- whole-batch checkers (bench.py's parity gate, tests): every frame / every match row of a batch on `nthreads`
 * host threads, outputs in fixed-capacity rows; counts[f] = {n, monoIndex} ({-2,-2}: capacity too small) ---- */
void or_extract_batch_mt(const uint8_t *frames, int nframes, int rows, int cols, int nfeatures, float scaleFactor,
                         int nlevels, int iniThFAST, int minThFAST, int lap0, int lap1, int nthreads, int capacity,
                         int *counts, OrKeyPoint *kps, uint8_t *desc);
void or_block_best2_batch_mt(const uint8_t *a, size_t a_stride, const int *na, const uint8_t *b, size_t b_stride,
                             const int *nb, int nblocks, int capacity, int nthreads, int *best, int *second, int *argbest);

/* ---- ORBmatcher ---- */
/* ORBmatcher::DescriptorDistance (ORBmatcher.cc:2047-2063) */
int or_descriptor_distance(const uint8_t *a, const uint8_t *b);
/* ComputeThreeMaxima (ORBmatcher.cc:2002-2043) on 30 bin sizes */
void or_three_maxima(const int *histoSizes, int L, int *ind1, int *ind2, int *ind3);

/* Frame grid (Frame.cc:521-553, 802-880): build + query on undistorted keypoints. */
typedef struct OrGrid OrGrid;
OrGrid *or_grid_build(const OrKeyPoint *kps, int n, float minX, float minY, float maxX, float maxY);
void or_grid_destroy(OrGrid *g);
int or_grid_query(const OrGrid *g, float x, float y, float r, int minLevel, int maxLevel, int *outIdx, int cap);

/* SearchByBoW(KeyFrame*, Frame&, ...) (ORBmatcher.cc:226-428), Nleft == -1 path.
 * FeatureVectors are CSR: node ids ascending (std::map order), off[nNodes+1], idx[].
 * kfValid[i] = (pMP != NULL && !pMP->isBad()).  matchF[realIdxF] = realIdxKF or -1. */
int or_search_by_bow_kf_f(const uint8_t *kfDesc, const float *kfAngle, const uint8_t *kfValid, int nKF,
                          const int *kfNodeId, const int *kfOff, const int *kfIdx, int kfNodes,
                          const uint8_t *fDesc, const float *fAngle, int nF, const int *fNodeId, const int *fOff,
                          const int *fIdx, int fNodes, float nnratio, int checkOri, int *matchF);

/* The same with F.Nleft given (fisheye stereo: a second best/second pair over the right-camera features,
 * ORBmatcher.cc:277-326, 362-389); nleftF = -1 is the monocular form above. */
int or_search_by_bow_kf_f_stereo(const uint8_t *kfDesc, const float *kfAngle, const uint8_t *kfValid, int nKF,
                                 const int *kfNodeId, const int *kfOff, const int *kfIdx, int kfNodes,
                                 const uint8_t *fDesc, const float *fAngle, int nF, int nleftF, const int *fNodeId,
                                 const int *fOff, const int *fIdx, int fNodes, float nnratio, int checkOri,
                                 int *matchF);

/* SearchByBoW(KeyFrame*, KeyFrame*, ...) (ORBmatcher.cc:758-900), NLeft == -1.
 * matches12[idx1] = idx2 or -1. */
int or_search_by_bow_kf_kf(const uint8_t *desc1, const float *angle1, const uint8_t *valid1, int n1,
                           const int *nodeId1, const int *off1, const int *idx1, int nodes1, const uint8_t *desc2,
                           const float *angle2, const uint8_t *valid2, int n2, const int *nodeId2, const int *off2,
                           const int *idx2, int nodes2, float nnratio, int checkOri, int *matches12);

/* SearchForTriangulation(KeyFrame*, KeyFrame*, vMatchedPairs, bOnlyStereo, bCoarse) (ORBmatcher.cc:902-1146),
 * NLeft == -1.  eligible = no MapPoint yet (and stereo when bOnlyStereo); pairOk/pairOff = the geometric predicate of
 * :1031-1071 for every pair of every shared node (NULL = always true, i.e. bCoarse without the epipole gate). */
int or_search_for_triangulation(const uint8_t *desc1, const float *angle1, const uint8_t *eligible1, int n1,
                                const int *nodeId1, const int *off1, const int *idx1, int nodes1, const uint8_t *desc2,
                                const float *angle2, const uint8_t *eligible2, int n2, const int *nodeId2,
                                const int *off2, const int *idx2, int nodes2, const uint32_t *pairOk,
                                const int *pairOff, int checkOri, int *matches12);

/* SearchByProjection(Frame &Cur, const Frame &Last, th, bMono) (ORBmatcher.cc:1667-1878),
 * Nleft == -1, with the geometry (projection, GetFeaturesInArea, mvuRight gate)
 * done by the caller: query q has candidate list cand[candOff[q]..candOff[q+1]).
 * trainBlocked[i] = Cur.mvpMapPoints[i] && Observations()>0 on entry (updated in place);
 * queryBlocks[q] = pMP_q->Observations()>0.  trainMatch[i] = q or unchanged (-1 init by caller). */
int or_search_by_projection_last(const uint8_t *qDesc, const float *qAngle, const uint8_t *queryBlocks, int nQ,
                                 const int *candOff, const int *candIdx, const uint8_t *tDesc, const float *tAngle,
                                 uint8_t *trainBlocked, int nT, int thHigh, int checkOri, int *trainMatch);

/* SearchByProjection(Frame &F, vpMapPoints, th, ...) (ORBmatcher.cc:42-216), Nleft == -1 (left block only).
 * Same flattening; tOctave = octave of each train keypoint; ratio test as in :123-141. */
int or_search_by_projection_local(const uint8_t *qDesc, const uint8_t *queryBlocks, int nQ, const int *candOff,
                                  const int *candIdx, const uint8_t *tDesc, const int *tOctave,
                                  uint8_t *trainBlocked, int nT, float nnratio, int *trainMatch);

/* Common core of the projection searches that keep only the best candidate: SearchByProjection(KF, Scw, ...)
 * (ORBmatcher.cc:430-528, :530-641), SearchByProjection(F, KF, ...) (:1880-2000) without the rotation check,
 * SearchBySim3's two directional passes (:1500-1640) and Fuse's candidate loop (:1148-1446): for each query in
 * order, best (strict '<') over the non-blocked candidates; qBestIdx/qBestDist per query (-1 / 256 if none);
 * accepted when dist <= thHigh: trainMatch[best] = q, trainBlocked[best] = queryBlocks[q] (NULL = never blocks). */
int or_search_window(const uint8_t *qDesc, const uint8_t *queryBlocks, int nQ, const int *candOff, const int *candIdx,
                     const uint8_t *tDesc, uint8_t *trainBlocked, int nT, int thHigh, int *qBestIdx, int *qBestDist,
                     int *trainMatch);

/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:380-415) for `ngroups` map points at once: group g owns the
 * descriptor rows off[g] .. off[g+1]-1 (its observations, in std::map iteration order); best[g] = row index inside
 * the group with the least median Hamming distance to the rest (first one on ties), -1 for an empty group. */
void or_distinctive_descriptors(const uint8_t *desc, const int *off, int ngroups, int *best);

/* SearchForInitialization (ORBmatcher.cc:643-756): octave-0 keypoints of F1, candidate lists per F1 kp
 * from F2.GetFeaturesInArea(prevMatched, windowSize, 0, 0).  candOff/candIdx as above (empty for octave>0). */
int or_search_for_initialization(const uint8_t *desc1, const float *angle1, const int *octave1, int n1,
                                 const int *candOff, const int *candIdx, const uint8_t *desc2, const float *angle2,
                                 int n2, float nnratio, int checkOri, int *matches12);

/* Brute-force best/second-best of each row of A against all of B (the inner loop of SearchByBoW
 * with one node and no greedy state): strict '<' scan order ties (Appendix B). */
void or_block_best2(const uint8_t *a, int na, const uint8_t *b, int nb, int *best, int *second, int *argbest);

/* ---- camera undistortion (undistort_oracle.cpp): Frame::UndistortKeyPoints (Frame.cc:891-921), Frame::ComputeImageBounds
 * (Frame.cc:924-955) and the [OCV 4.2] cv::undistortPoints(src, dst, K, D, Mat(), K) they call (5 fixed iterations, double).
 * K4 = {fx, fy, cx, cy} as the floats of mK; dist = the 4 or 5 floats of mDistCoef (k1, k2, p1, p2[, k3]).  Harness code
 * for the grid / search parity tests; parity unpinned like every OpenCV piece. */
void or_undistort_points(const float *xy_in, int n, const float K4[4], const float *dist, int ndist, float *xy_out);
void or_undistort_keypoints(const OrKeyPoint *keys, int n, const float K4[4], const float *dist, int ndist,
                            OrKeyPoint *keys_un);
void or_image_bounds(int cols, int rows, const float K4[4], const float *dist, int ndist, float out[4]);

/* ---- routine-level restatements on a flattened Frame / KeyFrame (routines_oracle.cpp): the functions follow the
 * reference's loops from the point where a map point has been projected; see the comments there. ---- */
typedef struct OrFrame OrFrame;
OrFrame *or_frame_create(const OrKeyPoint *keys, const uint8_t *desc, const float *uRight, int N, int Nleft, float minX,
                         float minY, float maxX, float maxY);
void or_frame_destroy(OrFrame *f);
int or_frame_grid(const OrFrame *f, int right, int *cell_start, int *entries);
int or_frame_features_in_area(const OrFrame *f, float x, float y, float r, int minLevel, int maxLevel, int bRight,
                              int kfForm, int *out, int cap);
int or_frame_search_by_projection(const OrFrame *F, int nMP, const uint8_t *mpDesc, const uint8_t *mpObserved,
                                  const uint8_t *mbTrackInView, const float *mTrackProjX, const float *mTrackProjY,
                                  const float *mTrackProjXR, const int *mnTrackScaleLevel, const float *mTrackViewCos,
                                  const uint8_t *mbTrackInViewR, const float *mTrackProjXR_r,
                                  const float *mTrackProjYR_r, const int *mnTrackScaleLevelR,
                                  const float *mTrackViewCosR, float th, float mfNNratio, const float *mvScaleFactors,
                                  const int *mvLeftToRightMatch, const int *mvRightToLeftMatch, uint8_t *trainBlocked,
                                  int *trainMatch);
int or_frame_search_by_projection_last(const OrFrame *Cur, int nQ, const uint8_t *mpDesc, const uint8_t *mpObserved,
                                       const float *u, const float *v, const float *ur, const float *uR,
                                       const float *vR, const int *nLastOctave, const float *kpLFangle, float th,
                                       int bForward, int bBackward, const float *mvScaleFactors, int mbCheckOrientation,
                                       uint8_t *trainBlocked, int *trainMatch);
int or_kf_search_by_projection_sim3(const OrFrame *pKF, int nQ, const uint8_t *mpDesc, const float *u, const float *v,
                                    const float *radius, const int *nPredictedLevel, float ratioHamming, int *matched);
int or_frame_search_by_projection_kf(const OrFrame *Cur, int nQ, const uint8_t *mpDesc, const float *u, const float *v,
                                     const float *radius, const int *nPredictedLevel, const float *kfAngle, int ORBdist,
                                     int mbCheckOrientation, uint8_t *occupied, int *trainMatch);
int or_kf_search_by_sim3(const OrFrame *pKF1, const OrFrame *pKF2, int nq1, const int *idx1, const uint8_t *desc1,
                         const float *u1, const float *v1, const float *radius1, const int *level1, int nq2,
                         const int *idx2, const uint8_t *desc2, const float *u2, const float *v2, const float *radius2,
                         const int *level2, int *matches12);
int or_kf_fuse(const OrFrame *pKF, int nQ, const int *queryMP, const uint8_t *mpDesc, const float *u, const float *v,
               const float *ur, const float *radius, const int *nPredictedLevel, int bRight,
               const float *mvInvLevelSigma2, int *slotMP, int *mpObs, uint8_t *mpBad, int *bestIdx, int *bestDist,
               int *action, int *other);
int or_kf_fuse_sim3(const OrFrame *pKF, int nQ, const int *queryMP, const uint8_t *mpDesc, const float *u,
                    const float *v, const float *radius, const int *nPredictedLevel, int *slotMP, int *mpObs,
                    uint8_t *mpBad, int *bestIdx, int *bestDist, int *action, int *other);
int or_frame_search_for_initialization(const OrFrame *F1, const OrFrame *F2, const float *prevX, const float *prevY,
                                       int windowSize, float mfNNratio, int mbCheckOrientation, int *vnMatches12);

#ifdef __cplusplus
}
#endif
#endif
