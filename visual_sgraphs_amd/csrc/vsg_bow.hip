// vsg_bow.hip -- DBoW2 vocabulary transform on the device: what Frame::ComputeBoW /
// KeyFrame::ComputeBoW call (Frame.cc:882-889, levelsup = 4) to produce the FeatureVector that SearchByBoW joins.
//   orb_slam3/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h  loadFromBinFile :1478-1552
//     transform(features, v, fv, levelsup) :1139-1212   transform(feature, word, weight, nid, levelsup) :1229-1271
//   BowVector.cpp:34-84, FeatureVector.cpp:31-45, FORB.cpp:81-101
// Device: one thread per descriptor walks the tree (k Hamming distances per level, first minimum wins, strict '<').
// Host (inside the C ABI): the std::map assembly in FEATURE ORDER, which fixes the floating-point sums of the
// BowVector (addWeight accumulates doubles in feature order; normalize() sums in ascending word id).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/vsg_orb.h"
#include "vsg_ctx.h"
#include "vsg_frame_int.h"

namespace {
// Device image of the vocabulary: one 48-byte record per node IN CHILD-LIST ORDER (the children of a node are
// contiguous, in the order transform() scans them): 32-byte descriptor | link = position of the node's first child |
// its child count << 24 | node id | word id (leaves).  The root's children start at position 0.
struct alignas(16) NodeRec {
  uint32_t d[8];
  uint32_t link;  // first child position | number of children << 24  (0 children: a leaf = a word)
  int32_t node_id, word;
  uint32_t pad;
};
static_assert(sizeof(NodeRec) == 48, "NodeRec layout");

// min over the G lanes of a group (G = 16 or 32, groups aligned to G lanes), every lane gets it: DPP row operations, the
// 32-lane form finishes with one xor-16 exchange
template <int G>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));  // row_half_mirror
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));  // row_mirror
  if (G == 32) v = min(v, (uint32_t)__shfl_xor((int)v, 16));
  return v;
}

// One group of G lanes per descriptor, one lane per child: a level of the descent is ONE round of loads (the k child
// records, contiguous), k Hamming distances in parallel, a group minimum of (distance << 8 | child) -- the first
// minimum, like the strict '<' scan of transform() (TemplatedVocabulary.h:1250-1259) -- and one broadcast of the
// winner's link.  The thread-per-descriptor form chained three dependent loads per level (child range, child ids,
// descriptors): 18 memory hops for the reference's L = 6 vocabulary against 6 here.
template <int G>
__global__ __launch_bounds__(256) void k_bow_descend(const NodeRec *__restrict__ rec, const double *__restrict__ weight,
                                                     uint32_t root_link, const uint8_t *__restrict__ desc, int n,
                                                     int nid_level, int *__restrict__ word_of, int *__restrict__ node_of,
                                                     double *__restrict__ weight_of, int *__restrict__ m_word,
                                                     int *__restrict__ m_node, double *__restrict__ m_weight) {
  const int i = (blockIdx.x * 256 + threadIdx.x) / G, j = threadIdx.x & (G - 1);
  if (i >= n) return;  // whole groups leave together
  const uint4 *f = (const uint4 *)(desc + (size_t)i * 32);
  const uint4 a0 = f[0], a1 = f[1];
  uint32_t link = root_link;
  int level = 0, nid = 0;
  while (link >> 24) {  // !isLeaf()
    ++level;
    const uint32_t first = link & 0xFFFFFFu, cnt = link >> 24;
    uint32_t key = 0xFFFFFFFFu, mylink = 0;
    int myid = 0, myword = 0;
    if ((uint32_t)j < cnt) {
      const uint4 *r = (const uint4 *)(rec + first + j);
      const uint4 b0 = r[0], b1 = r[1], m = r[2];
      const int dist = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                       __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
      key = ((uint32_t)dist << 8) | (uint32_t)j;
      mylink = m.x, myid = (int)m.y, myword = (int)m.z;
    }
    const uint32_t best = group_min_u32<G>(key);
    const int jw = (int)(best & 0xFFu);
    link = (uint32_t)__shfl((int)mylink, jw, G);
    if (level == nid_level) nid = __shfl(myid, jw, G);  // group-uniform: every lane keeps the node id of that level
    if (j == jw) {  // the winner's lane holds everything the outputs need
      if (!(link >> 24)) {
        if (nid_level > level || nid_level <= 0) nid = 0;  // levelsup beyond the tree height: nid stays 0
        const double w = weight[first + jw];
        word_of[i] = myword, node_of[i] = nid, weight_of[i] = w;
        if (m_word) m_word[i] = myword, m_node[i] = nid, m_weight[i] = w;  // pinned mirror: the per-feature outputs
      }
    }
  }
}

// ---- BowVector / FeatureVector assembly ON THE DEVICE (round 6; TemplatedVocabulary.h:1158-1206, BowVector.cpp:34-84,
// FeatureVector.cpp:31-45).  The reference inserts feature by feature into two std::maps; the same content falls out of
// two sorts of (id << 32 | feature) keys -- ascending ids, the features of an id in feature order.  Two launches: the rank
// of every key among the frame's keys, computed all over the chip (k_bow_rank; stopped words -- weight <= 0 -- take no part),
// then one workgroup per container (k_bow_assemble): group heads by a block scan, and the floating point exactly as the
// reference orders it:
//   * addWeight: every feature of a word carries the word's weight, so the running sum in feature order is the weight
//     added to itself (count - 1) times, left to right -- the head's thread does just that;
//   * normalize(): the L1 / L2 norm is ONE sequential sum in ascending word id (thread 0), then a correctly rounded
//     division (and square root) per entry -- IEEE operations, no contraction (-ffp-contract=off): bit-identical doubles.
// The host copies the result out of the pinned arena, nothing else; the FeatureVector also stays RESIDENT in the frame
// (Frame::mFeatVec), where the SearchByBoW kernels join it with another frame's without a host round trip.
enum { kAsmMax = 2048, kAsmThreads = 1024 };  // features a frame may hold for the device assembly (C4: 2000 + 24)
// BowVector::normalize (BowVector.cpp:62-84) is ONE dependent chain of n_bow double additions (the reference's loop order IS
// the result) followed by n_bow divisions: nothing a GPU has to offer -- one lane, ~10 cycles per dependent v_add_f64 at
// <= 2.4 GHz against 4 cycles at the host's clock.  Measured (profiles/r06_*_bow_norm_ab.txt): the chain on the device
// lengthens k_bow_assemble by more than the whole host pass takes.  So the kernel leaves the values as addWeight made them
// and the host normalises what it copies out; -DVSG_BOW_NORM_DEVICE=1 builds the all-device form (same bytes, tested).
#ifndef VSG_BOW_NORM_DEVICE
#define VSG_BOW_NORM_DEVICE 0
#endif
constexpr bool kNormOnDevice = VSG_BOW_NORM_DEVICE != 0;
struct BowOut {
  int *hdr;  // {n_bow, n_fv, features with a non-stopped word, 0}
  int *bow_ids;
  double *bow_vals;
  int *fv_node, *fv_off, *fv_idx;
  int *r_hdr, *r_fv_node, *r_fv_off, *r_fv_idx;  // the frame's resident copy (nullptr: none)
};

__device__ __forceinline__ int block_exclusive_scan_2(int v, int *wsum /*[16]*/, int *total) {
  // exclusive scan of one value per thread over kAsmThreads threads (16 waves): DPP inside the wave (VALU latency per step:
  // row shifts with zero fill, then the row broadcasts), one LDS word per wave across
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
  inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, true);   // row_shr:1
  inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, true);   // row_shr:2
  inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, true);   // row_shr:4
  inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, true);   // row_shr:8
  inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kAsmThreads / 64; k++) {
    const int t = wsum[k];
    base += k < wave ? t : 0;
    tot += t;
  }
  *total = tot;
  __syncthreads();
  return base + inc - v;
}

// Step 1, all over the chip: every key's RANK among the frame's keys -- rank(i) = #{j : key_j < key_i}, n^2 comparisons that
// no one workgroup should make (a bitonic network over 2 x 2048 64-bit keys on ONE compute unit was built first and took
// 25 us of instruction issue alone) but that ceil(n / 64) workgroups of 16 waves finish in about a microsecond: a workgroup
// owns 64 keys (one per lane), each of its waves compares them with a sixteenth of all keys (staged in LDS, read at
// wave-uniform addresses) and the partial counts meet in LDS.  Keys are unique (the feature index is in them), so the ranks
// are a permutation: the key goes straight to its sorted position.
enum { kRankChunk = kAsmMax / 16 };
__global__ __launch_bounds__(kAsmThreads) void k_bow_rank(const int *__restrict__ word_of, const int *__restrict__ node_of,
                                                          const double *__restrict__ weight_of, int n,
                                                          uint64_t *__restrict__ sorted_w, uint64_t *__restrict__ sorted_n) {
  __shared__ uint64_t jw[16][kRankChunk], jn[16][kRankChunk];
  __shared__ int cw[16][64], cn[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t SENT = ~0ull;
  const int J = (n + 15) / 16;
  const int i = blockIdx.x * 64 + lane;
  const bool on_i = i < n && weight_of[i] > 0;  // w > 0: not a stopped word (TemplatedVocabulary.h:1170)
  const uint64_t kwi = on_i ? ((uint64_t)(uint32_t)word_of[i] << 32) | (uint32_t)i : SENT;
  const uint64_t kni = on_i ? ((uint64_t)(uint32_t)node_of[i] << 32) | (uint32_t)i : SENT;
  for (int q = lane; q < J; q += 64) {
    const int j = w * J + q;
    const bool on = j < n && weight_of[j] > 0;
    jw[w][q] = on ? ((uint64_t)(uint32_t)word_of[j] << 32) | (uint32_t)j : SENT;
    jn[w][q] = on ? ((uint64_t)(uint32_t)node_of[j] << 32) | (uint32_t)j : SENT;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  int c1 = 0, c2 = 0;
  for (int q = 0; q < J; q++) c1 += jw[w][q] < kwi, c2 += jn[w][q] < kni;  // sentinels are never below a key
  cw[w][lane] = c1, cn[w][lane] = c2;
  __syncthreads();
  if (w == 0 && on_i) {
    int r1 = 0, r2 = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) r1 += cw[k][lane], r2 += cn[k][lane];
    sorted_w[r1] = kwi, sorted_n[r2] = kni;
  }
}

// Step 2, two workgroups side by side: block 0 turns the sorted word keys into the BowVector, block 1 the sorted node keys
// into the FeatureVector (the BowVector's sequential norm is the long pole; the FeatureVector finishes in its shadow).
__global__ __launch_bounds__(kAsmThreads) void k_bow_assemble(const uint64_t *__restrict__ sorted_w,
                                                              const uint64_t *__restrict__ sorted_n,
                                                              const double *__restrict__ weight_of, int n, int tf,
                                                              int scoring, int normalize, BowOut o) {
  __shared__ uint64_t key[kAsmMax];
  __shared__ double vals[kAsmMax];
  __shared__ int wsum[kAsmThreads / 64];
  __shared__ double s_norm;
  const int tid = threadIdx.x;
  const bool words = blockIdx.x == 0;
  // m = features with a non-stopped word = keys in either sorted array; every global load of the prologue is requested
  // before the first is used (positions >= m of the sorted arrays were never written: masked once m is known)
  const uint64_t *src = words ? sorted_w : sorted_n;
  const int i0 = tid, i1 = tid + kAsmThreads;
  const double w0 = i0 < n ? weight_of[i0] : 0.0, w1 = i1 < n ? weight_of[i1] : 0.0;
  const uint64_t s0 = i0 < n ? src[i0] : ~0ull, s1 = i1 < n ? src[i1] : ~0ull;
  int m;
  (void)block_exclusive_scan_2((int)(w0 > 0) + (int)(w1 > 0), wsum, &m);
  key[i0] = i0 < m ? s0 : ~0ull, key[i1] = i1 < m ? s1 : ~0ull;
  __syncthreads();
  // two consecutive sorted positions per thread
  const int p0 = 2 * tid;
  auto head = [&](int p) { return p < m && (p == 0 || (key[p] >> 32) != (key[p - 1] >> 32)); };
  const int h0 = head(p0), h1 = head(p0 + 1);
  int n_heads;
  const int ex = block_exclusive_scan_2(h0 + h1, wsum, &n_heads);
  if (words) {  // ---- BowVector
    const int n_bow = n_heads;
    for (int q = 0; q < 2; q++) {
      const int p = p0 + q;
      if (!(q ? h1 : h0)) continue;
      const int j = ex + (q ? h0 : 0);
      int e = p + 1;
      while (e < m && (key[e] >> 32) == (key[p] >> 32)) e++;
      const double wi = weight_of[(uint32_t)key[p]];
      double val = wi;
      if (tf)
        for (int r = p + 1; r < e; r++) val += wi;  // addWeight, feature by feature (BowVector.cpp:34-47)
      vals[j] = val;
      o.bow_ids[j] = (int)(key[p] >> 32);
    }
    __syncthreads();
    if (!normalize) {  // the values as addWeight left them; normalize() runs on the host (see bow_finish)
      for (int j = tid; j < n_bow; j += kAsmThreads) o.bow_vals[j] = vals[j];
      if (tid == 0) o.hdr[0] = n_bow, o.hdr[2] = m, o.hdr[3] = 0;
      return;
    }
    const bool must = scoring != 5;  // DotProductScoring: no normalisation (ScoringObject.h:73-89)
    if (tid == 0) {
      double norm = 0.0;
      if (tf && !must) {
        norm = (double)n_bow;  // TemplatedVocabulary.h:1181-1186: divide by the number of words
      } else if (must) {       // BowVector::normalize (:62-84): L2 for L2Scoring, L1 otherwise -- ONE sum in id order
        // sixteen values requested at a time, then added IN ORDER (one dependent chain of n_bow additions is what the
        // reference's loop is; the LDS round trips need not be part of it)
        const bool l2 = scoring == 1;
        int j = 0;
        for (; j + 16 <= n_bow; j += 16) {
          double v[16];
#pragma unroll
          for (int q = 0; q < 16; q++) v[q] = vals[j + q];
#pragma unroll
          for (int q = 0; q < 16; q++) norm += l2 ? v[q] * v[q] : fabs(v[q]);
        }
        for (; j < n_bow; j++) norm += l2 ? vals[j] * vals[j] : fabs(vals[j]);
        if (l2) norm = sqrt(norm);
      }
      s_norm = norm;
      o.hdr[0] = n_bow, o.hdr[2] = m, o.hdr[3] = 0;
    }
    __syncthreads();
    const double norm = s_norm;
    const bool divide = (tf && !must && n_bow > 0) || (must && norm > 0.0);
    for (int j = tid; j < n_bow; j += kAsmThreads) o.bow_vals[j] = divide ? vals[j] / norm : vals[j];
  } else {  // ---- FeatureVector: node id -> the features below it, ascending (FeatureVector.cpp:31-45)
    const int n_fv = n_heads;
    for (int q = 0; q < 2; q++) {
      const int p = p0 + q;
      if (p < m) {
        const int f = (int)(uint32_t)key[p];
        o.fv_idx[p] = f;
        if (o.r_hdr) o.r_fv_idx[p] = f;
      }
      if (!(q ? h1 : h0)) continue;
      const int j = ex + (q ? h0 : 0), id = (int)(key[p] >> 32);
      o.fv_node[j] = id, o.fv_off[j] = p;
      if (o.r_hdr) o.r_fv_node[j] = id, o.r_fv_off[j] = p;
    }
    if (tid == 0) {
      o.fv_off[n_fv] = m, o.hdr[1] = n_fv;
      if (o.r_hdr) o.r_fv_off[n_fv] = m, o.r_hdr[0] = n_fv, o.r_hdr[1] = m;
    }
  }
}
}  // namespace

struct vsg_vocab {
  int device = 0, k = 0, L = 0, scoring = 0, weighting = 0, nnodes = 0, nwords = 0, max_children = 0;
  uint32_t root_link = 0;
  NodeRec *d_rec = nullptr;    // [nnodes - 1] child-list order
  double *d_weight = nullptr;  // same order
};

#define B_TRY(expr)                               \
  do {                                            \
    if ((expr) != hipSuccess) return VSG_ERR_HIP; \
  } while (0)

extern "C" {

void vsg_vocab_destroy(vsg_vocab *v) {
  if (!v) return;
  hipSetDevice(v->device);
  hipFree(v->d_rec), hipFree(v->d_weight);
  delete v;
}

int vsg_vocab_load(int device, const uint8_t *blob, size_t size, vsg_vocab **out) {
  if (!out || !blob) return VSG_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VSG_ERR_NO_DEVICE;
  // parse like loadFromBinFile (:1478-1552)
  size_t pos = 0;
  auto rd = [&](void *dst, size_t n) {
    if (pos + n > size) return false;
    memcpy(dst, blob + pos, n);
    pos += n;
    return true;
  };
  int k = 0, L = 0, n1 = 0, n2 = 0;
  if (!rd(&k, 4) || !rd(&L, 4) || !rd(&n1, 4) || !rd(&n2, 4)) return VSG_ERR_INVALID;
  if (k < 2 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) return VSG_ERR_INVALID;
  const long expected = (long)((std::pow((double)k, (double)L + 1) - 1) / (k - 1));
  // pass 1: the nodes as the file lists them (parents before children, TemplatedVocabulary.h:1519-1545)
  std::vector<int> parent(1, 0), word(1, 0), nchild(1, 0);
  std::vector<uint8_t> desc(32, 0);
  std::vector<double> weight(1, 0.0);
  if (expected > 0 && expected < (1 << 26)) {
    parent.reserve(expected), word.reserve(expected), nchild.reserve(expected), weight.reserve(expected);
    desc.reserve((size_t)expected * 32);
  }
  int nwords = 0;
  while (pos < size && (long)parent.size() < expected) {
    const int nid = (int)parent.size();
    int pid = 0;
    unsigned char leaf = 0;
    uint8_t d[32];
    double w = 0;
    if (!rd(&pid, 4) || !rd(&leaf, 1) || !rd(d, 32) || !rd(&w, 8) || pid < 0 || pid >= nid) break;
    parent.push_back(pid);
    desc.insert(desc.end(), d, d + 32);
    weight.push_back(w);
    word.push_back(leaf > 0 ? nwords++ : 0);
    nchild.push_back(0);
    nchild[pid]++;
  }
  const int nn = (int)parent.size();
  // pass 2: child-list positions (children of a node contiguous, in file order = the order transform() scans them)
  std::vector<int> first(nn + 1, 0), fill(nn, 0), posn(nn, -1);
  int max_children = 0;
  for (int i = 0; i < nn; i++) {
    first[i + 1] = first[i] + nchild[i];
    if (nchild[i] > max_children) max_children = nchild[i];
  }
  if (nn >= (1 << 24) || max_children > 32) return VSG_ERR_UNSUPPORTED;  // link = position : 24 | children : 8; <= 32 lanes
  for (int i = 1; i < nn; i++) posn[i] = first[parent[i]] + fill[parent[i]]++;
  std::vector<NodeRec> rec((size_t)(nn > 1 ? nn - 1 : 1));
  std::vector<double> wpos((size_t)(nn > 1 ? nn - 1 : 1), 0.0);
  for (int i = 1; i < nn; i++) {
    NodeRec &r = rec[posn[i]];
    memcpy(r.d, &desc[(size_t)i * 32], 32);
    r.link = (uint32_t)first[i] | ((uint32_t)nchild[i] << 24);
    r.node_id = i, r.word = word[i], r.pad = 0;
    wpos[posn[i]] = weight[i];
  }
  B_TRY(hipSetDevice(device));
  vsg_vocab *v = new vsg_vocab();
  v->device = device, v->k = k, v->L = L, v->scoring = n1, v->weighting = n2, v->nnodes = nn, v->nwords = nwords;
  v->max_children = max_children;
  v->root_link = (uint32_t)first[0] | ((uint32_t)nchild[0] << 24);
  hipError_t e = hipMalloc(&v->d_rec, rec.size() * sizeof(NodeRec));
  if (e == hipSuccess) e = hipMalloc(&v->d_weight, wpos.size() * 8);
  if (e == hipSuccess) e = hipMemcpy(v->d_rec, rec.data(), rec.size() * sizeof(NodeRec), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(v->d_weight, wpos.data(), wpos.size() * 8, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    vsg_vocab_destroy(v);
    return VSG_ERR_HIP;
  }
  *out = v;
  return VSG_OK;
}

int vsg_vocab_info(const vsg_vocab *v, int *k, int *L, int *scoring, int *weighting, int *nnodes, int *nwords) {
  if (!v) return VSG_ERR_INVALID;
  if (k) *k = v->k;
  if (L) *L = v->L;
  if (scoring) *scoring = v->scoring;
  if (weighting) *weighting = v->weighting;
  if (nnodes) *nnodes = v->nnodes;
  if (nwords) *nwords = v->nwords;
  return VSG_OK;
}

}  // extern "C" (vocabulary object)

// TemplatedVocabulary.h:1181-1190 + BowVector::normalize (BowVector.cpp:62-84) on the n values of a BowVector in id order
static void bow_normalize_host(const vsg_vocab *voc, double *v, int n) {
  const bool tf = voc->weighting == 0 || voc->weighting == 1;
  const bool must = voc->scoring != 5;  // DotProductScoring: no normalisation (ScoringObject.h:73-89)
  if (tf && n > 0 && !must) {
    const double nd = (double)n;
    for (int i = 0; i < n; i++) v[i] /= nd;
  }
  if (must) {  // L2 for L2Scoring, L1 otherwise
    double norm = 0.0;
    if (voc->scoring != 1) {
      for (int i = 0; i < n; i++) norm += std::fabs(v[i]);
    } else {
      for (int i = 0; i < n; i++) norm += v[i] * v[i];
      norm = std::sqrt(norm);
    }
    if (norm > 0.0)
      for (int i = 0; i < n; i++) v[i] /= norm;
  }
}

// The same assembly on the host, for frames of more than kAsmMax features (the device form sorts in LDS).
static int bow_assemble_host(const vsg_vocab *voc, int n, const int *word, const int *node, const double *w,
                             int32_t *bow_ids, double *bow_vals, int bow_cap, int *n_bow, int32_t *fv_node,
                             int32_t *fv_off, int32_t *fv_idx, int fv_cap, int *n_fv) {
  // ---- BowVector / FeatureVector assembly (:1158-1206).  The reference inserts feature by feature into two std::maps;
  // the same content falls out of two sorts of (id << 32 | feature) keys: ascending ids, features of an id in feature
  // order.  Every feature of a word carries the word's weight, so addWeight's running sum in feature order is the
  // weight added to itself (count - 1) times, left to right -- bit-identical doubles.
  static thread_local std::vector<uint64_t> kw, kn;
  kw.clear(), kn.clear();
  for (int i = 0; i < n; i++) {
    if (!(w[i] > 0)) continue;  // stopped word
    kw.push_back(((uint64_t)(uint32_t)word[i] << 32) | (uint32_t)i);
    kn.push_back(((uint64_t)(uint32_t)node[i] << 32) | (uint32_t)i);
  }
  std::sort(kw.begin(), kw.end());
  std::sort(kn.begin(), kn.end());
  const bool tf = voc->weighting == 0 || voc->weighting == 1;  // TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist
  static thread_local std::vector<std::pair<unsigned, double>> v;
  v.clear();
  for (size_t a = 0; a < kw.size();) {
    size_t b = a + 1;
    while (b < kw.size() && (kw[b] >> 32) == (kw[a] >> 32)) b++;
    const double wi = w[(uint32_t)kw[a]];
    double val = wi;
    if (tf)
      for (size_t r = a + 1; r < b; r++) val += wi;
    v.emplace_back((unsigned)(kw[a] >> 32), val);
    a = b;
  }
  const bool must = voc->scoring != 5;  // DotProductScoring: no normalisation (ScoringObject.h:73-89)
  if (tf && !v.empty() && !must) {
    const double nd = (double)v.size();
    for (auto &kv : v) kv.second /= nd;
  }
  if (must) {  // BowVector::normalize (:62-84): L2 for L2Scoring, L1 otherwise
    double norm = 0.0;
    if (voc->scoring != 1) {
      for (auto &kv : v) norm += std::fabs(kv.second);
    } else {
      for (auto &kv : v) norm += kv.second * kv.second;
      norm = std::sqrt(norm);
    }
    if (norm > 0.0)
      for (auto &kv : v) kv.second /= norm;
  }
  *n_bow = (int)v.size();
  int bi = 0;
  for (auto &kv : v) {
    if (bi < bow_cap && bow_ids && bow_vals) bow_ids[bi] = (int)kv.first, bow_vals[bi] = kv.second;
    bi++;
  }
  int j = 0, o = 0;
  for (size_t a = 0; a < kn.size();) {
    size_t b = a + 1;
    while (b < kn.size() && (kn[b] >> 32) == (kn[a] >> 32)) b++;
    if (j < fv_cap && fv_node && fv_idx) {
      fv_node[j] = (int)(kn[a] >> 32);
      for (size_t r = a; r < b; r++) fv_idx[o++] = (int)(uint32_t)kn[r];
      fv_off[j + 1] = o;
    }
    j++;
    a = b;
  }
  *n_fv = j;
  return (*n_bow > bow_cap || *n_fv > fv_cap) ? VSG_ERR_CAPACITY : VSG_OK;
}

// ---- one ComputeBoW on the calling thread's stream, in two halves so that other work of the same Frame (stereo matches,
// SearchByBoW) can be enqueued behind it and share ONE wait (vsg_chain.hip): enqueue() lays its blocks out at the given
// arena offsets and launches the descent + the assembly; finish() runs after the stream has been waited for and copies
// the results out of the pinned arena (frames of more than kAsmMax features: the same assembly on the host).
namespace vsg {

int vocab_device(const vsg_vocab *v) { return v ? v->device : -1; }

void bow_sizes(int n, bool host_desc, size_t *pin_bytes, size_t *dev_bytes) {
  Stage p, d;
  const size_t N = (size_t)(n > 0 ? n : 1);
  p.add(host_desc ? 32 * N : 0);                                     // descriptors going up
  p.add(8 * N), p.add(4 * N), p.add(4 * N);                          // per-feature weight / word / node mirrors
  p.add(64), p.add(4 * N), p.add(8 * N), p.add(4 * N), p.add(4 * (N + 1)), p.add(4 * N);  // hdr, bow ids / vals, fv
  d.add(host_desc ? 32 * N : 0), d.add(8 * N), d.add(4 * N), d.add(4 * N), d.add(8 * N), d.add(8 * N);
  *pin_bytes = p.total, *dev_bytes = d.total;
}

int bow_enqueue(BowCall *b, vsg_vocab *voc, const uint8_t *desc, const uint8_t *d_desc, int n, int levelsup,
                vsg_frame *resident, ThreadCtx *c, size_t pin_base, size_t dev_base) {
  b->voc = voc, b->n = n, b->c = c, b->pin_base = pin_base, b->active = false, b->device_assembly = false;
  if (resident) resident->fv_valid = false;
  if (n == 0 || voc->nnodes <= 1) return VSG_OK;  // empty() vocabulary: v and fv stay empty (:1147-1150)
  Stage p, d;
  const size_t N = (size_t)n;
  const size_t oD = p.add(d_desc ? 0 : 32 * N);
  b->oW = p.add(8 * N), b->oWord = p.add(4 * N), b->oNode = p.add(4 * N);
  b->oHdr = p.add(64), b->oBowId = p.add(4 * N), b->oBowVal = p.add(8 * N), b->oFvNode = p.add(4 * N);
  b->oFvOff = p.add(4 * (N + 1)), b->oFvIdx = p.add(4 * N);
  const size_t dD = d.add(d_desc ? 0 : 32 * N), dW = d.add(8 * N), dWord = d.add(4 * N), dNode = d.add(4 * N);
  const size_t dSortW = d.add(8 * N), dSortN = d.add(8 * N);
  uint8_t *hp = c->h_pin + pin_base, *dp = c->d_pin + pin_base, *dv = c->d_buf + dev_base;
  if (!d_desc) {
    memcpy(hp + oD, desc, 32 * N);
    B_TRY(hipMemcpyAsync(dv + dD, hp + oD, 32 * N, hipMemcpyHostToDevice, c->stream));
    d_desc = dv + dD;
  }
  int *s_word = (int *)(dv + dWord), *s_node = (int *)(dv + dNode);
  double *s_w = (double *)(dv + dW);
  if (voc->max_children <= 16)
    hipLaunchKernelGGL(k_bow_descend<16>, dim3((n * 16 + 255) / 256), dim3(256), 0, c->stream, voc->d_rec, voc->d_weight,
                       voc->root_link, d_desc, n, voc->L - levelsup, s_word, s_node, s_w, (int *)(dp + b->oWord),
                       (int *)(dp + b->oNode), (double *)(dp + b->oW));
  else
    hipLaunchKernelGGL(k_bow_descend<32>, dim3((n * 32 + 255) / 256), dim3(256), 0, c->stream, voc->d_rec, voc->d_weight,
                       voc->root_link, d_desc, n, voc->L - levelsup, s_word, s_node, s_w, (int *)(dp + b->oWord),
                       (int *)(dp + b->oNode), (double *)(dp + b->oW));
  B_TRY(hipGetLastError());
  b->active = true;
  b->device_assembly = n <= kAsmMax;
  if (b->device_assembly) {
    BowOut o;
    o.hdr = (int *)(dp + b->oHdr), o.bow_ids = (int *)(dp + b->oBowId), o.bow_vals = (double *)(dp + b->oBowVal);
    o.fv_node = (int *)(dp + b->oFvNode), o.fv_off = (int *)(dp + b->oFvOff), o.fv_idx = (int *)(dp + b->oFvIdx);
    const bool res = resident && resident->capacity >= n;
    o.r_hdr = res ? resident->d_fv_hdr : nullptr, o.r_fv_node = res ? resident->d_fv_node : nullptr;
    o.r_fv_off = res ? resident->d_fv_off : nullptr, o.r_fv_idx = res ? resident->d_fv_idx : nullptr;
    const int tf = voc->weighting == 0 || voc->weighting == 1;  // TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist
    uint64_t *s_sw = (uint64_t *)(dv + dSortW), *s_sn = (uint64_t *)(dv + dSortN);
    hipLaunchKernelGGL(k_bow_rank, dim3((n + 63) / 64), dim3(kAsmThreads), 0, c->stream, s_word, s_node, s_w, n, s_sw, s_sn);
    B_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_bow_assemble, dim3(2), dim3(kAsmThreads), 0, c->stream, s_sw, s_sn, s_w, n, tf, voc->scoring,
                       (int)kNormOnDevice, o);
    B_TRY(hipGetLastError());
    if (res) {
      // the joins of the SearchByBoW kernels launch one wavefront per POSSIBLE node of the FeatureVector's level
      long bound = 1;
      for (int l = 0; l < voc->L - levelsup && bound < n; l++) bound *= voc->max_children;
      resident->fv_bound = (int)(bound < n ? bound : n);
      resident->fv_valid = true;
    }
  }
  return VSG_OK;
}

int bow_finish(BowCall *b, int32_t *bow_ids, double *bow_vals, int bow_cap, int *n_bow, int32_t *fv_node,
               int32_t *fv_off, int32_t *fv_idx, int fv_cap, int *n_fv, int32_t *word_of, int32_t *node_of,
               double *weight_of) {
  *n_bow = *n_fv = 0;
  fv_off[0] = 0;
  if (!b->active) return VSG_OK;
  const int n = b->n;
  const uint8_t *hp = b->c->h_pin + b->pin_base;
  const int *word = (const int *)(hp + b->oWord), *node = (const int *)(hp + b->oNode);
  const double *w = (const double *)(hp + b->oW);
  if (word_of) memcpy(word_of, word, 4 * (size_t)n);
  if (node_of) memcpy(node_of, node, 4 * (size_t)n);
  if (weight_of) memcpy(weight_of, w, 8 * (size_t)n);
  if (b->device_assembly) {  // the kernel left everything in final form: copy, nothing else
    const int *hdr = (const int *)(hp + b->oHdr);
    const int nb = hdr[0], nf = hdr[1], m = hdr[2];
    *n_bow = nb, *n_fv = nf;
    if (bow_ids && bow_vals && nb <= bow_cap) {
      memcpy(bow_ids, hp + b->oBowId, 4 * (size_t)nb);
      memcpy(bow_vals, hp + b->oBowVal, 8 * (size_t)nb);
      if (!kNormOnDevice) bow_normalize_host(b->voc, bow_vals, nb);
    }
    if (fv_node && fv_idx && nf <= fv_cap) {
      memcpy(fv_node, hp + b->oFvNode, 4 * (size_t)nf);
      memcpy(fv_off, hp + b->oFvOff, 4 * (size_t)(nf + 1));
      memcpy(fv_idx, hp + b->oFvIdx, 4 * (size_t)m);
    }
    return (nb > bow_cap || nf > fv_cap) ? VSG_ERR_CAPACITY : VSG_OK;
  }
  return bow_assemble_host(b->voc, n, word, node, w, bow_ids, bow_vals, bow_cap, n_bow, fv_node, fv_off, fv_idx, fv_cap, n_fv);
}

}  // namespace vsg

static int bow_transform(vsg_vocab *voc, const uint8_t *desc, vsg_frame *f, int n, int levelsup,
                         int32_t *bow_ids, double *bow_vals, int bow_cap, int *n_bow, int32_t *fv_node,
                         int32_t *fv_off, int32_t *fv_idx, int fv_cap, int *n_fv, int32_t *word_of, int32_t *node_of,
                         double *weight_of) {
  if (!voc || n < 0 || !n_bow || !n_fv || !fv_off) return VSG_ERR_INVALID;
  *n_bow = *n_fv = 0;
  fv_off[0] = 0;
  int rc = VSG_OK;
  vsg::ThreadCtx *c = vsg::thread_ctx(voc->device, &rc);
  if (!c) return rc;
  size_t pin = 0, dev = 0;
  vsg::bow_sizes(n, f == nullptr, &pin, &dev);
  rc = vsg::ctx_reserve(c, pin, dev);
  if (rc != VSG_OK) return rc;
  vsg::BowCall b;
  rc = vsg::bow_enqueue(&b, voc, desc, f ? f->d_desc : nullptr, n, levelsup, f, c, 0, 0);
  if (rc != VSG_OK) return rc;
  if (b.active) B_TRY(hipStreamSynchronize(c->stream));
  return vsg::bow_finish(&b, bow_ids, bow_vals, bow_cap, n_bow, fv_node, fv_off, fv_idx, fv_cap, n_fv, word_of, node_of,
                         weight_of);
}


extern "C" {

int vsg_bow_transform(vsg_vocab *voc, const uint8_t *desc, int n, int levelsup, int32_t *bow_ids, double *bow_vals,
                      int bow_cap, int *n_bow, int32_t *fv_node, int32_t *fv_off, int32_t *fv_idx, int fv_cap,
                      int *n_fv, int32_t *word_of, int32_t *node_of, double *weight_of) {
  if (n > 0 && !desc) return VSG_ERR_INVALID;
  return bow_transform(voc, desc, nullptr, n, levelsup, bow_ids, bow_vals, bow_cap, n_bow, fv_node, fv_off, fv_idx,
                       fv_cap, n_fv, word_of, node_of, weight_of);
}

// Frame::ComputeBoW (Frame.cc:882-889) on the descriptors of a resident frame: nothing goes up
int vsg_frame_bow_transform(vsg_vocab *voc, vsg_frame *f, int levelsup, int32_t *bow_ids, double *bow_vals,
                            int bow_cap, int *n_bow, int32_t *fv_node, int32_t *fv_off, int32_t *fv_idx, int fv_cap,
                            int *n_fv, int32_t *word_of, int32_t *node_of, double *weight_of) {
  if (!voc || !f || f->device != voc->device) return VSG_ERR_INVALID;
  return bow_transform(voc, nullptr, f, f->n, levelsup, bow_ids, bow_vals, bow_cap, n_bow, fv_node, fv_off,
                       fv_idx, fv_cap, n_fv, word_of, node_of, weight_of);
}

}  // extern "C"
