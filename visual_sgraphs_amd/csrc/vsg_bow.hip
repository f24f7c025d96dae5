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
                                                     double *__restrict__ weight_of) {
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
    if (j == jw) {  // the winner's lane holds everything the outputs need
      if (level == nid_level) nid = myid, node_of[i] = myid;
      if (!(link >> 24)) {
        word_of[i] = myword;
        weight_of[i] = weight[first + jw];
        if (nid_level > level || nid_level <= 0) node_of[i] = 0;  // levelsup beyond the tree height: nid stays 0
      }
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

// tree descent on the calling thread's stream; d_desc = descriptors on the device (resident frame) or nullptr (host
// descriptors are staged through the arena); per-feature word / node / weight come back through the pinned arena
static int bow_descend(vsg_vocab *voc, const uint8_t *desc, const uint8_t *d_desc, int n, int levelsup,
                       const int **word, const int **node, const double **w) {
  int rc = VSG_OK;
  vsg::ThreadCtx *c = vsg::thread_ctx(voc->device, &rc);
  if (!c) return rc;
  vsg::Stage st;
  const size_t oD = st.add(d_desc ? 0 : 32 * (size_t)n);
  const size_t in_bytes = st.total;
  const size_t oW = st.add(8 * (size_t)n), oWord = st.add(4 * (size_t)n), oNode = st.add(4 * (size_t)n);
  rc = vsg::ctx_reserve(c, st.total, in_bytes + 64);
  if (rc != VSG_OK) return rc;
  if (!d_desc) {
    memcpy(c->h_pin + oD, desc, 32 * (size_t)n);
    B_TRY(hipMemcpyAsync(c->d_buf, c->h_pin, in_bytes, hipMemcpyHostToDevice, c->stream));
    d_desc = c->d_buf + oD;
  }
  if (voc->max_children <= 16)
    hipLaunchKernelGGL(k_bow_descend<16>, dim3((n * 16 + 255) / 256), dim3(256), 0, c->stream, voc->d_rec, voc->d_weight,
                       voc->root_link, d_desc, n, voc->L - levelsup, (int *)(c->d_pin + oWord), (int *)(c->d_pin + oNode),
                       (double *)(c->d_pin + oW));
  else
    hipLaunchKernelGGL(k_bow_descend<32>, dim3((n * 32 + 255) / 256), dim3(256), 0, c->stream, voc->d_rec, voc->d_weight,
                       voc->root_link, d_desc, n, voc->L - levelsup, (int *)(c->d_pin + oWord), (int *)(c->d_pin + oNode),
                       (double *)(c->d_pin + oW));
  B_TRY(hipGetLastError());
  B_TRY(hipStreamSynchronize(c->stream));
  *word = (const int *)(c->h_pin + oWord);
  *node = (const int *)(c->h_pin + oNode);
  *w = (const double *)(c->h_pin + oW);
  return VSG_OK;
}

static int bow_transform(vsg_vocab *voc, const uint8_t *desc, const uint8_t *d_desc, int n, int levelsup,
                         int32_t *bow_ids, double *bow_vals, int bow_cap, int *n_bow, int32_t *fv_node,
                         int32_t *fv_off, int32_t *fv_idx, int fv_cap, int *n_fv, int32_t *word_of, int32_t *node_of,
                         double *weight_of) {
  if (!voc || n < 0 || !n_bow || !n_fv || !fv_off) return VSG_ERR_INVALID;
  *n_bow = *n_fv = 0;
  fv_off[0] = 0;
  if (n == 0 || voc->nnodes <= 1) return VSG_OK;  // empty() vocabulary: v and fv stay empty (:1147-1150)
  const int *word = nullptr, *node = nullptr;
  const double *w = nullptr;
  int rc = bow_descend(voc, desc, d_desc, n, levelsup, &word, &node, &w);
  if (rc != VSG_OK) return rc;
  // ---- BowVector / FeatureVector assembly (:1158-1206).  The reference inserts feature by feature into two std::maps;
  // the same content falls out of two sorts of (id << 32 | feature) keys: ascending ids, features of an id in feature
  // order.  Every feature of a word carries the word's weight, so addWeight's running sum in feature order is the
  // weight added to itself (count - 1) times, left to right -- bit-identical doubles.
  static thread_local std::vector<uint64_t> kw, kn;
  kw.clear(), kn.clear();
  for (int i = 0; i < n; i++) {
    if (word_of) word_of[i] = word[i];
    if (node_of) node_of[i] = node[i];
    if (weight_of) weight_of[i] = w[i];
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
  return bow_transform(voc, nullptr, f->d_desc, f->n, levelsup, bow_ids, bow_vals, bow_cap, n_bow, fv_node, fv_off,
                       fv_idx, fv_cap, n_fv, word_of, node_of, weight_of);
}

}  // extern "C"
