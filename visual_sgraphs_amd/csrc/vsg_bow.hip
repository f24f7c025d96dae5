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

#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/vsg_orb.h"
#include "vsg_ctx.h"
#include "vsg_frame_int.h"

namespace {
__global__ void k_bow_descend(const int *child_off, const int *child_list, const uint8_t *node_desc,
                              const double *node_weight, const int *node_word, const uint8_t *desc, int n, int nid_level,
                              int *word_of, int *node_of, double *weight_of) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4 *f = (const uint4 *)(desc + (size_t)i * 32);
  const uint4 a0 = f[0], a1 = f[1];
  int final_id = 0, nid = 0, level = 0;
  while (child_off[final_id + 1] > child_off[final_id]) {  // !isLeaf()
    ++level;
    const int c0 = child_off[final_id], c1 = child_off[final_id + 1];
    int best = 0x7FFFFFFF;
    for (int c = c0; c < c1; c++) {
      const int id = child_list[c];
      const uint4 *d = (const uint4 *)(node_desc + (size_t)id * 32);
      const uint4 b0 = d[0], b1 = d[1];
      const int dist = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                       __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
      if (dist < best) {  // first child initialises, later ones need a strictly smaller distance (:1250-1259)
        best = dist;
        final_id = id;
      }
    }
    if (level == nid_level) nid = final_id;
  }
  word_of[i] = node_word[final_id];
  node_of[i] = nid;
  weight_of[i] = node_weight[final_id];
}
}  // namespace

struct vsg_vocab {
  int device = 0, k = 0, L = 0, scoring = 0, weighting = 0, nnodes = 0, nwords = 0;
  int *d_child_off = nullptr, *d_child_list = nullptr, *d_word = nullptr;
  uint8_t *d_desc = nullptr;
  double *d_weight = nullptr;
};

#define B_TRY(expr)                               \
  do {                                            \
    if ((expr) != hipSuccess) return VSG_ERR_HIP; \
  } while (0)

extern "C" {

void vsg_vocab_destroy(vsg_vocab *v) {
  if (!v) return;
  hipSetDevice(v->device);
  hipFree(v->d_child_off), hipFree(v->d_child_list), hipFree(v->d_word), hipFree(v->d_desc), hipFree(v->d_weight);
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
  std::vector<int> parent(1, 0), word(1, 0);
  std::vector<uint8_t> desc(32, 0);
  std::vector<double> weight(1, 0.0);
  std::vector<std::vector<int>> children(1);
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
    children.push_back({});
    children[pid].push_back(nid);
  }
  const int nn = (int)parent.size();
  std::vector<int> off(nn + 1, 0), list;
  for (int i = 0; i < nn; i++) {
    off[i] = (int)list.size();
    list.insert(list.end(), children[i].begin(), children[i].end());
  }
  off[nn] = (int)list.size();
  B_TRY(hipSetDevice(device));
  vsg_vocab *v = new vsg_vocab();
  v->device = device, v->k = k, v->L = L, v->scoring = n1, v->weighting = n2, v->nnodes = nn, v->nwords = nwords;
  hipError_t e = hipMalloc(&v->d_child_off, 4 * (size_t)(nn + 1));
  if (e == hipSuccess) e = hipMalloc(&v->d_child_list, 4 * (list.size() + 1));
  if (e == hipSuccess) e = hipMalloc(&v->d_word, 4 * (size_t)nn);
  if (e == hipSuccess) e = hipMalloc(&v->d_desc, 32 * (size_t)nn);
  if (e == hipSuccess) e = hipMalloc(&v->d_weight, 8 * (size_t)nn);
  if (e == hipSuccess) e = hipMemcpy(v->d_child_off, off.data(), 4 * (size_t)(nn + 1), hipMemcpyHostToDevice);
  if (e == hipSuccess && !list.empty()) e = hipMemcpy(v->d_child_list, list.data(), 4 * list.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(v->d_word, word.data(), 4 * (size_t)nn, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(v->d_desc, desc.data(), 32 * (size_t)nn, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(v->d_weight, weight.data(), 8 * (size_t)nn, hipMemcpyHostToDevice);
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
  hipLaunchKernelGGL(k_bow_descend, dim3((n + 63) / 64), dim3(64), 0, c->stream, voc->d_child_off, voc->d_child_list,
                     voc->d_desc, voc->d_weight, voc->d_word, d_desc, n, voc->L - levelsup, (int *)(c->d_pin + oWord),
                     (int *)(c->d_pin + oNode), (double *)(c->d_pin + oW));
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
  // ---- BowVector / FeatureVector assembly in feature order (:1158-1206)
  std::map<unsigned, double> v;
  std::map<unsigned, std::vector<unsigned>> fv;
  const bool tf = voc->weighting == 0 || voc->weighting == 1;  // TF_IDF, TF
  for (int i = 0; i < n; i++) {
    if (word_of) word_of[i] = word[i];
    if (node_of) node_of[i] = node[i];
    if (weight_of) weight_of[i] = w[i];
    if (!(w[i] > 0)) continue;  // stopped word
    const unsigned id = (unsigned)word[i];
    auto it = v.lower_bound(id);
    if (it != v.end() && it->first == id) {
      if (tf) it->second += w[i];  // addWeight; addIfNotExist leaves it
    } else {
      v.insert(it, std::make_pair(id, w[i]));
    }
    fv[(unsigned)node[i]].push_back((unsigned)i);
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
  *n_fv = (int)fv.size();
  int j = 0, o = 0;
  for (auto &kv : fv) {
    if (j < fv_cap && fv_node && fv_idx) {
      fv_node[j] = (int)kv.first;
      for (unsigned f : kv.second) fv_idx[o++] = (int)f;
      fv_off[j + 1] = o;
    }
    j++;
  }
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
