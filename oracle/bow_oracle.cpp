/*
 * bow_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).  PARITY UNPINNED.
 * Restatement of the vendored, modified DBoW2 pieces that produce what ORBmatcher::SearchByBoW consumes
 * (SURVEY.md 8f N2):  orb_slam3/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h
 *   loadFromBinFile :1478-1552   transform(features, BowVector, FeatureVector, levelsup) :1139-1212
 *   transform(feature, word_id, weight, nid, levelsup) :1229-1271
 * BowVector.cpp:34-84 (addWeight / addIfNotExist / normalize), FeatureVector.cpp:31-45, FORB.cpp:81-101.
 * The real ORBvoc.txt.bin is a missing large blob, so tests use a synthetic vocabulary in the same binary layout.
 */
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "orb_oracle.h"

namespace {
struct Node {
  int parent = 0;
  std::vector<int> children;
  uint8_t descriptor[32] = {0};
  double weight = 0;
  int word_id = 0;
};
int forb_distance(const uint8_t *a, const uint8_t *b) {  // FORB.cpp:81-101
  int32_t pa[8], pb[8];
  memcpy(pa, a, 32);
  memcpy(pb, b, 32);
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    unsigned int v = pa[i] ^ pb[i];
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}
}  // namespace

struct OrVocab {
  int m_k = 0, m_L = 0, m_scoring = 0, m_weighting = 0;
  std::vector<Node> m_nodes;
  int nwords = 0;
};

extern "C" {

OrVocab *or_vocab_load(const uint8_t *blob, size_t size) {
  // TemplatedVocabulary::loadFromBinFile :1478-1552 on an in-memory image of the file
  OrVocab *v = new OrVocab();
  size_t pos = 0;
  auto rd = [&](void *dst, size_t n) {
    if (pos + n > size) return false;
    memcpy(dst, blob + pos, n);
    pos += n;
    return true;
  };
  int n1 = 0, n2 = 0;
  if (!rd(&v->m_k, 4) || !rd(&v->m_L, 4) || !rd(&n1, 4) || !rd(&n2, 4) || v->m_k < 0 || v->m_k > 20 || v->m_L < 1 ||
      v->m_L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
    delete v;
    return nullptr;
  }
  v->m_scoring = n1;
  v->m_weighting = n2;
  const int expected_nodes = (int)((pow((double)v->m_k, (double)v->m_L + 1) - 1) / (v->m_k - 1));
  v->m_nodes.resize(1);
  while (pos < size && (int)v->m_nodes.size() < expected_nodes) {
    const int nid = (int)v->m_nodes.size();
    int pid = 0;
    unsigned char leaf = 0;
    Node nd;
    if (!rd(&pid, 4) || !rd(&leaf, 1) || !rd(nd.descriptor, 32) || !rd(&nd.weight, 8) || pid < 0 || pid >= nid) break;
    nd.parent = pid;
    v->m_nodes.push_back(nd);
    v->m_nodes[pid].children.push_back(nid);
    if (leaf > 0) v->m_nodes[nid].word_id = v->nwords++;
  }
  return v;
}

void or_vocab_destroy(OrVocab *v) { delete v; }

int or_vocab_info(const OrVocab *v, int *k, int *L, int *scoring, int *weighting, int *nnodes, int *nwords) {
  *k = v->m_k, *L = v->m_L, *scoring = v->m_scoring, *weighting = v->m_weighting;
  *nnodes = (int)v->m_nodes.size(), *nwords = v->nwords;
  return 0;
}

int or_vocab_transform(const OrVocab *voc, const uint8_t *desc, int n, int levelsup, int *bow_ids, double *bow_vals,
                       int bow_cap, int *n_bow, int *fv_node, int *fv_off, int *fv_idx, int fv_cap, int *n_fv,
                       int *word_of, int *node_of, double *weight_of) {
  // TemplatedVocabulary::transform(features, v, fv, levelsup) :1139-1212
  std::map<unsigned, double> v;                  // BowVector
  std::map<unsigned, std::vector<unsigned>> fv;  // FeatureVector
  const std::vector<Node> &m_nodes = voc->m_nodes;
  const bool must = voc->m_scoring != 5;  // DotProductScoring does not normalize
  const bool l2 = voc->m_scoring == 1;
  for (int i_feature = 0; i_feature < n && m_nodes.size() > 1; i_feature++) {
    const uint8_t *feature = desc + (size_t)i_feature * 32;
    // transform(feature, word_id, weight, nid, levelsup) :1229-1271
    unsigned nid = 0;
    const int nid_level = voc->m_L - levelsup;
    unsigned final_id = 0;
    int current_level = 0;
    do {
      ++current_level;
      const std::vector<int> &nodes = m_nodes[final_id].children;
      final_id = nodes[0];
      double best_d = forb_distance(feature, m_nodes[final_id].descriptor);
      for (size_t c = 1; c < nodes.size(); c++) {
        const unsigned id = nodes[c];
        double d = forb_distance(feature, m_nodes[id].descriptor);
        if (d < best_d) {
          best_d = d;
          final_id = id;
        }
      }
      if (current_level == nid_level) nid = final_id;
    } while (!m_nodes[final_id].children.empty());
    const unsigned id = m_nodes[final_id].word_id;
    const double w = m_nodes[final_id].weight;
    if (word_of) word_of[i_feature] = (int)id;
    if (node_of) node_of[i_feature] = (int)nid;
    if (weight_of) weight_of[i_feature] = w;
    if (w > 0) {
      if (voc->m_weighting == 0 || voc->m_weighting == 1) {  // TF_IDF / TF: addWeight
        auto it = v.lower_bound(id);
        if (it != v.end() && it->first == id)
          it->second += w;
        else
          v.insert(it, std::make_pair(id, w));
      } else {  // IDF / BINARY: addIfNotExist
        if (v.find(id) == v.end()) v[id] = w;
      }
      fv[nid].push_back((unsigned)i_feature);
    }
  }
  if ((voc->m_weighting == 0 || voc->m_weighting == 1) && !v.empty() && !must) {
    const double nd = (double)v.size();
    for (auto &kv : v) kv.second /= nd;
  }
  if (must) {  // BowVector::normalize :62-84
    double norm = 0.0;
    if (!l2) {
      for (auto &kv : v) norm += fabs(kv.second);
    } else {
      for (auto &kv : v) norm += kv.second * kv.second;
      norm = sqrt(norm);
    }
    if (norm > 0.0)
      for (auto &kv : v) kv.second /= norm;
  }
  *n_bow = (int)v.size();
  int i = 0;
  for (auto &kv : v) {
    if (i < bow_cap) bow_ids[i] = (int)kv.first, bow_vals[i] = kv.second;
    i++;
  }
  *n_fv = (int)fv.size();
  int j = 0, o = 0;
  fv_off[0] = 0;
  for (auto &kv : fv) {
    if (j < fv_cap) {
      fv_node[j] = (int)kv.first;
      for (unsigned f : kv.second) fv_idx[o++] = (int)f;
      fv_off[j + 1] = o;
    }
    j++;
  }
  return 0;
}

}  // extern "C"
