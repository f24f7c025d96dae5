// vsg_match.hip -- Hamming searches of ORBmatcher (orb_slam3/src/ORBmatcher.cc) on flattened POD views.
//
// Integer XOR + popcount work (v_xor_b32 / v_bcnt_u32_b32), no MFMA.  Distances are evaluated lane-parallel;
// every greedy "already matched" decision of the reference stays ordered:
//   * SearchByBoW: vocabulary nodes are independent (an F feature belongs to one node), so one wavefront per
//     shared node walks that node's KF features in order, lanes over the node's F features;
//   * SearchByProjection / SearchForInitialization: one wavefront walks the queries in order, lanes over the
//     query's candidate list; the mutable state lives in global memory behind a wave-level fence.
// best / second-best follow the reference's strict '<' scan: packed keys (dist << 20 | scan position) make
// "earliest candidate wins ties" a plain integer minimum.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include <algorithm>

#include "../../include/vsg_orb.h"
#include "vsg_ctx.h"
#include "vsg_frame_int.h"
#include "vsg_math.h"
#include "vsg_walks.h"

namespace {

const int TH_HIGH = 100;      // ORBmatcher.cc:34
const int TH_LOW = 50;        // ORBmatcher.cc:35
const int HISTO_LENGTH = 30;  // ORBmatcher.cc:36
const uint32_t KEY_NONE = (256u << 20) | 0xFFFFFu;
// The reference's scans start from bestDist = 256 and update on a strict '<', so a candidate at distance 256 (every
// bit different) is never selected: any key whose distance field is 256 means "no best".
__device__ __forceinline__ bool key_is_none(uint32_t k) { return (k >> 20) >= 256u; }

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1) {
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}
__device__ __forceinline__ void load_desc(const uint8_t *base, int row, uint4 &lo, uint4 &hi) {
  const uint4 *p = (const uint4 *)(base + (size_t)row * 32);
  lo = p[0];
  hi = p[1];
}

// two smallest of {k1,k2} U {o1,o2}, each pair already ordered
__device__ __forceinline__ void merge2(uint32_t &k1, uint32_t &k2, uint32_t o1, uint32_t o2) {
  if (o1 < k1) {
    k2 = min(k1, o2);
    k1 = o1;
  } else {
    k2 = min(k2, o1);
  }
}
__device__ __forceinline__ void wave_best2(uint32_t &k1, uint32_t &k2) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o1 = __shfl_xor(k1, d), o2 = __shfl_xor(k2, d);
    merge2(k1, k2, o1, o2);
  }
}
// the same minimum over the DPP path (row shifts, then the row broadcasts; lanes without a source keep the identity): VALU
// latency per step instead of an LDS round trip per __shfl_xor -- what the ordered walk of k_search_by_bow is made of
__device__ __forceinline__ uint32_t wave_min_dpp(uint32_t v) {
  const int I = -1;
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(I, (int)v, 0x111, 0xF, 0xF, false));  // row_shr:1
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(I, (int)v, 0x112, 0xF, 0xF, false));  // row_shr:2
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(I, (int)v, 0x114, 0xF, 0xF, false));  // row_shr:4
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(I, (int)v, 0x118, 0xF, 0xF, false));  // row_shr:8
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(I, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1, 3
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(I, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_min(uint32_t k) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) k = min(k, __shfl_xor(k, d));
  return k;
}

// ---- DescriptorDistance over index pairs
__global__ void k_hamming_pairs(const uint8_t *a, const uint8_t *b, const int *ia, const int *ib, int n, int *dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint4 a0, a1, b0, b1;
  load_desc(a, ia[i], a0, a1);
  load_desc(b, ib[i], b0, b1);
  dist[i] = hamming256(a0, a1, b0, b1);
}

// ---- brute-force best / second best.  Packed keys (dist << 20 | index) are unique and keep the reference's strict-'<'
// scan semantics: the smallest key is the first minimum, the second smallest is "bestDist2".  With k1 <= k2 the update is
// k2 = median(k1, k2, key), k1 = min(k1, key): one v_med3_u32 and one v_min_u32 per pair.
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
  return min(max(a, b), max(min(a, b), c));  // folded to v_med3_u32
}
#ifndef VSG_MATCH_QW
#define VSG_MATCH_QW 2
#endif
enum { kMatchQW = VSG_MATCH_QW, kBest2Rows = 128 * kMatchQW };  // query sets of 32 per wave; A rows per workgroup

// ---- the best / second-best scan on the matrix cores.  A Hamming distance matrix is a GEMM over +-1 vectors:
// with s(x) = 2*bit - 1, sum_k s(a_k) * s(b_k) = 256 - 2 * dist.  The train rows are expanded with their bits
// INVERTED (so each product is negated), hence
//     D[i][j] = sum_k (-s(train_i,k)) * s(query_j,k) = 2 * dist(train_i, query_j) - 256
// The operands are FP4 (e2m1: +1 = 0x2, -1 = 0xA; block scales 2^0) for `v_mfma_scale_f32_32x32x64_f8f6f4`: K = 64 per
// instruction at the 32 cycles of the i8 form's K = 32 (tools/ubench.hip), so a 32 x 32 tile of 256-bit distances is FOUR
// instructions (round 4; rounds 1-3: eight v_mfma_i32_32x32x32_i8 -- 0.148 -> 0.111 ms per 512 x 1006^2 pairs, + 2.2 %
// frames/s in the same run), a descriptor dword expands into ONE 16-byte operand through a byte -> 8-nibble table and a
// tile of train rows is 4 KB of LDS.  Sums of +-1 products are exact in fp32.  A wave owns 32 queries (the MFMA's columns,
// kept expanded in 16 VGPRs for the whole scan); the workgroup expands each tile of 32 train rows once into LDS for its
// four waves.  C/D layout: lane = column + 32 * h, register g holds row (g & 3) + 8 * (g >> 2) + 4 * h, so a lane
// folds its 16 rows into a best/second pair of FLOAT keys 32 * acc + row offset (exact, lexicographic in (acc, row)) with
// v_fma_f32 + v_min_f32 + v_med3_f32 per pair, turns the two survivors into the packed integer keys
// (2 * dist << 19 | index == dist << 20 | index) and the two lanes of a column merge at the end: tie handling follows the
// packed-key rule above (an xor + popcount form of this scan measured 0.182 ms per 256 x 1006^2 pairs against 0.088 ms
// for the i8 form: DESIGN 8, round 1).
#ifndef VSG_MATCH_TR
#define VSG_MATCH_TR 1
#endif
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_block_best2_mfma(const uint8_t *a_base, const uint8_t *b_base,
                                                          size_t block_stride, const int *counts_a, const int *counts_b,
                                                          int count_stride, int fixed_na, int fixed_nb, int max_rows,
                                                          int *best, int *second, int *argbest) {
  constexpr int TR = VSG_MATCH_TR;  // 32-row train tiles per iteration (independent MFMA chains, one barrier)
  __shared__ i32x4 tiles[2][TR][8 * 32];  // double buffer of [tile][descriptor dword d = 2 * k-step + half][train row r]
  __shared__ uint32_t lut[256];           // byte -> its 8 bits as FP4 nibbles
  {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) m |= ((threadIdx.x >> i) & 1u) << (4 * i + 3);
    lut[threadIdx.x] = 0xAAAAAAAAu ^ m;  // bit set -> 0x2 (+1), bit clear -> 0xA (-1)
  }
  const int blk = blockIdx.y;
  const int na = counts_a ? min(counts_a[blk * count_stride], max_rows) : fixed_na;
  const int nb = counts_b ? min(counts_b[blk * count_stride], max_rows) : fixed_nb;
  const uint8_t *A = a_base + (size_t)blk * block_stride, *B = b_base + (size_t)blk * block_stride;
  if (blockIdx.x * kBest2Rows >= na) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  // A wave owns QW sets of 32 queries (round 6: two): the train operand read from LDS feeds QW matrix instructions, whose
  // accumulator chains are independent, and the workgroup's tile expansion serves twice the queries.  Per 1024 x 1006^2 pairs:
  // QW 1 / 2 / 3 / 4 = 0.204 / 0.186 / 0.199 / 0.221 ms at 78 / 114 / 154 / 186 VGPRs (6 / 4 / 3 / 2 waves per SIMD); two sets forced
  // into 96 registers spill and take 0.50 ms (profiles/r06_f_*)
  constexpr int QW = kMatchQW;
  const int q0 = blockIdx.x * kBest2Rows + wave * 32 * QW + r;  // this lane's query (column) of set 0; set u: + 32 u
  const bool wave_active = blockIdx.x * kBest2Rows + wave * 32 * QW < na;
  // queries: dword 2 s + h of the descriptor is the lane's 32 K-values of k-step s (expanded below, once the table is there)
  uint32_t qraw[QW][4];
#pragma unroll
  for (int u = 0; u < QW; u++) {
    const int q = q0 + 32 * u;
    const uint32_t *qd = (const uint32_t *)(A + (size_t)(q < na ? q : 0) * 32);
#pragma unroll
    for (int s = 0; s < 4; s++) qraw[u][s] = qd[2 * s + h];
  }
  uint32_t k1[QW], k2[QW];
#pragma unroll
  for (int u = 0; u < QW; u++) k1[u] = KEY_NONE, k2[u] = KEY_NONE;
  // tile expansion: thread = (train row, dword); the row runs along the lanes so that a wave's 16-byte LDS stores
  // are contiguous (the dword-fastest mapping put 8 lanes on the same banks: an 8-way conflict on every store)
  const int erow = tid & 31, es = tid >> 5;
  auto fetch = [&](int row0) {  // one dword of a train row, inverted (negated products); past the end: any valid row
    const int tr = row0 + erow;
    return ~((const uint32_t *)(B + (size_t)(tr < nb ? tr : 0) * 32))[es];
  };
  // The train dwords are prefetched a GROUP of kPf tiles ahead (register double bank): an iteration is ~600 cycles, a global
  // round trip several thousand, so a one-iteration prefetch left every iteration waiting on memory.  With two query sets per
  // wave: groups of 4 / 6 / 8 / 10 tiles = 0.185 / 0.200 / 0.181 / 0.188 ms per 1024 x 1006^2 pairs (122 VGPRs at 8: still 4 waves
  // per SIMD; profiles/r06_f_*)
#ifndef VSG_MATCH_PF
#define VSG_MATCH_PF 8
#endif
  constexpr int kPf = VSG_MATCH_PF;
  uint32_t wa[kPf], wn[kPf];
#pragma unroll
  for (int j = 0; j < kPf; j++) wa[j] = fetch(32 * j), wn[j] = 0u;
  __syncthreads();  // lut
  const f32x16 zerof = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto expand4 = [&](uint32_t w) {
    return (i32x8){(int)lut[w & 255], (int)lut[(w >> 8) & 255], (int)lut[(w >> 16) & 255], (int)lut[w >> 24], 0, 0, 0, 0};
  };
  i32x8 Q[QW][4];
#pragma unroll
  for (int u = 0; u < QW; u++)
#pragma unroll
    for (int s = 0; s < 4; s++) Q[u][s] = expand4(qraw[u][s]);
  const int kScale = 0x7F7F7F7F;  // e8m0 block scales: 2^0
  int buf = 0;
  for (int tg = 0; tg < nb; tg += 32 * kPf) {
    if (tg + 32 * kPf < nb) {
#pragma unroll
      for (int j = 0; j < kPf; j++) wn[j] = fetch(tg + 32 * (kPf + j));
    }
#pragma unroll
    for (int j = 0; j < kPf; j += TR) {
      const int t0 = tg + 32 * j;
      if (t0 >= nb) break;
#pragma unroll
      for (int t = 0; t < TR; t++) {
        i32x4 *tile = tiles[buf][t];
        const uint32_t ww = wa[j + t];
        tile[es * 32 + erow] = (i32x4){(int)lut[ww & 255], (int)lut[(ww >> 8) & 255], (int)lut[(ww >> 16) & 255], (int)lut[ww >> 24]};
      }
      // one barrier per iteration: nobody can overwrite this buffer before every wave has passed the NEXT
      // barrier, i.e. finished reading it
      __syncthreads();
      if (wave_active) {
        // acc = sum of negated products = 2 * dist - 256 (C operand: inline 0)
        f32x16 acc[QW][TR];
        auto opA = [&](int t, int s) {
          const i32x4 v = tiles[buf][t][(s * 2 + h) * 32 + r];
          return (i32x8){v.x, v.y, v.z, v.w, 0, 0, 0, 0};
        };
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
          for (int t = 0; t < TR; t++) {
            const i32x8 a = opA(t, s);  // ONE operand read for the QW instructions of this k-step
#pragma unroll
            for (int u = 0; u < QW; u++)
              acc[u][t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, Q[u][s], s == 0 ? zerof : acc[u][t], 4, 4, 0, kScale, 0,
                                                                          kScale);
          }
#pragma unroll
        for (int t = 0; t < TR; t++) {
          const int tb = t0 + 32 * t + 4 * h;  // row of register 0 of this lane
          if (tb - 4 * h + 32 <= nb) {
            // full tile: the best two of the lane's 16 rows on FLOAT keys 32 * acc + row offset (exact: |32 acc + 27| < 2^14;
            // the offset < 32 keeps (acc, row) in lexicographic order): one v_fma_f32, one v_min_f32, one v_med3_f32 per
            // pair; the two survivors become the integer keys (dist << 20 | row) the rest of the kernel works on
#pragma unroll
            for (int u = 0; u < QW; u++) {
              float f1 = __builtin_inff(), f2 = __builtin_inff();
#pragma unroll
              for (int g = 0; g < 16; g++) {
                const float kf = __builtin_fmaf(acc[u][t][g], 32.0f, (float)((g & 3) + 8 * (g >> 2)));
                f2 = __builtin_amdgcn_fmed3f(f1, f2, kf);
                f1 = __builtin_fminf(f1, kf);
              }
              const int i1 = (int)f1, i2 = (int)f2;  // 32 acc + offset: acc = i >> 5 (floor), offset = i & 31
              const uint32_t l1 = ((uint32_t)((i1 >> 5) + 256) << 19) + (uint32_t)((i1 & 31) + tb);
              const uint32_t l2 = ((uint32_t)((i2 >> 5) + 256) << 19) + (uint32_t)((i2 & 31) + tb);
              k2[u] = min(max(k1[u], l1), min(k2[u], l2));
              k1[u] = min(k1[u], l1);
            }
          } else {
#pragma unroll
            for (int u = 0; u < QW; u++)
#pragma unroll
              for (int g = 0; g < 16; g++) {
                const int row = tb + (g & 3) + 8 * (g >> 2);
                uint32_t key = ((uint32_t)((int)acc[u][t][g] + 256) << 19) | (uint32_t)row;
                if (row >= nb) key = KEY_NONE;
                k2[u] = umed3(k1[u], k2[u], key);
                k1[u] = min(k1[u], key);
              }
          }
        }
      }
      buf ^= 1;
    }
#pragma unroll
    for (int j = 0; j < kPf; j++) wa[j] = wn[j];
  }
#pragma unroll
  for (int u = 0; u < QW; u++) {
    const uint32_t o1 = __shfl_xor(k1[u], 32), o2 = __shfl_xor(k2[u], 32);
    merge2(k1[u], k2[u], o1, o2);
    const int q = q0 + 32 * u;
    if (h == 0 && q < na) {
      const size_t o = (size_t)blk * max_rows + q;
      best[o] = (int)(k1[u] >> 20);
      second[o] = (int)(k2[u] >> 20);
      argbest[o] = key_is_none(k1[u]) ? -1 : (int)(k1[u] & 0xFFFFF);
    }
  }
}

// ---- SearchByBoW: one wavefront per shared vocabulary node (node pairs merged on the host).
// mode 0: KF -> Frame  (ORBmatcher.cc:254-392): skip F features already in match_f; accept bestDist1 <= TH_LOW
// mode 1: KF -> KF     (ORBmatcher.cc:790-864): skip vbMatched2 / invalid; accept bestDist1 <  TH_LOW
struct NodePair {
  int a_begin, a_end, b_begin, b_end;
};
enum { kBowNodeSide = 128 };  // nodes up to this many features a side are matched on a distance matrix in LDS

// A FeatureVector resident on the device (vsg_frame: Frame::mFeatVec written by k_bow_assemble): hdr = {nodes, features}
struct FvDev {
  const int *hdr, *node, *off, *idx;
};

// One WORKGROUP per shared vocabulary node (round 6; rounds 1-5: one wavefront, whose ordered walk chained three dependent
// global loads and a fence per KeyFrame feature -- 37 us per call, all of it the largest node's walk).
//   * pairs == nullptr: BOTH FeatureVectors are resident and the join of ORBmatcher.cc:247-405 happens here -- block b takes
//     node b of A and finds the same id in B's ascending node list by bisection (a node is shared or it is not: the
//     merge-join's lower_bound jumps and this lookup name the same pairs; nodes are independent, their order is irrelevant).
//   * The node's descriptors go to LDS once (na + nb rows, not na x nb row pairs), every distance of the node is computed
//     from there in one sweep of all 256 lanes.
//   * The ordered, greedy walk (a KeyFrame feature sees the claims of the ones BEFORE it) as a fixed point, one lane per
//     KeyFrame feature: in round t every feature decides on the claims that the features before it made in round t - 1.
//     By induction feature i is final from round i + 1 on, and a round that changes no decision is the walk's result
//     (feature 0 never depended on anyone, feature 1 is consistent with feature 0's final claim, ...).  Conflicts between
//     neighbours are rare, so a node settles in 2-4 rounds of ~nb LDS reads per lane instead of na dependent reductions.
// matchA / matchB are WRITE-ONLY (pre-filled with -1 by whoever owns them: they may be pinned host memory); every feature
// belongs to exactly one node, so a node's block is the only writer and the only reader of its claims.
__global__ __launch_bounds__(256) void k_search_by_bow(const NodePair *pairs, int npairs, FvDev fa, FvDev fb,
                                                       const uint8_t *descA,
                                                       const uint8_t *validA, const int *idxA, const uint8_t *descB,
                                                       const uint8_t *validB, const int *idxB, float nnratio, int mode,
                                                       int nleftB /*mode0: F.Nleft or -1*/,
                                                       int *matchA /*mode1: matches12*/, int *matchB /*mode0: match_f*/,
                                                       int *claimB /*device scratch [nB]: large nodes only*/) {
  const int node = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  NodePair np;
  if (pairs) {
    if (node >= npairs) return;
    np = pairs[node];
  } else {
    // everything the join needs is requested at once (B's node list 256 entries at a time, whatever the headers say: the
    // arrays hold capacity + 1 entries): ONE memory round trip in front of the offsets instead of a bisection's eight
    __shared__ int s_pos;
    const int nA_nodes = fa.hdr[0], nB_nodes = fb.hdr[0];
    const int id = node < nA_nodes ? fa.node[node] : -1;
    if (tid == 0) s_pos = -1;
    __syncthreads();
    for (int base = 0; base < nB_nodes; base += 256)  // block-uniform; one pass for the reference's ~100 nodes
      if (base + tid < nB_nodes && fb.node[base + tid] == id) s_pos = base + tid;  // ids are unique: at most one writer
    __syncthreads();
    const int lo = s_pos;
    if (node >= nA_nodes || lo < 0) return;
    np.a_begin = fa.off[node], np.a_end = fa.off[node + 1], np.b_begin = fb.off[lo], np.b_end = fb.off[lo + 1];
    idxA = fa.idx, idxB = fb.idx;
  }
  const int nb = np.b_end - np.b_begin, na = np.a_end - np.a_begin;
  __shared__ uint4 s_da[kBowNodeSide][2], s_db[kBowNodeSide][2];
  __shared__ uint16_t s_dist[kBowNodeSide * kBowNodeSide];
  __shared__ int s_rb[kBowNodeSide], s_ra[kBowNodeSide], s_by0[kBowNodeSide], s_by[kBowNodeSide];
  __shared__ int s_changed[2];  // by round parity: a flag is cleared a whole round before it is written again
  if (na <= kBowNodeSide && nb <= kBowNodeSide) {
    const int kFree = 0x7FFFFFFF;
    {  // rows of both sides: thread = (row, half)
      const int r = tid >> 1, h = tid & 1;
      if (r < na) {
        const int ra = idxA[np.a_begin + r];
        s_da[r][h] = ((const uint4 *)(descA + (size_t)ra * 32))[h];
        if (h == 0) s_ra[r] = validA[ra] ? ra : -1;  // !pMP || pMP->isBad()
      }
      if (tid < 2) s_changed[tid] = 0;
      if (r < nb) {
        const int rb = idxB[np.b_begin + r];
        s_db[r][h] = ((const uint4 *)(descB + (size_t)rb * 32))[h];
        if (h == 0) s_rb[r] = rb, s_by0[r] = s_by[r] = (mode == 1 && !validB[rb]) ? -1 : kFree;  // -1: never a candidate
      }
    }
    __syncthreads();
    const float inv_nb = 1.0f / (float)nb;
    for (int p = tid; p < na * nb; p += 256) {
      const int i = min((int)(((float)p + 0.5f) * inv_nb), na - 1), j = p - i * nb;
      s_dist[p] = (uint16_t)hamming256(s_da[i][0], s_da[i][1], s_db[j][0], s_db[j][1]);
    }
    __syncthreads();
    const int i = tid;
    const int ra = i < na ? s_ra[i] : -1;
    const bool active = ra >= 0;
    int decL = -1, decR = -1;
    for (int round = 0; round <= na; round++) {  // block-uniform
      int newL = -1, newR = -1;
      if (active) {
        uint32_t k1 = KEY_NONE, k2 = KEY_NONE, r1 = KEY_NONE;
        const uint16_t *row = s_dist + i * nb;
        // branch-free, so that the LDS reads of several candidates are in flight together (a taken candidate -- by a feature
        // before this one, or not a candidate at all -- contributes the neutral key)
        if (nleftB < 0) {
#pragma unroll 4
          for (int j = 0; j < nb; j++) {
            const uint32_t key = s_by[j] < i ? KEY_NONE : ((uint32_t)row[j] << 20) | (uint32_t)j;
            k2 = umed3(k1, k2, key), k1 = min(k1, key);
          }
        } else {
#pragma unroll 4
          for (int j = 0; j < nb; j++) {
            const uint32_t key = s_by[j] < i ? KEY_NONE : ((uint32_t)row[j] << 20) | (uint32_t)j;
            const bool isleft = s_rb[j] < nleftB;
            const uint32_t kl = isleft ? key : KEY_NONE, kr = isleft ? KEY_NONE : key;
            k2 = umed3(k1, k2, kl), k1 = min(k1, kl);
            r1 = min(r1, kr);
          }
        }
        const int bestDist1 = (int)(k1 >> 20), bestDist2 = (int)(k2 >> 20);
        // right block, nested in the left test, no ratio test (:362-389)
        if (nleftB >= 0 && bestDist1 <= TH_LOW && (int)(r1 >> 20) <= TH_LOW) newR = (int)(r1 & 0xFFFFF);
        const bool pass = mode == 0 ? bestDist1 <= TH_LOW : bestDist1 < TH_LOW;
        if (pass && (float)bestDist1 < nnratio * (float)bestDist2) newL = (int)(k1 & 0xFFFFF);  // :335-337 / :843-845
      }
      const bool changed = newL != decL || newR != decR;
      decL = newL, decR = newR;
      __syncthreads();  // every lane has read the claims of the previous round (and that round's flag)
      if (tid < nb) s_by[tid] = s_by0[tid];
      if (tid == 0) s_changed[(round + 1) & 1] = 0;
      __syncthreads();
      if (newL >= 0) atomicMin(&s_by[newL], i);
      if (newR >= 0) atomicMin(&s_by[newR], i);
      if (changed) s_changed[round & 1] = 1;
      __syncthreads();
      if (!s_changed[round & 1]) break;
    }
    if (active) {
      if (mode == 0) {
        if (decL >= 0) matchB[s_rb[decL]] = ra;  // vpMapPointMatches[bestIdxF] = pMP
        if (decR >= 0) matchB[s_rb[decR]] = ra;
      } else if (decL >= 0) {
        matchA[ra] = s_rb[decL];                 // vpMatches12[idx1] = vpMapPoints2[bestIdx2]
      }
    }
    return;
  }
  if (tid >= 64) return;
  // ---- large nodes: one scan of the node's Frame features per KeyFrame feature; the claims in device scratch, set up here
  for (int j = lane; j < nb; j += 64) {
    const int rb = idxB[np.b_begin + j];
    claimB[rb] = mode == 1 ? (int)(!validB[rb]) : 0;
  }
  __threadfence_block();
  for (int ia = np.a_begin; ia < np.a_end; ia++) {
    const int ra = idxA[ia];
    if (!validA[ra]) continue;  // !pMP || pMP->isBad()
    uint4 a0, a1;
    load_desc(descA, ra, a0, a1);
    uint32_t k1 = KEY_NONE, k2 = KEY_NONE, r1 = KEY_NONE, r2 = KEY_NONE;  // left / right-camera (fisheye) candidates
    for (int j = lane; j < nb; j += 64) {
      const int rb = idxB[np.b_begin + j];
      if (__hip_atomic_load(&claimB[rb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) continue;
      uint4 b0, b1;
      load_desc(descB, rb, b0, b1);
      const uint32_t key = ((uint32_t)hamming256(a0, a1, b0, b1) << 20) | (uint32_t)j;
      if (nleftB < 0 || rb < nleftB)
        merge2(k1, k2, key, KEY_NONE);
      else
        merge2(r1, r2, key, KEY_NONE);
    }
    wave_best2(k1, k2);
    const int bestDist1 = (int)(k1 >> 20), bestDist2 = (int)(k2 >> 20);
    if (nleftB >= 0 && bestDist1 <= TH_LOW) {  // right block, nested in the left test, no ratio test (:362-389)
      r1 = wave_min(r1);
      if ((int)(r1 >> 20) <= TH_LOW) {
        const int rb = idxB[np.b_begin + (int)(r1 & 0xFFFFF)];
        if (lane == 0) matchB[rb] = ra, claimB[rb] = 1;
        __threadfence_block();  // later KF features of this node must see the claim
      }
    }
    const bool pass = mode == 0 ? bestDist1 <= TH_LOW : bestDist1 < TH_LOW;
    if (pass && (float)bestDist1 < nnratio * (float)bestDist2) {  // :335-337 / :843-845
      const int rb = idxB[np.b_begin + (int)(k1 & 0xFFFFF)];
      if (lane == 0) {
        if (mode == 0)
          matchB[rb] = ra;  // vpMapPointMatches[bestIdxF] = pMP
        else
          matchA[ra] = rb;  // vpMatches12[idx1] = vpMapPoints2[bestIdx2]
        claimB[rb] = 1;     // (mode 1: vbMatched2[bestIdx2] = true)
      }
      __threadfence_block();
    }
  }
}

// ---- SearchForTriangulation (ORBmatcher.cc:902-1146): one wavefront per shared vocabulary node.  For every eligible
// KF1 feature of the node: best KF2 feature of the node with dist <= TH_LOW whose (idx1, idx2) pair passes the
// caller's geometric predicate bit; the reference's `dist > bestDist -> continue` lets a LATER equal distance win, so
// keys are (dist << 20 | 0xFFFFF - position) and the wave takes their minimum.  No greedy state (vbMatched2 is never
// set by the reference), so nodes and features are independent.
__global__ __launch_bounds__(256) void k_search_triangulation(const NodePair *pairs, int npairs, const uint8_t *descA,
                                                              const uint8_t *eligA, const int *idxA,
                                                              const uint8_t *descB, const uint8_t *eligB,
                                                              const int *idxB, const uint32_t *pairOk,
                                                              const int *pairOff, int *matches12) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= npairs) return;
  const NodePair np = pairs[wave];
  const int nb = np.b_end - np.b_begin;
  const long long bit0 = pairOk ? (long long)pairOff[wave] : 0;
  for (int ia = np.a_begin; ia < np.a_end; ia++) {
    const int ra = idxA[ia];
    if (!eligA[ra]) continue;
    uint4 a0, a1;
    load_desc(descA, ra, a0, a1);
    uint32_t k = KEY_NONE;
    for (int j = lane; j < nb; j += 64) {
      const int rb = idxB[np.b_begin + j];
      if (!eligB[rb]) continue;
      if (pairOk) {
        const long long bit = bit0 + (long long)(ia - np.a_begin) * nb + j;
        if (!((pairOk[bit >> 5] >> (bit & 31)) & 1u)) continue;
      }
      uint4 b0, b1;
      load_desc(descB, rb, b0, b1);
      const int dist = hamming256(a0, a1, b0, b1);
      if (dist > TH_LOW) continue;
      k = min(k, ((uint32_t)dist << 20) | (0xFFFFFu - (uint32_t)j));
    }
    k = wave_min(k);
    if (lane == 0 && k != KEY_NONE) matches12[ra] = idxB[np.b_begin + (int)(0xFFFFFu - (k & 0xFFFFFu))];
  }
}

// ---- windowed searches on HOST candidate lists (the caller ran GetFeaturesInArea itself): every candidate entry's
// Hamming distance in ONE data-parallel pass, written as the packed entry the ordered host passes read
// (vsg_walks.h: index | distance << 15 | octave << 24).  The walk over the queries that follows is inherently
// sequential (a claimed feature blocks later queries) and touches a dozen candidates per query: it runs on the host
// over these entries, like the reference's own loop -- the first version walked the queries with a single wavefront
// on the device and took 1.3 ms for 1000 queries, 25x the CPU.  (Resident frames: k_window_search, vsg_frame.hip.)
__global__ __launch_bounds__(256) void k_cand_dist(const uint8_t *qDesc, const int *qOf, const int *candIdx,
                                                   const uint8_t *tDesc, const uint8_t *tOct, int ncand,
                                                   uint32_t *ent) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncand) return;
  uint4 a0, a1, b0, b1;
  const int i = candIdx[c];
  load_desc(qDesc, qOf[c], a0, a1);
  load_desc(tDesc, i, b0, b1);
  ent[c] = (uint32_t)i | ((uint32_t)hamming256(a0, a1, b0, b1) << 15) | ((uint32_t)(tOct ? tOct[i] & 15 : 0) << 24);
}

// ---- MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:380-415): one workgroup per map point.  The N x N
// distance matrix lives in LDS; thread i finds the element of rank floor(0.5*(N-1)) of row i by counting (no
// sort needed: the median VALUE does not depend on how ties are ordered); the first row with the least median
// wins (strict '<').
enum { kDistinctMaxN = 128 };
__global__ __launch_bounds__(128) void k_distinctive(const uint8_t *desc, const int *off, int *best) {
  __shared__ uint16_t D[kDistinctMaxN * kDistinctMaxN];
  __shared__ uint32_t s_key[kDistinctMaxN];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int o = off[g], N = off[g + 1] - o;
  if (N <= 0) {
    if (tid == 0) best[g] = -1;
    return;
  }
  for (int i = tid; i < N; i += 128) {
    uint4 a0, a1;
    load_desc(desc, o + i, a0, a1);
    for (int j = 0; j < N; j++) {
      uint4 b0, b1;
      load_desc(desc, o + j, b0, b1);
      D[i * N + j] = (uint16_t)hamming256(a0, a1, b0, b1);
    }
  }
  __syncthreads();
  const int k = (int)(0.5 * (N - 1));
  for (int i = tid; i < N; i += 128) {
    int median = 0;
    for (int j = 0; j < N; j++) {
      const int v = D[i * N + j];
      int less = 0, le = 0;
      for (int t = 0; t < N; t++) {
        const int x = D[i * N + t];
        less += x < v;
        le += x <= v;
      }
      if (less <= k && k < le) median = v;
    }
    s_key[i] = ((uint32_t)median << 16) | (uint32_t)i;  // smallest key = least median, first index
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t b = 0xFFFFFFFFu;
    for (int i = 0; i < N; i++) b = min(b, s_key[i]);
    best[g] = (int)(b & 0xFFFF);
  }
}

// ------------------------------------------------------------------------------------------------ host side
// Every entry point runs on the calling thread's stream and stages through its arenas (vsg_ctx.h): inputs are laid
// out in the pinned arena and go up in ONE DMA into the device arena (they are read many times by the kernels),
// results come back through the pinned arena.  No allocation, no NULL-stream launch in steady state.

#define M_TRY(expr)                      \
  do {                                   \
    hipError_t _e = (expr);              \
    if (_e != hipSuccess) return VSG_ERR_HIP; \
  } while (0)

using vsg::ThreadCtx;
using vsg::Stage;
namespace walk = vsg::walk;

int use_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return VSG_ERR_NO_DEVICE;
  return hipSetDevice(device) == hipSuccess ? VSG_OK : VSG_ERR_NO_DEVICE;
}

// merge-join of two FeatureVectors (ORBmatcher.cc:247-405 loop skeleton incl. lower_bound jumps)
void join_nodes(const int *idA, const int *offA, int nA, const int *idB, const int *offB, int nB,
                std::vector<NodePair> &out) {
  int i = 0, j = 0;
  while (i != nA && j != nB) {
    if (idA[i] == idB[j]) {
      out.push_back({offA[i], offA[i + 1], offB[j], offB[j + 1]});
      i++, j++;
    } else if (idA[i] < idB[j]) {
      i = (int)(std::lower_bound(idA, idA + nA, idB[j]) - idA);
    } else {
      j = (int)(std::lower_bound(idB, idB + nB, idA[i]) - idB);
    }
  }
}

// Packed candidate entries (vsg_walks.h) for host-provided CSR candidate lists.  On return *ent points into the
// calling thread's pinned arena (valid until its next call).
static int candidate_entries(int device, const uint8_t *q_desc, int n_q, const int32_t *cand_off,
                             const int32_t *cand_idx, const uint8_t *t_desc, const int32_t *t_octave, int n_t,
                             const uint32_t **ent) {
  int rc = VSG_OK;
  ThreadCtx *c = vsg::thread_ctx(device, &rc);
  if (!c) return rc;
  const int ncand = cand_off[n_q];
  *ent = nullptr;
  if (ncand == 0) return VSG_OK;
  if (n_t > 32767) return VSG_ERR_UNSUPPORTED;
  for (int k = 0; k < ncand; k++)
    if (cand_idx[k] < 0 || cand_idx[k] >= n_t) return VSG_ERR_INVALID;
  Stage st;
  const size_t oQ = st.add((size_t)n_q * 32), oT = st.add((size_t)n_t * 32), oOct = st.add(t_octave ? (size_t)n_t : 0),
               oOf = st.add((size_t)ncand * 4), oIdx = st.add((size_t)ncand * 4);
  const size_t in_bytes = st.total;
  const size_t oEnt = st.add((size_t)ncand * 4);
  rc = vsg::ctx_reserve(c, st.total, in_bytes);
  if (rc != VSG_OK) return rc;
  uint8_t *h = c->h_pin;
  memcpy(h + oQ, q_desc, (size_t)n_q * 32);
  memcpy(h + oT, t_desc, (size_t)n_t * 32);
  if (t_octave)
    for (int i = 0; i < n_t; i++) h[oOct + i] = (uint8_t)t_octave[i];
  int32_t *qof = (int32_t *)(h + oOf);
  for (int q = 0; q < n_q; q++)
    for (int k = cand_off[q]; k < cand_off[q + 1]; k++) qof[k] = q;
  memcpy(h + oIdx, cand_idx, (size_t)ncand * 4);
  M_TRY(hipMemcpyAsync(c->d_buf, h, in_bytes, hipMemcpyHostToDevice, c->stream));
  const uint8_t *d = c->d_buf;
  hipLaunchKernelGGL(k_cand_dist, dim3((ncand + 255) / 256), dim3(256), 0, c->stream, d + oQ, (const int *)(d + oOf),
                     (const int *)(d + oIdx), d + oT, t_octave ? d + oOct : (const uint8_t *)nullptr, ncand,
                     (uint32_t *)(c->d_pin + oEnt));
  M_TRY(hipGetLastError());
  M_TRY(hipStreamSynchronize(c->stream));
  *ent = (const uint32_t *)(h + oEnt);
  return VSG_OK;
}

static walk::CandView csr_view(const uint32_t *ent, const int32_t *cand_off) {
  walk::CandView cv;
  cv.ent = ent;
  cv.off = cand_off;
  return cv;
}

// rotation-consistency filter of the BoW / triangulation searches: every match sits in exactly one bin
template <class AngleA, class AngleB>
static int bow_rotation_filter(int *out, int nOut, int mode, AngleA angleA, AngleB angleB, bool checkOri) {
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int i = 0; i < nOut; i++) {
    if (out[i] < 0) continue;
    nmatches++;
    if (checkOri) {
      const int bin = mode == 0 ? walk::rot_bin(angleA(out[i]), angleB(i)) : walk::rot_bin(angleA(i), angleB(out[i]));
      rotHist[bin].push_back(i);
    }
  }
  if (checkOri)
    walk::filter_rotation(rotHist, [&](int i) {
      out[i] = -1;
      nmatches--;
    });
  return nmatches;
}

// SearchByBoW on descriptors that are either host arrays (descA/descB, staged) or already on the device (dDescA/dDescB)
template <class AngleA, class AngleB>
static int search_by_bow(int device, int mode, int nleftB, const uint8_t *descA, const uint8_t *dDescA, AngleA angleA,
                         const uint8_t *validA, int nA, const int *idA, const int *offA, const int *idxA, int nodesA,
                         const uint8_t *descB, const uint8_t *dDescB, AngleB angleB, const uint8_t *validB, int nB,
                         const int *idB, const int *offB, const int *idxB, int nodesB, float nnratio, int checkOri,
                         int *out) {
  int rc = VSG_OK;
  ThreadCtx *c = vsg::thread_ctx(device, &rc);
  if (!c) return rc;
  const int nOut = mode == 0 ? nB : nA;
  for (int i = 0; i < nOut; i++) out[i] = -1;
  std::vector<NodePair> pairs;
  join_nodes(idA, offA, nodesA, idB, offB, nodesB, pairs);
  if (pairs.empty() || nA == 0 || nB == 0) return 0;
  const int nIdxA = offA[nodesA], nIdxB = offB[nodesB], npairs = (int)pairs.size();
  Stage st;
  const size_t oP = st.add(pairs.size() * sizeof(NodePair)), oVA = st.add((size_t)nA), oIA = st.add((size_t)nIdxA * 4),
               oVB = st.add(mode == 1 ? (size_t)nB : 0), oIB = st.add((size_t)nIdxB * 4),
               oDA = st.add(dDescA ? 0 : (size_t)nA * 32), oDB = st.add(dDescB ? 0 : (size_t)nB * 32);
  const size_t in_bytes = st.total;
  Stage dv;  // device-only scratch behind the inputs
  dv.total = in_bytes;
  const size_t nAB = (size_t)(nA > nB ? nA : nB);
  const size_t oMA = dv.add(nAB * 4), oMB = dv.add(nAB * 4);
  rc = vsg::ctx_reserve(c, in_bytes + (size_t)nOut * 4 + 64, dv.total);
  if (rc != VSG_OK) return rc;
  uint8_t *h = c->h_pin;
  memcpy(h + oP, pairs.data(), pairs.size() * sizeof(NodePair));
  memcpy(h + oVA, validA, (size_t)nA);
  memcpy(h + oIA, idxA, (size_t)nIdxA * 4);
  if (mode == 1) memcpy(h + oVB, validB, (size_t)nB);
  memcpy(h + oIB, idxB, (size_t)nIdxB * 4);
  if (!dDescA) memcpy(h + oDA, descA, (size_t)nA * 32);
  if (!dDescB) memcpy(h + oDB, descB, (size_t)nB * 32);
  M_TRY(hipMemcpyAsync(c->d_buf, h, in_bytes, hipMemcpyHostToDevice, c->stream));
  uint8_t *d = c->d_buf;
  int *dMA = (int *)(d + oMA), *dMB = (int *)(d + oMB);  // mode 0: dMB = match_f, dMA = the claim scratch; mode 1: the reverse
  M_TRY(hipMemsetAsync(mode == 0 ? dMB : dMA, 0xFF, (size_t)(mode == 0 ? nB : nA) * 4, c->stream));  // -1
  hipLaunchKernelGGL(k_search_by_bow, dim3(npairs), dim3(256), 0, c->stream, (const NodePair *)(d + oP), npairs,
                     FvDev{}, FvDev{}, dDescA ? dDescA : d + oDA, d + oVA, (const int *)(d + oIA), dDescB ? dDescB : d + oDB, d + oVB,
                     (const int *)(d + oIB), nnratio, mode, nleftB, dMA, mode == 0 ? dMB : (int *)nullptr,
                     mode == 0 ? dMA : dMB);
  M_TRY(hipGetLastError());
  int *hOut = (int *)(h + in_bytes);
  M_TRY(hipMemcpyAsync(hOut, mode == 0 ? dMB : dMA, (size_t)nOut * 4, hipMemcpyDeviceToHost, c->stream));
  M_TRY(hipStreamSynchronize(c->stream));
  memcpy(out, hOut, (size_t)nOut * 4);
  // rotation consistency (:407-425 / :879-897)
  return bow_rotation_filter(out, nOut, mode, angleA, angleB, checkOri != 0);
}

// ---- SearchByBoW on two frames whose descriptors AND FeatureVectors are resident (round 6): nothing but the KeyFrame's
// "has a map point" flags goes up, the join runs in the kernel, and the call comes in two halves so that it can share one
// wait with the ComputeStereoMatches / ComputeBoW of the same Frame (vsg_chain.hip).
}  // namespace (reopened below: the two halves are called from vsg_chain.hip)
namespace vsg {

void bow_search_sizes(int nA, int nB, int mode, size_t *pin_bytes, size_t *dev_bytes) {
  Stage p, d;
  p.add((size_t)nA), p.add(mode == 1 ? (size_t)nB : 0), p.add(4 * (size_t)(mode == 0 ? nB : nA) + 64);
  d.add(4 * (size_t)nB + 64);
  *pin_bytes = p.total, *dev_bytes = d.total;
}

// ONE launch and nothing else on the stream: the flags are read and the matches written where they lie in the pinned arena
// (a few hundred scattered bytes each way over PCIe, in parallel over the node waves) -- the DMA of the flags, the fill of the
// match array and its copy back cost 4.4 us of stream time EACH (rocprofv3: __amd_rocclr_copyBuffer / fillBufferAligned).
int bow_search_enqueue(BowSearchCall *s, int mode, vsg_frame *A, const uint8_t *validA, vsg_frame *B, const uint8_t *validB,
                       float nnratio, ThreadCtx *c, size_t pin_base, size_t dev_base) {
  s->c = c, s->mode = mode, s->A = A, s->B = B, s->active = false, s->pin_base = pin_base;
  s->nOut = mode == 0 ? B->n : A->n;
  if (!A->fv_valid || !B->fv_valid) return VSG_ERR_INVALID;  // ComputeBoW first (Frame.cc:882-889)
  if (A->n == 0 || B->n == 0) return VSG_OK;
  const int nA = A->n, nB = B->n;
  Stage p;
  const size_t oVA = p.add((size_t)nA), oVB = p.add(mode == 1 ? (size_t)nB : 0);
  s->oOut = p.add(4 * (size_t)s->nOut + 64);
  uint8_t *hp = c->h_pin + pin_base, *dp = c->d_pin + pin_base;
  memcpy(hp + oVA, validA, (size_t)nA);
  if (mode == 1) memcpy(hp + oVB, validB, (size_t)nB);
  memset(hp + s->oOut, 0xFF, 4 * (size_t)s->nOut);  // -1: the kernel only writes matches
  const FvDev fa{A->d_fv_hdr, A->d_fv_node, A->d_fv_off, A->d_fv_idx}, fb{B->d_fv_hdr, B->d_fv_node, B->d_fv_off, B->d_fv_idx};
  const int waves = A->fv_bound > 0 ? A->fv_bound : 1;
  int *out = (int *)(dp + s->oOut);
  hipLaunchKernelGGL(k_search_by_bow, dim3(waves), dim3(256), 0, c->stream, (const NodePair *)nullptr, 0, fa, fb,
                     A->d_desc, dp + oVA, (const int *)nullptr, B->d_desc, dp + oVB, (const int *)nullptr, nnratio, mode,
                     mode == 0 ? B->nleft : -1, mode == 1 ? out : (int *)nullptr, mode == 0 ? out : (int *)nullptr,
                     (int *)(c->d_buf + dev_base));
  M_TRY(hipGetLastError());
  s->active = true;
  return VSG_OK;
}

// after the stream has been waited for: the matches + the rotation-consistency filter (:407-425 / :879-897)
int bow_search_finish(BowSearchCall *s, int check_orientation, int32_t *out) {
  for (int i = 0; i < s->nOut; i++) out[i] = -1;
  if (!s->active) return 0;
  memcpy(out, s->c->h_pin + s->pin_base + s->oOut, 4 * (size_t)s->nOut);
  const vsg_keypoint *ka = s->A->h_kps.data(), *kb = s->B->h_kps.data();
  return bow_rotation_filter(out, s->nOut, s->mode, [&](int i) { return ka[i].angle; }, [&](int i) { return kb[i].angle; },
                             check_orientation != 0);
}

}  // namespace vsg
namespace {

static int search_by_bow_resident(int mode, vsg_frame *A, const uint8_t *validA, vsg_frame *B, const uint8_t *validB,
                                  float nnratio, int check_orientation, int32_t *out) {
  int rc = VSG_OK;
  ThreadCtx *c = vsg::thread_ctx(A->device, &rc);
  if (!c) return rc;
  size_t pin = 0, dev = 0;
  vsg::bow_search_sizes(A->n, B->n, mode, &pin, &dev);
  rc = vsg::ctx_reserve(c, pin, dev);
  if (rc != VSG_OK) return rc;
  vsg::BowSearchCall s;
  rc = vsg::bow_search_enqueue(&s, mode, A, validA, B, validB, nnratio, c, 0, 0);
  if (rc != VSG_OK) return rc;
  if (s.active) M_TRY(hipStreamSynchronize(c->stream));
  return vsg::bow_search_finish(&s, check_orientation, out);
}

// SearchForTriangulation on descriptors that are host arrays (desc1/desc2, staged) or already resident (dDesc1/dDesc2)
template <class Angle1, class Angle2>
static int search_triangulation(int device, const uint8_t *desc1, const uint8_t *dDesc1, Angle1 angle1,
                                const uint8_t *eligible1, int n1, const int32_t *node_id1, const int32_t *off1,
                                const int32_t *idx1, int nodes1, const uint8_t *desc2, const uint8_t *dDesc2,
                                Angle2 angle2, const uint8_t *eligible2, int n2, const int32_t *node_id2,
                                const int32_t *off2, const int32_t *idx2, int nodes2, const uint32_t *pair_ok,
                                const int32_t *pair_off, int check_orientation, int32_t *matches12) {
  int rc = VSG_OK;
  ThreadCtx *c = vsg::thread_ctx(device, &rc);
  if (!c) return rc;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  std::vector<NodePair> pairs;
  join_nodes(node_id1, off1, nodes1, node_id2, off2, nodes2, pairs);
  if (pairs.empty() || n1 == 0 || n2 == 0) return 0;
  const int npairs = (int)pairs.size();
  const int nI1 = off1[nodes1], nI2 = off2[nodes2];
  // bits of shared node s start at pair_off[s]; the last node ends at pair_off[npairs]
  const size_t ok_words = pair_ok ? (size_t)(((long long)pair_off[npairs] + 31) / 32 + 1) : 0;
  Stage st;
  const size_t oP = st.add(pairs.size() * sizeof(NodePair)), oD1 = st.add(dDesc1 ? 0 : (size_t)n1 * 32),
               oE1 = st.add((size_t)n1), oI1 = st.add((size_t)nI1 * 4), oD2 = st.add(dDesc2 ? 0 : (size_t)n2 * 32),
               oE2 = st.add((size_t)n2), oI2 = st.add((size_t)nI2 * 4), oOk = st.add(ok_words * 4),
               oOff = st.add(pair_ok ? (size_t)(npairs + 1) * 4 : 0);
  const size_t in_bytes = st.total;
  const size_t oM = st.add((size_t)n1 * 4);
  rc = vsg::ctx_reserve(c, st.total, st.total);
  if (rc != VSG_OK) return rc;
  uint8_t *h = c->h_pin;
  memcpy(h + oP, pairs.data(), pairs.size() * sizeof(NodePair));
  if (!dDesc1) memcpy(h + oD1, desc1, (size_t)n1 * 32);
  memcpy(h + oE1, eligible1, (size_t)n1);
  memcpy(h + oI1, idx1, (size_t)nI1 * 4);
  if (!dDesc2) memcpy(h + oD2, desc2, (size_t)n2 * 32);
  memcpy(h + oE2, eligible2, (size_t)n2);
  memcpy(h + oI2, idx2, (size_t)nI2 * 4);
  if (pair_ok) {
    memcpy(h + oOk, pair_ok, ok_words * 4);
    memcpy(h + oOff, pair_off, (size_t)(npairs + 1) * 4);
  }
  M_TRY(hipMemcpyAsync(c->d_buf, h, in_bytes, hipMemcpyHostToDevice, c->stream));
  uint8_t *d = c->d_buf;
  M_TRY(hipMemsetAsync(d + oM, 0xFF, (size_t)n1 * 4, c->stream));
  hipLaunchKernelGGL(k_search_triangulation, dim3((npairs + 3) / 4), dim3(256), 0, c->stream, (const NodePair *)(d + oP),
                     npairs, dDesc1 ? dDesc1 : d + oD1, d + oE1, (const int *)(d + oI1), dDesc2 ? dDesc2 : d + oD2, d + oE2,
                     (const int *)(d + oI2), pair_ok ? (const uint32_t *)(d + oOk) : (const uint32_t *)nullptr,
                     pair_ok ? (const int *)(d + oOff) : (const int *)nullptr, (int *)(d + oM));
  M_TRY(hipGetLastError());
  M_TRY(hipMemcpyAsync(h + oM, d + oM, (size_t)n1 * 4, hipMemcpyDeviceToHost, c->stream));
  M_TRY(hipStreamSynchronize(c->stream));
  memcpy(matches12, h + oM, (size_t)n1 * 4);
  return bow_rotation_filter(matches12, n1, 1, angle1, angle2, check_orientation != 0);
}

}  // namespace

extern "C" {

int vsg_hamming_pairs(int device, const uint8_t *a, int na, const uint8_t *b, int nb, const int32_t *ia,
                      const int32_t *ib, int npairs, int32_t *dist) {
  if (!a || !b || !ia || !ib || !dist || npairs < 0 || na < 0 || nb < 0) return VSG_ERR_INVALID;
  int rc = VSG_OK;
  ThreadCtx *c = vsg::thread_ctx(device, &rc);
  if (!c) return rc;
  if (npairs == 0) return VSG_OK;
  Stage st;
  const size_t oA = st.add((size_t)na * 32), oB = st.add((size_t)nb * 32), oIA = st.add((size_t)npairs * 4),
               oIB = st.add((size_t)npairs * 4);
  const size_t in_bytes = st.total;
  const size_t oD = st.add((size_t)npairs * 4);
  rc = vsg::ctx_reserve(c, st.total, in_bytes);
  if (rc != VSG_OK) return rc;
  uint8_t *h = c->h_pin;
  memcpy(h + oA, a, (size_t)na * 32);
  memcpy(h + oB, b, (size_t)nb * 32);
  memcpy(h + oIA, ia, (size_t)npairs * 4);
  memcpy(h + oIB, ib, (size_t)npairs * 4);
  M_TRY(hipMemcpyAsync(c->d_buf, h, in_bytes, hipMemcpyHostToDevice, c->stream));
  const uint8_t *d = c->d_buf;
  hipLaunchKernelGGL(k_hamming_pairs, dim3((npairs + 255) / 256), dim3(256), 0, c->stream, d + oA, d + oB,
                     (const int *)(d + oIA), (const int *)(d + oIB), npairs, (int *)(c->d_pin + oD));
  M_TRY(hipGetLastError());
  M_TRY(hipStreamSynchronize(c->stream));
  memcpy(dist, h + oD, (size_t)npairs * 4);
  return VSG_OK;
}

int vsg_hamming_block_best2_device(int device, const uint8_t *d_a, const uint8_t *d_b, size_t block_stride_bytes,
                                   const int32_t *d_counts_a, const int32_t *d_counts_b, int count_stride,
                                   int nblocks, int max_rows, int32_t *d_best, int32_t *d_second,
                                   int32_t *d_argbest, void *stream) {
  if (!d_a || !d_b || !d_best || !d_second || !d_argbest || nblocks < 1 || max_rows < 1) return VSG_ERR_INVALID;
  int rc = use_device(device);
  if (rc != VSG_OK) return rc;
  dim3 grid((max_rows + kBest2Rows - 1) / kBest2Rows, nblocks);
  // stream == NULL is the caller's NULL stream (see vsg_orb_extract_batch_device for how the extractor orders itself
  // against it)
  hipLaunchKernelGGL(k_block_best2_mfma, grid, dim3(256), 0, (hipStream_t)stream, d_a, d_b, block_stride_bytes,
                     d_counts_a, d_counts_b, count_stride, max_rows, max_rows, max_rows, d_best, d_second, d_argbest);
  M_TRY(hipGetLastError());
  return VSG_OK;
}

int vsg_hamming_block_best2(int device, const uint8_t *a, int na, const uint8_t *b, int nb, int32_t *best,
                            int32_t *second, int32_t *argbest) {
  if (!a || !b || !best || !second || !argbest || na < 0 || nb < 0) return VSG_ERR_INVALID;
  int rc = VSG_OK;
  ThreadCtx *c = vsg::thread_ctx(device, &rc);
  if (!c) return rc;
  if (na == 0) return VSG_OK;
  Stage st;
  const size_t oA = st.add((size_t)na * 32), oB = st.add((size_t)(nb > 0 ? nb : 1) * 32);
  const size_t in_bytes = st.total;
  const size_t o1 = st.add((size_t)na * 4), o2 = st.add((size_t)na * 4), o3 = st.add((size_t)na * 4);
  rc = vsg::ctx_reserve(c, st.total, in_bytes);
  if (rc != VSG_OK) return rc;
  uint8_t *h = c->h_pin;
  memcpy(h + oA, a, (size_t)na * 32);
  if (nb) memcpy(h + oB, b, (size_t)nb * 32);
  M_TRY(hipMemcpyAsync(c->d_buf, h, in_bytes, hipMemcpyHostToDevice, c->stream));
  const uint8_t *d = c->d_buf;
  int *d1 = (int *)(c->d_pin + o1), *d2 = (int *)(c->d_pin + o2), *d3 = (int *)(c->d_pin + o3);
  const dim3 grid((na + kBest2Rows - 1) / kBest2Rows, 1);
  hipLaunchKernelGGL(k_block_best2_mfma, grid, dim3(256), 0, c->stream, d + oA, d + oB, (size_t)0,
                     (const int *)nullptr, (const int *)nullptr, 0, na, nb, na, d1, d2, d3);
  M_TRY(hipGetLastError());
  M_TRY(hipStreamSynchronize(c->stream));
  memcpy(best, h + o1, (size_t)na * 4);
  memcpy(second, h + o2, (size_t)na * 4);
  memcpy(argbest, h + o3, (size_t)na * 4);
  return VSG_OK;
}

int vsg_search_for_triangulation(int device, const uint8_t *desc1, const float *angle1, const uint8_t *eligible1,
                                 int n1, const int32_t *node_id1, const int32_t *off1, const int32_t *idx1, int nodes1,
                                 const uint8_t *desc2, const float *angle2, const uint8_t *eligible2, int n2,
                                 const int32_t *node_id2, const int32_t *off2, const int32_t *idx2, int nodes2,
                                 const uint32_t *pair_ok, const int32_t *pair_off, int check_orientation,
                                 int32_t *matches12) {
  if (!matches12 || n1 < 0 || n2 < 0 || (pair_ok && !pair_off)) return VSG_ERR_INVALID;
  return search_triangulation(device, desc1, nullptr, [&](int i) { return angle1[i]; }, eligible1, n1, node_id1, off1, idx1,
                              nodes1, desc2, nullptr, [&](int i) { return angle2[i]; }, eligible2, n2, node_id2, off2, idx2,
                              nodes2, pair_ok, pair_off, check_orientation, matches12);
}

// both KeyFrames resident (vsg_frame): only the FeatureVectors, the eligibility flags and the predicate bits go up
int vsg_frame_search_for_triangulation(vsg_frame *kf1, const uint8_t *eligible1, const int32_t *node_id1,
                                       const int32_t *off1, const int32_t *idx1, int nodes1, vsg_frame *kf2,
                                       const uint8_t *eligible2, const int32_t *node_id2, const int32_t *off2,
                                       const int32_t *idx2, int nodes2, const uint32_t *pair_ok, const int32_t *pair_off,
                                       int check_orientation, int32_t *matches12) {
  if (!kf1 || !kf2 || !matches12 || kf1->device != kf2->device || !eligible1 || !eligible2 || (pair_ok && !pair_off))
    return VSG_ERR_INVALID;
  const vsg_keypoint *ka = kf1->h_kps.data(), *kb = kf2->h_kps.data();
  return search_triangulation(kf1->device, nullptr, kf1->d_desc, [&](int i) { return ka[i].angle; }, eligible1, kf1->n,
                              node_id1, off1, idx1, nodes1, nullptr, kf2->d_desc, [&](int i) { return kb[i].angle; },
                              eligible2, kf2->n, node_id2, off2, idx2, nodes2, pair_ok, pair_off, check_orientation,
                              matches12);
}

int vsg_search_by_bow_kf_f(int device, const uint8_t *kf_desc, const float *kf_angle, const uint8_t *kf_valid,
                           int n_kf, const int32_t *kf_node_id, const int32_t *kf_off, const int32_t *kf_idx,
                           int kf_nodes, const uint8_t *f_desc, const float *f_angle, int n_f,
                           const int32_t *f_node_id, const int32_t *f_off, const int32_t *f_idx, int f_nodes,
                           float nnratio, int check_orientation, int32_t *match_f) {
  return vsg_search_by_bow_kf_f_stereo(device, kf_desc, kf_angle, kf_valid, n_kf, kf_node_id, kf_off, kf_idx, kf_nodes,
                                       f_desc, f_angle, n_f, -1, f_node_id, f_off, f_idx, f_nodes, nnratio,
                                       check_orientation, match_f);
}

int vsg_search_by_bow_kf_f_stereo(int device, const uint8_t *kf_desc, const float *kf_angle, const uint8_t *kf_valid,
                                  int n_kf, const int32_t *kf_node_id, const int32_t *kf_off, const int32_t *kf_idx,
                                  int kf_nodes, const uint8_t *f_desc, const float *f_angle, int n_f, int f_nleft,
                                  const int32_t *f_node_id, const int32_t *f_off, const int32_t *f_idx, int f_nodes,
                                  float nnratio, int check_orientation, int32_t *match_f) {
  if (!match_f || n_kf < 0 || n_f < 0 || f_nleft < -1 || f_nleft > n_f) return VSG_ERR_INVALID;
  return search_by_bow(device, 0, f_nleft, kf_desc, nullptr, [&](int i) { return kf_angle[i]; }, kf_valid, n_kf,
                       kf_node_id, kf_off, kf_idx, kf_nodes, f_desc, nullptr, [&](int i) { return f_angle[i]; }, nullptr,
                       n_f, f_node_id, f_off, f_idx, f_nodes, nnratio, check_orientation, match_f);
}

int vsg_search_by_bow_kf_kf(int device, const uint8_t *desc1, const float *angle1, const uint8_t *valid1, int n1,
                            const int32_t *node_id1, const int32_t *off1, const int32_t *idx1, int nodes1,
                            const uint8_t *desc2, const float *angle2, const uint8_t *valid2, int n2,
                            const int32_t *node_id2, const int32_t *off2, const int32_t *idx2, int nodes2,
                            float nnratio, int check_orientation, int32_t *matches12) {
  if (!matches12 || n1 < 0 || n2 < 0) return VSG_ERR_INVALID;
  return search_by_bow(device, 1, -1, desc1, nullptr, [&](int i) { return angle1[i]; }, valid1, n1, node_id1, off1, idx1,
                       nodes1, desc2, nullptr, [&](int i) { return angle2[i]; }, valid2, n2, node_id2, off2, idx2, nodes2,
                       nnratio, check_orientation, matches12);
}

// the same two searches with both descriptor sets resident (vsg_frame): only FeatureVectors + flags go up
int vsg_frame_search_by_bow_kf_f(vsg_frame *kf, const uint8_t *kf_valid, const int32_t *kf_node_id,
                                 const int32_t *kf_off, const int32_t *kf_idx, int kf_nodes, vsg_frame *f,
                                 const int32_t *f_node_id, const int32_t *f_off, const int32_t *f_idx, int f_nodes,
                                 float nnratio, int check_orientation, int32_t *match_f) {
  if (!kf || !f || !match_f || kf->device != f->device || !kf_valid) return VSG_ERR_INVALID;
  if (!kf_node_id && !f_node_id)  // both FeatureVectors resident (ComputeBoW ran on both frames): nothing goes up but the flags
    return search_by_bow_resident(0, kf, kf_valid, f, nullptr, nnratio, check_orientation, match_f);
  if (!kf_node_id || !f_node_id || !kf_off || !f_off || !kf_idx || !f_idx) return VSG_ERR_INVALID;
  const vsg_keypoint *ka = kf->h_kps.data(), *kb = f->h_kps.data();
  return search_by_bow(kf->device, 0, f->nleft, nullptr, kf->d_desc, [&](int i) { return ka[i].angle; }, kf_valid, kf->n,
                       kf_node_id, kf_off, kf_idx, kf_nodes, nullptr, f->d_desc, [&](int i) { return kb[i].angle; },
                       nullptr, f->n, f_node_id, f_off, f_idx, f_nodes, nnratio, check_orientation, match_f);
}

int vsg_frame_search_by_bow_kf_kf(vsg_frame *kf1, const uint8_t *valid1, const int32_t *node_id1, const int32_t *off1,
                                  const int32_t *idx1, int nodes1, vsg_frame *kf2, const uint8_t *valid2,
                                  const int32_t *node_id2, const int32_t *off2, const int32_t *idx2, int nodes2,
                                  float nnratio, int check_orientation, int32_t *matches12) {
  if (!kf1 || !kf2 || !matches12 || kf1->device != kf2->device || !valid1 || !valid2) return VSG_ERR_INVALID;
  if (!node_id1 && !node_id2) return search_by_bow_resident(1, kf1, valid1, kf2, valid2, nnratio, check_orientation, matches12);
  if (!node_id1 || !node_id2 || !off1 || !off2 || !idx1 || !idx2) return VSG_ERR_INVALID;
  const vsg_keypoint *ka = kf1->h_kps.data(), *kb = kf2->h_kps.data();
  return search_by_bow(kf1->device, 1, -1, nullptr, kf1->d_desc, [&](int i) { return ka[i].angle; }, valid1, kf1->n,
                       node_id1, off1, idx1, nodes1, nullptr, kf2->d_desc, [&](int i) { return kb[i].angle; }, valid2,
                       kf2->n, node_id2, off2, idx2, nodes2, nnratio, check_orientation, matches12);
}

int vsg_search_by_projection_last(int device, const uint8_t *q_desc, const float *q_angle,
                                  const uint8_t *query_blocks, int n_q, const int32_t *cand_off,
                                  const int32_t *cand_idx, const uint8_t *t_desc, const float *t_angle,
                                  uint8_t *train_blocked, int n_t, int th_high, int check_orientation,
                                  int32_t *train_match) {
  if (!cand_off || !train_blocked || !train_match || n_q < 0 || n_t < 0) return VSG_ERR_INVALID;
  if (n_q == 0 || n_t == 0) return use_device(device) == VSG_OK ? 0 : VSG_ERR_NO_DEVICE;
  const uint32_t *ent = nullptr;
  int rc = candidate_entries(device, q_desc, n_q, cand_off, cand_idx, t_desc, nullptr, n_t, &ent);
  if (rc != VSG_OK) return rc;
  // ORBmatcher.cc:1686-1784 (left / mono block) + the rotation filter :1855-1875
  return walk::search_last(csr_view(ent, cand_off), n_q, -1, q_angle, query_blocks, [&](int i) { return t_angle[i]; },
                           th_high, check_orientation != 0, train_blocked, train_match);
}

int vsg_search_by_projection_local(int device, const uint8_t *q_desc, const uint8_t *query_blocks, int n_q,
                                   const int32_t *cand_off, const int32_t *cand_idx, const uint8_t *t_desc,
                                   const int32_t *t_octave, uint8_t *train_blocked, int n_t, float nnratio,
                                   int32_t *train_match) {
  if (!cand_off || !train_blocked || !train_match || !t_octave || n_q < 0 || n_t < 0) return VSG_ERR_INVALID;
  if (n_q == 0 || n_t == 0) return use_device(device) == VSG_OK ? 0 : VSG_ERR_NO_DEVICE;
  const uint32_t *ent = nullptr;
  int rc = candidate_entries(device, q_desc, n_q, cand_off, cand_idx, t_desc, t_octave, n_t, &ent);
  if (rc != VSG_OK) return rc;
  // ORBmatcher.cc:48-144 (left block): every query is "in view"
  std::vector<uint8_t> in_view((size_t)n_q, 1);
  return walk::search_local(csr_view(ent, cand_off), n_q, -1, in_view.data(), nullptr, nullptr, query_blocks, nnratio,
                            nullptr, nullptr, train_blocked, train_match);
}

int vsg_distinctive_descriptors(int device, const uint8_t *desc, const int32_t *off, int ngroups, int32_t *best) {
  if (!off || !best || ngroups < 0) return VSG_ERR_INVALID;
  int rc = VSG_OK;
  ThreadCtx *c = vsg::thread_ctx(device, &rc);
  if (!c) return rc;
  if (ngroups == 0) return VSG_OK;
  for (int g = 0; g < ngroups; g++)
    if (off[g + 1] - off[g] > kDistinctMaxN) return VSG_ERR_UNSUPPORTED;
  const int n = off[ngroups];
  Stage st;
  const size_t oD = st.add((size_t)n * 32), oOff = st.add((size_t)(ngroups + 1) * 4);
  const size_t in_bytes = st.total;
  const size_t oB = st.add((size_t)ngroups * 4);
  rc = vsg::ctx_reserve(c, st.total, in_bytes);
  if (rc != VSG_OK) return rc;
  uint8_t *h = c->h_pin;
  if (n) memcpy(h + oD, desc, (size_t)n * 32);
  memcpy(h + oOff, off, (size_t)(ngroups + 1) * 4);
  M_TRY(hipMemcpyAsync(c->d_buf, h, in_bytes, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_distinctive, dim3(ngroups), dim3(128), 0, c->stream, c->d_buf + oD, (const int *)(c->d_buf + oOff),
                     (int *)(c->d_pin + oB));
  M_TRY(hipGetLastError());
  M_TRY(hipStreamSynchronize(c->stream));
  memcpy(best, h + oB, (size_t)ngroups * 4);
  return VSG_OK;
}

int vsg_search_window(int device, const uint8_t *q_desc, const uint8_t *query_blocks, int n_q,
                      const int32_t *cand_off, const int32_t *cand_idx, const uint8_t *t_desc, uint8_t *train_blocked,
                      int n_t, int th_high, int32_t *q_best_idx, int32_t *q_best_dist, int32_t *train_match) {
  if (!cand_off || !q_best_idx || !q_best_dist || n_q < 0 || n_t < 0) return VSG_ERR_INVALID;
  for (int q = 0; q < n_q; q++) q_best_idx[q] = -1, q_best_dist[q] = 256;
  if (n_q == 0 || n_t == 0) return use_device(device) == VSG_OK ? 0 : VSG_ERR_NO_DEVICE;
  const uint32_t *ent = nullptr;
  int rc = candidate_entries(device, q_desc, n_q, cand_off, cand_idx, t_desc, nullptr, n_t, &ent);
  if (rc != VSG_OK) return rc;
  const walk::CandView cv = csr_view(ent, cand_off);
  int nmatches = 0;
  for (int q = 0; q < n_q; q++) {
    int bestDist, bestIdx;
    walk::scan_best(cv.begin(q), cv.size(q), [&](int idx) { return train_blocked && train_blocked[idx]; }, 256, bestDist,
                    bestIdx);
    q_best_idx[q] = bestIdx;
    q_best_dist[q] = bestDist;
    if (bestIdx >= 0 && bestDist <= th_high) {
      if (train_match) train_match[bestIdx] = q;
      if (train_blocked) train_blocked[bestIdx] = query_blocks ? query_blocks[q] : 0;
      nmatches++;
    }
  }
  return nmatches;
}

int vsg_search_for_initialization(int device, const uint8_t *desc1, const float *angle1, const int32_t *octave1,
                                  int n1, const int32_t *cand_off, const int32_t *cand_idx, const uint8_t *desc2,
                                  const float *angle2, int n2, float nnratio, int check_orientation,
                                  int32_t *matches12) {
  if (!cand_off || !matches12 || !octave1 || n1 < 0 || n2 < 0) return VSG_ERR_INVALID;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (n1 == 0 || n2 == 0) return use_device(device) == VSG_OK ? 0 : VSG_ERR_NO_DEVICE;
  const uint32_t *ent = nullptr;
  int rc = candidate_entries(device, desc1, n1, cand_off, cand_idx, desc2, nullptr, n2, &ent);
  if (rc != VSG_OK) return rc;
  // ORBmatcher.cc:643-756
  return walk::search_initialization(csr_view(ent, cand_off), n1, n2, octave1, [&](int i) { return angle1[i]; },
                                     [&](int i) { return angle2[i]; }, nnratio, check_orientation != 0, matches12);
}

}  // extern "C"
