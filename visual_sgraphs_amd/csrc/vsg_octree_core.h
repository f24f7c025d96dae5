// vsg_octree_core.h -- ORBextractor::DistributeOctTree (ORBextractor.cc:482-785) as a data-parallel,
// array-based algorithm for ONE workgroup per (frame, level).
//
// The reference walks a std::list, push_front()s children, and switches to a "careful" phase that
// std::sort()s (size, node*) pairs and splits the largest nodes one at a time until the quota N is
// reached.  Its output ORDER is the list order, which this code reproduces exactly:
//   * a pass that splits the nodes proc[0..nEff) (in that order) creates their non-empty children in
//     order n1..n4; the new list is reverse(creation order) ++ (old list minus the split nodes);
//   * main passes use proc = every node with >1 point in list order; careful passes use the libstdc++
//     introsort order (vsg_introsort.h) back to front and stop at the first prefix reaching size >= N;
//   * vSizeAndPointerToNode (the sort input) is the creation order of children with >1 point.
// Points are never moved: each keeps the list position of its node (node_of[]), relabelled per pass.
// The candidate order only matters for "first maximum wins" (ORBextractor.cc:774), which is carried by
// an explicit rank = position in the reference's cell-major / row-major candidate order.
//
// The algorithm is written against a Group concept (tid, nthreads, sync, LDS atomics, block scan) so
// the same source runs as a 256-thread workgroup on the GPU and as a 1-thread group in the host unit
// tests (tests/_hostcore), which compare it with the oracle's literal std::list restatement.
#pragma once
#include <stdint.h>

#include "vsg_common.h"
#include "vsg_introsort.h"

#if defined(__HIPCC__)
#define VSG_OCT_HD __host__ __device__ inline
#define VSG_OCT_UNROLL _Pragma("unroll")
#else
#define VSG_OCT_HD inline
#define VSG_OCT_UNROLL
#endif

namespace vsg {
namespace octree {

// ---- single-thread group (host tests; also valid on device for debugging)
struct SerialGroup {
  int tid = 0, nthreads = 1;
  VSG_OCT_HD void sync() {}
  VSG_OCT_HD int atomic_add(int *p, int v) {
    int o = *p;
    *p = o + v;
    return o;
  }
  VSG_OCT_HD void atomic_max(uint32_t *p, uint32_t v) {
    if (v > *p) *p = v;
  }
  VSG_OCT_HD void atomic_max64(uint64_t *p, uint64_t v) {
    if (v > *p) *p = v;
  }
  VSG_OCT_HD void atomic_min(int *p, int v) {
    if (v < *p) *p = v;
  }
  // p[0..4) += the group's sums of four per-thread counts
  VSG_OCT_HD void add4(int *p, int c0, int c1, int c2, int c3) { p[0] += c0, p[1] += c1, p[2] += c2, p[3] += c3; }
  // in-place exclusive prefix sum of a[0..n); returns the total
  VSG_OCT_HD int exclusive_scan(int *a, int n) {
    int s = 0;
    for (int i = 0; i < n; i++) {
      int v = a[i];
      a[i] = s;
      s += v;
    }
    return s;
  }
  VSG_OCT_HD int exclusive_scan2(int *a, int *b, int n, int *total_b) {
    *total_b = exclusive_scan(b, n);
    return exclusive_scan(a, n);
  }
  // one value pair per THREAD (tid = element index): exclusive prefixes of a and b over the group, totals returned
  VSG_OCT_HD int exclusive_scan2_one(int a, int b, int *ex_a, int *ex_b, int *total_b) {
    *ex_a = 0, *ex_b = 0, *total_b = b;
    return a;
  }
  // std::sort's quicksort half on items[0..n) (see vsg_introsort.h); posA/posB: n + 1 uint16 of scratch each.
  // The caller syncs afterwards.
  VSG_OCT_HD void sort_partition_phase(introsort::item_t *items, int n, uint16_t *, uint16_t *) {
    introsort::partition_phase(items, n);
  }
  // whole std::sort + the back-to-front processing order in one step, where the group has a faster form (false: use the
  // two-step form: sort_partition_phase, then the stable ranks)
  VSG_OCT_HD bool sort_to_proc(introsort::item_t *, int, uint16_t *) { return false; }
};

struct Params {
  int N;                 // quota for this level
  int height;            // maxY - minY
  int nIni;
  const int *iniUL;      // nIni + 1
  const int *iniThresh;  // nIni
  int nCols, wCell, hCell;  // FAST cell grid, for the candidate rank
};

// Capacity (in nodes) the workspace must provide for quota N.  The list never exceeds
// max(N + 3, 4 * nIni): the first pass makes <= 4*nIni nodes; a further main pass only runs when
// size + 3*nToExpand <= N (:696) and adds <= 3 per split node; the careful phase adds <= 3 per split
// and stops at the first size >= N (:753).
VSG_OCT_HD int node_capacity(int N) { return (N + 3 > 4 * kMaxIniNodes ? N + 3 : 4 * kMaxIniNodes) + 4; }

// ---- where a thread keeps "its" points (p = tid, tid + nthreads, ...) between passes.
// MemPts: candidate words and node labels live in memory (cand[] read-only, node_of[] scratch) -- any npts.
// RegPts<K>: both live in the thread's registers (npts <= K * nthreads): the per-pass sweeps then touch only the
// node arrays in LDS, which takes the global-memory round trips out of the (latency-bound) pass loop.
template <int U>
struct MemPtsT {
  const uint32_t *cand;
  uint16_t *node_of;
  template <class G>
  VSG_OCT_HD void load(G &, int) {}
  // The sweeps take their points U at a time -- U candidate words and U labels requested, then U bodies, then the changed
  // labels stored: written one point per trip every trip waited for its own loads (the label store of one trip may alias the
  // loads of the next, so the compiler keeps them in order), ~700 cycles x 78 trips per sweep over the 20 k candidates of a
  // photograph's level 0 at 1280x720 (profiles/r06_u_*).  The points of a sweep are independent of each other (shared state
  // is only touched through the group's atomics), so the order inside a batch does not matter.  U is the kernel's choice: the
  // batched bodies cost registers, and the launch that also carries the blur and the register form of every 640x480 level
  // (k_octree_blur) measured 1.2 % slower on the default content with U = 4 where the stand-alone k_octree gains
  // (vsg_kernels.hip: kOctMemBatch*).
  template <class G, class F>
  VSG_OCT_HD void for_each(G &g, int npts, F f) {
    const int step = g.nthreads;
    int p = g.tid;
    for (; p + (U - 1) * step < npts; p += U * step) {
      uint32_t c[U];
      int n[U], n0[U];
      VSG_OCT_UNROLL
      for (int u = 0; u < U; u++) c[u] = cand[p + u * step], n0[u] = n[u] = node_of[p + u * step];
      VSG_OCT_UNROLL
      for (int u = 0; u < U; u++) f(c[u], n[u]);
      VSG_OCT_UNROLL
      for (int u = 0; u < U; u++)
        if (n[u] != n0[u]) node_of[p + u * step] = (uint16_t)n[u];
    }
    for (; p < npts; p += step) {
      int n = node_of[p];
      const int n0 = n;
      f(cand[p], n);
      if (n != n0) node_of[p] = (uint16_t)n;
    }
  }
  template <class G, class F>
  VSG_OCT_HD void init_each(G &g, int npts, F f) {  // first sweep: labels are written, not read
    const int step = g.nthreads;
    int p = g.tid;
    for (; p + (U - 1) * step < npts; p += U * step) {
      uint32_t c[U];
      int n[U];
      VSG_OCT_UNROLL
      for (int u = 0; u < U; u++) c[u] = cand[p + u * step], n[u] = 0;
      VSG_OCT_UNROLL
      for (int u = 0; u < U; u++) f(c[u], n[u]);
      VSG_OCT_UNROLL
      for (int u = 0; u < U; u++) node_of[p + u * step] = (uint16_t)n[u];
    }
    for (; p < npts; p += step) {
      int n = 0;
      f(cand[p], n);
      node_of[p] = (uint16_t)n;
    }
  }
};
using MemPts = MemPtsT<1>;

// where candidate p comes from: a contiguous array, or (device) the per-cell segments k_fast_cells writes
struct PtrSrc {
  const uint32_t *cand;
  VSG_OCT_HD uint32_t operator()(int p) const { return cand[p]; }
};

template <int K, class Src = PtrSrc>
struct RegPts {
  Src src;
  uint32_t c[K];
  int n[K];
  template <class G>
  VSG_OCT_HD void load(G &g, int npts) {
    VSG_OCT_UNROLL
    for (int k = 0; k < K; k++) {
      const int p = g.tid + k * g.nthreads;
      c[k] = p < npts ? src(p) : 0u;
      n[k] = 0;
    }
  }
  template <class G, class F>
  VSG_OCT_HD void for_each(G &g, int npts, F f) {
    VSG_OCT_UNROLL
    for (int k = 0; k < K; k++)
      if (g.tid + k * g.nthreads < npts) f(c[k], n[k]);
  }
  template <class G, class F>
  VSG_OCT_HD void init_each(G &g, int npts, F f) {
    for_each(g, npts, f);
  }
};

// Workspace: carve from one byte buffer (LDS on the device).  Layout is 8-byte aligned.
struct Work {
  // The two generations (ping-pong index b) of the node boxes / counts are addressed by arithmetic, not through
  // pointer arrays: `ptr[b]` with a run-time b made the compiler keep the whole struct in scratch memory (every
  // access a scratch load, and the kernel a scratch user).
  int16_t *box;  // [b][ulx|uly|urx|bly][capa]
  int *cntb;     // [b][capa]
  int capa;
  VSG_OCT_HD int16_t *ulx(int b) const { return box + (size_t)(4 * b + 0) * capa; }
  VSG_OCT_HD int16_t *uly(int b) const { return box + (size_t)(4 * b + 1) * capa; }
  VSG_OCT_HD int16_t *urx(int b) const { return box + (size_t)(4 * b + 2) * capa; }
  VSG_OCT_HD int16_t *bly(int b) const { return box + (size_t)(4 * b + 3) * capa; }
  VSG_OCT_HD int *cnt(int b) const { return cntb + (size_t)b * capa; }
  int *childcnt;        // 4 per node; aliased by sortbuf (item_t per node) and bestkey
  uint16_t *childpos;   // 4 per node
  uint16_t *keeppos;
  uint16_t *proc;
  uint16_t *V;
  uint8_t *divided;
  int *scanA, *scanB;
  int *ctrl;            // [0]=nL [1]=nV [2]=break index
  // histogram mode (hist_* below): point counts of every quadtree cell down to depth D under each of up to kFuseRoots
  // initial nodes (D = 5 under one root, 4 under two to four: kHistInts either way), the depth-D cell -> list position table,
  // and each node's (root, depth, path) word in two generations
  int *hist;            // [roots][4 + 16 + ... + 4^D]
  uint16_t *cellpos;    // [roots][4^D]
  int histInts, cellCap;  // capacity of hist / cellpos (two workspace classes, see work_bytes)
  // node words root << 13 | depth << 10 | path (2 bits per level), generation b in the first two quarters of childpos: the
  // child positions are what the LABEL-based passes relabel the points through -- not live in histogram mode, and written
  // before they are read once it has been left
  VSG_OCT_HD uint16_t *ncode(int b) const { return childpos + (size_t)b * capa; }
};

// Two workspace classes: the BIG histogram (depth 5 under one root, 4 under up to four: 5.5 + 2 KB) where a workgroup's LDS
// share has the room, the SMALL one (depth 4 under one root, 3 under up to four: 1.4 + 0.5 KB, the size rounds 5's depth-3
// tables had) where the node arrays already fill it (1280x720 / 2000).
enum { kFuseRoots = 4, kHistIntsBig = 4 + 16 + 64 + 256 + 1024, kHistCellsBig = 1024, kHistIntsSmall = 4 + 16 + 64 + 256,
       kHistCellsSmall = 256 };
// cells of levels 1 .. d - 1 = where level d starts inside a root's histogram; a root's histogram holds levels 1 .. D
VSG_OCT_HD int hist_off(int d) { return ((1 << (2 * d)) - 4) / 3; }
// the deepest histogram the workspace holds for nRoots initial nodes (0: none)
VSG_OCT_HD int hist_depth(const Work &W, int nRoots) {
  int D = 5;
  while (D > 0 && (nRoots * hist_off(D + 1) > W.histInts || nRoots * (1 << (2 * D)) > W.cellCap)) D--;
  return D;
}

VSG_OCT_HD size_t work_bytes(int cap, bool big = true) {
  size_t capa = (size_t)((cap + 3) & ~3);
  return capa * (2 * 4 * 2 + 2 * 4 + 16 + 8 + 2 + 2 + 2 + 1 + 4 + 4) + 64 +
         (big ? kHistIntsBig * 4 + kHistCellsBig * 2 : kHistIntsSmall * 4 + kHistCellsSmall * 2);
}

VSG_OCT_HD void carve(Work &W, void *buf, int cap, bool big = true) {
  W.histInts = big ? kHistIntsBig : kHistIntsSmall;
  W.cellCap = big ? kHistCellsBig : kHistCellsSmall;
  size_t capa = (size_t)((cap + 3) & ~3);
  uint8_t *p = (uint8_t *)buf;
  W.childcnt = (int *)p;
  p += capa * 16;
  W.capa = (int)capa;
  W.cntb = (int *)p;
  p += 2 * capa * 4;
  W.scanA = (int *)p;
  p += capa * 4;
  W.scanB = (int *)p;
  p += capa * 4;
  W.ctrl = (int *)p;
  p += 64;
  W.childpos = (uint16_t *)p;
  p += capa * 8;
  W.box = (int16_t *)p;
  p += 8 * capa * 2;
  W.keeppos = (uint16_t *)p;
  p += capa * 2;
  W.proc = (uint16_t *)p;
  p += capa * 2;
  W.V = (uint16_t *)p;
  p += capa * 2;
  W.cellpos = (uint16_t *)p;
  p += (size_t)W.cellCap * 2;
  W.hist = (int *)p;  // 4-byte aligned: everything before it is a multiple of 4 bytes (capa is a multiple of 4)
  p += (size_t)W.histInts * 4;
  W.divided = (uint8_t *)p;
}

VSG_OCT_HD int quadrant(const Work &W, int b, int n, int x, int y) {
  const int ulx = W.ulx(b)[n], uly = W.uly(b)[n];
  const int midX = ulx + ((W.urx(b)[n] - ulx + 1) >> 1);  // UL.x + ceil((UR.x-UL.x)/2)  (:484)
  const int midY = uly + ((W.bly(b)[n] - uly + 1) >> 1);
  return (x >= midX ? 1 : 0) | (y >= midY ? 2 : 0);       // n1,n2,n3,n4 (:513-527)
}

// position of a candidate in the reference's candidate order (cells row-major, pixels row-major
// inside a cell): ORBextractor.cc:811-875 + FAST_t's row-major output.
// a / b for 0 <= a < 2^15, 0 < b < 2^12 through the float reciprocal (an integer division is ~40 instructions on the device
// and this runs twice per candidate): the float quotient is off by at most one, two compares make it exact
VSG_OCT_HD int div_exact(int a, int b, float inv_b) {
  int q = (int)((float)a * inv_b);
  q -= q * b > a;
  q += (q + 1) * b <= a;
  return q;
}
VSG_OCT_HD uint32_t cand_rank(const Params &P, int x, int y, float inv_w, float inv_h) {
  const int cx = x - 3, cy = y - 3;  // valid region of a cell starts 3 px inside it
  const int j = div_exact(cx, P.wCell, inv_w), i = div_exact(cy, P.hCell, inv_h);
  return (uint32_t)(((i * P.nCols + j) * P.hCell + (cy - i * P.hCell)) * P.wCell + (cx - j * P.wCell));
}

// One splitting pass over proc[0..nProc).  Returns new list length; *nV_out = |V|.
template <class G, class PT>
VSG_OCT_HD int run_pass(G &g, const Params &P, Work &W, int &cur, int nL, int nProc, bool careful, PT &pts, int npts,
                        int *nV_out) {
  const int b = cur, nb = cur ^ 1;
  for (int i = g.tid; i < nL; i += g.nthreads) W.divided[i] = 0;
  g.sync();
  for (int t = g.tid; t < nProc; t += g.nthreads) {
    const int n = W.proc[t];
    W.divided[n] = 1;
    W.childcnt[4 * n + 0] = 0;
    W.childcnt[4 * n + 1] = 0;
    W.childcnt[4 * n + 2] = 0;
    W.childcnt[4 * n + 3] = 0;
  }
  if (g.tid == 0) W.ctrl[2] = nProc;
  g.sync();
  pts.for_each(g, npts, [&](uint32_t c, int &n) {
    if (W.divided[n]) g.atomic_add(&W.childcnt[4 * n + quadrant(W, b, n, VSG_CAND_X(c), VSG_CAND_Y(c))], 1);
  });
  g.sync();
  for (int t = g.tid; t < nProc; t += g.nthreads) {
    const int n = W.proc[t];
    int k = 0, e = 0;
    for (int c = 0; c < 4; c++) {
      const int cc = W.childcnt[4 * n + c];
      k += cc > 0;
      e += cc > 1;
    }
    W.scanA[t] = k | (e << 16);
  }
  g.sync();
  const int total = g.exclusive_scan(W.scanA, nProc);
  int nEff = nProc;
  if (careful) {
    // `if ((int)lNodes.size() >= N) break;` after each split (:753)
    for (int t = g.tid; t < nProc; t += g.nthreads) {
      const int n = W.proc[t];
      int k = 0;
      for (int c = 0; c < 4; c++) k += W.childcnt[4 * n + c] > 0;
      const int sizeAfter = nL + (W.scanA[t] & 0xFFFF) + k - (t + 1);
      if (sizeAfter >= P.N) g.atomic_min(&W.ctrl[2], t + 1);
    }
    g.sync();
    nEff = W.ctrl[2];
    g.sync();
    for (int t = nEff + g.tid; t < nProc; t += g.nthreads) W.divided[W.proc[t]] = 0;
  }
  const int packed = nEff < nProc ? W.scanA[nEff] : total;
  const int K = packed & 0xFFFF, E = packed >> 16;
  g.sync();
  for (int i = g.tid; i < nL; i += g.nthreads) W.scanB[i] = W.divided[i] ? 0 : 1;
  g.sync();
  const int kept = g.exclusive_scan(W.scanB, nL);
  const int newL = K + kept;
  for (int t = g.tid; t < nEff; t += g.nthreads) {
    const int n = W.proc[t];
    const int q0 = W.scanA[t] & 0xFFFF, e0 = W.scanA[t] >> 16;
    const int ulx = W.ulx(b)[n], uly = W.uly(b)[n], urx = W.urx(b)[n], bly = W.bly(b)[n];
    const int midX = ulx + ((urx - ulx + 1) >> 1), midY = uly + ((bly - uly + 1) >> 1);
    int m = 0, ev = 0;
    for (int c = 0; c < 4; c++) {
      const int cc = W.childcnt[4 * n + c];
      if (cc > 0) {
        const int pos = K - 1 - (q0 + m);  // push_front => reversed creation order
        W.ulx(nb)[pos] = (int16_t)((c & 1) ? midX : ulx);
        W.urx(nb)[pos] = (int16_t)((c & 1) ? urx : midX);
        W.uly(nb)[pos] = (int16_t)((c & 2) ? midY : uly);
        W.bly(nb)[pos] = (int16_t)((c & 2) ? bly : midY);
        W.cnt(nb)[pos] = cc;
        W.childpos[4 * n + c] = (uint16_t)pos;
        if (cc > 1) W.V[e0 + ev++] = (uint16_t)pos;
        m++;
      }
    }
  }
  for (int i = g.tid; i < nL; i += g.nthreads) {
    if (!W.divided[i]) {
      const int pos = K + W.scanB[i];
      W.ulx(nb)[pos] = W.ulx(b)[i];
      W.urx(nb)[pos] = W.urx(b)[i];
      W.uly(nb)[pos] = W.uly(b)[i];
      W.bly(nb)[pos] = W.bly(b)[i];
      W.cnt(nb)[pos] = W.cnt(b)[i];
      W.keeppos[i] = (uint16_t)pos;
    }
  }
  g.sync();
  pts.for_each(g, npts, [&](uint32_t c, int &n) {
    n = W.divided[n] ? W.childpos[4 * n + quadrant(W, b, n, VSG_CAND_X(c), VSG_CAND_Y(c))] : W.keeppos[n];
  });
  g.sync();
  cur = nb;
  *nV_out = E;
  return newL;
}

// A MAIN pass (:617-690): every node with more than one point is split, in list order.  Same result as
// run_pass(proc = those nodes, careful = false) but with one fused scan -- children created before a node (K),
// children with > 1 point before it (E, the sort input order) and kept nodes before it all ride the same pair of
// barriers -- 7 barriers per pass instead of ~20 (the passes are latency-bound: barriers and LDS round trips).
template <class G, class PT>
VSG_OCT_HD int run_main_pass(G &g, const Params &P, Work &W, int &cur, int nL, PT &pts, int npts, int *nV_out) {
  (void)P;
  const int b = cur, nb = cur ^ 1;
  for (int i = g.tid; i < nL; i += g.nthreads) {
    const bool d = W.cnt(b)[i] > 1;
    W.divided[i] = d;
    if (d) {
      W.childcnt[4 * i + 0] = 0;
      W.childcnt[4 * i + 1] = 0;
      W.childcnt[4 * i + 2] = 0;
      W.childcnt[4 * i + 3] = 0;
    }
  }
  g.sync();
  pts.for_each(g, npts, [&](uint32_t c, int &n) {
    if (W.divided[n]) g.atomic_add(&W.childcnt[4 * n + quadrant(W, b, n, VSG_CAND_X(c), VSG_CAND_Y(c))], 1);
  });
  g.sync();
  for (int i = g.tid; i < nL; i += g.nthreads) {
    int k = 0, e = 0;
    if (W.divided[i]) {
      for (int c = 0; c < 4; c++) {
        const int cc = W.childcnt[4 * i + c];
        k += cc > 0;
        e += cc > 1;
      }
    }
    W.scanA[i] = k | (e << 16);
    W.scanB[i] = W.divided[i] ? 0 : 1;
  }
  g.sync();
  int kept = 0;
  const int total = g.exclusive_scan2(W.scanA, W.scanB, nL, &kept);
  const int K = total & 0xFFFF, E = total >> 16;
  const int newL = K + kept;
  for (int i = g.tid; i < nL; i += g.nthreads) {
    const int ulx = W.ulx(b)[i], uly = W.uly(b)[i], urx = W.urx(b)[i], bly = W.bly(b)[i];
    if (W.divided[i]) {
      const int q0 = W.scanA[i] & 0xFFFF, e0 = W.scanA[i] >> 16;
      const int midX = ulx + ((urx - ulx + 1) >> 1), midY = uly + ((bly - uly + 1) >> 1);
      int m = 0, ev = 0;
      for (int c = 0; c < 4; c++) {
        const int cc = W.childcnt[4 * i + c];
        if (cc > 0) {
          const int pos = K - 1 - (q0 + m);  // push_front => reversed creation order
          W.ulx(nb)[pos] = (int16_t)((c & 1) ? midX : ulx);
          W.urx(nb)[pos] = (int16_t)((c & 1) ? urx : midX);
          W.uly(nb)[pos] = (int16_t)((c & 2) ? midY : uly);
          W.bly(nb)[pos] = (int16_t)((c & 2) ? bly : midY);
          W.cnt(nb)[pos] = cc;
          W.childpos[4 * i + c] = (uint16_t)pos;
          if (cc > 1) W.V[e0 + ev++] = (uint16_t)pos;
          m++;
        }
      }
    } else {
      const int pos = K + W.scanB[i];
      W.ulx(nb)[pos] = (int16_t)ulx;
      W.urx(nb)[pos] = (int16_t)urx;
      W.uly(nb)[pos] = (int16_t)uly;
      W.bly(nb)[pos] = (int16_t)bly;
      W.cnt(nb)[pos] = W.cnt(b)[i];
      W.keeppos[i] = (uint16_t)pos;
    }
  }
  g.sync();
  pts.for_each(g, npts, [&](uint32_t c, int &n) {
    n = W.divided[n] ? W.childpos[4 * n + quadrant(W, b, n, VSG_CAND_X(c), VSG_CAND_Y(c))] : W.keeppos[n];
  });
  g.sync();
  cur = nb;
  *nV_out = E;
  return newL;
}

// ---- Histogram mode: passes without point sweeps.
// A pass splits nodes into four, so which nodes exist after k passes -- and in which list order -- only depends on how many
// points lie in each quadtree cell of depth <= k: the boxes are pure geometry (a node's box is its root's box halved along
// its path, :484-527), the points are never moved, and the list surgery of a pass (children in creation order n1..n4, pushed
// to the front, :617-690; the careful phase's sorted order and its stop test, :696-757) only looks at counts.  So: ONE sweep
// over the points computes each point's path down to depth D in registers (no LDS gathers) and counts it into its depth-D
// cell; the coarser counts are sums of four; every pass -- main or careful -- whose nodes sit above depth D then runs on the
// node arrays alone, the same statements as run_main_pass / run_pass with the child counts read from the histogram; ONE more
// sweep (hist_leave) gives every point the list position of the node its depth-D cell ended up in.  A node's depth is at
// most the number of passes so far, so the first D passes qualify whatever they are; a level that needs more leaves the
// mode and goes on with the label-based passes.
// Rounds 5-6: depth 3 and main passes only (the one-frame octree: 29 of 81 k cycles in those passes) -> depth 5 under one
// root / 4 under up to four, careful passes included: a photograph's level 0 (5.7 k points, 165 k cycles in the memory form)
// ran its fourth main pass and its careful pass with two point sweeps each; the default content's careful pass took its
// child counts from a sweep.
// Requires nL <= kFuseRoots initial nodes in generation `cur`, labels n = list position.  Labels become root * 4^D + path.
template <class G, class PT>
VSG_OCT_HD void hist_setup(G &g, Work &W, int cur, int nL, PT &pts, int npts, int D) {
  const int nRoots = nL, S = hist_off(D + 1), cellsD = 1 << (2 * D);
  for (int i = g.tid; i < nRoots * S; i += g.nthreads) W.hist[i] = 0;
  for (int i = g.tid; i < nL; i += g.nthreads) W.ncode(cur)[i] = (uint16_t)(i << 13);
  g.sync();
  {
    const int b = cur, offD = hist_off(D);
    pts.for_each(g, npts, [&](uint32_t c, int &n) {
      const int x = VSG_CAND_X(c), y = VSG_CAND_Y(c);
      int ulx = W.ulx(b)[n], uly = W.uly(b)[n], urx = W.urx(b)[n], bly = W.bly(b)[n], path = 0;
      for (int d = 0; d < D; d++) {
        const int midX = ulx + ((urx - ulx + 1) >> 1), midY = uly + ((bly - uly + 1) >> 1);
        const int qx = x >= midX, qy = y >= midY;
        path = path * 4 + (qx | (qy << 1));
        if (qx) ulx = midX; else urx = midX;
        if (qy) uly = midY; else bly = midY;
      }
      g.atomic_add(&W.hist[n * S + offD + path], 1);
      n = n * cellsD + path;
    });
  }
  g.sync();
  for (int d = D - 1; d >= 1; d--) {  // level d = sums of four level d + 1 cells
    const int cells = 1 << (2 * d), od = hist_off(d), oc = hist_off(d + 1);
    for (int i = g.tid; i < nRoots * cells; i += g.nthreads) {
      const int r = i >> (2 * d), c = i & (cells - 1);
      const int *h = &W.hist[r * S + oc + 4 * c];
      W.hist[r * S + od + c] = h[0] + h[1] + h[2] + h[3];
    }
    g.sync();
  }
}

// child counts of the node with word `code` (its depth < D)
VSG_OCT_HD const int *hist_children(const Work &W, int code, int S) {
  const int r = code >> 13, d = (code >> 10) & 7, path = code & 1023;
  return &W.hist[r * S + hist_off(d + 1) + 4 * path];
}
VSG_OCT_HD uint16_t hist_child_code(int code, int c) {
  return (uint16_t)((code & 0xE000) | ((((code >> 10) & 7) + 1) << 10) | ((code & 1023) * 4 + c));
}

// leave the mode: depth-D cell -> list position of the node that holds it (a node at depth d covers 4^(D - d) consecutive
// paths), then every point's label through that table
template <class G, class PT>
VSG_OCT_HD void hist_leave(G &g, Work &W, int cur, int nL, PT &pts, int npts, int D) {
  const int cellsD = 1 << (2 * D);
  for (int i = g.tid; i < nL; i += g.nthreads) {
    const int code = W.ncode(cur)[i], r = code >> 13, d = (code >> 10) & 7, path = code & 1023;
    const int span = 1 << (2 * (D - d)), first = path * span;
    for (int j = 0; j < span; j++) W.cellpos[r * cellsD + first + j] = (uint16_t)i;
  }
  g.sync();
  pts.for_each(g, npts, [&](uint32_t, int &n) { n = W.cellpos[n]; });
  g.sync();
}

// A MAIN pass (:617-690) in histogram mode.  Returns the new list length; *nV_out = |V|.
template <class G>
VSG_OCT_HD int hist_main_pass(G &g, Work &W, int &cur, int nL, int D, int *nV_out) {
  const int b = cur, nb = cur ^ 1, S = hist_off(D + 1);
  if (nL <= g.nthreads) {
    // One node per thread (on the device the common case: <= 256 nodes enter a pass): the node, its four child counts and
    // its scan values stay in the thread's registers from the first read to the last write -- one block scan and one
    // barrier per pass instead of three LDS round trips through divided / childcnt / scanA / scanB and four barriers.
    const int i = g.tid;
    const bool have = i < nL;
    int ulx = 0, uly = 0, urx = 0, bly = 0, code = 0, cnt = 0, k = 0, e = 0, cc[4] = {0, 0, 0, 0};
    bool d = false;
    if (have) {
      cnt = W.cnt(b)[i];
      ulx = W.ulx(b)[i], uly = W.uly(b)[i], urx = W.urx(b)[i], bly = W.bly(b)[i];
      code = W.ncode(b)[i];
      d = cnt > 1;
      if (d) {
        const int *h = hist_children(W, code, S);
        for (int c = 0; c < 4; c++) {
          cc[c] = h[c];
          k += cc[c] > 0;
          e += cc[c] > 1;
        }
      }
    }
    int ex = 0, exKept = 0, kept = 0;
    const int total = g.exclusive_scan2_one(k | (e << 16), have && !d ? 1 : 0, &ex, &exKept, &kept);
    const int K = total & 0xFFFF, E = total >> 16;
    if (have) {
      if (d) {
        const int q0 = ex & 0xFFFF, e0 = ex >> 16;
        const int midX = ulx + ((urx - ulx + 1) >> 1), midY = uly + ((bly - uly + 1) >> 1);
        int m = 0, ev = 0;
        for (int c = 0; c < 4; c++) {
          if (cc[c] > 0) {
            const int pos = K - 1 - (q0 + m);  // push_front => reversed creation order
            W.ulx(nb)[pos] = (int16_t)((c & 1) ? midX : ulx);
            W.urx(nb)[pos] = (int16_t)((c & 1) ? urx : midX);
            W.uly(nb)[pos] = (int16_t)((c & 2) ? midY : uly);
            W.bly(nb)[pos] = (int16_t)((c & 2) ? bly : midY);
            W.cnt(nb)[pos] = cc[c];
            W.ncode(nb)[pos] = hist_child_code(code, c);
            if (cc[c] > 1) W.V[e0 + ev++] = (uint16_t)pos;
            m++;
          }
        }
      } else {
        const int pos = K + exKept;
        W.ulx(nb)[pos] = (int16_t)ulx;
        W.urx(nb)[pos] = (int16_t)urx;
        W.uly(nb)[pos] = (int16_t)uly;
        W.bly(nb)[pos] = (int16_t)bly;
        W.cnt(nb)[pos] = cnt;
        W.ncode(nb)[pos] = (uint16_t)code;
      }
    }
    g.sync();
    cur = nb;
    *nV_out = E;
    return K + kept;
  }
  // counts of the four children of every node that splits
  for (int i = g.tid; i < nL; i += g.nthreads) {
    const bool d = W.cnt(b)[i] > 1;
    W.divided[i] = d;
    int k = 0, e = 0;
    if (d) {
      const int *h = hist_children(W, W.ncode(b)[i], S);
      for (int c = 0; c < 4; c++) {
        const int cc = h[c];
        W.childcnt[4 * i + c] = cc;
        k += cc > 0;
        e += cc > 1;
      }
    }
    W.scanA[i] = k | (e << 16);
    W.scanB[i] = d ? 0 : 1;
  }
  g.sync();
  int kept = 0;
  const int total = g.exclusive_scan2(W.scanA, W.scanB, nL, &kept);
  const int K = total & 0xFFFF, E = total >> 16;
  const int newL = K + kept;
  for (int i = g.tid; i < nL; i += g.nthreads) {
    const int ulx = W.ulx(b)[i], uly = W.uly(b)[i], urx = W.urx(b)[i], bly = W.bly(b)[i];
    const int code = W.ncode(b)[i];
    if (W.divided[i]) {
      const int q0 = W.scanA[i] & 0xFFFF, e0 = W.scanA[i] >> 16;
      const int midX = ulx + ((urx - ulx + 1) >> 1), midY = uly + ((bly - uly + 1) >> 1);
      int m = 0, ev = 0;
      for (int c = 0; c < 4; c++) {
        const int cc = W.childcnt[4 * i + c];
        if (cc > 0) {
          const int pos = K - 1 - (q0 + m);  // push_front => reversed creation order
          W.ulx(nb)[pos] = (int16_t)((c & 1) ? midX : ulx);
          W.urx(nb)[pos] = (int16_t)((c & 1) ? urx : midX);
          W.uly(nb)[pos] = (int16_t)((c & 2) ? midY : uly);
          W.bly(nb)[pos] = (int16_t)((c & 2) ? bly : midY);
          W.cnt(nb)[pos] = cc;
          W.ncode(nb)[pos] = hist_child_code(code, c);
          if (cc > 1) W.V[e0 + ev++] = (uint16_t)pos;
          m++;
        }
      }
    } else {
      const int pos = K + W.scanB[i];
      W.ulx(nb)[pos] = (int16_t)ulx;
      W.urx(nb)[pos] = (int16_t)urx;
      W.uly(nb)[pos] = (int16_t)uly;
      W.bly(nb)[pos] = (int16_t)bly;
      W.cnt(nb)[pos] = W.cnt(b)[i];
      W.ncode(nb)[pos] = (uint16_t)code;
    }
  }
  g.sync();
  cur = nb;
  *nV_out = E;
  return newL;
}

// One CAREFUL pass over proc[0..nProc) (:696-757: the nodes in sorted order, `break` at the first size >= N) in histogram
// mode: run_pass with the child counts read from the histogram and without its two point sweeps.
template <class G>
VSG_OCT_HD int hist_careful_pass(G &g, const Params &P, Work &W, int &cur, int nL, int nProc, int D, int *nV_out) {
  const int b = cur, nb = cur ^ 1, S = hist_off(D + 1);
  for (int i = g.tid; i < nL; i += g.nthreads) W.divided[i] = 0;
  if (g.tid == 0) W.ctrl[2] = nProc;
  g.sync();
  for (int t = g.tid; t < nProc; t += g.nthreads) {
    const int n = W.proc[t];
    W.divided[n] = 1;
    const int *h = hist_children(W, W.ncode(b)[n], S);
    int k = 0, e = 0;
    for (int c = 0; c < 4; c++) {
      const int cc = h[c];
      W.childcnt[4 * n + c] = cc;
      k += cc > 0;
      e += cc > 1;
    }
    W.scanA[t] = k | (e << 16);
  }
  g.sync();
  const int total = g.exclusive_scan(W.scanA, nProc);
  // `if ((int)lNodes.size() >= N) break;` after each split (:753)
  for (int t = g.tid; t < nProc; t += g.nthreads) {
    const int n = W.proc[t];
    int k = 0;
    for (int c = 0; c < 4; c++) k += W.childcnt[4 * n + c] > 0;
    const int sizeAfter = nL + (W.scanA[t] & 0xFFFF) + k - (t + 1);
    if (sizeAfter >= P.N) g.atomic_min(&W.ctrl[2], t + 1);
  }
  g.sync();
  const int nEff = W.ctrl[2];
  g.sync();
  for (int t = nEff + g.tid; t < nProc; t += g.nthreads) W.divided[W.proc[t]] = 0;
  const int packed = nEff < nProc ? W.scanA[nEff] : total;
  const int K = packed & 0xFFFF, E = packed >> 16;
  g.sync();
  for (int i = g.tid; i < nL; i += g.nthreads) W.scanB[i] = W.divided[i] ? 0 : 1;
  g.sync();
  const int kept = g.exclusive_scan(W.scanB, nL);
  const int newL = K + kept;
  for (int t = g.tid; t < nEff; t += g.nthreads) {
    const int n = W.proc[t];
    const int q0 = W.scanA[t] & 0xFFFF, e0 = W.scanA[t] >> 16;
    const int ulx = W.ulx(b)[n], uly = W.uly(b)[n], urx = W.urx(b)[n], bly = W.bly(b)[n];
    const int midX = ulx + ((urx - ulx + 1) >> 1), midY = uly + ((bly - uly + 1) >> 1);
    const int code = W.ncode(b)[n];
    int m = 0, ev = 0;
    for (int c = 0; c < 4; c++) {
      const int cc = W.childcnt[4 * n + c];
      if (cc > 0) {
        const int pos = K - 1 - (q0 + m);  // push_front => reversed creation order
        W.ulx(nb)[pos] = (int16_t)((c & 1) ? midX : ulx);
        W.urx(nb)[pos] = (int16_t)((c & 1) ? urx : midX);
        W.uly(nb)[pos] = (int16_t)((c & 2) ? midY : uly);
        W.bly(nb)[pos] = (int16_t)((c & 2) ? bly : midY);
        W.cnt(nb)[pos] = cc;
        W.ncode(nb)[pos] = hist_child_code(code, c);
        if (cc > 1) W.V[e0 + ev++] = (uint16_t)pos;
        m++;
      }
    }
  }
  for (int i = g.tid; i < nL; i += g.nthreads) {
    if (!W.divided[i]) {
      const int pos = K + W.scanB[i];
      W.ulx(nb)[pos] = W.ulx(b)[i];
      W.urx(nb)[pos] = W.urx(b)[i];
      W.uly(nb)[pos] = W.uly(b)[i];
      W.bly(nb)[pos] = W.bly(b)[i];
      W.cnt(nb)[pos] = W.cnt(b)[i];
      W.ncode(nb)[pos] = W.ncode(b)[i];
    }
  }
  g.sync();
  cur = nb;
  *nV_out = E;
  return newL;
}

// DistributeOctTree.  cand[0..npts): packed candidates in ANY order.  node_of: npts uint16 scratch.
// sel_out: receives the chosen candidate of every final node, in the reference's list order.
// Returns the number of selected keypoints.  Must be called by every thread of the group.
template <class G, class PT>
VSG_OCT_HD int distribute_pts(G &g, const Params &P, PT &pts, int npts, Work &W, uint32_t *sel_out) {
  int cur = 0;
  pts.load(g, npts);
  // initial nodes (:575-586)
  int nL;
  if (P.nIni == 1) {
    // one initial node (width / height rounds to 1: every 4:3 and 16:10 level): it holds every point -- no counting sweep
    // with npts atomics on ONE LDS word, no compaction, no relabelling (5.7 of the one-frame octree's 81 k cycles)
    if (g.tid == 0) {
      W.ulx(1)[0] = (int16_t)P.iniUL[0];
      W.urx(1)[0] = (int16_t)P.iniUL[1];
      W.uly(1)[0] = 0;
      W.bly(1)[0] = (int16_t)P.height;
      W.cnt(1)[0] = npts;
    }
    pts.init_each(g, npts, [&](uint32_t, int &n) { n = 0; });
    cur = 1;
    nL = npts > 0 ? 1 : 0;
    g.sync();
  } else if (P.nIni <= 4) {
    // two to four initial nodes (16:9, 752x480, ...): thresholds and corners in registers up front -- read inside the sweep they
    // were re-loaded from the level's table for every point (its stores may alias them) --, the counts summed per wave
    // (256 same-word LDS atomics before), every thread places the nodes from the four counts, and the relabelling sweep only
    // runs when an initial node is empty (one-frame 752x480: 13.1 k -> 4.0 k cycles, profiles/r06_i_*)
    const int nI = P.nIni;
    const int t1 = P.iniThresh[1], t2 = nI > 2 ? P.iniThresh[2] : 0x7FFFFFFF, t3 = nI > 3 ? P.iniThresh[3] : 0x7FFFFFFF;
    const int u0 = P.iniUL[0], u1 = P.iniUL[1], u2 = P.iniUL[2], u3 = nI > 2 ? P.iniUL[3] : 0, u4 = nI > 3 ? P.iniUL[4] : 0;
    for (int i = g.tid; i < 4; i += g.nthreads) W.cnt(0)[i] = 0;
    g.sync();
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    pts.init_each(g, npts, [&](uint32_t c, int &n) {
      const int x = VSG_CAND_X(c);
      const int idx = (x >= t1) + (x >= t2) + (x >= t3);
      n = idx;
      c0 += idx == 0, c1 += idx == 1, c2 += idx == 2, c3 += idx == 3;
    });
    g.add4(&W.cnt(0)[0], c0, c1, c2, c3);
    g.sync();
    // erase empty initial nodes, keep order (:597-608)
    const int k0 = W.cnt(0)[0], k1 = W.cnt(0)[1], k2 = W.cnt(0)[2], k3 = W.cnt(0)[3];
    const int p1 = k0 > 0, p2 = p1 + (k1 > 0), p3 = p2 + (k2 > 0);
    nL = p3 + (k3 > 0);
    for (int i = g.tid; i < nI; i += g.nthreads) {
      const int ki = i == 0 ? k0 : i == 1 ? k1 : i == 2 ? k2 : k3;
      if (ki > 0) {
        const int pos = i == 0 ? 0 : i == 1 ? p1 : i == 2 ? p2 : p3;
        W.ulx(1)[pos] = (int16_t)(i == 0 ? u0 : i == 1 ? u1 : i == 2 ? u2 : u3);
        W.urx(1)[pos] = (int16_t)(i == 0 ? u1 : i == 1 ? u2 : i == 2 ? u3 : u4);
        W.uly(1)[pos] = 0;
        W.bly(1)[pos] = (int16_t)P.height;
        W.cnt(1)[pos] = ki;
      }
    }
    if (nL != nI)  // group-uniform
      pts.for_each(g, npts, [&](uint32_t, int &n) { n = n == 0 ? 0 : n == 1 ? p1 : n == 2 ? p2 : p3; });
    cur = 1;
    g.sync();
  } else {
    for (int i = g.tid; i < P.nIni; i += g.nthreads) {
      W.ulx(0)[i] = (int16_t)P.iniUL[i];
      W.urx(0)[i] = (int16_t)P.iniUL[i + 1];
      W.uly(0)[i] = 0;
      W.bly(0)[i] = (int16_t)P.height;
      W.cnt(0)[i] = 0;
    }
    g.sync();
    // vpIniNodes[kp.pt.x / hX] (:589-593), five and more initial nodes (no such geometry among the reference's settings)
    pts.init_each(g, npts, [&](uint32_t c, int &n) {
      const int x = VSG_CAND_X(c);
      int idx = 0;
      for (int i = 1; i < P.nIni; i++) idx += (x >= P.iniThresh[i]);
      n = idx;
      g.atomic_add(&W.cnt(0)[idx], 1);
    });
    g.sync();
    // erase empty initial nodes, keep order (:597-608)
    if (g.tid == 0) {
      int pos = 0;
      for (int i = 0; i < P.nIni; i++) {
        if (W.cnt(0)[i] > 0) {
          W.ulx(1)[pos] = W.ulx(0)[i];
          W.urx(1)[pos] = W.urx(0)[i];
          W.uly(1)[pos] = W.uly(0)[i];
          W.bly(1)[pos] = W.bly(0)[i];
          W.cnt(1)[pos] = W.cnt(0)[i];
          W.keeppos[i] = (uint16_t)pos;
          pos++;
        }
      }
      W.ctrl[0] = pos;
    }
    g.sync();
    pts.for_each(g, npts, [&](uint32_t, int &n) { n = W.keeppos[n]; });
    cur = 1;
    nL = W.ctrl[0];
    g.sync();
  }

  bool finish = false;
  // histogram mode for the first hist_D passes (hist_* above); hist_D = 0: the label-based passes
  int hist_D = nL >= 1 && nL <= kFuseRoots ? hist_depth(W, nL) : 0, hist_pass = 0;
#ifdef VSG_OCT_NO_FUSE
  hist_D = 0;  // A/B builds (tools/build_variant.sh): the regular passes from the start
#endif
  if (hist_D) hist_setup(g, W, cur, nL, pts, npts, hist_D);
  while (!finish) {  // (:617)
    const int prevSize = nL;
    int nV = 0;
    if (hist_D && hist_pass >= hist_D) {
      hist_leave(g, W, cur, nL, pts, npts, hist_D);
      hist_D = 0;
    }
    if (hist_D) {
      nL = hist_main_pass(g, W, cur, nL, hist_D, &nV);
      hist_pass++;
    } else {
      nL = run_main_pass(g, P, W, cur, nL, pts, npts, &nV);
    }
    if (nL >= P.N || nL == prevSize) {  // (:692)
      finish = true;
    } else if (nL + nV * 3 > P.N) {  // (:696)
      while (!finish) {
        const int prev2 = nL;
        introsort::item_t *sortbuf = (introsort::item_t *)W.childcnt;
        for (int t = g.tid; t < nV; t += g.nthreads) {
          const int n = W.V[t];
          // compareNodes: (size, UL.x) ascending (:539-560)
          const uint32_t key = ((uint32_t)W.cnt(cur)[n] << 13) | (uint32_t)(uint16_t)W.ulx(cur)[n];
          sortbuf[t] = ((introsort::item_t)key << 32) | (uint32_t)n;
        }
        g.sync();
        // (Round 6 tried a counting sort for the case that all keys differ -- the sorted order is then unique, whatever std::sort
        // does with ties: on a photograph's level 0 two of the ~200 (size, UL.x) keys are equal almost always, the attempt cost
        // 22 k cycles and the replay below ran anyway; removed.  profiles/r06_h_octree_large_levels.txt)
        // std::sort (:707) = serial quicksort partitioning + a stable sort of what it leaves (vsg_introsort.h);
        // the stable part is a rank computation spread over the group, written straight into the back-to-front
        // processing order of (:708).
        if (g.sort_to_proc(sortbuf, nV, W.proc)) {
          g.sync();
        } else {
        g.sort_partition_phase(sortbuf, nV, (uint16_t *)W.scanA, (uint16_t *)W.scanB);
        g.sync();
        for (int t = g.tid; t < nV; t += g.nthreads) {
          const introsort::item_t it = sortbuf[t];
          const uint32_t key = (uint32_t)(it >> 32);
          // The partition phase leaves consecutive runs of <= 16 items, every run >= the ones before it (a
          // heap-sorted run is already in order), so only the 15 neighbours on either side can change an item's
          // stable rank: everything further left counts, nothing further right does.
          // a fixed trip count with clamped indices: the 30 loads are independent and issue back to back (with run-time
          // bounds every iteration waited for its own load: 5.1 k cycles for 64 items on the one-frame path)
          const int lo = t > 15 ? t - 15 : 0, hi = t + 15 < nV - 1 ? t + 15 : nV - 1;
          int rank = lo;
          VSG_OCT_UNROLL
          for (int d = -15; d <= 15; d++) {
            if (d == 0) continue;
            const int j = t + d, jc = j < lo ? lo : j > hi ? hi : j;
            const uint32_t kj = (uint32_t)(sortbuf[jc] >> 32);
            rank += (j >= lo) & (j <= hi) & ((kj < key) | ((kj == key) & (d < 0)));
          }
          W.proc[nV - 1 - rank] = (uint16_t)(uint32_t)it;
        }
        g.sync();
        }
        int nV2 = 0;
        if (hist_D && hist_pass >= hist_D) {
          hist_leave(g, W, cur, nL, pts, npts, hist_D);
          hist_D = 0;
        }
        if (hist_D) {
          nL = hist_careful_pass(g, P, W, cur, nL, nV, hist_D, &nV2);
          hist_pass++;
        } else {
          nL = run_pass(g, P, W, cur, nL, nV, true, pts, npts, &nV2);
        }
        nV = nV2;
        if (nL >= P.N || nL == prev2) finish = true;  // (:757)
      }
    }
  }
  if (hist_D) hist_leave(g, W, cur, nL, pts, npts, hist_D);  // the points' labels: list positions of the final nodes

  // retain the best point of every node, first maximum wins (:763-782)
  // ONE sweep: the key (response, then the EARLIER candidate) rides above the candidate word in a 64-bit LDS maximum, so the
  // winner's word is what the maximum leaves behind (two sweeps and a barrier between them before: key, then compare)
  uint64_t *bestkey = (uint64_t *)W.childcnt;  // 8 of the 16 bytes per node
  for (int i = g.tid; i < nL; i += g.nthreads) bestkey[i] = 0;
  g.sync();
  const float inv_w = 1.0f / (float)P.wCell, inv_h = 1.0f / (float)P.hCell;
  pts.for_each(g, npts, [&](uint32_t c, int &n) {
    const uint32_t key = ((uint32_t)VSG_CAND_R(c) << 24) | (0xFFFFFFu - cand_rank(P, VSG_CAND_X(c), VSG_CAND_Y(c), inv_w, inv_h));
    g.atomic_max64(&bestkey[n], ((uint64_t)key << 32) | c);
  });
  g.sync();
  for (int i = g.tid; i < nL; i += g.nthreads) sel_out[i] = (uint32_t)bestkey[i];
  g.sync();
  return nL;
}

// Points in memory: works for any npts (the host tests and the device fallback).
template <int U = 1, class G>
VSG_OCT_HD int distribute(G &g, const Params &P, const uint32_t *cand, int npts, uint16_t *node_of, Work &W,
                          uint32_t *sel_out) {
  MemPtsT<U> pts{cand, node_of};
  return distribute_pts(g, P, pts, npts, W, sel_out);
}

// Points in registers: npts <= K * g.nthreads.
template <int K, class G>
VSG_OCT_HD int distribute_reg(G &g, const Params &P, const uint32_t *cand, int npts, Work &W, uint32_t *sel_out) {
  RegPts<K> pts;
  pts.src.cand = cand;
  return distribute_pts(g, P, pts, npts, W, sel_out);
}
template <int K, class Src, class G>
VSG_OCT_HD int distribute_reg_src(G &g, const Params &P, const Src &src, int npts, Work &W, uint32_t *sel_out) {
  RegPts<K, Src> pts;
  pts.src = src;
  return distribute_pts(g, P, pts, npts, W, sel_out);
}

}  // namespace octree
}  // namespace vsg
