// vio_bow.hip — the bag-of-words query of the loop-closure producer: DBoW2 as the app uses it from
// LoopClosure::startLoopClosure (VINS_ios/loop/loop_closure.cpp:20-36) -> TemplatedLoopDetector::detectLoop
// (loop/TemplatedLoopDetector.h:668-700):
//   * vocabulary in the app's binary layout      loop/VocabularyBinary.hpp:17-50, TemplatedVocabulary::loadBin
//                                                ThirdParty/DBoW/TemplatedVocabulary.h:1505-1554
//   * descriptor -> word                         TemplatedVocabulary::transform(feature, id, weight) :1213-1253: descend the
//                                                k-ary tree, at every level the child with the smallest Hamming distance
//                                                (FBrief::distance, ThirdParty/DBoW/FBrief.cpp:53-57), first one on ties
//   * descriptors of a keyframe -> BowVector     transform(features, v) :1061-1117 (TF_IDF / TF accumulate the word's weight
//                                                per occurrence, IDF / BINARY take it once) + BowVector::normalize(L1)
//                                                (ThirdParty/DBoW/BowVector.cpp:57-80)
//   * database add / query                       TemplatedDatabase::add :439-470, queryL1 :651-720 (L1 score of the query
//                                                against every entry below max_id, best max_results)
// Kernels, many keyframes / queries per launch:
//   bow_lookup_kernel   16 lanes per descriptor: every lane takes children (256-bit XOR + popcount), the key
//                       distance << 20 | child position is min-reduced inside the 16-lane row; one tree level per
//                       dependent global fetch, thousands of descriptors in flight
//   bow_vector_kernel   one workgroup per keyframe: bitonic sort of the word ids in LDS, run heads -> unique words, the
//                       value of a word by the reference's own sequence of additions, L1 norm summed in ascending word
//                       order by one lane (the order of std::map iteration: bit-identical values)
//   bow_insert_kernel / bow_candidates_kernel / bow_score_kernel   the database: an inverted file (sorted postings) finds the
//                       entries that share a word with a query, one wave per such entry sums |q - d| - |q| - |d| over the
//                       common words in ascending word order (= the order in which queryL1 meets them: bit-identical sums)
//                       from the entry's BowVector in the direct file; see the comment above bow_insert_kernel
// The final sort / cut / scaling of queryL1 (a few hundred candidates) runs on the host inside the ABI call.
// Scoring other than L1_NORM is refused (the app's vocabulary is TF_IDF / L1_NORM, DBoW2's defaults).
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "vio_amd.h"
#include "vio_device.h"

namespace {

#define HIP_OK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "vio_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return VIO_ENODEV;                                                                   \
    }                                                                                      \
  } while (0)

constexpr int kMaxBowFeatures = 8192;  // descriptors per keyframe the BowVector kernel sorts in LDS

template <class T>
struct Buf {
  T *p = nullptr;
  size_t n = 0;
  int ensure(size_t count) {
    if (count <= n && p) return VIO_OK;
    if (p) (void)hipFree(p);
    p = nullptr, n = 0;
    if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return VIO_ENOMEM;
    n = count;
    return VIO_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr, n = 0;
  }
};

// 16 lanes per descriptor, 4 descriptors per wave
__global__ __launch_bounds__(256) void bow_lookup_kernel(const unsigned long long *node_desc, const double *node_weight, const int *node_word,
                                                          const int *child_off, const int *child, const unsigned long long *desc,
                                                          int n, int depth, int *word, double *weight) {
  const int g = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 4), l = threadIdx.x & 15;
  const bool have = g < n;
  const unsigned long long *f = desc + 4 * (size_t)(have ? g : 0);
  const unsigned long long f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3];
  int node = 0;
  for (int level = 0; level < depth; level++) {  // (depth = the tree's height, checked at load: the walk always ends)
    const int c0 = child_off[node], nc = child_off[node + 1] - c0;
    if (nc <= 0) break;  // leaf (every lane of the row walks the same path)
    unsigned best = 0xffffffffu;
    for (int c = l; c < nc; c += 16) {
      const unsigned long long *d = node_desc + 4 * (size_t)child[c0 + c];
      const unsigned dist = __popcll(f0 ^ d[0]) + __popcll(f1 ^ d[1]) + __popcll(f2 ^ d[2]) + __popcll(f3 ^ d[3]);
      const unsigned key = (dist << 20) | (unsigned)c;  // smallest distance, first child on ties (`d < best_d`)
      best = key < best ? key : best;
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
      const unsigned o = (unsigned)__shfl_xor((int)best, m, 16);
      best = o < best ? o : best;
    }
    node = child[c0 + (int)(best & 0xfffffu)];
  }
  if (have && l == 0) word[g] = node_word[node], weight[g] = node_weight[node];
}

// one workgroup per keyframe: its descriptors' (word, weight) -> ascending unique words with their L1-normalised values
__global__ __launch_bounds__(256) void bow_vector_kernel(const int *kf_off, const int *word, const double *weight, const double *word_weight,
                                                          int accumulate, int *bow_count, int *bow_word, double *bow_value, int stride) {
  __shared__ int key[kMaxBowFeatures];
  __shared__ int scan[257];
  __shared__ double norm_s;
  const int kf = blockIdx.x, o = kf_off[kf], n = kf_off[kf + 1] - o, tid = threadIdx.x, nt = blockDim.x;
  int np2 = nt;
  while (np2 < n) np2 <<= 1;
  for (int i = tid; i < np2; i += nt) key[i] = (i < n && weight[o + i] > 0.0) ? word[o + i] : 0x7fffffff;  // stopped words drop out
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < np2; i += nt) {
        const int p = i ^ j;
        if (p > i) {
          const int a = key[i], b = key[p];
          if (((i & k) == 0) == (a > b)) key[i] = b, key[p] = a;
        }
      }
      __syncthreads();
    }
  // heads of runs of equal words: every lane owns a contiguous piece of the sorted list
  const int chunk = np2 / nt, i0 = tid * chunk;
  int cnt = 0;
  for (int i = i0; i < i0 + chunk; i++) cnt += key[i] != 0x7fffffff && (i == 0 || key[i] != key[i - 1]);
  scan[tid + 1] = cnt;
  if (tid == 0) scan[0] = 0;
  __syncthreads();
  if (tid == 0)
    for (int t = 0; t < nt; t++) scan[t + 1] += scan[t];
  __syncthreads();
  const int u = scan[nt];
  int *ow = bow_word + (size_t)kf * stride;
  double *ov = bow_value + (size_t)kf * stride;
  if (u > stride) {  // caller's capacity too small: report the count, write nothing
    if (tid == 0) bow_count[kf] = -u;
    return;
  }
  int q = scan[tid];
  for (int i = i0; i < i0 + chunk; i++) {
    const int wd = key[i];
    if (wd == 0x7fffffff || (i > 0 && wd == key[i - 1])) continue;
    int i1 = i + 1;
    while (i1 < np2 && key[i1] == wd) i1++;
    // BowVector::addWeight adds the word's weight once per occurrence (in that order: w + w + ...), addIfNotExist keeps it once
    const double wv = word_weight[wd];
    double sv = wv;
    if (accumulate)
      for (int r = i + 1; r < i1; r++) sv += wv;
    ow[q] = wd, ov[q] = sv;
    q++;
  }
  __syncthreads();
  if (tid == 0) {
    double norm = 0.0;
    for (int r = 0; r < u; r++) norm += fabs(ov[r]);  // ascending word order, like BowVector::normalize over the std::map
    norm_s = norm;
    bow_count[kf] = u;
  }
  __syncthreads();
  const double norm = norm_s;
  if (norm > 0.0)
    for (int r = tid; r < u; r += nt) ov[r] /= norm;
}

// ---- the database: inverted file + direct file ----------------------------------------------------------------------
// TemplatedDatabase keeps, per word, the list of (entry, value) pairs that hold it (m_ifile, TemplatedDatabase.h:439-470)
// and queryL1 walks the lists of the query's words (:651-720): work proportional to the postings touched, not to the
// database. Here the inverted file is ONE sorted array of postings, key = word << 32 | entry: an entry arrives with
// ascending unique words and an id above every id in the file, so its postings go to the END of their words' runs and an
// add is one out-of-place merge (a streaming pass at HBM speed; lists that grow in place would need an allocator on the
// device). A query is two kernels:
//   bow_candidates_kernel   one wave per query word: the word's run by two binary searches (cut at max_id), lanes over the
//                           run; the first posting that reaches an entry puts it on the query's candidate list
//   bow_score_kernel        one wave per candidate: lanes over the entry's words in the direct file (ascending), each looks
//                           its word up in the query (staged in LDS); the matched terms are added in ascending word order
//                           (lane order, chunk after chunk) = the order in which queryL1 meets them: bit-identical sums
// so entries that share no word with the query (most of a session's database under a 10^6-word vocabulary) cost nothing.
__global__ __launch_bounds__(256) void bow_insert_kernel(const unsigned long long *old_key, int n_old, const int *new_word, int n_new, int entry,
                                                          unsigned long long *out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < n_old) {
    const unsigned long long key = old_key[t];
    const int wk = (int)(key >> 32);
    int lo = 0, hi = n_new;  // new words below this posting's word
    while (lo < hi) {
      const int m = (lo + hi) >> 1;
      if (new_word[m] < wk) lo = m + 1;
      else hi = m;
    }
    out[t + lo] = key;
  } else if (t < n_old + n_new) {
    const int j = t - n_old, w = new_word[j];
    const unsigned long long bound = (unsigned long long)(unsigned)(w + 1) << 32;
    int lo = 0, hi = n_old;  // old postings of words <= w
    while (lo < hi) {
      const int m = (lo + hi) >> 1;
      if (old_key[m] < bound) lo = m + 1;
      else hi = m;
    }
    out[lo + j] = (unsigned long long)(unsigned)w << 32 | (unsigned)entry;
  }
}

__device__ __forceinline__ int posting_lower_bound(const unsigned long long *key, int n, unsigned long long k) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int m = (lo + hi) >> 1;
    if (key[m] < k) lo = m + 1;
    else hi = m;
  }
  return lo;
}

// flag [n_queries][n_entries] (zeroed), n_cand [n_queries] (zeroed), cand [n_queries][n_entries]
__global__ __launch_bounds__(256) void bow_candidates_kernel(const unsigned long long *key, int n_post, int n_entries, const int *q_count,
                                                              const int *q_word, int q_stride, const int *max_id, int *flag, int *n_cand,
                                                              int *cand) {
  const int q = blockIdx.y, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), n_waves = gridDim.x * 4;
  const int ni = q_count[q], mid = max_id[q];
  for (int i = wave; i < ni; i += n_waves) {
    const int w = q_word[(size_t)q * q_stride + i];
    const unsigned long long base = (unsigned long long)(unsigned)w << 32;
    const unsigned long long top = mid == -1 ? (unsigned long long)(unsigned)(w + 1) << 32 : base | (unsigned)max(mid, 0);  // entries below max_id
    const int lo = posting_lower_bound(key, n_post, base), hi = posting_lower_bound(key, n_post, top);
    for (int p = lo + lane; p < hi; p += 64) {
      const int e = (int)(unsigned)(key[p] & 0xffffffffull);
      if (atomicExch(&flag[(size_t)q * n_entries + e], 1) == 0) cand[(size_t)q * n_entries + atomicAdd(&n_cand[q], 1)] = e;
    }
  }
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
  return __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));
}

constexpr int kBowQueryLds = 8192;  // query words staged in LDS (a BowVector has at most kMaxBowFeatures words)
// wave c of query q: candidate cand[q][c] -> cscore[q][c] = sum over the common words, ascending, of |q - d| - |q| - |d|
__global__ __launch_bounds__(256) void bow_score_kernel(const int *db_off, const int *db_word, const double *db_value, int n_entries,
                                                         const int *q_count, const int *q_word, const double *q_value, int q_stride,
                                                         const int *n_cand, const int *cand, double *cscore) {
  __shared__ int qw_s[kBowQueryLds];
  const int q = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (tid >> 6)), n_waves = gridDim.x * 4;
  const int ni = q_count[q];
  const int *qw = q_word + (size_t)q * q_stride;
  const double *qv = q_value + (size_t)q * q_stride;
  const bool staged = ni <= kBowQueryLds;
  if (staged)
    for (int i = tid; i < ni; i += 256) qw_s[i] = qw[i];
  __syncthreads();
  const int nc = n_cand[q];
  for (int c = wave; c < nc; c += n_waves) {
    const int e = cand[(size_t)q * n_entries + c];
    const int j0 = db_off[e], j1 = db_off[e + 1];
    double s = 0.0;
    for (int jb = j0; jb < j1; jb += 64) {
      const int j = jb + lane;
      bool hit = false;
      double t = 0.0;
      if (j < j1) {
        const int b = db_word[j];
        int lo = 0, hi = ni;
        if (staged) {
          while (lo < hi) {
            const int m = (lo + hi) >> 1;
            if (qw_s[m] < b) lo = m + 1;
            else hi = m;
          }
          hit = lo < ni && qw_s[lo] == b;
        } else {
          while (lo < hi) {
            const int m = (lo + hi) >> 1;
            if (qw[m] < b) lo = m + 1;
            else hi = m;
          }
          hit = lo < ni && qw[lo] == b;
        }
        if (hit) {
          const double qq = qv[lo], dd = db_value[j];
          t = fabs(qq - dd) - fabs(qq) - fabs(dd);
        }
      }
      unsigned long long m = __ballot(hit);
      while (m) {  // ascending lanes = ascending words
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        s += readlane_f64(t, l);
      }
    }
    if (lane == 0) cscore[(size_t)q * n_entries + c] = s;
  }
}

}  // namespace

struct vio_vocabulary {
  int device = -1;
  int32_t k = 0, L = 0, scoring = 0, weighting = 0, n_nodes = 0, n_words = 0;  // n_nodes incl. the root
  int height = 0;  // levels below the root: bounds the descent of bow_lookup_kernel
  hipStream_t stream = nullptr;
  Buf<unsigned long long> d_desc;
  Buf<double> d_weight, d_wweight;  // per node; per word
  Buf<int> d_word, d_child_off, d_child;
  // transform scratch
  Buf<unsigned long long> t_desc;
  Buf<int> t_word, t_off, t_bcount, t_bword;
  Buf<double> t_weight, t_bvalue;
};

// (A database takes the word count from its vocabulary at create and owns its stream: it keeps working, and can be
// destroyed, after the vocabulary is gone, and two threads may use a vocabulary and its database side by side.)
struct vio_bow_database {
  int32_t voc_words = 0;
  hipStream_t stream = nullptr;
  int device = -1;
  int max_entries = 0, n_entries = 0;
  size_t max_words = 0, n_words = 0;
  std::vector<int> h_off;  // [n_entries + 1]
  Buf<int> d_off, d_word, q_count, q_word, q_max;  // d_off / d_word / d_value: the direct file (every entry's BowVector)
  Buf<double> d_value, q_value;
  Buf<unsigned long long> inv[2];  // the inverted file: postings sorted by (word, entry); inv[cur] is the live copy
  int cur = 0;
  Buf<int> flag, n_cand, cand;  // query scratch: [n_queries][n_entries] marks, [n_queries], [n_queries][n_entries]
  Buf<double> cscore;           // [n_queries][n_entries] raw score of candidate c
};

extern "C" {

int vio_vocabulary_create(const void *blob, size_t bytes, vio_vocabulary_t **out) {
  if (!blob || !out) return VIO_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    fprintf(stderr, "vio_amd: no HIP device visible; the bag-of-words query has no CPU fallback\n");
    return VIO_ENODEV;
  }
  const unsigned char *p = (const unsigned char *)blob;
  if (bytes < 24) return VIO_EINVAL;
  int32_t hdr[6];
  memcpy(hdr, p, 24);  // Vocabulary::staticDataSize(): k, L, scoringType, weightingType, nNodes, nWords
  const int32_t nNodes = hdr[4], nWords = hdr[5];
  if (nNodes < 1 || nWords < 1 || bytes < 24 + (size_t)nNodes * 48 + (size_t)nWords * 8) return VIO_EINVAL;
  if (hdr[2] != 0) return VIO_EINVAL;                  // scoring: only L1_NORM (ScoringType 0) is implemented
  if (hdr[3] < 0 || hdr[3] > 3) return VIO_EINVAL;     // weighting: TF_IDF, TF, IDF, BINARY
  try {
    const size_t N = (size_t)nNodes + 1;
    std::vector<unsigned long long> desc(4 * N, 0);
    std::vector<double> weight(N, 0.0);
    std::vector<int> word(N, 0), parent(N, 0), cnt(N + 1, 0), order(nNodes);
    std::vector<char> seen(N, 0);
    const unsigned char *q = p + 24;
    for (int i = 0; i < nNodes; i++, q += 48) {  // struct Node { int32 nodeId, parentId; double weight; uint64 descriptor[4]; }
      int32_t nid, pid;
      memcpy(&nid, q, 4), memcpy(&pid, q + 4, 4);
      if (nid < 1 || nid > nNodes || pid < 0 || pid > nNodes || pid == nid) return VIO_EINVAL;
      if (seen[nid]) return VIO_EINVAL;  // a node id twice: two records would share one slot of their parent's child list
      seen[nid] = 1;
      memcpy(&weight[nid], q + 8, 8);
      memcpy(&desc[4 * (size_t)nid], q + 16, 32);
      parent[nid] = pid, order[i] = nid, cnt[pid + 1]++;
    }
    for (size_t i = 0; i < N; i++) cnt[i + 1] += cnt[i];  // children of a node, in FILE order (loadBin pushes them back in that order)
    std::vector<int> fill(cnt.begin(), cnt.end() - 1), child(nNodes);
    std::vector<double> wweight(nWords, 0.0);
    for (int i = 0; i < nNodes; i++) child[fill[parent[order[i]]]++] = order[i];
    for (size_t i = 0; i < N; i++)
      if (fill[i] != cnt[i + 1]) return VIO_EINVAL;
    // the records must form ONE tree under node 0: every node reaches the root (a cycle of parent links never does), and
    // the height of the tree bounds the descent of the lookup kernel
    int height = 0;
    {
      std::vector<int> dep(N, -1);
      dep[0] = 0;
      std::vector<int> path;
      for (int i = 1; i <= nNodes; i++) {
        path.clear();
        int a = i;
        while (dep[a] < 0) {
          path.push_back(a);
          if ((int)path.size() > nNodes) return VIO_EINVAL;
          a = parent[a];
        }
        int d = dep[a];
        for (size_t k = path.size(); k-- > 0;) dep[path[k]] = ++d;
        height = std::max(height, dep[i]);
      }
    }
    for (int i = 0; i < nWords; i++, q += 8) {  // struct Word { int32 nodeId, wordId; }
      int32_t nid, wid;
      memcpy(&nid, q, 4), memcpy(&wid, q + 4, 4);
      if (nid < 1 || nid > nNodes || wid < 0 || wid >= nWords) return VIO_EINVAL;
      word[nid] = wid, wweight[wid] = weight[nid];
    }
    if (cnt[1] - cnt[0] < 1) return VIO_EINVAL;  // the root has no child: an empty vocabulary
    vio_vocabulary *v = new vio_vocabulary();
    v->device = vio::current_device();
    v->k = hdr[0], v->L = hdr[1], v->scoring = hdr[2], v->weighting = hdr[3], v->n_nodes = (int32_t)N, v->n_words = nWords;
    v->height = height;
    bool ok = hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && v->d_desc.ensure(4 * N) == VIO_OK && v->d_weight.ensure(N) == VIO_OK && v->d_word.ensure(N) == VIO_OK &&
         v->d_child_off.ensure(N + 1) == VIO_OK && v->d_child.ensure(nNodes) == VIO_OK && v->d_wweight.ensure(nWords) == VIO_OK;
    ok = ok && hipMemcpy(v->d_desc.p, desc.data(), 32 * N, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->d_weight.p, weight.data(), 8 * N, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->d_word.p, word.data(), 4 * N, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->d_wweight.p, wweight.data(), 8 * (size_t)nWords, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->d_child_off.p, cnt.data(), 4 * (N + 1), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->d_child.p, child.data(), 4 * (size_t)nNodes, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
      vio_vocabulary_destroy(v);
      return VIO_ENOMEM;
    }
    *out = v;
    return VIO_OK;
  } catch (const std::bad_alloc &) {
    return VIO_ENOMEM;
  }
}

int vio_vocabulary_load(const char *path, vio_vocabulary_t **out) {
  if (!path || !out) return VIO_EINVAL;
  FILE *f = fopen(path, "rb");
  if (!f) return VIO_EINVAL;
  std::vector<unsigned char> blob;
  try {
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 24) {
      fclose(f);
      return VIO_EINVAL;
    }
    blob.resize((size_t)sz);
    const size_t got = fread(blob.data(), 1, blob.size(), f);
    fclose(f);
    if (got != blob.size()) return VIO_EINVAL;
  } catch (const std::bad_alloc &) {
    fclose(f);
    return VIO_ENOMEM;
  }
  return vio_vocabulary_create(blob.data(), blob.size(), out);
}

void vio_vocabulary_destroy(vio_vocabulary_t *v) {
  if (!v) return;
  vio::DeviceScope scope(v->device);
  if (v->stream) (void)hipStreamSynchronize(v->stream), (void)hipStreamDestroy(v->stream);
  v->d_desc.release(), v->d_weight.release(), v->d_wweight.release(), v->d_word.release(), v->d_child_off.release(), v->d_child.release();
  v->t_desc.release(), v->t_word.release(), v->t_off.release(), v->t_bcount.release(), v->t_bword.release();
  v->t_weight.release(), v->t_bvalue.release();
  delete v;
}

int vio_vocabulary_info(const vio_vocabulary_t *v, int32_t info[6]) {
  if (!v || !info) return VIO_EINVAL;
  info[0] = v->k, info[1] = v->L, info[2] = v->scoring, info[3] = v->weighting, info[4] = v->n_nodes - 1, info[5] = v->n_words;
  return VIO_OK;
}

int vio_vocabulary_get_device(const vio_vocabulary_t *v, int32_t *device) {
  if (!v || !device) return VIO_EINVAL;
  *device = v->device;
  return VIO_OK;
}

int vio_vocabulary_transform(vio_vocabulary_t *v, int32_t n_keyframes, const int32_t *n_desc, const uint64_t *desc, int32_t *word_id,
                             double *word_weight, int32_t *bow_count, int32_t *bow_word, double *bow_value, int32_t bow_stride) {
  if (!v || n_keyframes < 1 || !n_desc || !bow_count || !bow_word || !bow_value || bow_stride < 1) return VIO_EINVAL;
  VIO_ON_DEVICE_OF(v);
  try {
    std::vector<int> off(n_keyframes + 1, 0);
    for (int k = 0; k < n_keyframes; k++) {
      if (n_desc[k] < 0) return VIO_EINVAL;
      if (n_desc[k] > kMaxBowFeatures) return VIO_ECAP;
      off[k + 1] = off[k] + n_desc[k];
    }
    const int total = off[n_keyframes];
    if (total > 0 && !desc) return VIO_EINVAL;
    if (v->t_desc.ensure(4 * (size_t)std::max(total, 1)) != VIO_OK || v->t_word.ensure(std::max(total, 1)) != VIO_OK ||
        v->t_weight.ensure(std::max(total, 1)) != VIO_OK || v->t_off.ensure(n_keyframes + 1) != VIO_OK ||
        v->t_bcount.ensure(n_keyframes) != VIO_OK || v->t_bword.ensure((size_t)n_keyframes * bow_stride) != VIO_OK ||
        v->t_bvalue.ensure((size_t)n_keyframes * bow_stride) != VIO_OK)
      return VIO_ENOMEM;
    hipStream_t st = v->stream;
    if (total > 0) HIP_OK(hipMemcpyAsync(v->t_desc.p, desc, 32 * (size_t)total, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(v->t_off.p, off.data(), 4 * (size_t)(n_keyframes + 1), hipMemcpyHostToDevice, st));
    if (total > 0)
      hipLaunchKernelGGL(bow_lookup_kernel, dim3((total * 16 + 255) / 256), dim3(256), 0, st, v->d_desc.p, v->d_weight.p, v->d_word.p,
                         v->d_child_off.p, v->d_child.p, v->t_desc.p, total, v->height + 1, v->t_word.p, v->t_weight.p);
    const int accumulate = v->weighting == 0 || v->weighting == 1;  // TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist
    hipLaunchKernelGGL(bow_vector_kernel, dim3(n_keyframes), dim3(256), 0, st, v->t_off.p, v->t_word.p, v->t_weight.p, v->d_wweight.p, accumulate,
                       v->t_bcount.p, v->t_bword.p, v->t_bvalue.p, bow_stride);
    HIP_OK(hipGetLastError());
    if (total > 0 && word_id) HIP_OK(hipMemcpyAsync(word_id, v->t_word.p, 4 * (size_t)total, hipMemcpyDeviceToHost, st));
    if (total > 0 && word_weight) HIP_OK(hipMemcpyAsync(word_weight, v->t_weight.p, 8 * (size_t)total, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(bow_count, v->t_bcount.p, 4 * (size_t)n_keyframes, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(bow_word, v->t_bword.p, 4 * (size_t)n_keyframes * bow_stride, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(bow_value, v->t_bvalue.p, 8 * (size_t)n_keyframes * bow_stride, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (int k = 0; k < n_keyframes; k++)
      if (bow_count[k] < 0) return VIO_ECAP;  // (bow_count[k] = -(entries needed))
    return VIO_OK;
  } catch (const std::bad_alloc &) {
    return VIO_ENOMEM;
  }
}

int vio_bow_database_create(vio_vocabulary_t *v, int32_t max_entries, int32_t max_total_words, vio_bow_database_t **out) {
  if (!v || !out || max_entries < 1 || max_total_words < 1) return VIO_EINVAL;
  VIO_ON_DEVICE_OF(v);
  vio_bow_database *d = new (std::nothrow) vio_bow_database();
  if (!d) return VIO_ENOMEM;
  d->voc_words = v->n_words, d->device = v->device, d->max_entries = max_entries, d->max_words = (size_t)max_total_words;
  try {
    d->h_off.assign((size_t)max_entries + 1, 0);
  } catch (const std::bad_alloc &) {
    delete d;
    return VIO_ENOMEM;
  }
  if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) {
    delete d;
    return VIO_ENODEV;
  }
  if (d->d_off.ensure((size_t)max_entries + 1) != VIO_OK || d->d_word.ensure(d->max_words) != VIO_OK || d->d_value.ensure(d->max_words) != VIO_OK ||
      d->inv[0].ensure(d->max_words) != VIO_OK || d->inv[1].ensure(d->max_words) != VIO_OK) {
    vio_bow_database_destroy(d);
    return VIO_ENOMEM;
  }
  *out = d;
  return VIO_OK;
}

void vio_bow_database_destroy(vio_bow_database_t *d) {
  if (!d) return;
  vio::DeviceScope scope(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream), (void)hipStreamDestroy(d->stream);
  d->d_off.release(), d->d_word.release(), d->d_value.release(), d->q_count.release(), d->q_word.release(), d->q_max.release();
  d->q_value.release(), d->inv[0].release(), d->inv[1].release(), d->flag.release(), d->n_cand.release(), d->cand.release(), d->cscore.release();
  delete d;
}

int vio_bow_database_size(const vio_bow_database_t *d, int32_t *n_entries) {
  if (!d || !n_entries) return VIO_EINVAL;
  *n_entries = d->n_entries;
  return VIO_OK;
}

int vio_bow_database_add(vio_bow_database_t *d, int32_t n, const int32_t *word, const double *value, int32_t *entry_id) {
  if (!d || n < 0 || (n > 0 && (!word || !value))) return VIO_EINVAL;
  if (d->n_entries >= d->max_entries || d->n_words + (size_t)n > d->max_words) return VIO_ECAP;
  for (int i = 0; i < n; i++)
    if (word[i] < 0 || word[i] >= d->voc_words || (i > 0 && word[i] <= word[i - 1])) return VIO_EINVAL;  // a BowVector: ascending unique words
  VIO_ON_DEVICE_OF(d);
  hipStream_t st = d->stream;
  d->h_off[d->n_entries + 1] = (int)(d->n_words + (size_t)n);  // (assigned, not appended: a failed copy below leaves nothing behind)
  if (n > 0) {
    HIP_OK(hipMemcpyAsync(d->d_word.p + d->n_words, word, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d->d_value.p + d->n_words, value, 8 * (size_t)n, hipMemcpyHostToDevice, st));
  }
  HIP_OK(hipMemcpyAsync(d->d_off.p + d->n_entries, d->h_off.data() + d->n_entries, 8, hipMemcpyHostToDevice, st));
  if (n > 0) {  // the entry's postings into the inverted file: merge into the other copy
    const size_t total = d->n_words + (size_t)n;
    hipLaunchKernelGGL(bow_insert_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d->inv[d->cur].p, (int)d->n_words,
                       d->d_word.p + d->n_words, n, d->n_entries, d->inv[1 - d->cur].p);
    HIP_OK(hipGetLastError());
  }
  HIP_OK(hipStreamSynchronize(st));  // (the caller's buffers are pageable)
  if (entry_id) *entry_id = d->n_entries;
  if (n > 0) d->cur = 1 - d->cur;
  d->n_entries++, d->n_words += (size_t)n;
  return VIO_OK;
}

int vio_bow_database_query(vio_bow_database_t *d, int32_t n_queries, const int32_t *bow_count, const int32_t *bow_word, const double *bow_value,
                           int32_t bow_stride, const int32_t *max_id, int32_t max_results, int32_t *n_results, int32_t *entry, double *score,
                           int32_t result_stride) {
  if (!d || n_queries < 1 || !bow_count || !bow_word || !bow_value || bow_stride < 1 || !max_id || !n_results || !entry || !score ||
      result_stride < 1)
    return VIO_EINVAL;
  for (int q = 0; q < n_queries; q++)
    if (bow_count[q] < 0 || bow_count[q] > bow_stride) return VIO_EINVAL;
  VIO_ON_DEVICE_OF(d);
  const int N = d->n_entries;
  if (N == 0) {
    for (int q = 0; q < n_queries; q++) n_results[q] = 0;
    return VIO_OK;
  }
  try {
    if (d->q_count.ensure(n_queries) != VIO_OK || d->q_max.ensure(n_queries) != VIO_OK ||
        d->q_word.ensure((size_t)n_queries * bow_stride) != VIO_OK || d->q_value.ensure((size_t)n_queries * bow_stride) != VIO_OK ||
        d->flag.ensure((size_t)n_queries * N) != VIO_OK || d->n_cand.ensure(n_queries) != VIO_OK ||
        d->cand.ensure((size_t)n_queries * N) != VIO_OK || d->cscore.ensure((size_t)n_queries * N) != VIO_OK)
      return VIO_ENOMEM;
    hipStream_t st = d->stream;
    int max_words = 0;
    for (int q = 0; q < n_queries; q++) max_words = std::max(max_words, (int)bow_count[q]);
    HIP_OK(hipMemcpyAsync(d->q_count.p, bow_count, 4 * (size_t)n_queries, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d->q_max.p, max_id, 4 * (size_t)n_queries, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d->q_word.p, bow_word, 4 * (size_t)n_queries * bow_stride, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d->q_value.p, bow_value, 8 * (size_t)n_queries * bow_stride, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemsetAsync(d->flag.p, 0, 4 * (size_t)n_queries * N, st));
    HIP_OK(hipMemsetAsync(d->n_cand.p, 0, 4 * (size_t)n_queries, st));
    std::vector<int> nc(n_queries, 0);
    if (max_words > 0 && d->n_words > 0) {
      hipLaunchKernelGGL(bow_candidates_kernel, dim3(std::min((max_words + 3) / 4, 256), n_queries), dim3(256), 0, st, d->inv[d->cur].p,
                         (int)d->n_words, N, d->q_count.p, d->q_word.p, bow_stride, d->q_max.p, d->flag.p, d->n_cand.p, d->cand.p);
      hipLaunchKernelGGL(bow_score_kernel, dim3(std::min((N + 3) / 4, 1024), n_queries), dim3(256), 0, st, d->d_off.p, d->d_word.p, d->d_value.p, N,
                         d->q_count.p, d->q_word.p, d->q_value.p, bow_stride, d->n_cand.p, d->cand.p, d->cscore.p);
      HIP_OK(hipGetLastError());
      HIP_OK(hipMemcpyAsync(nc.data(), d->n_cand.p, 4 * (size_t)n_queries, hipMemcpyDeviceToHost, st));
    }
    HIP_OK(hipStreamSynchronize(st));
    const int widest = *std::max_element(nc.begin(), nc.end());
    std::vector<int> h_cand((size_t)n_queries * std::max(widest, 1));
    std::vector<double> h_score((size_t)n_queries * std::max(widest, 1));
    if (widest > 0) {  // only the candidates come back: rows of `widest` out of rows of N
      HIP_OK(hipMemcpy2DAsync(h_cand.data(), 4 * (size_t)widest, d->cand.p, 4 * (size_t)N, 4 * (size_t)widest, n_queries, hipMemcpyDeviceToHost, st));
      HIP_OK(hipMemcpy2DAsync(h_score.data(), 8 * (size_t)widest, d->cscore.p, 8 * (size_t)N, 8 * (size_t)widest, n_queries, hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
    }
    // the tail of queryL1 (TemplatedDatabase.h:696-719): ascending raw score (-2 best .. 0 worst), cut, scale to [0, 1]
    std::vector<std::pair<double, int>> ret;
    for (int q = 0; q < n_queries; q++) {
      ret.clear();
      for (int c = 0; c < nc[q]; c++) {
        const double s = h_score[(size_t)q * widest + c];
        if (s <= 0.0) ret.push_back(std::make_pair(s, h_cand[(size_t)q * widest + c]));
      }
      std::sort(ret.begin(), ret.end());  // (equal scores: ascending entry id; the reference's std::sort leaves them unspecified)
      if (max_results > 0 && (int)ret.size() > max_results) ret.resize(max_results);
      if ((int)ret.size() > result_stride) return VIO_ECAP;
      n_results[q] = (int)ret.size();
      for (size_t i = 0; i < ret.size(); i++) entry[(size_t)q * result_stride + i] = ret[i].second, score[(size_t)q * result_stride + i] = -ret[i].first / 2.0;
    }
    return VIO_OK;
  } catch (const std::bad_alloc &) {
    return VIO_ENOMEM;
  }
}

}  // extern "C"
