// oracle/vio_oracle_dbow.cpp — TEST INFRASTRUCTURE ONLY (tests/, never the product path).
//
// CPU restatement of the bag-of-words query of the loop-closure producer:
//   * the vocabulary file layout          VINS_ios/loop/VocabularyBinary.hpp:17-50, VocabularyBinary.cpp:29-43
//   * TemplatedVocabulary::loadBin        ThirdParty/DBoW/TemplatedVocabulary.h:1505-1554 (children in file order)
//   * transform(feature, id, weight)      TemplatedVocabulary.h:1213-1253 (descend the k-ary tree; first minimum wins)
//   * transform(features, BowVector)      TemplatedVocabulary.h:1061-1117 (TF_IDF / TF: addWeight; IDF / BINARY:
//                                         addIfNotExist; L1 normalisation, BowVector.cpp:57-80)
//   * TemplatedDatabase::add / queryL1    ThirdParty/DBoW/TemplatedDatabase.h:439-470, 651-720 (inverted file)
//   * FBrief::distance                    ThirdParty/DBoW/FBrief.cpp:53-57 (Hamming distance of 256-bit BRIEF)
// PARITY UNPINNED: DBoW2's templates need boost::dynamic_bitset and OpenCV's FileStorage, neither of which is in the
// image, and the app's vocabulary (brief_k10L6.bin) is not part of the reference tree; the restatement follows the
// sources line by line and is checked against independent formulations in tests/test_dbow.py.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

namespace {

struct Node {
  int parent = 0;
  double weight = 0;
  std::vector<int> children;
  uint64_t desc[4] = {0, 0, 0, 0};
  int word_id = 0;
};

struct Voc {
  int k = 0, L = 0, scoring = 0, weighting = 0;
  std::vector<Node> nodes;      // [nNodes + 1], 0 = root
  std::vector<int> word_node;   // word id -> node id
};

int hamming(const uint64_t *a, const uint64_t *b) {
  int d = 0;
  for (int i = 0; i < 4; i++) d += __builtin_popcountll(a[i] ^ b[i]);
  return d;
}

// TemplatedVocabulary::transform(feature, word_id, weight): TemplatedVocabulary.h:1213-1253
void transform_one(const Voc &v, const uint64_t *f, int &word, double &weight) {
  int final_id = 0;
  do {
    const std::vector<int> &ch = v.nodes[final_id].children;
    final_id = ch[0];
    double best_d = (double)hamming(f, v.nodes[final_id].desc);
    for (size_t i = 1; i < ch.size(); i++) {
      const double d = (double)hamming(f, v.nodes[ch[i]].desc);
      if (d < best_d) best_d = d, final_id = ch[i];
    }
  } while (!v.nodes[final_id].children.empty());
  word = v.nodes[final_id].word_id;
  weight = v.nodes[final_id].weight;
}

typedef std::map<int, double> Bow;

// transform(features, v): TemplatedVocabulary.h:1061-1117 with L1Scoring::mustNormalize -> L1 (ScoringObject.cpp)
void transform_all(const Voc &v, const uint64_t *desc, int n, Bow &out) {
  out.clear();
  if (v.nodes.size() <= 1) return;
  const bool tf = v.weighting == 0 || v.weighting == 1;  // enum WeightingType { TF_IDF, TF, IDF, BINARY }
  for (int i = 0; i < n; i++) {
    int id;
    double w;
    transform_one(v, desc + 4 * (size_t)i, id, w);
    if (!(w > 0)) continue;  // stopped word
    Bow::iterator it = out.lower_bound(id);
    if (it != out.end() && it->first == id) {
      if (tf) it->second += w;  // addWeight; addIfNotExist leaves an existing entry alone
    } else {
      out.insert(it, Bow::value_type(id, w));
    }
  }
  // (must == true for L1_NORM: the "divide by size" branch does not run)
  double norm = 0.0;
  for (Bow::iterator it = out.begin(); it != out.end(); ++it) norm += fabs(it->second);
  if (norm > 0.0)
    for (Bow::iterator it = out.begin(); it != out.end(); ++it) it->second /= norm;
}

struct Db {
  const Voc *voc;
  int n_entries = 0;
  std::vector<std::vector<std::pair<int, double>>> ifile;  // word -> (entry, weight), ascending entry id
};

}  // namespace

extern "C" {

void *oracle_voc_create(const void *blob, size_t bytes) {
  const unsigned char *p = (const unsigned char *)blob;
  if (bytes < 24) return nullptr;
  int32_t hdr[6];
  memcpy(hdr, p, 24);
  const int nNodes = hdr[4], nWords = hdr[5];
  if (nNodes < 0 || nWords < 0 || bytes < 24 + (size_t)nNodes * 48 + (size_t)nWords * 8) return nullptr;
  Voc *v = new Voc();
  v->k = hdr[0], v->L = hdr[1], v->scoring = hdr[2], v->weighting = hdr[3];
  v->nodes.resize((size_t)nNodes + 1);
  const unsigned char *q = p + 24;
  for (int i = 0; i < nNodes; i++, q += 48) {
    int32_t nid, pid;
    double wgt;
    memcpy(&nid, q, 4), memcpy(&pid, q + 4, 4), memcpy(&wgt, q + 8, 8);
    if (nid < 1 || nid > nNodes || pid < 0 || pid > nNodes) {
      delete v;
      return nullptr;
    }
    v->nodes[nid].parent = pid, v->nodes[nid].weight = wgt;
    memcpy(v->nodes[nid].desc, q + 16, 32);
    v->nodes[pid].children.push_back(nid);
  }
  v->word_node.assign(nWords, 0);
  for (int i = 0; i < nWords; i++, q += 8) {
    int32_t nid, wid;
    memcpy(&nid, q, 4), memcpy(&wid, q + 4, 4);
    if (nid < 1 || nid > nNodes || wid < 0 || wid >= nWords) {
      delete v;
      return nullptr;
    }
    v->nodes[nid].word_id = wid;
    v->word_node[wid] = nid;
  }
  return v;
}
void oracle_voc_destroy(void *v) { delete (Voc *)v; }

int oracle_voc_transform(void *vv, const uint64_t *desc, int n, int32_t *word, double *weight) {
  const Voc &v = *(const Voc *)vv;
  for (int i = 0; i < n; i++) {
    int id;
    double w;
    transform_one(v, desc + 4 * (size_t)i, id, w);
    word[i] = id, weight[i] = w;
  }
  return 0;
}

// -> number of BoW entries (ascending word id), or -1 if cap is too small
int oracle_voc_bow(void *vv, const uint64_t *desc, int n, int32_t *out_word, double *out_value, int cap) {
  Bow b;
  transform_all(*(const Voc *)vv, desc, n, b);
  if ((int)b.size() > cap) return -1;
  int i = 0;
  for (Bow::iterator it = b.begin(); it != b.end(); ++it, ++i) out_word[i] = it->first, out_value[i] = it->second;
  return i;
}

void *oracle_db_create(void *vv) {
  Db *d = new Db();
  d->voc = (const Voc *)vv;
  d->ifile.resize(d->voc->word_node.size());
  return d;
}
void oracle_db_destroy(void *d) { delete (Db *)d; }

// TemplatedDatabase::add(BowVector): TemplatedDatabase.h:439-470 -> entry id
int oracle_db_add(void *dd, const int32_t *word, const double *value, int n) {
  Db &d = *(Db *)dd;
  const int id = d.n_entries++;
  for (int i = 0; i < n; i++) d.ifile[word[i]].push_back(std::make_pair(id, value[i]));
  return id;
}

// TemplatedDatabase::queryL1: TemplatedDatabase.h:651-720. -> number of results (best first), scores in [0, 1]
int oracle_db_query(void *dd, const int32_t *word, const double *value, int n, int max_results, int max_id, int32_t *out_entry,
                    double *out_score, int cap) {
  const Db &d = *(const Db *)dd;
  std::map<int, double> pairs;
  for (int i = 0; i < n; i++) {
    const double q = value[i];
    const std::vector<std::pair<int, double>> &row = d.ifile[word[i]];
    for (size_t r = 0; r < row.size(); r++) {
      const int e = row[r].first;
      const double dv = row[r].second;
      if (e < max_id || max_id == -1) {
        const double val = fabs(q - dv) - fabs(q) - fabs(dv);
        std::map<int, double>::iterator pit = pairs.lower_bound(e);
        if (pit != pairs.end() && pit->first == e) pit->second += val;
        else pairs.insert(pit, std::make_pair(e, val));
      }
    }
  }
  std::vector<std::pair<double, int>> ret;  // (score, entry): "the lower the better"
  for (std::map<int, double>::iterator it = pairs.begin(); it != pairs.end(); ++it) ret.push_back(std::make_pair(it->second, it->first));
  // (std::sort on Score alone in the reference: equal scores come out in an unspecified order; ties broken by entry id here)
  std::sort(ret.begin(), ret.end());
  if (max_results > 0 && (int)ret.size() > max_results) ret.resize(max_results);
  if ((int)ret.size() > cap) return -1;
  for (size_t i = 0; i < ret.size(); i++) out_entry[i] = ret[i].second, out_score[i] = -ret[i].first / 2.0;
  return (int)ret.size();
}

}  // extern "C"
