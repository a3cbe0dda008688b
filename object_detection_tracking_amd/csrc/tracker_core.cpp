// DeepSORT tracker core, native (host C++ + the HIP cosine kernel): Kalman filter, gating,
// matching cascade, IoU matching, rectangular assignment, track life cycle, appearance gallery.
//
// "Next" row 2 of SURVEY.md 8(f).  Restates, in float64 like the reference:
//   deep_sort/kalman_filter.py:23-232   (constant-velocity filter on (x, y, a, h); chi2 gating)
//   deep_sort/track.py:15-170           (Track state machine)
//   deep_sort/iou_matching.py:8-84      (IoU cost)
//   deep_sort/linear_assignment.py:12-194 (min_cost_matching, matching_cascade, gate_cost_matrix)
//   deep_sort/tracker.py:40-138         (Tracker.predict / update / _match / _initiate_track)
//   deep_sort/nn_matching.py:99-177     (gallery bookkeeping; the distance itself is the HIP
//                                        kernel of tracker.hip)
// and scipy.optimize.linear_sum_assignment (un-vendored dependency, scipy >= 1.4: the shortest
// augmenting path algorithm of Crouse 2016 as implemented in rectangular_lsap.cpp, including
// its tie rules: columns scanned from the highest index down, an unassigned column preferred
// among equal reduced costs).  Results match the reference to summation-order rounding of the
// 4x4 Cholesky solves (LAPACK in the reference), i.e. ~1e-12 relative; identities, match lists
// and the order in which new track ids are handed out are reproduced exactly.
#include "../../include/odt.h"

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "odt_common.hpp"

namespace odt {

// ------------------------------------------------------------------ assignment
// scipy.optimize.linear_sum_assignment.  cost: nr x nc row-major.  Returns pairs sorted by row.
int lsap(const double* cost_in, int nr, int nc, std::vector<int>* rows, std::vector<int>* cols) {
  rows->clear(); cols->clear();
  if (nr == 0 || nc == 0) return 0;
  const bool transpose = nc < nr;
  std::vector<double> ct;
  const double* cost = cost_in;
  int R = nr, Cn = nc;
  if (transpose) {
    ct.resize((size_t)nr * nc);
    for (int i = 0; i < nr; ++i)
      for (int j = 0; j < nc; ++j) ct[(size_t)j * nr + i] = cost_in[(size_t)i * nc + j];
    cost = ct.data(); R = nc; Cn = nr;
  }
  for (size_t i = 0; i < (size_t)R * Cn; ++i)
    if (cost[i] != cost[i] || cost[i] == -std::numeric_limits<double>::infinity()) {
      set_error("lsap: cost matrix contains invalid numeric entries");
      return 1;
    }
  const double INF = std::numeric_limits<double>::infinity();
  std::vector<double> u(R, 0.0), v(Cn, 0.0), spc(Cn);
  std::vector<int> path(Cn, -1), col4row(R, -1), row4col(Cn, -1), remaining(Cn);
  std::vector<char> SR(R), SC(Cn);
  for (int cur = 0; cur < R; ++cur) {
    double minVal = 0.0;
    int num_remaining = Cn;
    for (int it = 0; it < Cn; ++it) remaining[it] = Cn - it - 1;
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(spc.begin(), spc.end(), INF);
    int sink = -1, i = cur;
    while (sink == -1) {
      int index = -1;
      double lowest = INF;
      SR[i] = 1;
      for (int it = 0; it < num_remaining; ++it) {
        const int j = remaining[it];
        const double r = minVal + cost[(size_t)i * Cn + j] - u[i] - v[j];
        if (r < spc[j]) { path[j] = i; spc[j] = r; }
        if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
      }
      minVal = lowest;
      if (minVal == INF) { set_error("lsap: cost matrix is infeasible"); return 1; }
      const int j = remaining[index];
      if (row4col[j] == -1) sink = j; else i = row4col[j];
      SC[j] = 1;
      remaining[index] = remaining[--num_remaining];
    }
    u[cur] += minVal;
    for (int r2 = 0; r2 < R; ++r2)
      if (SR[r2] && r2 != cur) u[r2] += minVal - spc[col4row[r2]];
    for (int j = 0; j < Cn; ++j)
      if (SC[j]) v[j] -= minVal - spc[j];
    int j = sink;
    for (;;) {
      const int r2 = path[j];
      row4col[j] = r2;
      std::swap(col4row[r2], j);
      if (r2 == cur) break;
    }
  }
  if (transpose) {
    std::vector<std::pair<int, int>> pr;
    for (int r2 = 0; r2 < R; ++r2) pr.emplace_back(col4row[r2], r2);   // (original row, original col)
    std::sort(pr.begin(), pr.end());
    for (auto& q : pr) { rows->push_back(q.first); cols->push_back(q.second); }
  } else {
    for (int r2 = 0; r2 < R; ++r2) { rows->push_back(r2); cols->push_back(col4row[r2]); }
  }
  return 0;
}

namespace {

// ---------------------------------------------------------------------- Kalman
constexpr double kStdPos = 1.0 / 20, kStdVel = 1.0 / 160;
constexpr double kChi2_4 = 9.4877;           // kalman_filter.py:11-20 chi2inv95[4]
constexpr double kInftyCost = 1e+5;          // linear_assignment.py:9

inline double sq(double x) { return x * x; }

void kf_initiate(const double* z, double* mean, double* cov) {      // kalman_filter.py:57-86
  for (int i = 0; i < 4; ++i) { mean[i] = z[i]; mean[4 + i] = 0.0; }
  const double h = z[3];
  const double std_[8] = {2 * kStdPos * h, 2 * kStdPos * h, 1e-2, 2 * kStdPos * h,
                          10 * kStdVel * h, 10 * kStdVel * h, 1e-5, 10 * kStdVel * h};
  std::memset(cov, 0, 64 * sizeof(double));
  for (int i = 0; i < 8; ++i) cov[i * 8 + i] = sq(std_[i]);
}

void kf_predict(double* mean, double* cov) {                         // kalman_filter.py:88-124
  const double h = mean[3];
  const double q[8] = {sq(kStdPos * h), sq(kStdPos * h), sq(1e-2), sq(kStdPos * h),
                       sq(kStdVel * h), sq(kStdVel * h), sq(1e-5), sq(kStdVel * h)};
  for (int i = 0; i < 4; ++i) mean[i] = mean[i] + mean[4 + i];
  // F P F^T with F = [[I, I],[0, I]] (dt = 1): X = P F^T, then F X (only two non-zero terms per
  // entry, so this equals numpy's multi_dot bit for bit)
  double X[64];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) X[i * 8 + j] = j < 4 ? cov[i * 8 + j] + cov[i * 8 + j + 4] : cov[i * 8 + j];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) cov[i * 8 + j] = i < 4 ? X[i * 8 + j] + X[(i + 4) * 8 + j] : X[i * 8 + j];
  for (int i = 0; i < 8; ++i) cov[i * 8 + i] += q[i];
}

void kf_project(const double* mean, const double* cov, double* pm, double* pc) {   // :126-154
  const double h = mean[3];
  const double r[4] = {sq(kStdPos * h), sq(kStdPos * h), sq(1e-1), sq(kStdPos * h)};
  for (int i = 0; i < 4; ++i) {
    pm[i] = mean[i];
    for (int j = 0; j < 4; ++j) pc[i * 4 + j] = cov[i * 8 + j] + (i == j ? r[i] : 0.0);
  }
}

bool chol4(const double* a, double* L) {          // lower Cholesky of a 4x4 SPD matrix
  std::memset(L, 0, 16 * sizeof(double));
  for (int j = 0; j < 4; ++j) {
    double d = a[j * 4 + j];
    for (int k = 0; k < j; ++k) d -= L[j * 4 + k] * L[j * 4 + k];
    if (!(d > 0.0)) return false;
    L[j * 4 + j] = std::sqrt(d);
    for (int i = j + 1; i < 4; ++i) {
      double s = a[i * 4 + j];
      for (int k = 0; k < j; ++k) s -= L[i * 4 + k] * L[j * 4 + k];
      L[i * 4 + j] = s / L[j * 4 + j];
    }
  }
  return true;
}

bool kf_update(double* mean, double* cov, const double* z) {          // kalman_filter.py:156-189
  double pm[4], S[16], L[16];
  kf_project(mean, cov, pm, S);
  if (!chol4(S, L)) return false;
  // K^T = S^-1 (P H^T)^T : solve S x = b for the 8 right-hand sides b = row i of P[:, :4]
  double K[32];                                   // [8][4]
  for (int i = 0; i < 8; ++i) {
    double y[4], x[4];
    for (int r = 0; r < 4; ++r) {                 // forward: L y = b
      double s = cov[i * 8 + r];
      for (int k = 0; k < r; ++k) s -= L[r * 4 + k] * y[k];
      y[r] = s / L[r * 4 + r];
    }
    for (int r = 3; r >= 0; --r) {                // backward: L^T x = y
      double s = y[r];
      for (int k = r + 1; k < 4; ++k) s -= L[k * 4 + r] * x[k];
      x[r] = s / L[r * 4 + r];
    }
    for (int r = 0; r < 4; ++r) K[i * 4 + r] = x[r];
  }
  double innov[4];
  for (int r = 0; r < 4; ++r) innov[r] = z[r] - pm[r];
  for (int i = 0; i < 8; ++i) {
    double s = 0.0;
    for (int r = 0; r < 4; ++r) s += innov[r] * K[i * 4 + r];
    mean[i] += s;
  }
  double SK[32];                                  // S K^T : [4][8]
  for (int r = 0; r < 4; ++r)
    for (int j = 0; j < 8; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += S[r * 4 + k] * K[j * 4 + k];
      SK[r * 8 + j] = s;
    }
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += K[i * 4 + k] * SK[k * 8 + j];
      cov[i * 8 + j] -= s;
    }
  return true;
}

// squared Mahalanobis distance of every measurement to the projected track state (:191-232)
bool kf_gating(const double* mean, const double* cov, const double* meas, int n, double* out) {
  double pm[4], S[16], L[16];
  kf_project(mean, cov, pm, S);
  if (!chol4(S, L)) return false;
  for (int m = 0; m < n; ++m) {
    double y[4], acc = 0.0;
    for (int r = 0; r < 4; ++r) {
      double s = meas[m * 4 + r] - pm[r];
      for (int k = 0; k < r; ++k) s -= L[r * 4 + k] * y[k];
      y[r] = s / L[r * 4 + r];
      acc += y[r] * y[r];
    }
    out[m] = acc;
  }
  return true;
}

enum { kTentative = 1, kConfirmed = 2, kDeleted = 3 };

struct Trk {
  double mean[8], cov[64];
  int id, hits, age, tsu, state;
  std::vector<std::vector<float>> features;     // not yet flushed into the gallery
  void tlwh(double* o) const {                  // track.py:71-84
    o[0] = mean[0]; o[1] = mean[1]; o[2] = mean[2] * mean[3]; o[3] = mean[3];
    o[0] -= o[2] / 2; o[1] -= o[3] / 2;
  }
};

struct Det { double tlwh[4], xyah[4], conf; const float* feat; };

}  // namespace
}  // namespace odt

using namespace odt;

struct odt_tracker {
  double max_cos, max_iou;
  int budget, max_age, n_init, device, D = 0;
  int next_id = 1;
  std::vector<Trk> tracks;
  std::map<int, std::deque<std::vector<float>>> samples;     // nn_matching.py:125-154
  // the appearance cost of ONE update(): every confirmed track x every detection, computed by one kernel
  // call on the tracker's own stream (persistent pinned + device scratch: CosineCtx); the cascade levels
  // read their sub-matrices from it (the reference recomputes the same entries level by level,
  // tracker.py:94-104 -> nn_matching.py:156-177: an entry depends on its track and detection only)
  CosineCtx cos;
  std::vector<double> app_cost;          // [confirmed tracks at update start][N]
  std::map<int, int> app_row;            // track id -> row of app_cost
  int app_n = 0;
};

namespace {

// NearestNeighborDistanceMetric.distance of every confirmed track's gallery against all N detections
// (one HIP kernel call per update)
int appearance_costs(odt_tracker* t, const std::vector<Det>& dets, const std::vector<int>& confirmed) {
  const int T = (int)confirmed.size(), N = (int)dets.size(), D = t->D;
  t->app_row.clear(); t->app_n = N;
  t->app_cost.assign((size_t)T * N, 0.0);
  if (T == 0 || N == 0) return 0;
  std::vector<const float*> gal, det(N);
  std::vector<int> seg(1, 0);
  for (int r = 0; r < T; ++r) {
    const int id = t->tracks[confirmed[r]].id;
    auto it = t->samples.find(id);
    ODT_CHECK(it != t->samples.end() && !it->second.empty(), "tracker: confirmed track without gallery samples");
    for (const auto& f : it->second) gal.push_back(f.data());
    seg.push_back((int)gal.size());
    t->app_row[id] = r;
  }
  for (int j = 0; j < N; ++j) det[j] = dets[j].feat;
  return t->cos.run(t->device, gal.data(), (int)gal.size(), seg.data(), T, det.data(), N, D, t->app_cost.data());
}

typedef int (*CostFn)(odt_tracker*, const std::vector<Det>&, const std::vector<int>&, const std::vector<int>&,
                      std::vector<double>*);

// tracker.py:94-104 gated_metric
int gated_metric_cost(odt_tracker* t, const std::vector<Det>& dets, const std::vector<int>& trk_idx,
                      const std::vector<int>& det_idx, std::vector<double>* cost) {
  const int N = (int)det_idx.size();
  cost->assign(trk_idx.size() * (size_t)N, 0.0);
  for (size_t r = 0; r < trk_idx.size(); ++r) {
    auto it = t->app_row.find(t->tracks[trk_idx[r]].id);
    ODT_CHECK(it != t->app_row.end(), "tracker: appearance cost of an unconfirmed track requested");
    const double* src = &t->app_cost[(size_t)it->second * t->app_n];
    for (int j = 0; j < N; ++j) (*cost)[r * N + j] = src[det_idx[j]];
  }
  std::vector<double> meas((size_t)N * 4), g(N);
  for (int j = 0; j < N; ++j) std::memcpy(&meas[(size_t)j * 4], dets[det_idx[j]].xyah, 4 * sizeof(double));
  for (size_t r = 0; r < trk_idx.size(); ++r) {            // linear_assignment.py:148-194
    const Trk& tr = t->tracks[trk_idx[r]];
    ODT_CHECK(kf_gating(tr.mean, tr.cov, meas.data(), N, g.data()), "tracker: projected covariance not positive definite");
    for (int j = 0; j < N; ++j)
      if (g[j] > kChi2_4) (*cost)[r * N + j] = kInftyCost;
  }
  return 0;
}

// iou_matching.py:42-84
int iou_cost(odt_tracker* t, const std::vector<Det>& dets, const std::vector<int>& trk_idx,
             const std::vector<int>& det_idx, std::vector<double>* cost) {
  const int N = (int)det_idx.size();
  cost->assign(trk_idx.size() * (size_t)N, 0.0);
  for (size_t r = 0; r < trk_idx.size(); ++r) {
    const Trk& tr = t->tracks[trk_idx[r]];
    if (tr.tsu > 1) {
      for (int j = 0; j < N; ++j) (*cost)[r * N + j] = kInftyCost;
      continue;
    }
    double b[4];
    tr.tlwh(b);
    const double bx2 = b[0] + b[2], by2 = b[1] + b[3], area_b = b[2] * b[3];
    for (int j = 0; j < N; ++j) {
      const double* c = dets[det_idx[j]].tlwh;
      const double tlx = std::max(b[0], c[0]), tly = std::max(b[1], c[1]);
      const double brx = std::min(bx2, c[0] + c[2]), bry = std::min(by2, c[1] + c[3]);
      const double w = std::max(0.0, brx - tlx), h = std::max(0.0, bry - tly);
      const double inter = w * h, area_c = c[2] * c[3];
      (*cost)[r * N + j] = 1.0 - inter / (area_b + area_c - inter);
    }
  }
  return 0;
}

struct Matching {
  std::vector<std::pair<int, int>> matches;
  std::vector<int> unmatched_tracks, unmatched_dets;
};

// linear_assignment.py:12-79
int min_cost_matching(odt_tracker* t, CostFn fn, double max_distance, const std::vector<Det>& dets,
                      const std::vector<int>& trk_idx, const std::vector<int>& det_idx, Matching* out) {
  out->matches.clear(); out->unmatched_tracks.clear(); out->unmatched_dets.clear();
  if (det_idx.empty() || trk_idx.empty()) {
    out->unmatched_tracks = trk_idx; out->unmatched_dets = det_idx;
    return 0;
  }
  std::vector<double> cost;
  if (fn(t, dets, trk_idx, det_idx, &cost)) return 1;
  const int R = (int)trk_idx.size(), N = (int)det_idx.size();
  for (double& c : cost)
    if (c > max_distance) c = max_distance + 1e-5;
  std::vector<int> rows, cols;
  if (lsap(cost.data(), R, N, &rows, &cols)) return 1;
  std::vector<char> col_used(N, 0), row_used(R, 0);
  for (size_t q = 0; q < rows.size(); ++q) { row_used[rows[q]] = 1; col_used[cols[q]] = 1; }
  for (int c = 0; c < N; ++c)
    if (!col_used[c]) out->unmatched_dets.push_back(det_idx[c]);
  for (int r = 0; r < R; ++r)
    if (!row_used[r]) out->unmatched_tracks.push_back(trk_idx[r]);
  for (size_t q = 0; q < rows.size(); ++q) {
    const int ti = trk_idx[rows[q]], di = det_idx[cols[q]];
    if (cost[(size_t)rows[q] * N + cols[q]] > max_distance) {
      out->unmatched_tracks.push_back(ti);
      out->unmatched_dets.push_back(di);
    } else {
      out->matches.emplace_back(ti, di);
    }
  }
  return 0;
}

}  // namespace

extern "C" {

int odt_tracker_nms(const double* boxes, const double* scores, const int32_t* order_in, int n, double max_overlap, int32_t* pick,
                    int* npick) {
  ODT_CHECK(npick != nullptr && (n == 0 || (boxes != nullptr && pick != nullptr)), "odt_tracker_nms: null argument");
  *npick = 0;
  if (n <= 0) return 0;
  std::vector<double> x2(n), y2(n), area(n);
  for (int i = 0; i < n; ++i) {
    x2[i] = boxes[4 * i] + boxes[4 * i + 2]; y2[i] = boxes[4 * i + 1] + boxes[4 * i + 3];
    area[i] = (x2[i] - boxes[4 * i] + 1) * (y2[i] - boxes[4 * i + 1] + 1);
  }
  // visiting order: the caller's np.argsort (ties then keep numpy's order), else a stable ascending sort
  std::vector<int> order(n);
  if (order_in != nullptr) {
    std::vector<char> seen(n, 0);
    for (int i = 0; i < n; ++i) {
      ODT_CHECK(order_in[i] >= 0 && order_in[i] < n && !seen[order_in[i]], "odt_tracker_nms: order is not a permutation");
      seen[order_in[i]] = 1; order[i] = order_in[i];
    }
  } else {
    for (int i = 0; i < n; ++i) order[i] = i;
    const double* key = scores != nullptr ? scores : y2.data();
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
  }
  while (!order.empty()) {
    const int i = order.back();
    order.pop_back();
    pick[(*npick)++] = i;
    std::vector<int> rest;
    rest.reserve(order.size());
    for (int j : order) {
      const double iw = std::max(0.0, std::min(x2[i], x2[j]) - std::max(boxes[4 * i], boxes[4 * j]) + 1);
      const double ih = std::max(0.0, std::min(y2[i], y2[j]) - std::max(boxes[4 * i + 1], boxes[4 * j + 1]) + 1);
      if (!((iw * ih) / area[j] > max_overlap)) rest.push_back(j);
    }
    order.swap(rest);
  }
  return 0;
}

int odt_lsap(const double* cost, int nr, int nc, int32_t* rows, int32_t* cols, int* n) {
  ODT_CHECK(rows && cols && n && (cost || nr * nc == 0), "odt_lsap: null argument");
  std::vector<int> r, c;
  if (lsap(cost, nr, nc, &r, &c)) return 1;
  *n = (int)r.size();
  for (size_t i = 0; i < r.size(); ++i) { rows[i] = r[i]; cols[i] = c[i]; }
  return 0;
}

int odt_tracker_create(double max_cosine_distance, int nn_budget, double max_iou_distance, int max_age,
                       int n_init, int device, odt_tracker_handle* out) {
  ODT_CHECK(out != nullptr, "odt_tracker_create: null argument");
  ODT_CHECK(max_age >= 1 && n_init >= 1, "odt_tracker_create: bad parameters");
  knobs_reload();
  odt_tracker* t = new odt_tracker();
  t->max_cos = max_cosine_distance; t->budget = nn_budget; t->max_iou = max_iou_distance;
  t->max_age = max_age; t->n_init = n_init; t->device = device;
  *out = t;
  return 0;
}

int odt_tracker_destroy(odt_tracker_handle t) {
  delete t;
  return 0;
}

int odt_tracker_predict(odt_tracker_handle t) {           // tracker.py:50-56, track.py:112-126
  ODT_CHECK(t != nullptr, "null tracker");
  for (Trk& tr : t->tracks) {
    kf_predict(tr.mean, tr.cov);
    tr.age += 1;
    tr.tsu += 1;
  }
  return 0;
}

int odt_tracker_update(odt_tracker_handle t, const double* tlwh, const double* conf, const float* feats,
                       int N, int D) {                    // tracker.py:57-138
  ODT_CHECK(t != nullptr, "null tracker");
  ODT_CHECK(N == 0 || (tlwh && conf && feats && D > 0), "odt_tracker_update: null argument");
  if (N > 0) {
    ODT_CHECK(t->D == 0 || t->D == D, "odt_tracker_update: feature dimension changed");
    t->D = D;
  }
  static const bool timing = env_knob(K_TRACKER_TIMING).set;     // tuning aid: where an update's wall time goes (stderr, every 160 updates)
  static double tacc[4] = {0, 0, 0, 0}; static long tn = 0;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double u0 = timing ? now() : 0;
  std::vector<Det> dets(N);
  for (int j = 0; j < N; ++j) {
    Det& d = dets[j];
    std::memcpy(d.tlwh, tlwh + (size_t)j * 4, 4 * sizeof(double));
    d.xyah[0] = d.tlwh[0] + d.tlwh[2] / 2; d.xyah[1] = d.tlwh[1] + d.tlwh[3] / 2;    // detection.py:42-49
    d.xyah[2] = d.tlwh[2] / d.tlwh[3]; d.xyah[3] = d.tlwh[3];
    d.conf = conf[j]; d.feat = feats + (size_t)j * D;
  }
  // ---- _match (tracker.py:92-138)
  std::vector<int> confirmed, unconfirmed;
  for (int i = 0; i < (int)t->tracks.size(); ++i)
    (t->tracks[i].state == kConfirmed ? confirmed : unconfirmed).push_back(i);
  const double u1 = timing ? now() : 0;
  if (appearance_costs(t, dets, confirmed)) return 1;
  const double u2 = timing ? now() : 0;
  // matching_cascade (linear_assignment.py:82-145)
  std::vector<int> unmatched_dets(N);
  for (int j = 0; j < N; ++j) unmatched_dets[j] = j;
  std::vector<std::pair<int, int>> matches;
  for (int level = 0; level < t->max_age; ++level) {
    if (unmatched_dets.empty()) break;
    std::vector<int> lvl;
    for (int k : confirmed)
      if (t->tracks[k].tsu == 1 + level) lvl.push_back(k);
    if (lvl.empty()) continue;
    Matching m;
    if (min_cost_matching(t, gated_metric_cost, t->max_cos, dets, lvl, unmatched_dets, &m)) return 1;
    matches.insert(matches.end(), m.matches.begin(), m.matches.end());
    unmatched_dets = m.unmatched_dets;
  }
  std::set<int> matched_a;
  for (auto& pr : matches) matched_a.insert(pr.first);
  std::vector<int> iou_cand = unconfirmed, unmatched_tracks;
  for (int k : confirmed) {
    if (matched_a.count(k)) continue;
    if (t->tracks[k].tsu == 1) iou_cand.push_back(k); else unmatched_tracks.push_back(k);
  }
  Matching mb;
  if (min_cost_matching(t, iou_cost, t->max_iou, dets, iou_cand, unmatched_dets, &mb)) return 1;
  matches.insert(matches.end(), mb.matches.begin(), mb.matches.end());
  unmatched_tracks.insert(unmatched_tracks.end(), mb.unmatched_tracks.begin(), mb.unmatched_tracks.end());
  const double u3 = timing ? now() : 0;
  // ---- update track set (tracker.py:70-78)
  for (auto& pr : matches) {
    Trk& tr = t->tracks[pr.first];
    const Det& d = dets[pr.second];
    ODT_CHECK(kf_update(tr.mean, tr.cov, d.xyah), "tracker: innovation covariance not positive definite");
    tr.features.emplace_back(d.feat, d.feat + D);
    tr.hits += 1;
    tr.tsu = 0;
    if (tr.state == kTentative && tr.hits >= t->n_init) tr.state = kConfirmed;
  }
  for (int k : unmatched_tracks) {                       // track.py:147-153
    Trk& tr = t->tracks[k];
    if (tr.state == kTentative) tr.state = kDeleted;
    else if (tr.tsu > t->max_age) tr.state = kDeleted;
  }
  for (int j : mb.unmatched_dets) {                      // tracker.py:133-138
    Trk tr;
    kf_initiate(dets[j].xyah, tr.mean, tr.cov);
    tr.id = t->next_id++; tr.hits = 1; tr.age = 1; tr.tsu = 0; tr.state = kTentative;
    tr.features.emplace_back(dets[j].feat, dets[j].feat + D);
    t->tracks.push_back(std::move(tr));
  }
  t->tracks.erase(std::remove_if(t->tracks.begin(), t->tracks.end(),
                                 [](const Trk& tr) { return tr.state == kDeleted; }), t->tracks.end());
  // ---- metric.partial_fit (tracker.py:80-90, nn_matching.py:137-154)
  std::set<int> active;
  for (Trk& tr : t->tracks) {
    if (tr.state != kConfirmed) continue;
    active.insert(tr.id);
    auto& bucket = t->samples[tr.id];
    for (auto& f : tr.features) {
      bucket.push_back(std::move(f));
      if (t->budget > 0)
        while ((int)bucket.size() > t->budget) bucket.pop_front();
    }
    tr.features.clear();
  }
  for (auto it = t->samples.begin(); it != t->samples.end();)
    it = active.count(it->first) ? std::next(it) : t->samples.erase(it);
  if (timing) {
    const double u4 = now();
    tacc[0] += u1 - u0; tacc[1] += u2 - u1; tacc[2] += u3 - u2; tacc[3] += u4 - u3;
    if (++tn % 160 == 0) {
      fprintf(stderr, "[tracker] per update over 160 (us): setup %.1f  appearance %.1f  matching %.1f  track set + gallery %.1f  (tracks %d, N %d)\n",
              tacc[0] / 160 * 1e6, tacc[1] / 160 * 1e6, tacc[2] / 160 * 1e6, tacc[3] / 160 * 1e6, (int)t->tracks.size(), N);
      for (double& v : tacc) v = 0;
    }
  }
  return 0;
}

int odt_tracker_tracks(odt_tracker_handle t, int cap, int32_t* ids, int32_t* state, int32_t* time_since_update,
                       int32_t* hits, int32_t* age, double* mean, double* covariance, int* n) {
  ODT_CHECK(t != nullptr && n != nullptr, "odt_tracker_tracks: null argument");
  *n = (int)t->tracks.size();
  for (int i = 0; i < *n && i < cap; ++i) {
    const Trk& tr = t->tracks[i];
    if (ids) ids[i] = tr.id;
    if (state) state[i] = tr.state;
    if (time_since_update) time_since_update[i] = tr.tsu;
    if (hits) hits[i] = tr.hits;
    if (age) age[i] = tr.age;
    if (mean) std::memcpy(mean + (size_t)i * 8, tr.mean, 8 * sizeof(double));
    if (covariance) std::memcpy(covariance + (size_t)i * 64, tr.cov, 64 * sizeof(double));
  }
  return 0;
}

}  // extern "C"


// =================================================================================================
// TMOT / JDE tracker core (reference tmot/multitracker.py, tmot/matching.py, tmot/kalman_filter.py,
// tmot/basetrack.py; used by obj_detect_tracking_multi_queuer_tmot.py:543-583).  "Next" row 3b of
// SURVEY.md 8(f).  Same Kalman model as deep_sort (the constants above); association: embedding
// distance (Euclidean between the EMA-smoothed, L2-normalised track feature and the detection
// feature) fused with the Mahalanobis gate, then two IoU stages; assignment = lap.lapjv(cost,
// extend_cost=True, cost_limit=thresh), i.e. the cost matrix extended to (n+m)^2 with thresh/2 in
// the off-diagonal blocks and 0 in the lower-right block, solved exactly (lap 0.4.0 _lapjv.pyx) --
// here with the scipy-exact solver above; IoU = cython_bbox.bbox_overlaps ("+1" widths/heights).
// =================================================================================================
namespace odt {
namespace {

enum { kNew = 0, kTracked = 1, kLost = 2, kRemoved = 3 };       // basetrack.py:5-9

struct STrk {
  int id = 0, state = kNew;
  bool activated = false, has_kf = false;
  double mean[8], cov[64];
  double tlwh0[4];                       // _tlwh
  double score = 0.0;
  int tracklet_len = 0, frame_id = 0, start_frame = 0;
  std::vector<float> smooth, curr;
  double det_tlwh[4], det_conf = 0.0;
  double alpha = 0.9;                    // Python float (multitracker.py:34); cast per use, as numpy does

  void tlwh(double* o) const {           // multitracker.py:119-130
    if (!has_kf) { for (int i = 0; i < 4; ++i) o[i] = tlwh0[i]; return; }
    o[0] = mean[0]; o[1] = mean[1]; o[2] = mean[2] * mean[3]; o[3] = mean[3];
    o[0] -= o[2] / 2; o[1] -= o[3] / 2;
  }
  void tlbr(double* o) const { tlwh(o); o[2] += o[0]; o[3] += o[1]; }
};
typedef std::shared_ptr<STrk> STrkP;

void tlwh_to_xyah(const double* t, double* o) {                  // multitracker.py:143-151
  o[0] = t[0] + t[2] / 2; o[1] = t[1] + t[3] / 2; o[2] = t[2] / t[3]; o[3] = t[3];
}

// feat /= np.linalg.norm(feat)  (float32 array; the norm is rounded to float32)
void normalize_f32(std::vector<float>& f) {
  double s = 0.0;
  for (float x : f) s += (double)x * (double)x;
  const float n = (float)std::sqrt(s);
  for (float& x : f) x = x / n;
}

// STrack.update_features (multitracker.py:36-45), including its aliasing: on the first call
// smooth_feat IS the feature array, so the final in-place normalisation hits curr_feat too
void update_features(STrk& t, std::vector<float> feat, std::vector<float>* feat_inout) {
  normalize_f32(feat);
  if (feat_inout) *feat_inout = feat;    // the caller's array was normalised in place
  t.curr = feat;
  if (t.smooth.empty()) {
    normalize_f32(t.curr);               // smooth_feat /= norm on the shared array
    t.smooth = t.curr;
    if (feat_inout) *feat_inout = t.curr;
  } else {
    const float a = (float)t.alpha, b = (float)(1.0 - t.alpha);     // float32(0.9), float32(1 - 0.9) = float32(0.1)
    for (size_t i = 0; i < feat.size(); ++i) t.smooth[i] = a * t.smooth[i] + b * feat[i];
    normalize_f32(t.smooth);
  }
}

double iou_plus1(const double* a, const double* b) {             // cython_bbox.bbox_overlaps
  const double iw = std::min(a[2], b[2]) - std::max(a[0], b[0]) + 1;
  if (iw <= 0) return 0.0;
  const double ih = std::min(a[3], b[3]) - std::max(a[1], b[1]) + 1;
  if (ih <= 0) return 0.0;
  const double ua = (a[2] - a[0] + 1) * (a[3] - a[1] + 1) + (b[2] - b[0] + 1) * (b[3] - b[1] + 1) - iw * ih;
  return iw * ih / ua;
}

void iou_distance(const std::vector<STrkP>& a, const std::vector<STrkP>& b, std::vector<double>* cost) {
  cost->assign(a.size() * b.size(), 0.0);                        // matching.py:61-79
  for (size_t i = 0; i < a.size(); ++i) {
    double ba[4]; a[i]->tlbr(ba);
    for (size_t j = 0; j < b.size(); ++j) {
      double bb[4]; b[j]->tlbr(bb);
      (*cost)[i * b.size() + j] = 1.0 - iou_plus1(ba, bb);
    }
  }
}

// matching.linear_assignment (matching.py:26-37) = lap.lapjv(extend_cost=True, cost_limit=thresh)
int linear_assignment(const std::vector<double>& cost, int nr, int nc, double thresh,
                      std::vector<std::pair<int, int>>* matches, std::vector<int>* ua, std::vector<int>* ub) {
  matches->clear(); ua->clear(); ub->clear();
  if (nr == 0 || nc == 0) {
    for (int i = 0; i < nr; ++i) ua->push_back(i);
    for (int j = 0; j < nc; ++j) ub->push_back(j);
    return 0;
  }
  const int n = nr + nc;
  std::vector<double> ext((size_t)n * n, thresh / 2.0);
  for (int i = nr; i < n; ++i)
    for (int j = nc; j < n; ++j) ext[(size_t)i * n + j] = 0.0;
  for (int i = 0; i < nr; ++i)
    for (int j = 0; j < nc; ++j) ext[(size_t)i * n + j] = cost[(size_t)i * nc + j];
  std::vector<int> rows, cols;
  if (lsap(ext.data(), n, n, &rows, &cols)) return 1;
  std::vector<int> x(nr, -1), y(nc, -1);
  for (size_t k = 0; k < rows.size(); ++k) {
    const int r = rows[k], c = cols[k];
    if (r < nr && c < nc) { x[r] = c; y[c] = r; }
  }
  for (int i = 0; i < nr; ++i) { if (x[i] >= 0) matches->emplace_back(i, x[i]); else ua->push_back(i); }
  for (int j = 0; j < nc; ++j) if (y[j] < 0) ub->push_back(j);
  return 0;
}

std::vector<STrkP> joint_stracks(const std::vector<STrkP>& a, const std::vector<STrkP>& b) {   // :345-356
  std::set<int> seen; std::vector<STrkP> r;
  for (auto& t : a) { seen.insert(t->id); r.push_back(t); }
  for (auto& t : b) if (!seen.count(t->id)) { seen.insert(t->id); r.push_back(t); }
  return r;
}

std::vector<STrkP> sub_stracks(const std::vector<STrkP>& a, const std::vector<STrkP>& b) {     // :358-367
  // dict keyed by track_id (insertion order kept; a later duplicate id replaces the VALUE in place)
  std::vector<int> order; std::map<int, STrkP> d;
  for (auto& t : a) { if (!d.count(t->id)) order.push_back(t->id); d[t->id] = t; }
  for (auto& t : b) d.erase(t->id);
  std::vector<STrkP> r;
  for (int id : order) { auto it = d.find(id); if (it != d.end()) r.push_back(it->second); }
  return r;
}

}  // namespace
}  // namespace odt

struct odt_tmot {
  double det_thresh, max_frame_lost, emb_max_dist, iou1, iou2;
  double alpha;
  int frame_id = 0;
  std::vector<odt::STrkP> tracked, lost, removed, output;
};

extern "C" {

int odt_tmot_create(double conf_thres, double track_max_second_lost, double emb_max_dist, double iou_max_dist1,
                    double iou_max_dist2, double emb_smooth_alpha, double frame_gap, double frame_rate,
                    odt_tmot_handle* out) {
  ODT_CHECK(out != nullptr && frame_gap > 0, "odt_tmot_create: bad argument");
  odt_tmot* t = new odt_tmot();
  t->det_thresh = conf_thres;
  t->max_frame_lost = track_max_second_lost * frame_rate / frame_gap;      // multitracker.py:187
  t->emb_max_dist = emb_max_dist; t->iou1 = iou_max_dist1; t->iou2 = iou_max_dist2;
  t->alpha = emb_smooth_alpha;
  *out = t;
  return 0;
}

int odt_tmot_destroy(odt_tmot_handle t) { delete t; return 0; }

int odt_tmot_reset(odt_tmot_handle t) {                                      // multitracker.py:198-206
  ODT_CHECK(t != nullptr, "odt_tmot_reset: null handle");
  t->tracked.clear(); t->lost.clear(); t->removed.clear(); t->output.clear();
  t->frame_id = 0;
  return 0;
}

// JDETracker.update (multitracker.py:208-343).  id_counter: BaseTrack._count, owned by the caller
// because the reference shares it between all tracker instances of the process.
int odt_tmot_update(odt_tmot_handle t, const double* tlwh, const double* conf, const float* feats, int n, int dim,
                    int* id_counter, int* n_out) {
  using namespace odt;
  ODT_CHECK(t != nullptr && id_counter != nullptr, "odt_tmot_update: null argument");
  ODT_CHECK(n == 0 || (tlwh && conf && feats && dim > 0), "odt_tmot_update: null detections");
  t->frame_id += 1;
  std::vector<STrkP> activated, refind, lost_now, removed_now;
  std::vector<STrkP> dets;
  for (int i = 0; i < n; ++i) {                                               // STrack.__init__
    STrkP d(new STrk());
    for (int k = 0; k < 4; ++k) { d->tlwh0[k] = tlwh[i * 4 + k]; d->det_tlwh[k] = tlwh[i * 4 + k]; }
    d->score = conf[i]; d->det_conf = conf[i]; d->alpha = t->alpha;
    update_features(*d, std::vector<float>(feats + (size_t)i * dim, feats + (size_t)(i + 1) * dim), nullptr);
    dets.push_back(d);
  }
  std::vector<STrkP> unconfirmed, tracked;
  for (auto& tr : t->tracked) (tr->activated ? tracked : unconfirmed).push_back(tr);
  // ---- step 2: first association, embedding + motion
  std::vector<STrkP> pool = joint_stracks(tracked, t->lost);
  for (auto& tr : pool) {                                                     // multi_predict (:53-66)
    if (tr->state != kTracked) tr->mean[7] = 0;
    kf_predict(tr->mean, tr->cov);
  }
  std::vector<double> cost((size_t)pool.size() * dets.size(), 0.0);
  const size_t nd = dets.size();
  if (!pool.empty() && nd) {
    for (size_t i = 0; i < pool.size(); ++i)                                  // matching.embedding_distance
      for (size_t j = 0; j < nd; ++j) {
        double s = 0.0;
        const std::vector<float>& a = pool[i]->smooth; const std::vector<float>& b = dets[j]->curr;
        for (size_t k = 0; k < a.size(); ++k) { const double d = (double)a[k] - (double)b[k]; s += d * d; }
        cost[i * nd + j] = std::max(0.0, std::sqrt(s));
      }
    std::vector<double> meas(nd * 4), gd(nd);                                 // matching.fuse_motion
    for (size_t j = 0; j < nd; ++j) { double b[4]; dets[j]->tlwh(b); tlwh_to_xyah(b, &meas[j * 4]); }
    for (size_t i = 0; i < pool.size(); ++i) {
      ODT_CHECK(kf_gating(pool[i]->mean, pool[i]->cov, meas.data(), (int)nd, gd.data()),
                "tmot: projected covariance is not positive definite");
      for (size_t j = 0; j < nd; ++j) {
        if (gd[j] > kChi2_4) cost[i * nd + j] = std::numeric_limits<double>::infinity();
        cost[i * nd + j] = 0.98 * cost[i * nd + j] + (1 - 0.98) * gd[j];
      }
    }
  }
  std::vector<std::pair<int, int>> matches; std::vector<int> u_track, u_det;
  if (linear_assignment(cost, (int)pool.size(), (int)nd, t->emb_max_dist, &matches, &u_track, &u_det)) return 1;
  auto do_update = [&](STrkP& tr, STrkP& det) {                              // STrack.update (:96-117)
    tr->frame_id = t->frame_id; tr->tracklet_len += 1;
    double z[4], b[4]; det->tlwh(b); tlwh_to_xyah(b, z);
    if (!kf_update(tr->mean, tr->cov, z)) return 1;
    tr->state = kTracked; tr->activated = true; tr->score = det->score;
    update_features(*tr, det->curr, &det->curr);
    for (int k = 0; k < 4; ++k) tr->det_tlwh[k] = det->det_tlwh[k];
    tr->det_conf = det->det_conf;
    return 0;
  };
  auto do_reactivate = [&](STrkP& tr, STrkP& det) {                          // STrack.re_activate (:80-94)
    double z[4], b[4]; det->tlwh(b); tlwh_to_xyah(b, z);
    if (!kf_update(tr->mean, tr->cov, z)) return 1;
    update_features(*tr, det->curr, &det->curr);
    tr->tracklet_len = 0; tr->state = kTracked; tr->activated = true; tr->frame_id = t->frame_id;
    for (int k = 0; k < 4; ++k) tr->det_tlwh[k] = det->det_tlwh[k];
    tr->det_conf = det->det_conf;
    return 0;
  };
  for (auto& m : matches) {
    STrkP& tr = pool[m.first]; STrkP& det = dets[m.second];
    if (tr->state == kTracked) { ODT_CHECK(!do_update(tr, det), "tmot: Kalman update failed"); activated.push_back(tr); }
    else { ODT_CHECK(!do_reactivate(tr, det), "tmot: Kalman update failed"); refind.push_back(tr); }
  }
  // ---- step 3: second association, IoU, tracked-but-unmatched only
  std::vector<STrkP> dets2; for (int j : u_det) dets2.push_back(dets[j]);
  std::vector<STrkP> r_tracked; for (int i : u_track) if (pool[i]->state == kTracked) r_tracked.push_back(pool[i]);
  iou_distance(r_tracked, dets2, &cost);
  if (linear_assignment(cost, (int)r_tracked.size(), (int)dets2.size(), t->iou1, &matches, &u_track, &u_det)) return 1;
  for (auto& m : matches) {
    STrkP& tr = r_tracked[m.first]; STrkP& det = dets2[m.second];
    if (tr->state == kTracked) { ODT_CHECK(!do_update(tr, det), "tmot: Kalman update failed"); activated.push_back(tr); }
    else { ODT_CHECK(!do_reactivate(tr, det), "tmot: Kalman update failed"); refind.push_back(tr); }
  }
  for (int i : u_track) {
    STrkP& tr = r_tracked[i];
    if (tr->state != kLost) { tr->state = kLost; lost_now.push_back(tr); }
  }
  // ---- unconfirmed tracks (one beginning frame)
  std::vector<STrkP> dets3; for (int j : u_det) dets3.push_back(dets2[j]);
  iou_distance(unconfirmed, dets3, &cost);
  std::vector<int> u_unc;
  if (linear_assignment(cost, (int)unconfirmed.size(), (int)dets3.size(), t->iou2, &matches, &u_unc, &u_det)) return 1;
  for (auto& m : matches) {
    ODT_CHECK(!do_update(unconfirmed[m.first], dets3[m.second]), "tmot: Kalman update failed");
    activated.push_back(unconfirmed[m.first]);
  }
  for (int i : u_unc) { unconfirmed[i]->state = kRemoved; removed_now.push_back(unconfirmed[i]); }
  // ---- step 4: new tracks
  for (int j : u_det) {
    STrkP& d = dets3[j];
    if (d->score < t->det_thresh) continue;
    d->id = ++(*id_counter);                                                  // activate (:68-78)
    double z[4]; tlwh_to_xyah(d->tlwh0, z);
    kf_initiate(z, d->mean, d->cov); d->has_kf = true;
    d->tracklet_len = 0; d->state = kTracked; d->frame_id = t->frame_id; d->start_frame = t->frame_id;
    activated.push_back(d);
  }
  // ---- step 5: state update
  for (auto& tr : t->lost)
    if (t->frame_id - tr->frame_id > t->max_frame_lost) { tr->state = kRemoved; removed_now.push_back(tr); }
  std::vector<STrkP> keep;
  for (auto& tr : t->tracked) if (tr->state == kTracked) keep.push_back(tr);
  t->tracked = joint_stracks(joint_stracks(keep, activated), refind);
  t->lost = sub_stracks(t->lost, t->tracked);
  t->lost.insert(t->lost.end(), lost_now.begin(), lost_now.end());
  t->lost = sub_stracks(t->lost, t->removed);
  t->removed.insert(t->removed.end(), removed_now.begin(), removed_now.end());
  {   // remove_duplicate_stracks (:369-383)
    iou_distance(t->tracked, t->lost, &cost);
    std::set<int> dupa, dupb;
    const size_t nl = t->lost.size();
    for (size_t p2 = 0; p2 < t->tracked.size(); ++p2)
      for (size_t q = 0; q < nl; ++q)
        if (cost[p2 * nl + q] < 0.15) {
          const int timep = t->tracked[p2]->frame_id - t->tracked[p2]->start_frame;
          const int timeq = t->lost[q]->frame_id - t->lost[q]->start_frame;
          if (timep > timeq) dupb.insert((int)q); else dupa.insert((int)p2);
        }
    std::vector<STrkP> ra, rb;
    for (size_t i = 0; i < t->tracked.size(); ++i) if (!dupa.count((int)i)) ra.push_back(t->tracked[i]);
    for (size_t i = 0; i < nl; ++i) if (!dupb.count((int)i)) rb.push_back(t->lost[i]);
    t->tracked.swap(ra); t->lost.swap(rb);
  }
  t->output.clear();
  for (auto& tr : t->tracked) if (tr->activated) t->output.push_back(tr);
  if (n_out) *n_out = (int)t->output.size();
  return 0;
}

// which: 0 output_stracks of the last update, 1 tracked_stracks, 2 lost_stracks, 3 removed_stracks
int odt_tmot_tracks(odt_tmot_handle t, int which, int cap, int32_t* ids, int32_t* state, int32_t* activated,
                    double* tlwh, double* det_tlwh, double* det_conf, double* score, int32_t* tracklet_len,
                    int32_t* start_frame, int32_t* frame_id, int* n) {
  ODT_CHECK(t != nullptr && n != nullptr && which >= 0 && which <= 3, "odt_tmot_tracks: bad argument");
  const std::vector<odt::STrkP>& v = which == 0 ? t->output : which == 1 ? t->tracked : which == 2 ? t->lost : t->removed;
  *n = (int)v.size();
  for (int i = 0; i < *n && i < cap; ++i) {
    const odt::STrk& k = *v[i];
    if (ids) ids[i] = k.id;
    if (state) state[i] = k.state;
    if (activated) activated[i] = k.activated ? 1 : 0;
    if (tlwh) k.tlwh(tlwh + i * 4);
    if (det_tlwh) std::memcpy(det_tlwh + i * 4, k.det_tlwh, 4 * sizeof(double));
    if (det_conf) det_conf[i] = k.det_conf;
    if (score) score[i] = k.score;
    if (tracklet_len) tracklet_len[i] = k.tracklet_len;
    if (start_frame) start_frame[i] = k.start_frame;
    if (frame_id) frame_id[i] = k.frame_id;
  }
  return 0;
}

}  // extern "C"
