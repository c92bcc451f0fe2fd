// epv_model.cpp -- see epv_model.hpp.  Every routine follows the arithmetic order of
// the reference function it cites so that the written param file matches to the bit
// (tests/test_host_model.py pins this against the linked reference).
#include "epv_model.hpp"

#include <cassert>
#include <cmath>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>

namespace epv {

namespace {

inline double &at(std::array<double, 4> &m, int r, int c) { return m[2 * r + c]; }
inline double at(const std::array<double, 4> &m, int r, int c) { return m[2 * r + c]; }

// EpiEvoModel.cpp:79-92: Gibbs pair-wise potentials from T
void potential_from_T(const std::array<double, 4> &T, std::array<double, 4> &Q) {
  Q = T;
  at(Q, 0, 0) = 1.0 - at(T, 0, 1);
  at(Q, 0, 1) = std::sqrt(at(T, 0, 1) * at(T, 1, 0));
  at(Q, 1, 0) = at(Q, 0, 1);
  at(Q, 1, 1) = 1.0 - at(T, 1, 0);
}

// EpiEvoModel.cpp:98-108: potentials (up to scale) from the rates, phi(0,1) = 0
void potential_from_rates(const std::array<double, 8> &r, std::array<double, 4> &Q) {
  Q = {1.0, 1.0, 1.0, 1.0};
  const double death_birth = r[2] / r[0];
  const double expand_contract = r[1] / r[3];
  at(Q, 0, 0) = at(Q, 0, 1) * std::sqrt(death_birth);
  at(Q, 1, 1) = at(Q, 0, 1) * std::sqrt(death_birth) * expand_contract;
}

// EpiEvoModel.cpp:111-132
void T_from_potential(const std::array<double, 4> &Q, std::array<double, 4> &T) {
  const double d = at(Q, 0, 0) - at(Q, 1, 1);
  const double delta = std::sqrt(d * d + 4 * at(Q, 0, 1) * at(Q, 1, 0));
  T = Q;
  const double diag_denom = at(Q, 0, 0) + at(Q, 1, 1) + delta;
  at(T, 1, 1) = 2 * at(Q, 1, 1) / diag_denom;
  at(T, 0, 0) = 2 * at(Q, 0, 0) / diag_denom;
  at(T, 0, 1) = 1.0 - at(T, 0, 0);
  at(T, 1, 0) = 1.0 - at(T, 1, 1);
}

// EpiEvoModel.cpp:390-411
void rates_from_potential(const std::array<double, 4> &Q, const std::array<double, 4> &bl,
                          std::array<double, 8> &r) {
  double e = std::exp(at(bl, 0, 0));
  r[0] = at(Q, 0, 1) * at(Q, 1, 0) * e;  // 000
  r[2] = at(Q, 0, 0) * at(Q, 0, 0) * e;  // 010
  e = std::exp(at(bl, 0, 1));
  r[1] = at(Q, 0, 1) * at(Q, 1, 1) * e;  // 001
  r[3] = at(Q, 0, 0) * at(Q, 0, 1) * e;  // 011
  e = std::exp(at(bl, 1, 0));
  r[4] = at(Q, 1, 1) * at(Q, 1, 0) * e;  // 100
  r[6] = at(Q, 1, 0) * at(Q, 0, 0) * e;  // 110
  e = std::exp(at(bl, 1, 1));
  r[5] = at(Q, 1, 1) * at(Q, 1, 1) * e;  // 101
  r[7] = at(Q, 1, 0) * at(Q, 0, 1) * e;  // 111
}

// ParamEstimation.cpp:131-143
double log_likelihood(const double *J, const double *D, const std::array<double, 8> &rates) {
  double ll = 0;
  for (int i = 0; i < 8; ++i) ll += J[i] * std::log(rates[i]) - D[i] * rates[i];
  return ll;
}

// ParamEstimation.cpp:147-184: gradient w.r.t. log-rates under the constraints
// l100 = l001, l110 = l011, l111 = l000 l101 l011^2 / (l010 l001^2)
void gradient_of(const double *J, const double *D, const std::array<double, 8> &r,
                 std::array<double, 8> &g) {
  g.fill(0.0);
  const double f111 = J[7] - D[7] * r[7];
  g[0] += J[0] - D[0] * r[0] + f111;
  g[2] += J[2] - D[2] * r[2] - f111;
  g[1] += J[1] + J[4] - (D[1] + D[4]) * r[1] - 2 * f111;
  g[4] = g[1];
  g[3] += J[3] + J[6] - (D[3] + D[6]) * r[3] + 2 * f111;
  g[6] = g[3];
  g[5] += J[5] - D[5] * r[5] + f111;
}

// ParamEstimation.cpp:200-218
void candidate_rates(double step, const std::array<double, 8> &g,
                     const std::array<double, 8> &r, std::array<double, 8> &u) {
  for (int i = 0; i < 7; ++i) u[i] = std::exp(std::log(r[i]) + g[i] * step);
  u[7] = std::exp(std::log(u[0]) + std::log(u[5]) + 2 * std::log(u[3]) - std::log(u[2]) -
                  2 * std::log(u[1]));
}

// ParamEstimation.cpp:256-276: one step of projected gradient ascent with step halving
bool gradient_ascent(double tol, const double *J, const double *D, double llh,
                     const std::array<double, 8> &r, double &new_llh,
                     std::array<double, 8> &new_r) {
  std::array<double, 8> g;
  gradient_of(J, D, r, g);
  double l1 = 0.0;
  for (double x : g) l1 += std::fabs(x);
  double step = 1.0 / l1;
  new_llh = std::numeric_limits<double>::lowest();
  while (new_llh < llh && step > tol) {
    candidate_rates(step, g, r, new_r);
    new_llh = log_likelihood(J, D, new_r);
    step *= 0.5;
  }
  return new_llh > llh;
}

// ParamEstimation.cpp:279-315: collapse branches, iterate to convergence
double fit_rates(double tol, int n_nodes, const double *J, const double *D,
                 const std::array<double, 8> &in, std::array<double, 8> &out) {
  double Jc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, Dc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = 1; b < n_nodes; ++b)
    for (int i = 0; i < 8; ++i) {
      Jc[i] += J[(b - 1) * 8 + i];
      Dc[i] += D[(b - 1) * 8 + i];
    }
  double llh = log_likelihood(Jc, Dc, in);
  out = in;
  std::array<double, 8> tmp = out;
  double tmp_llh = llh;
  while (gradient_ascent(tol, Jc, Dc, llh, out, tmp_llh, tmp)) {
    llh = tmp_llh;
    out.swap(tmp);
  }
  return llh;
}

}  // namespace

double rate_scaling_factor(const std::array<double, 8> &rates) {
  std::array<double, 4> Q, T;
  potential_from_rates(rates, Q);
  T_from_potential(Q, T);
  double pi[2];
  pi[1] = (1.0 - at(T, 0, 0)) / (2.0 - at(T, 0, 0) - at(T, 1, 1));
  pi[0] = 1.0 - pi[1];
  double mu = 0.0;
  for (int i = 0; i < 8; ++i) {
    const int l = (i >> 2) & 1, m = (i >> 1) & 1, r = i & 1;
    mu += pi[l] * at(T, l, m) * at(T, m, r) * rates[i];
  }
  return mu;
}

Model Model::read(const std::string &param_file) {
  std::ifstream in(param_file);
  if (!in) throw std::runtime_error("Could not open file: " + param_file);
  Model m;
  std::string label;
  in >> label;
  if (label == "stationary") {
    in >> at(m.T, 0, 0) >> at(m.T, 1, 1);
    at(m.T, 1, 0) = 1.0 - at(m.T, 1, 1);
    at(m.T, 0, 1) = 1.0 - at(m.T, 0, 0);
    in >> label;
    if (label != "baseline") throw std::runtime_error("bad param file: " + param_file);
    in >> at(m.baseline, 0, 0) >> at(m.baseline, 1, 1);
    potential_from_T(m.T, m.Q);
    rates_from_potential(m.Q, m.baseline, m.rates);
  } else {
    if (label != "000") throw std::runtime_error("bad param file: " + param_file);
    std::array<double, 8> r;
    in >> r[0];
    for (int i = 1; i < 8; ++i) in >> label >> r[i];
    r[4] = r[1];
    r[6] = r[3];
    r[7] = (r[0] * r[6] * r[6] * r[5]) / (r[2] * r[4] * r[4]);
    m.rebuild_from_triplet_rates(r);
  }
  return m;
}

void Model::scale_triplet_rates() {
  const double mu = rate_scaling_factor(rates);
  for (double &r : rates) r /= mu;
}

void Model::rebuild_from_triplet_rates(const std::array<double, 8> &updated) {
  assert(updated[1] == updated[4] && updated[3] == updated[6]);
  rates = updated;
  std::array<double, 4> Qp;
  potential_from_rates(rates, Qp);
  T_from_potential(Qp, T);
  potential_from_T(T, Q);
  const double log_Q01 = std::log(at(Q, 0, 1));
  const double log_Q10 = std::log(at(Q, 1, 0));
  const double log_Q11 = std::log(at(Q, 1, 1));
  at(baseline, 0, 0) = std::log(rates[0]) - (log_Q01 + log_Q10);
  at(baseline, 0, 1) = std::log(rates[1]) - (log_Q01 + log_Q11);
  at(baseline, 1, 0) = std::log(rates[4]) - (log_Q11 + log_Q10);
  at(baseline, 1, 1) = std::log(rates[7]) - (log_Q10 + log_Q01);
  const double centre = at(baseline, 0, 1);
  for (double &b : baseline) b -= centre;
}

std::string Model::format_for_param_file() const {
  std::ostringstream oss;
  oss << "stationary\t" << at(T, 0, 0) << '\t' << at(T, 1, 1) << std::endl
      << "baseline\t" << at(baseline, 0, 0) << '\t' << at(baseline, 1, 1);
  return oss.str();
}

void set_one_change_per_site_per_unit_time(std::array<double, 8> &rates,
                                           std::vector<double> &branches) {
  const double f = rate_scaling_factor(rates);
  for (double &b : branches) b = b * f;
  for (double &r : rates) r = r / f;
}

double estimate_rates(double param_tol, int n_nodes, const double *J, const double *D,
                      Model &model) {
  std::array<double, 8> updated;
  const double llh = fit_rates(param_tol, n_nodes, J, D, model.rates, updated);
  model.rebuild_from_triplet_rates(updated);
  return llh;
}

double estimate_rates_and_branches(double param_tol, int n_nodes, const double *J,
                                   const double *D, std::vector<double> &branches,
                                   Model &model) {
  std::array<double, 8> updated;
  fit_rates(param_tol, n_nodes, J, D, model.rates, updated);
  // ParamEstimation.cpp:224-240: branch scale = sum_c J_c / sum_c D_c * rate_c
  std::vector<double> scale(branches.size(), 1.0);
  for (int b = 1; b < n_nodes; ++b) {
    double num = 0.0, denom = 0.0;
    for (int i = 0; i < 8; ++i) num += J[(b - 1) * 8 + i];
    for (int i = 0; i < 8; ++i) denom += D[(b - 1) * 8 + i] * updated[i];
    scale[b] = num / denom;
  }
  std::vector<double> new_branches(branches.size());
  for (size_t b = 0; b < branches.size(); ++b) new_branches[b] = scale[b] * branches[b];
  set_one_change_per_site_per_unit_time(updated, new_branches);
  model.rebuild_from_triplet_rates(updated);
  branches = new_branches;
  double Jc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, Dc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = 1; b < n_nodes; ++b)
    for (int i = 0; i < 8; ++i) {
      Jc[i] += J[(b - 1) * 8 + i];
      Dc[i] += scale[b] * D[(b - 1) * 8 + i];
    }
  return log_likelihood(Jc, Dc, updated);
}

}  // namespace epv
