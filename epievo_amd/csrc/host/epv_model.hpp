// epv_model.hpp -- host-side model parameters and M-step (O(8) work; stays on the
// host by design).
//
// Mirrors the reference's EpiEvoModel (src/libepievo/EpiEvoModel.hpp:31-66) and the
// M-step entry points of src/libepievo/ParamEstimation.hpp:37-81, re-stated over
// plain arrays so that the same values can cross the C ABI unchanged:
//   T, Q, baseline : row-major 2x2 {00,01,10,11}
//   rates          : the 8 triplet rates indexed 4*l + 2*m + r
//   J, D           : per-branch sufficient statistics, [(b-1)*8 + ctx], b = 1..n_nodes-1
#ifndef EPV_MODEL_HPP
#define EPV_MODEL_HPP

#include <array>
#include <string>
#include <vector>

namespace epv {

struct Model {
  std::array<double, 4> T{};         // stationary horizontal transition probs
  std::array<double, 4> Q{};         // pair-wise potentials
  std::array<double, 4> baseline{};  // stationary log baseline
  std::array<double, 8> rates{};     // triplet rates

  // EpiEvoModel.cpp:319-370 (both the "stationary/baseline" and the 8-line
  // "000 <rate>" forms); throws std::runtime_error when the file cannot be opened
  static Model read(const std::string &param_file);
  // EpiEvoModel.cpp:372-377
  void scale_triplet_rates();
  // EpiEvoModel.cpp:420-449
  void rebuild_from_triplet_rates(const std::array<double, 8> &updated);
  // EpiEvoModel.cpp:192-200 (default stream precision, no trailing newline)
  std::string format_for_param_file() const;
  // EpiEvoModel.cpp:281-298: first-order Markov chain root sequence
  // (the uniform draws are supplied by the caller)
};

// EpiEvoModel.cpp:173-189: expected changes per site per unit time
double rate_scaling_factor(const std::array<double, 8> &rates);

// ParamEstimation.cpp:318-334
void set_one_change_per_site_per_unit_time(std::array<double, 8> &rates,
                                           std::vector<double> &branches);

// ParamEstimation.cpp:337-353 (J/D per branch -> collapsed -> gradient ascent);
// updates the model, returns the log-likelihood
double estimate_rates(double param_tol, int n_nodes, const double *J, const double *D,
                      Model &model);

// ParamEstimation.cpp:383-422; `branches` (size n_nodes, index 0 = root) is updated
double estimate_rates_and_branches(double param_tol, int n_nodes, const double *J,
                                   const double *D, std::vector<double> &branches,
                                   Model &model);

}  // namespace epv

#endif
