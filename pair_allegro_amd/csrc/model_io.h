// Host-side model description + AHIP blob reader (no HIP, no torch).
#pragma once
#include <map>
#include <string>
#include <vector>

namespace ahip {

struct HostTensor {
  std::vector<int> shape;
  std::vector<double> data;
  long long numel() const { long long n = 1; for (int s : shape) n *= s; return n; }
};

struct HostModel {
  std::string model_dtype = "float32";
  std::vector<std::string> type_names;
  std::string type_names_joined;            // whitespace separated (reference metadata form)
  double r_max = 0;
  std::vector<double> per_edge_type_cutoff; // [T*T] model index, empty = r_max everywhere
  int num_types = 0, num_bessels = 0, poly_p = 0, l_max = 0, num_layers = 0;
  int S = 0, U = 0, mlp_depth = 0, mlp_width = 0, readout_depth = 0, readout_width = 0;
  double avg_num_neighbors = 1;
  int allow_tf32 = 0;                       // the reference's fifth metadata key (pair_nequip_allegro.cpp:267-270): the model file permits TF32-class matrix arithmetic
  long long seed = 0;
  std::map<std::string, HostTensor> tensors;

  const HostTensor &get(const std::string &name) const;   // throws std::runtime_error
};

// Reads `path` (.nequip.pth zip archive with an allegro_hip.bin member, or bare .ahip blob).
// Throws std::runtime_error with a user-facing message.
HostModel load_model_file(const std::string &path);

// The same model with its widths zero-padded to the fixed widths of a fused kernel (round 6): S scalar features -> SF, U tensor features -> UF (the columns of the
// (l, u) weight vectors move from l U + u to l UF + u), MLP width W -> WF, read-out width R -> RF.  Padded features are exact zeros all the way (no biases, silu(0) = 0,
// zero path weights, zero rows / columns in every linear), so the padded model computes the same energies and forces; needs S <= SF, U <= UF, W <= WF, R <= RF.
HostModel pad_host_model(const HostModel &h, int SF, int UF, int WF, int RF);

// Parses an in-memory blob.
HostModel parse_blob(const unsigned char *p, size_t n, const std::string &origin);

}  // namespace ahip
