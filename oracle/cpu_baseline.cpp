// ORACLE / CPU BASELINE HARNESS (test + measurement infrastructure, never linked into the product).
//
// Runs a contract-conforming `*.nequip.pth` TorchScript model through libtorch on the CPU with exactly the call sequence of
// the reference pair style, and times it the way SURVEY.md section 8d prescribes (3 warm-up + >= 10 timed evaluations, thread
// count stated, model-only and glue-inclusive figures):
//
//   torch::jit::load(path, device, metadata)        /root/reference/pair_nequip_allegro.cpp:214-222
//   eval(); hasattr("training") -> freeze            :225-231
//   preprocess(): count, prefix sum, fill            :488-519, :566-629   (oracle/glue_oracle.c, linked in)
//   tensors: pos f64 [N,3], edge_index i64 [2,E], atom_types i64 [N]                 :524-533, :638-641
//   forward(vector<IValue>{Dict}) -> GenericDict -> Dict<string,Tensor>              :419-430
//   scatter: f += forces, eng = sum local atomic_energy, virial unpack               :358-393
//
// Input: a flat binary system file (the format of tests/test_lammps_cpp.py::_write_system: header {nlocal, nghost, ntypes,
// nneigh}, x f64 [nall][3], type i32 [nall], tag i32 [nall], numneigh i32 [nall], flat neighbour list i32 [nneigh]), the model
// path, and the LAMMPS type names in deck order.  Output: one JSON line on stdout, and (optionally) forces / energies / virial to
// a binary file for a parity check.
#include <torch/script.h>
#include <torch/torch.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

extern "C" {
long long ref_count_edges(int nlocal, const int *ilist, const int *numneigh, const int *const *firstneigh, const double *x,
                          const int *type, int ntypes, const double *cutoff_matrix, int *neigh_per_atom);
void ref_prefix_sum(int nlocal, const int *neigh_per_atom, int *cumsum);
void ref_fill_edges(int nlocal, int ntotal, const int *ilist, const int *numneigh, const int *const *firstneigh, const double *x,
                    const int *type, int ntypes, const double *cutoff_matrix, const int *type_mapper, const int *cumsum,
                    long long nedges, double *pos, long long *edges, long long *atom_types);
double ref_scatter(int inum, int ntotal, const int *ilist, const double *forces, const double *atomic_energies, int eflag_atom,
                   double *f, double *eatom);
void ref_virial_unpack(const double *v, double *virial);
}

template <typename T> static std::vector<T> rd(FILE *f, size_t n) {
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != n) { perror("read"); exit(3); }
  return v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: cpu_baseline system.bin model.nequip.pth [--out f.bin] [--warmup 3] [--reps 10] [--budget 20] [--threads N] names...\n"); return 2; }
  std::string sysf = argv[1], model_path = argv[2], outf;
  int warmup = 3, min_reps = 10, threads = 0;
  double budget = 20.0;
  std::vector<std::string> names;
  for (int k = 3; k < argc; ++k) {
    std::string a = argv[k];
    if (a == "--out" && k + 1 < argc) outf = argv[++k];
    else if (a == "--warmup" && k + 1 < argc) warmup = atoi(argv[++k]);
    else if (a == "--reps" && k + 1 < argc) min_reps = atoi(argv[++k]);
    else if (a == "--budget" && k + 1 < argc) budget = atof(argv[++k]);
    else if (a == "--threads" && k + 1 < argc) threads = atoi(argv[++k]);
    else names.push_back(a);
  }
  // intra-op threads: the host's full core count oversubscribes this model badly (128 threads: 500 ms for 64 atoms, 8 threads: 30 ms),
  // so bench.py sweeps the count and reports the best (the reference leaves it to OMP_NUM_THREADS, README.md:142-146)
  if (threads > 0) at::set_num_threads(threads);
  FILE *f = fopen(sysf.c_str(), "rb");
  int hdr[4];
  if (!f || fread(hdr, sizeof(int), 4, f) != 4) { fprintf(stderr, "cannot read %s\n", sysf.c_str()); return 3; }
  const int nlocal = hdr[0], nghost = hdr[1], ntypes = hdr[2], nneigh = hdr[3], ntotal = nlocal + nghost;
  auto x = rd<double>(f, (size_t)ntotal * 3);
  auto type = rd<int>(f, ntotal);
  auto tag = rd<int>(f, ntotal);
  auto numneigh = rd<int>(f, ntotal);
  auto flat = rd<int>(f, nneigh);
  fclose(f);
  if ((int)names.size() != ntypes) { fprintf(stderr, "need %d type names\n", ntypes); return 2; }
  std::vector<const int *> first(ntotal);
  std::vector<int> ilist(ntotal);
  size_t off = 0;
  for (int i = 0; i < ntotal; i++) { first[i] = flat.data() + off; off += numneigh[i]; ilist[i] = i; }

  // ---- coeff(): load + metadata + type mapping (:214-232, :267-328) ----
  std::unordered_map<std::string, std::string> metadata = {{"r_max", ""}, {"per_edge_type_cutoff", ""}, {"type_names", ""}, {"num_types", ""}, {"allow_tf32", ""}};
  torch::Device device = torch::kCPU;
  torch::jit::Module model = torch::jit::load(model_path, device, metadata);
  model.eval();
  if (model.hasattr("training")) model = torch::jit::freeze(model);
  const double cutoff = std::stod(metadata["r_max"]);
  std::vector<std::string> model_names;
  { std::stringstream ss(metadata["type_names"]); std::string t; while (ss >> t) model_names.push_back(t); }
  std::vector<int> type_mapper(ntypes, -1);
  for (size_t i = 0; i < model_names.size(); i++)
    for (int it = 0; it < ntypes; it++)
      if (model_names[i] == names[it]) type_mapper[it] = (int)i;
  std::vector<double> cutoff_matrix((size_t)ntypes * ntypes, cutoff);
  if (!metadata["per_edge_type_cutoff"].empty()) {
    std::vector<double> pc;
    { std::stringstream ss(metadata["per_edge_type_cutoff"]); double v; while (ss >> v) pc.push_back(v); }
    const int T = (int)model_names.size();
    for (int a = 0; a < ntypes; a++)
      for (int b = 0; b < ntypes; b++)
        if (type_mapper[a] >= 0 && type_mapper[b] >= 0) cutoff_matrix[(size_t)a * ntypes + b] = pc[(size_t)type_mapper[a] * T + type_mapper[b]];
  }

  std::vector<double> fr((size_t)ntotal * 3, 0.0), eatom(ntotal, 0.0);
  double eng = 0, virial[6] = {0, 0, 0, 0, 0, 0};
  long long nedges = 0;
  double t_model = 0, t_total = 0;
  int reps = 0;
  const double t_start = now();
  for (int it = 0;; ++it) {
    const bool timed = it >= warmup;
    if (timed && reps >= min_reps) break;
    if (timed && reps >= 3 && now() - t_start > budget) break;
    const double t0 = now();
    // ---- preprocess (:457-650) ----
    std::vector<int> npa(nlocal), cumsum(nlocal);
    nedges = ref_count_edges(nlocal, ilist.data(), numneigh.data(), first.data(), x.data(), type.data(), ntypes, cutoff_matrix.data(), npa.data());
    ref_prefix_sum(nlocal, npa.data(), cumsum.data());
    torch::Tensor pos_t = torch::zeros({ntotal, 3}, torch::kFloat64);
    torch::Tensor edges_t = torch::zeros({2, nedges}, torch::kInt64);
    torch::Tensor types_t = torch::zeros({ntotal}, torch::kInt64);
    ref_fill_edges(nlocal, ntotal, ilist.data(), numneigh.data(), first.data(), x.data(), type.data(), ntypes, cutoff_matrix.data(),
                   type_mapper.data(), cumsum.data(), nedges, pos_t.data_ptr<double>(), (long long *)edges_t.data_ptr<int64_t>(),
                   (long long *)types_t.data_ptr<int64_t>());
    c10::Dict<std::string, torch::Tensor> input;
    input.insert("pos", pos_t.to(device));
    input.insert("edge_index", edges_t.to(device));
    input.insert("atom_types", types_t.to(device));
    // ---- call (:409-430) ----
    const double t1 = now();
    std::vector<torch::IValue> input_vector(1, input);
    auto generic = model.forward(input_vector).toGenericDict();
    c10::Dict<std::string, torch::Tensor> output;
    for (const auto &item : generic) output.insert(item.key().toStringRef(), item.value().toTensor());
    const double t2 = now();
    // ---- scatter (:358-393) ----
    torch::Tensor forces = output.at("forces").cpu().contiguous();
    torch::Tensor ae = output.at("atomic_energy").cpu().contiguous();
    torch::Tensor v = output.at("virial").cpu().contiguous();
    std::fill(fr.begin(), fr.end(), 0.0);
    eng = ref_scatter(nlocal, ntotal, ilist.data(), forces.data_ptr<double>(), ae.data_ptr<double>(), 1, fr.data(), eatom.data());
    ref_virial_unpack(v.data_ptr<double>(), virial);
    const double t3 = now();
    if (timed) { t_model += t2 - t1; t_total += t3 - t0; ++reps; }
  }
  if (!outf.empty()) {
    FILE *o = fopen(outf.c_str(), "wb");
    fwrite(&eng, sizeof(double), 1, o);
    fwrite(virial, sizeof(double), 6, o);
    fwrite(fr.data(), sizeof(double), fr.size(), o);
    fwrite(eatom.data(), sizeof(double), eatom.size(), o);
    fclose(o);
  }
  printf("{\"nlocal\": %d, \"nghost\": %d, \"nedges\": %lld, \"threads\": %d, \"warmup\": %d, \"reps\": %d, \"ms_model\": %.3f, \"ms_total\": %.3f, \"eng\": %.12g}\n",
         nlocal, nghost, nedges, at::get_num_threads(), warmup, reps, 1e3 * t_model / reps, 1e3 * t_total / reps, eng);
  return 0;
}
