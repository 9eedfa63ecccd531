/* allegro_hip.h -- C-ABI of liballegro_hip.so: the MI355X-native `pair_style allegro` hot path.
 *
 * Drop-in boundary.  The reference pair style (mir-group/pair_allegro,
 * pair_nequip_allegro.{h,cpp}) does, per MPI rank:
 *
 *     coeff()      : torch::jit::load(model) + metadata            pair_nequip_allegro.cpp:174-330
 *     compute()    : preprocess() -> call() -> scatter             pair_nequip_allegro.cpp:333-407
 *
 * with preprocess()/call() living in libtorch.  This library replaces everything below
 * `Pair::coeff` / `Pair::compute`: the model file reader, the cutoff filter of the LAMMPS
 * (skin-inflated) full neighbor list, the Allegro model itself (hand-written HIP kernels,
 * analytic backward) and the force / energy / virial read-out.  A LAMMPS `Pair` subclass only
 * marshals pointers into these entry points (pair_allegro_amd/lammps/pair_allegro_hip.cpp;
 * binding recipe in INTEGRATION.md).
 *
 * Conventions: every function returns 0 on success, non-zero on failure; the message is then
 * available from ahip_last_error() (thread-local).  Nothing here throws, exits or prints
 * (except the explicit debug dump).  Plain pointers and sizes only, no torch / LAMMPS types.
 * Pointers are HOST pointers unless the function name ends in `_dev`.
 * There is NO CPU fallback: every entry point that computes needs a gfx950 device.
 */
#ifndef ALLEGRO_HIP_H
#define ALLEGRO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ahip_model ahip_model;

#define AHIP_OK 0
#define AHIP_ERR_ARG 1      /* bad argument / deck error  (-> LAMMPS error->all)            */
#define AHIP_ERR_FILE 2     /* model file problems (reference throws std::runtime_error, :205) */
#define AHIP_ERR_DEVICE 3   /* HIP runtime failure / no GPU                                  */
#define AHIP_ERR_STATE 4    /* call order (compute before neigh_update, ...)                */
#define AHIP_ERR_UNSUPPORTED 5

/* Message of the last failure on this thread ("" if none). */
const char *ahip_last_error(void);

/* Number of visible HIP devices; replaces torch::cuda::is_available()/device_count()
 * (pair_nequip_allegro.cpp:92,102).  The caller maps node-local rank -> device index exactly
 * like pair_nequip_allegro.cpp:93-120 (modulo wrap-around only in debug mode). */
int ahip_device_count(int *count);

/* Load `*.nequip.pth` (TorchScript archive carrying the `allegro_hip.bin` section) or a bare
 * `*.ahip` blob and upload the weights to `device`.  Replaces torch::jit::load + freeze
 * (pair_nequip_allegro.cpp:214-232).  Any other extension -> AHIP_ERR_FILE, like :197-206. */
int ahip_model_load(const char *path, int device, ahip_model **out);
void ahip_model_free(ahip_model *m);

/* Model metadata = the five keys the reference reads from the archive
 * (pair_nequip_allegro.cpp:214-220; consumed :267-328).  Any out-pointer may be NULL.
 *   type_names: whitespace-separated, model-type order (:275-294)
 *   per_edge_type_cutoff: row-major [num_types][num_types] in MODEL type index, or NULL when the
 *                         model has a single r_max (:303-328)
 * Returned pointers live as long as the model. */
int ahip_model_meta(const ahip_model *m, double *r_max, int *num_types, const char **type_names,
                    const double **per_edge_type_cutoff, int *l_max, int *num_tensor_features,
                    int *num_scalar_features, int *num_layers, const char **model_dtype);

/* Options (string key/value):
 *   "path"      = "auto" | "fused" | "generic"   kernel family (auto: fused when the model shape
 *                                                 and the per-atom edge counts allow it)
 *   "precision" = "model" | "float64"            compute dtype; float64 is the debug/parity build
 *   "chunk_edges" = "<n>"                        max edges processed per pass (workspace bound)
 *   "reserve_wgs" = "<n>"                        half-CU workgroup slots the persistent fused kernels leave unoccupied, so that
 *                                                 kernels launched on OTHER streams (ghost exchange) can run beside them (default 0)
 *   "cutoff_compare" = "le" | "lt"               edge kept iff rsq <= cut^2 (default; pair_nequip_allegro.cpp:507) or rsq < cut^2
 *                                                 (the KOKKOS path of the reference, pair_nequip_allegro_kokkos.cpp:174)
 *   "fused_arith" = "auto" | "f32" | "f16x2" | "bf16x3" | "tf32eq"   arithmetic of the dense contractions of the fused kernels: f32-input MFMA (exact float32
 *                                                 fmaf chains); two float16 terms per operand, three f16-MFMA products (float32-equivalent inside float16's exponent
 *                                                 range; an evaluation that leaves it returns AHIP_ERR_STATE: host-pointer calls at once, _dev calls at the next
 *                                                 evaluation); three-term bf16 split (float32-equivalent; l_max = 1 kernel); two-term bf16 split (TF32-class; l_max = 1
 *                                                 kernel).  auto = f16x2 unless the model file says allow_tf32 = 1 (pair_nequip_allegro.cpp:267-270), then tf32eq.
 *                                                 auto is never less robust than the reference's float32 (which only ever RELAXES precision, :267-270): a model the split
 *                                                 cannot carry (a weight beyond float16's range, a linear inside its subnormals), a first evaluation that disagrees with
 *                                                 the float32 instance by more than 1e-5 max|F|, or a float16-range alarm switch the model to the f32 instance for the
 *                                                 rest of its life -- host-pointer calls re-evaluate inside the same call, _dev calls report the invalid earlier
 *                                                 evaluation once (AHIP_ERR_STATE) and continue on f32.  ahip_arith_note says what was decided.  Only an explicit
 *                                                 "f16x2" keeps the hard errors.
 *   "fused_tb"  = "table" | "mlp"                two-body embedding of the fused kernels: tabulated cubic splines (default) or the MLP itself
 *   "edge_schedule" = "auto" | "static" | "dynamic"   unit schedule of the single-pass edge build (dynamic: safe beside other resident kernels)
 *   "tile_pack" = "auto" | "separate" | "fused"  tile packing of the fused kernels inside the edge build or as its own kernels
 *   "timing"    = "0" | "1"                      record per-stage HIP events (ahip_get_timings)
 * Unknown keys and values are errors (AHIP_ERR_ARG).
 */
int ahip_set_option(ahip_model *m, const char *key, const char *value);

/* Hand over the LAMMPS full neighbor list -- only on steps where LAMMPS rebuilt it.
 * Replaces the list walk of preprocess() (pair_nequip_allegro.cpp:469-480,489-496).
 *   inum      : number of centre atoms (== nlocal, asserted at :470)
 *   nall      : nlocal + nghost (size of x / f / type)
 *   ilist     : [inum] centre atom indices
 *   numneigh  : indexed by ATOM index i (LAMMPS convention), numneigh[i]
 *   firstneigh: indexed by atom index i, firstneigh[i][0..numneigh[i])
 *   neighmask : LAMMPS NEIGHMASK; every j is stored as (j & neighmask) (:496)
 */
int ahip_neigh_update(ahip_model *m, int inum, int nall, const int *ilist, const int *numneigh,
                      const int *const *firstneigh, int neighmask);

/* Same, from a flat CSR list: neighbours of centre ii are neigh[offsets[ii] .. offsets[ii+1]). */
int ahip_neigh_update_csr(ahip_model *m, int inum, int nall, const int *ilist,
                          const long long *offsets, const int *neigh, int neighmask);
/* Same, CSR arrays already on the device (int32 offsets [inum+1]); no copy is made of `neigh`
 * -- the caller keeps it alive until the next update.  Synchronises the device once (the longest row is measured here: it bounds
 * every centre's degree until the next hand-over, which is what lets ahip_compute_dev* run without any device -> host read-back).
 * The CSR arrays MUST NOT change between hand-overs: a row that has grown past the measured bound is detected (the edge build's overflow word is
 * inspected, without waiting, by the following calls and by every accessor) and reported as AHIP_ERR_STATE by a LATER call -- the forces of the
 * evaluation that met it are invalid. */
int ahip_neigh_update_dev(ahip_model *m, int inum, int nall, const int *ilist_dev,
                          const int *offsets_dev, const int *neigh_dev, long long nneigh_total);

/* Same, from the device-resident list of the LAMMPS KOKKOS package: a padded 2-D table d_neighbors(i, jj) addressed as
 * neighbors_dev[i * stride_atom + jj * stride_slot] (column-major on a GPU build: stride_atom = 1), numneigh_dev indexed by ATOM
 * index, ilist_dev [inum].  Replaces the per-step table walk of pair_nequip_allegro_kokkos.cpp:128-131,142-181 (there every
 * step; here once per list rebuild: the table is compacted on the device into the library's own CSR rows, nothing is kept of
 * the caller's arrays).  Synchronises `stream` (two 4-byte read-backs). */
int ahip_neigh_update_dev_table(ahip_model *m, int inum, int nall, const int *ilist_dev, const int *numneigh_dev,
                                const int *neighbors_dev, long long stride_atom, long long stride_slot, int neighmask,
                                void *stream);

/* LAMMPS types (1-based, device) -> model types (device) through the pair_coeff mapping `type_mapper` [ntypes] (host, -1 =
 * unmapped); what pair_nequip_allegro_kokkos.cpp:222-223 does every step.  Fails if an atom carries an unmapped type.
 * Synchronises `stream`; callers run it on list-rebuild steps (the type of an atom index cannot change in between). */
int ahip_map_types_dev(ahip_model *m, int n, const int *type_dev, int ntypes, const int *type_mapper, int *mtype_dev,
                       void *stream);

/* One force evaluation = PairNequIPAllegro<false>::compute (pair_nequip_allegro.cpp:333-407).
 *   x            [nall][3] f64 positions, locals first then ghosts (atom->x)
 *   type         [nall] LAMMPS types, 1-based (atom->type)
 *   ntypes       atom->ntypes
 *   type_mapper  [ntypes] LAMMPS type-1 -> model type, -1 = unmapped (:274-294)
 *   cutoff_matrix[ntypes*ntypes] in LAMMPS type index (:303-328); edge kept iff rsq <= cut^2 (:507)
 *   f            [nall][3]  forces are ADDED (f[i] += ...) for locals AND ghosts (:370-377)
 *   eatom        [nall] or NULL; eatom[i] = E_i is written for the inum centre atoms (:378)
 *   eng          out: sum of E_i over centre atoms only (:379)
 *   virial       out or NULL: xx,yy,zz,xy,xz,yz, no sign change (:387-392)
 */
int ahip_compute(ahip_model *m, int nlocal, int nghost, const double *x, const int *type, int ntypes,
                 const int *type_mapper, const double *cutoff_matrix, double *f, double *eatom,
                 double *eng, double *virial);

/* Device-resident variant (the pattern of pair_nequip_allegro_kokkos.cpp:109-126,266-268,305-319):
 * x_dev/f_dev/eatom_dev are device pointers, mtype_dev holds MODEL types (already mapped),
 * cutoff_matrix_model is a HOST [num_types^2] matrix in model-type index (NULL = r_max).
 * Forces are accumulated into f_dev; eng_vir_dev receives 7 doubles {eng, xx,yy,zz,xy,xz,yz}.
 * Asynchronous on `stream` (a hipStream_t, NULL = default).  The reference's Kokkos path reads its edge total back in every step
 * (:203-206); here the counters of the edge build (edge total, largest degree, centres left to the layer-at-a-time kernels) travel to
 * page-locked memory behind the edge build and the host waits for them only when it needs a value: never on the l_max = 1 fused path with
 * list rows of at most 128 entries (the row length measured at ahip_neigh_update* bounds every degree; tile shape and tile count are
 * decided on the device), after the model kernel has been enqueued on the wide fused paths (number of centres with more than 64 edges),
 * before the model on the layer-at-a-time path, and in the getters (ahip_get_edges, ahip_last_max_degree, ahip_last_tile_occupancy).
 * With outputs registered through ahip_output_register the call additionally keeps host copies of them
 * (pair_nequip_allegro_kokkos.cpp:342-344) and synchronises the stream. */
int ahip_compute_dev(ahip_model *m, int nlocal, int nghost, const double *x_dev, const int *mtype_dev,
                     const double *cutoff_matrix_model, double *f_dev, double *eatom_dev,
                     double *eng_vir_dev, void *stream);

/* Same evaluation restricted to the centre atoms ilist[centre_begin .. centre_end) of the installed list (their edges, their
 * energies; forces still go to every atom they touch).  eng_vir_dev receives this range's partial sums.  This is what lets a
 * caller overlap the ghost exchange with the force evaluation (SURVEY 8e "Overlap"): centres whose whole neighbourhood is
 * local are evaluated while ghost positions are in flight, the boundary centres afterwards
 * (the reference evaluates all centres in one libtorch call after the exchange: pair_nequip_allegro.cpp:333-407). */
int ahip_compute_dev_range(ahip_model *m, int centre_begin, int centre_end, int nlocal, int nghost, const double *x_dev,
                           const int *mtype_dev, const double *cutoff_matrix_model, double *f_dev, double *eatom_dev,
                           double *eng_vir_dev, void *stream);

/* Number of entries of the installed (skin-inflated) neighbor list: sum of numneigh over the centre atoms
 * (the walk of pair_nequip_allegro.cpp:489-496 visits exactly these). */
long long ahip_last_list_size(ahip_model *m);

/* `compute allegro` / `compute allegro/atom` support: named entries of the model's output dict kept from the last
 * ahip_compute call (reference: the pair style stashes `output.at(name)` for every name registered through
 * add_custom_output, pair_nequip_allegro.cpp:403-406,681-684; compute/compute_allegro.cpp:81,114,145 reads them).
 * Entries this model's graph returns: "atomic_energy" [nlocal+nghost] (ghosts carry only their per-type shift, as in the
 * TorchScript model), "forces" [nlocal+nghost][3] (this evaluation's forces, before they are added to f),
 * "virial" [3][3], "total_energy" [1] (sum of atomic_energy over locals AND ghosts, cf. compute/README.md).
 * ahip_output_register: unknown names are accepted here and fail at the next ahip_compute, like the reference's
 * `output.at(name)`.  ahip_output_get copies the flattened tensor (capacity in doubles; *count = its length; pass out = NULL
 * to query the length).  Host-pointer ahip_compute only. */
int ahip_output_register(ahip_model *m, const char *name);
int ahip_output_get(ahip_model *m, const char *name, double *out, long long capacity, long long *count);

/* Edge list of the last compute: the tensors the reference would have handed to the model
 * (edge_index i64 [2][E], pair_nequip_allegro.cpp:601-602) plus |r_ij|.  Pass NULL buffers to
 * query nedges only.  This is what the `_NEQUIP_LOG_LEVEL=DEBUG` dump prints (:562-565,625). */
int ahip_get_edges(ahip_model *m, long long *nedges, long long *edge_index, double *rij);

/* Print "Allegro edges: i j rij" ... "end Allegro edges" to stdout with 0-based tag-1 ids and
 * %.10g distances, the exact format of pair_nequip_allegro.cpp:564,625,632.  tag may be NULL
 * (then atom indices are printed). */
int ahip_debug_dump_edges(ahip_model *m, const int *tag);

/* Per-stage device timings (option timing=1; HIP events on the launch stream): for every stage the SUM of its durations (ms) over the calls
 * since the previous ahip_get_timings -- called after every compute it is that compute's timing; names is a ';'-separated list valid until
 * the next call.  Waits for the recorded stages to finish; the compute calls themselves never wait for their events.
 * ahip_get_timing_counts: the number of launches behind each of those sums (same order), valid until the next ahip_get_timings. */
int ahip_get_timings(ahip_model *m, const char **names, const double **ms, int *n);
int ahip_get_timing_counts(ahip_model *m, const double **counts, int *n);

/* Diagnostics: edge slots of the tiles of the last fused evaluation (tiles x slots per tile) and the number that held an edge (= edges of
 * that call); their ratio is the packing efficiency bench.py reports as slots_used / slots_total.  0 / 0 after a non-fused evaluation. */
int ahip_last_tile_occupancy(ahip_model *m, long long *slots_used, long long *slots_total);

/* The model file's fifth metadata key, `allow_tf32` ("0" / "1").  The reference hands it to libtorch
 * (pair_nequip_allegro.cpp:267-270: at::globalContext().setAllowTF32CuBLAS / CuDNN): 1 = the model's author permits TF32-class
 * matrix arithmetic.  Here: with 1 (and option fused_arith=auto, the default) the fused model-S kernel runs its linears on the bf16
 * matrix cores with a two-term split (ahip_last_path reports "fused_tf32eq"); with 0 every path is float32-exact, float32-equivalent ("fused_f16x2", the default of the fused kernels) or better. */
int ahip_model_allow_tf32(const ahip_model *m, int *allow);

/* Kernel family and arithmetic used by the last compute: "generic_f32" | "generic_f64" | "fused_f16x2" | "fused_f32" | "fused_bf16x3" | "fused_tf32eq" ("" before). */
const char *ahip_last_path(ahip_model *m);
/* What fused_arith=auto decided for this model and why, one line ("" before the first evaluation): "fused_arith=auto: f16x2 kept: first evaluation within
 * ... of the float32 instance" or "fused_arith=auto: float32 instance (f32-input MFMA) selected: <reason>". */
const char *ahip_arith_note(const ahip_model *m);
/* Largest number of edges of any centre atom in the last compute. */
int ahip_last_max_degree(ahip_model *m);

/* Diagnostic: single-wave self-test of the fused path's register-chain MFMA primitive,
 * out[32][N] = in[32][K] @ W[K][N] (W row-major f64, in/out f32 host buffers). */
int ahip_debug_fused_linear(int K, int N, const double *W, const float *in, float *out);

/* Diagnostic (environment AHIP_FUSED_DBG=1): per-edge {g[3], dE/dd, dE/dfc, dE/dY1..3} of the last fused
 * compute, [nedges][8] floats, edge order = ahip_get_edges. */
int ahip_debug_fused_edges(ahip_model *m, float *out, long long nedges);

/* ---- mini-MD helpers used by the stand-alone driver / bench (device-resident) ------------- */

/* Build a full neighbor list with cutoff rc_list (= r_max + skin) for nlocal centre atoms among
 * nall atoms (ghost images already materialised), binned cell list on the GPU.  The result is
 * owned by the model and installed as its current list (as if ahip_neigh_update_dev was called). */
int ahip_build_neighbors_dev(ahip_model *m, int nlocal, int nall, const double *x_dev,
                             const double *lo, const double *hi, double rc_list, void *stream);

/* Re-neighboring criterion of a stand-alone driver (LAMMPS' `neigh_modify check yes` test, Neighbor::check_distance, evaluated
 * one step ahead): flag_dev[0] = 1 iff max_i |x_i - xhold_i| + 2 dt max_i |v_i| > half_skin over the n local atoms.  Two launches on
 * `stream`, no host synchronisation (the caller all-reduces the flag and reads it a step later). */
int ahip_reneighbor_flag_dev(ahip_model *m, int n, const double *x_dev, const double *xhold_dev, const double *v_dev, double dt,
                             double half_skin, int *flag_dev, void *stream);

/* Ghost atoms of ONE rank that owns the whole periodic box (the stand-alone driver's `borders` step at a re-neighboring; LAMMPS' Comm::borders on a
 * 1 x 1 x 1 grid): every periodic image of an owned atom that lies inside the halo of thickness rc around [lo, hi) -- shift +box_d allowed where
 * x_d < lo_d + rc, -box_d where x_d >= hi_d - rc, every combination but (0, 0, 0).  Writes, per ghost, its position, model type, the index of the
 * owned atom it images (int64) and the shift (what ahip_comm_set_plan_local takes); *nghost is the number of images, which may exceed `capacity`
 * (then only the first `capacity` were written: call again with larger arrays).  Needs box_d >= rc.  Synchronises `stream` (one 4-byte read-back). */
int ahip_borders_local_dev(ahip_model *m, int nlocal, const double *x_dev, const int *mtype_dev, const double *lo, const double *hi,
                           const double *box, double rc, int capacity, double *xg_dev, int *mtg_dev, long long *src_dev, double *shift_dev,
                           int *nghost, void *stream);

/* v += dtf*f/m ; x += dt*v  style velocity-Verlet half steps on device arrays (NVE).
 *   mode 0: v += 0.5*dt*f*ftm2v/mass[type]; x += dt*v      (initial_integrate)
 *   mode 1: v += 0.5*dt*f*ftm2v/mass[type]                  (final_integrate)  */
int ahip_nve_dev(ahip_model *m, int mode, int n, double *x_dev, double *v_dev, const double *f_dev,
                 const int *mtype_dev, const double *mass_by_mtype, double dt, double ftm2v, void *stream);
/* mode 0 for the nlocal owned atoms AND f[0 .. nall) = 0 behind it (the array the next force evaluation accumulates into, pair_nequip_allegro.cpp:370-377
 * adds to whatever f holds): one launch instead of an integrate and a zero-fill per step of the stand-alone driver. */
int ahip_nve_first_dev(ahip_model *m, int nlocal, int nall, double *x_dev, double *v_dev, double *f_dev, const int *mtype_dev,
                       const double *mass_by_mtype, double dt, double ftm2v, void *stream);

/* ---- ghost exchange of a spatially decomposed system (csrc/comm.hip) ---------------------------
 * Replaces what the reference gets from LAMMPS: ghost positions through Comm::forward_comm before compute(), and the forces the
 * model puts on ghost atoms (pair_nequip_allegro.cpp:370-377 adds to all nlocal + nghost rows) returned to their owners by the
 * reverse communication of `newton_pair on` (pair_nequip_allegro.cpp:149 requests it, :366-368 relies on it).
 * HIP pack / unpack kernels + RCCL ncclSend / ncclRecv groups on the caller's stream (one group per dimension, both directions);
 * librccl.so is opened at the first RCCL call.  The hosted transport moves the same packed buffers through a host callback
 * (CPU tests through the emulation build; several ranks sharing one GPU). */
typedef struct ahip_comm ahip_comm;
typedef struct ahip_xfer_op {
  int kind;          /* 0 = send, 1 = receive, 2 = in-place all-reduce float64 sum, 3 = in-place all-reduce int32 max */
  int peer;          /* rank (send / receive) */
  void *ptr;         /* the buffer as the library sees it (a device pointer on the GPU) */
  long long bytes;
} ahip_xfer_op;
/* performs all ops of one group (they may only complete together: post the receives before waiting for the sends); 0 = success */
typedef int (*ahip_xfer_fn)(void *user, int nops, const ahip_xfer_op *ops);

/* ncclGetVersion() of the librccl.so the RCCL transport opens (0: not available).  Reporting only (bench.py prints it per rank). */
int ahip_comm_rccl_version(void);

/* 128-byte RCCL unique id (ncclGetUniqueId): rank 0 creates it, the host program hands it to every rank */
int ahip_comm_unique_id(unsigned char id[128]);
int ahip_comm_create_rccl(int rank, int nranks, const unsigned char id[128], int device, ahip_comm **out);
int ahip_comm_create_hosted(int rank, int nranks, ahip_xfer_fn fn, void *user, ahip_comm **out);
void ahip_comm_free(ahip_comm *c);

/* Exchange plan, set after every re-neighboring (LAMMPS Comm::borders): nswaps directed swaps, two per dimension in the order
 * (dim 0: -,+), (dim 1: -,+), (dim 2: -,+).  Swap s sends rows send_idx_dev[s][0..nsend[s]) of x, shifted by shift[s] along dim[s],
 * to sendrank[s] and receives nrecv[s] rows from recvrank[s] into rows [first_recv[s], first_recv[s] + nrecv[s]) of x.  Later swaps
 * may send rows received by earlier dimensions.  The index arrays stay owned by the caller and must live until the next plan. */
int ahip_comm_set_plan(ahip_comm *c, int nswaps, const int *dim, const int *sendrank, const int *recvrank, const double *shift,
                       const int *nsend, const int *nrecv, const int *first_recv, const long long *const *send_idx_dev);
/* One rank: ghost g (row nlocal + g) is the image of local row src_dev[g] displaced by shift_dev[g][3]. */
int ahip_comm_set_plan_local(ahip_comm *c, int nlocal, int nghost, const long long *src_dev, const double *shift_dev);
/* Re-neighboring inside the library (round 6) -- what LAMMPS' Comm::exchange and Comm::borders give the reference at every re-neighboring
 * (pair_nequip_allegro.cpp:366-368 relies on the ghost shell and the reverse communication they set up).  Brick decomposition: `grid` ranks per dimension,
 * this rank at `coord`, rank = (cx * gy + cy) * gz + cz, brick [lo, hi), periodic `box`.  Both calls are collective over the communicator, synchronise `stream`
 * a few times (counts travel to the host) and agree on overflow TOGETHER (no rank is left waiting in an exchange).
 * ahip_comm_migrate: wraps the nlocal owned positions into the box and hands the atoms that left the brick (at most one brick per re-neighboring) to the
 *   neighbour brick with their velocity, tag and model type; the arrays have room for `capacity` rows; *nlocal_new is the new owned count, or -(rows needed)
 *   on every rank when some brick would overflow (the arrays are then partly migrated: treat as fatal, start with more room).
 * ahip_comm_borders: rebuilds the ghost shell of thickness rc behind the owned rows of x_dev / mtype_dev (capacity rows) by the six directed swaps of
 *   ahip_comm_set_plan -- send lists in ascending row order, later dimensions forwarding what earlier ones received -- and INSTALLS that plan in the communicator
 *   (the lists live in the library until the next call).  *nall = owned + ghost rows; a value above `capacity` means nothing may be used and every rank got that
 *   answer: call again with larger arrays (the owned rows are untouched). */
int ahip_comm_migrate(ahip_comm *c, int nlocal, double *x_dev, double *v_dev, long long *tag_dev, int *mtype_dev, int capacity, const double *box,
                      const int *grid, const int *coord, int *nlocal_new, void *stream);
int ahip_comm_borders(ahip_comm *c, int nlocal, double *x_dev, int *mtype_dev, int capacity, const double *lo, const double *hi, const double *box,
                      double rc, const int *grid, const int *coord, int *nall, void *stream);
/* x_dev [nall][3]: ghost rows refreshed from their owners (forward);  f_dev [nall][3]: ghost rows added to their owners (reverse). */
int ahip_comm_forward(ahip_comm *c, double *x_dev, void *stream);
int ahip_comm_reverse(ahip_comm *c, double *f_dev, void *stream);
/* in-place all-reduce over the ranks: kind 0 = float64 sum, 1 = int32 max (thermo sums, the re-neighboring flag) */
int ahip_comm_allreduce(ahip_comm *c, void *buf_dev, int count, int kind, void *stream);
/* ring send / receive + both all-reduces with known values through every transport entry point; AHIP_ERR_STATE on a wrong value */
int ahip_comm_selftest(ahip_comm *c, int n, void *stream);
/* hipMemsetAsync(ptr, 0, bytes) on the stream (the driver zeroes its force array with it) */
int ahip_fill_zero_dev(void *ptr_dev, long long bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
