/* ORACLE (test infrastructure, never linked into the product): plain-C restatement of the
 * reference pair style's host glue, /root/reference/pair_nequip_allegro.cpp.
 *
 *   ref_count_edges   pass 1 of preprocess()          :488-512
 *   ref_prefix_sum    serial exclusive prefix sum     :515-519
 *   ref_fill_edges    pass 2 of preprocess()          :566-629  (allegro branch: ghost index kept, :602)
 *   ref_scatter       force / energy scatter          :369-380
 *   ref_virial_unpack [1,3,3] -> xx,yy,zz,xy,xz,yz    :387-392
 *
 * Used by tests/ to pin oracle/glue.py and the HIP edge build bit-for-bit (integer outputs) on
 * the same neighbor lists.  Parity pinned by construction against the reference text; the
 * reference's own tests hold no golden vectors for this path.
 */
#include <stddef.h>

#define REF_NEIGHMASK 0x1FFFFFFF

/* returns total number of edges; neigh_per_atom[ii] filled (:486,:508) */
long long ref_count_edges(int nlocal, const int *ilist, const int *numneigh, const int *const *firstneigh,
                          const double *x /*[n][3]*/, const int *type, int ntypes, const double *cutoff_matrix,
                          int *neigh_per_atom) {
  long long nedges = 0;
  for (int ii = 0; ii < nlocal; ii++) {
    int i = ilist[ii];
    int jnum = numneigh[i];
    const int *jlist = firstneigh[i];
    neigh_per_atom[ii] = 0;
    for (int jj = 0; jj < jnum; jj++) {
      int j = jlist[jj] & REF_NEIGHMASK;
      double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
      double rsq = dx * dx + dy * dy + dz * dz;
      double cutij = cutoff_matrix[(type[i] - 1) * ntypes + (type[j] - 1)];
      if (rsq <= cutij * cutij) { neigh_per_atom[ii]++; nedges++; }
    }
  }
  return nedges;
}

void ref_prefix_sum(int nlocal, const int *neigh_per_atom, int *cumsum) {
  if (nlocal > 0) cumsum[0] = 0;
  for (int ii = 1; ii < nlocal; ii++) cumsum[ii] = cumsum[ii - 1] + neigh_per_atom[ii - 1];
}

/* pos/atom_types for all ntotal atoms (:572-577), edges for locals (:584-628) */
void ref_fill_edges(int nlocal, int ntotal, const int *ilist, const int *numneigh, const int *const *firstneigh,
                    const double *x, const int *type, int ntypes, const double *cutoff_matrix,
                    const int *type_mapper, const int *cumsum, long long nedges, double *pos /*[ntotal][3]*/,
                    long long *edges /*[2][nedges]*/, long long *atom_types /*[ntotal]*/) {
  for (int ii = 0; ii < ntotal; ii++) {
    int i = ilist[ii];
    int itype = type[i];
    pos[3 * i] = x[3 * i]; pos[3 * i + 1] = x[3 * i + 1]; pos[3 * i + 2] = x[3 * i + 2];
    atom_types[i] = type_mapper[itype - 1];
    if (ii >= nlocal) continue;
    int jnum = numneigh[i];
    const int *jlist = firstneigh[i];
    long long edge_counter = cumsum[ii];
    for (int jj = 0; jj < jnum; jj++) {
      int j = jlist[jj] & REF_NEIGHMASK;
      double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
      double rsq = dx * dx + dy * dy + dz * dz;
      double cutij = cutoff_matrix[(itype - 1) * ntypes + (type[j] - 1)];
      if (rsq > cutij * cutij) continue;
      edges[edge_counter] = i;
      edges[nedges + edge_counter] = j;
      edge_counter++;
    }
  }
}

/* returns eng_vdwl */
double ref_scatter(int inum, int ntotal, const int *ilist, const double *forces /*[ntotal][3]*/,
                   const double *atomic_energies /*[ntotal]*/, int eflag_atom, double *f, double *eatom) {
  double eng_vdwl = 0.0;
  for (int ii = 0; ii < ntotal; ii++) {
    int i = ilist[ii];
    f[3 * i] += forces[3 * i]; f[3 * i + 1] += forces[3 * i + 1]; f[3 * i + 2] += forces[3 * i + 2];
    if (eflag_atom && ii < inum) eatom[i] = atomic_energies[i];
    if (ii < inum) eng_vdwl += atomic_energies[i];
  }
  return eng_vdwl;
}

void ref_virial_unpack(const double *v /*[1][3][3]*/, double *virial /*[6]*/) {
  virial[0] = v[0]; virial[1] = v[4]; virial[2] = v[8];
  virial[3] = v[1]; virial[4] = v[2]; virial[5] = v[5];
}
