// TEST INFRASTRUCTURE ONLY: single-rank stand-in for the three MPI calls of the pair style's constructor.
#pragma once
typedef int MPI_Comm;
typedef int MPI_Info;
#define MPI_COMM_TYPE_SHARED 1
#define MPI_INFO_NULL 0
inline int MPI_Comm_split_type(MPI_Comm, int, int, MPI_Info, MPI_Comm *out) { *out = 0; return 0; }
inline int MPI_Comm_rank(MPI_Comm, int *r) { *r = 0; return 0; }
inline int MPI_Comm_free(MPI_Comm *) { return 0; }
