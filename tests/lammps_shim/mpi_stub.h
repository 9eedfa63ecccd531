// TEST INFRASTRUCTURE ONLY: single-rank stand-in for the MPI calls of the pair style's constructor and of the compute.
#pragma once
typedef int MPI_Comm;
typedef int MPI_Info;
#define MPI_COMM_TYPE_SHARED 1
#define MPI_INFO_NULL 0
inline int MPI_Comm_split_type(MPI_Comm, int, int, MPI_Info, MPI_Comm *out) { *out = 0; return 0; }
inline int MPI_Comm_rank(MPI_Comm, int *r) { *r = 0; return 0; }
inline int MPI_Comm_free(MPI_Comm *) { return 0; }
typedef int MPI_Datatype;
typedef int MPI_Op;
#define MPI_IN_PLACE ((void *) 1)
#define MPI_DOUBLE 1
#define MPI_SUM 1
inline int MPI_Allreduce(const void *, void *, int, MPI_Datatype, MPI_Op, MPI_Comm) { return 0; }    // one rank: in place
